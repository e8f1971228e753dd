#!/bin/bash
# round 2, third GPU call: GPU tests on the LDS-FFT build, the default bench line (both halves of the
# metric), and kernel-trace stats of the chain at 250 kS/s and 10 MS/s
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02c; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then echo "test run killed (rc $rc)"; tail -5 $OUT/gpu_tests.log; exit 1; fi
tail -30 $OUT/gpu_tests.log
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain250k -- python3 bench.py --workload chain --no-cpu --steps 3 --warmup 1 > $OUT/chain250k.json 2> $OUT/chain250k.err || { tail -5 $OUT/chain250k.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain10M -- python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 > $OUT/chain10M.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT -name "*kernel_stats.csv"); do echo $f; head -16 $f | cut -c1-160; done
cat $OUT/chain250k.json $OUT/chain10M.json
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 3 --warmup 1 > $OUT/chain10M_noprof.json 2>&1; cat $OUT/chain10M_noprof.json
