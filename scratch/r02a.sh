#!/bin/bash
# round 2, first GPU call: full GPU test suite on the hardened build, then "before" kernel-trace stats of the chain at
# 250 kS/s (configs[2]) and 10 MS/s (configs[4] shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02a; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -40 $OUT/gpu_tests.log; exit 1; }
tail -3 $OUT/gpu_tests.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain250k -- python3 bench.py --workload chain --no-cpu --steps 3 --warmup 1 > $OUT/chain250k.json 2> $OUT/chain250k.err || { tail -5 $OUT/chain250k.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain10M -- python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 > $OUT/chain10M.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT -name "*kernel_stats.csv"); do echo $f; head -16 $f | cut -c1-160; done
cat $OUT/chain250k.json $OUT/chain10M.json
