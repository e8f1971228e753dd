#!/bin/bash
# round 3, first call: GPU suite on the re-created tree, default bench line, kernel trace of it, 2-rank rehearsal through bench.py's own launcher
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03a; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
cut -c1-600 $OUT/bench_default.json
ISEE3_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu > $OUT/bench_2ranks_one_device.json 2> $OUT/bench_2ranks.err || { tail -5 $OUT/bench_2ranks.err; exit 1; }
cut -c1-400 $OUT/bench_2ranks_one_device.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu > $OUT/trace_bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $f $OUT/bench_kernel_stats.csv; head -12 $OUT/bench_kernel_stats.csv | cut -c1-200
rm -rf $OUT/trace
