#!/bin/bash
# kernel-trace stats of the chain workload (60 s of 250 kS/s IQ through libisee3chain.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/prof_chain; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload chain --no-cpu > $OUT/bench_chain_under_rocprof.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
find $OUT/trace -name "*kernel_trace.csv" -delete
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -25 $f | cut -c1-200; done
cat $OUT/bench_chain_under_rocprof.json
