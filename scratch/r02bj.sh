#!/bin/bash
# last build of round 2: kernel stats of the chain at 250 kS/s and 10 MS/s
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bj; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain250k -- python3 bench.py --workload chain --no-cpu --steps 3 --warmup 1 > $OUT/chain250k.json 2> $OUT/chain250k.err || { tail -5 $OUT/chain250k.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain10M -- python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 > $OUT/chain10M.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT/chain250k $OUT/chain10M -name "*kernel_stats.csv"); do echo $f; head -6 $f | cut -c1-120; done
