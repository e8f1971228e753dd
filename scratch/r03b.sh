#!/bin/bash
# round 3: DSP suite after the stepped carrier / 32-column first pass, 10 MS/s chain line + kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03b; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -5 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for v in 0 1; do
ISEE3DSP_CARRIER_CLOSED=$v timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/chain10M_48s_closed$v.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
cut -c1-420 $OUT/chain10M_48s_closed$v.json
done
timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/chain250k.json 2> $OUT/chain250k.err || { tail -5 $OUT/chain250k.err; exit 1; }
cut -c1-420 $OUT/chain250k.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 --no-cpu > $OUT/trace_chain10M.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $f $OUT/chain10M_kernel_stats.csv; head -24 $OUT/chain10M_kernel_stats.csv | cut -c1-60,200-330
rm -rf $OUT/trace
