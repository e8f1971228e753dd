#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02p; rm -rf $OUT; mkdir -p $OUT
for sp in 1 2; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$sp -- python3 bench.py --split $sp --symbols 300000 --steps 1 --warmup 1 --no-cpu --chain-steps 6 > $OUT/b$sp.json 2> $OUT/b$sp.err || { tail -5 $OUT/b$sp.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b$sp.json'));c=d['chain']; print('split $sp: chain', c['value'], c['ms_per_step'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
f=$(find $OUT/s$sp -name "*kernel_stats.csv"); head -7 $f | cut -c1-60,150-260
find $OUT/s$sp -name "*kernel_trace.csv" -delete
done
