#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02o; rm -rf $OUT; mkdir -p $OUT
for q in 8 16 24; do
export GPU_MAX_HW_QUEUES=$q
timeout -k 10 300 python3 bench.py --split 1 --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 4 > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('queues $q: split 1 bench', d['value']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 300 python3 bench.py --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 4 > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('queues $q: default bench', d['value']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
done
