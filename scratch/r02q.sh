#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r02q
export ISEE3DSP_NORMAL_PRIORITY=1
for c in X Z W V; do timeout -k 10 120 python scratch/queue_cases.py $c 2>&1 | grep case; done
echo "--- benches, normal priority, 8 queues"
OUT=gpurun_out/r02q
timeout -k 10 300 python3 bench.py --split 1 --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 4 > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('split 1 bench', d['value']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 300 python3 bench.py --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 4 > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('default bench', d['value']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain only', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
