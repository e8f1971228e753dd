#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03z3
( time timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 > gpurun_out/r03z3/bench_default.json 2> gpurun_out/r03z3/err.txt ) 2>&1 | grep real
python3 -c "
import json; r=json.load(open('gpurun_out/r03z3/bench_default.json')); s=r['stress']
print(r['value'], r['chain']['value'], s['value'], s['ms_per_step'], s['config']['processed_rate_Msamples_per_s'], s['roofline']['frac'], s['config']['seams'], s['check'], s['capture_generated_in_s'], s['config']['workload'][:160])"
