#!/bin/bash
# round 3: one more box for the pair rate of the last build (short bench line: no stress, no CPU baseline)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03bd; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 140 python3 bench.py --steps 2 --warmup 1 --no-stress --no-cpu --no-frames > $OUT/bench_short.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
python3 -c "
import json; r=json.load(open('$OUT/bench_short.json')); ro=r['roofline']
print('value', r['value'], 'single', r['config']['split']['single_decoder']['value'], 'frac', ro['frac'], 'pair_ms', ro['launch_pair_ms'], 'chain', r['chain']['value'])"
