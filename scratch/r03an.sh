#!/bin/bash
# round 3: timeline of the 250 kS/s chain (V224HIP_VERBOSE): when the cut is placed, when the front end ends, when each decoder ends
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03an; rm -rf $OUT; mkdir -p $OUT
V224HIP_VERBOSE=1 timeout -k 10 400 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/p.json 2> $OUT/p.err || { tail -5 $OUT/p.err; exit 1; }
grep -E "isee3chain: last symbol|v224hip progressive" $OUT/p.err | head -12 | tee -a $OUT/timeline.txt
python3 -c "
import json; d=json.loads([l for l in open('$OUT/p.json') if l.startswith('{')][-1]); st=d['roofline']['stages']
print(d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/timeline.txt
