#!/bin/bash
# round 3: 10 MS/s chain with the search transform: timeline (V224HIP_VERBOSE) and per-stage engine times, progressive and whole
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03aj; rm -rf $OUT; mkdir -p $OUT
for m in progressive whole; do
  ISEE3_CHAIN_MODE=$m V224HIP_VERBOSE=1 timeout -k 10 400 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/$m.json 2> $OUT/$m.err || { tail -5 $OUT/$m.err; exit 1; }
  grep -E "isee3chain: last symbol|v224hip progressive: 2" $OUT/$m.err | head -4 | sed "s/^/$m: /" | tee -a $OUT/timeline.txt
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/$m.json') if l.startswith('{')][-1]); st=d['roofline']['stages']
print('$m', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/timeline.txt
done
