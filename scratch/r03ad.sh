#!/bin/bash
# round 3: timeline of the 10 MS/s chain (48 s capture): when symdemod's last symbol leaves, where the cut falls, when each decoder ends
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ad; rm -rf $OUT; mkdir -p $OUT
for m in progressive whole; do
  ISEE3_CHAIN_MODE=$m V224HIP_VERBOSE=1 timeout -k 10 400 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/$m.json 2> $OUT/$m.err || { tail -5 $OUT/$m.err; exit 1; }
  grep -E "isee3chain: last symbol|v224hip progressive|isee3chain/vdecode" $OUT/$m.err | sed "s/^/$m: /" | tee -a $OUT/timeline.txt
  python3 -c "import json,sys; d=json.loads([l for l in open('$OUT/$m.json') if l.startswith('{')][-1]); print('$m', d['value'], d['ms_per_step'], d.get('config'))" | tee -a $OUT/timeline.txt
done
