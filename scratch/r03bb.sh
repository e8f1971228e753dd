#!/bin/bash
# round 3: stream chunk of the chain's decoders 510 (warm-up 1 020) against 1 020 (warm-up 2 040), both chains; then smoke() of the last build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03bb; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do for v in "" "V224HIP_CHUNK=510" "V224HIP_CHUNK=765"; do
  env $v timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); print('[${v:-default}] 10M chain', d['value'], d['ms_per_step'], d['check'] if 'check' in d else '')" | tee -a $OUT/ab.txt
  env $v timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); print('[${v:-default}] 250k chain', d['value'], d['ms_per_step'], d['check'] if 'check' in d else '')" | tee -a $OUT/ab.txt
done; done
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
