#!/bin/bash
# round 3: stream priorities in the chain once more, now that the front end is shorter: decoders high, front end high
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03am; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do for v in "" "V224HIP_STREAM_PRIORITY=high" "V224HIP_STREAM_PRIORITY=low" "ISEE3DSP_HIGH_PRIORITY=1"; do
  env $v timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); st=d['roofline']['stages']; print('[${v:-default}] 10M chain', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/ab.txt
  env $v timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); st=d['roofline']['stages']; print('[${v:-default}] 250k chain', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/ab.txt
done; done
