#!/bin/bash
# round 3: pmdemod's two-handle pipeline in the chain again (ISEE3_CHAIN_PM_PIPELINE=1), now that a block is six launches; new CLI test first
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ar; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_dsp.py -x -q -k "whichever or search_transform" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do for v in "" "ISEE3_CHAIN_PM_PIPELINE=1"; do
  env $v timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); st=d['roofline']['stages']; print('[${v:-default}] 10M chain', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/ab.txt
  env $v timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); st=d['roofline']['stages']; print('[${v:-default}] 250k chain', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/ab.txt
done; done
