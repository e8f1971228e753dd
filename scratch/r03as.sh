#!/bin/bash
# round 3: the window's arg-max inside k_window_demod (one launch less per symdemod window), pmdemod pipeline on by default: DSP tests, chains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03as; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -3 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); st=d['roofline']['stages']; print('10M chain', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/ab.txt
  timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); st=d['roofline']['stages']; print('250k chain', d['value'], d['ms_per_step'], {k: st[k]['engine_ms'] for k in ('pmdemod','symdemod','viterbi')})" | tee -a $OUT/ab.txt
done
