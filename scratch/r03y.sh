#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03y; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -3 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "10M chain: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms']['pmdemod'], r['stage_engine_ms']['symdemod'], r['stage_engine_ms']['vdecode'])")" | tee -a $OUT/out.txt
timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "250k chain: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms']['pmdemod'], r['stage_engine_ms']['symdemod'], r['stage_engine_ms']['vdecode'])")" | tee -a $OUT/out.txt
done
