#!/bin/bash
# round 3: pmdemod's two-handle pipeline in the chain (PMDEMOD_SERIAL=1: the one-handle loop)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03z; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -3 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for ser in 1 0 1 0; do
timeout -k 10 300 env PMDEMOD_SERIAL=$ser python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "10M serial=$ser: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms']['pmdemod'], r['stage_engine_ms']['symdemod'], r['stage_engine_ms']['vdecode'], r['host_capture']['value'])")" | tee -a $OUT/out.txt
timeout -k 10 300 env PMDEMOD_SERIAL=$ser python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "250k serial=$ser: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms']['pmdemod'], r['stage_engine_ms']['symdemod'], r['stage_engine_ms']['vdecode'], r['host_capture']['value'])")" | tee -a $OUT/out.txt
done
