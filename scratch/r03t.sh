#!/bin/bash
# round 3: symdemod's fused window call (symd_window): DSP suite, then the chain with and without it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03t; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -3 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for sw in 1 0 1 0; do
  SYMDEMOD_STEPWISE=$sw V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err
  echo "250k stepwise=$sw: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms']['pmdemod'], r['stage_engine_ms']['symdemod'], r['stage_engine_ms']['vdecode'])")" | tee -a $OUT/fused.txt
  grep -E "v224hip progressive: 3" $OUT/c.err | sed -n 2,2p | cut -c1-260 | tee -a $OUT/fused.txt
done
for sw in 1 0; do
  SYMDEMOD_STEPWISE=$sw timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err
  echo "10M stepwise=$sw: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms'])")" | tee -a $OUT/fused.txt
done
