#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03z2; rm -rf $OUT; mkdir -p $OUT
for rr in 0 1 0 1; do
V224HIP_VERBOSE=1 timeout -k 10 300 env ISEE3DSP_FFT_REGISTER_RADIX=$rr python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "10M register-radix FFT=$rr: $(python3 -c "import json; r=json.load(open('$OUT/c.json')); print(r['value'], r['ms_per_step'], r['stage_engine_ms']['pmdemod'], r['stage_engine_ms']['symdemod'], r['stage_engine_ms']['vdecode'])")" | tee -a $OUT/out.txt
grep -E "v224hip progressive: at end" $OUT/c.err | sed -n 2,3p | cut -c1-230 | tee -a $OUT/out.txt
done
