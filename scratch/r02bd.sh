#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bd; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -m gpu -q -x > $OUT/t.log 2>&1; rc=$?; tail -4 $OUT/t.log
[ $rc -ne 0 ] && exit 1
run() { local label="$1"; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload chain --no-cpu $EXTRA > $OUT/c.json 2> $OUT/c.err || { echo "$label: failed"; tail -5 $OUT/c.err; return; }
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('%-34s' % '$label', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
}
EXTRA="--steps 4 --warmup 2"
run "250k peak on the last pass" X=1
run "250k separate peak kernel" ISEE3DSP_PEAK_SEPARATE=1
EXTRA="--chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1"
run "10M peak on the last pass" X=1
run "10M separate peak kernel" ISEE3DSP_PEAK_SEPARATE=1
run "10M peak on the last pass" X=1
run "10M separate peak kernel" ISEE3DSP_PEAK_SEPARATE=1
