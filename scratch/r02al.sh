#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02al; rm -rf $OUT; mkdir -p $OUT
run() { local label="$1"; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 $EXTRA > $OUT/c.json 2> $OUT/c.err || { echo "$label: failed"; tail -5 $OUT/c.err; return; }
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('%-34s' % '$label', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'}, c['check'], c['host_capture']['identical_output'])"
}
run "progressive (default)" X=1
run "block mode" ISEE3_CHAIN_MODE=block
run "progressive, verbose" V224HIP_VERBOSE=1
grep "progressive on" $OUT/c.err | tail -2
EXTRA="--chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1" 
run "10M 48 s progressive" X=1
run "10M 48 s block" ISEE3_CHAIN_MODE=block
EXTRA="--chain-rate 10000000 --chain-seconds 12 --steps 3 --warmup 1"
run "10M 12 s progressive" X=1
run "10M 12 s block" ISEE3_CHAIN_MODE=block
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py tests/test_gpu_viterbi.py -m gpu -q -x -k "chain or progressive or config" 2>&1 | tail -5
