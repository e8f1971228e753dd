#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bk; rm -rf $OUT; mkdir -p $OUT
run() { local label="$1"; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1 > $OUT/c.json 2> $OUT/c.err || { echo "$label: failed"; return; }
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('%-34s' % '$label', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
}
run "10M 48 s block (default)" X=1
run "10M 48 s progressive" ISEE3_CHAIN_MODE=progressive V224HIP_VERBOSE=1
grep "v224hip progressive" $OUT/c.err | tail -2
run "10M 48 s block, never share" ISEE3_CHAIN_SHARE=0
