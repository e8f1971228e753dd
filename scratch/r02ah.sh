#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02ah; rm -rf $OUT; mkdir -p $OUT
run() { # label, env...
  local label="$1"; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 > $OUT/c.json 2>/dev/null || { echo "$label: failed"; return; }
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('%-44s' % '$label', c['value'], c['ms_per_step'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'}, c['check'])"
}
run "A baseline (share after front end)" X=1
run "B share always" ISEE3_CHAIN_SHARE=2
run "C share always, d1 low" ISEE3_CHAIN_SHARE=2 ISEE3_CHAIN_D1_LOW=1
run "D share always, both decoders low" ISEE3_CHAIN_SHARE=2 V224HIP_STREAM_PRIORITY=low
run "E share always, front end high" ISEE3_CHAIN_SHARE=2 ISEE3DSP_HIGH_PRIORITY=1
run "F baseline, both decoders low" V224HIP_STREAM_PRIORITY=low
run "G baseline, front end high" ISEE3DSP_HIGH_PRIORITY=1
run "H share always, decoders low + fe high" ISEE3_CHAIN_SHARE=2 V224HIP_STREAM_PRIORITY=low ISEE3DSP_HIGH_PRIORITY=1
