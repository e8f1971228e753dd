#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02am; rm -rf $OUT; mkdir -p $OUT
V224HIP_VERBOSE=1 timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 > $OUT/c.json 2> $OUT/c.err
grep "v224hip progressive" $OUT/c.err | tail -4
python3 -c "import json;c=json.load(open('$OUT/c.json'));print(c['value'], c['ms_per_step'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
