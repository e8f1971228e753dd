#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02as; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; rc=$?; tail -4 $OUT/gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
done
