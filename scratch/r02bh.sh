#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bh; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -m gpu -q -x > $OUT/t.log 2>&1; rc=$?; tail -3 $OUT/t.log
[ $rc -ne 0 ] && exit 1
for a in "--chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1" "--chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1" "--steps 4 --warmup 2"; do
timeout -k 10 200 python3 bench.py --workload chain --no-cpu $a > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain $a', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
done
