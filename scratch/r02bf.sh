#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bf; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_dsp.py -m gpu -q -x -k "progressive or chain or config2" > $OUT/t.log 2>&1; rc=$?; tail -4 $OUT/t.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 scratch/soak_progressive.py > $OUT/soak_progressive.txt 2>&1; tail -5 $OUT/soak_progressive.txt
for i in 1 2; do
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
done
