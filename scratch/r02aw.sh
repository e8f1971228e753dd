#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02aw; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_dropin.py -m gpu -q -x > $OUT/t.log 2>&1; rc=$?; tail -4 $OUT/t.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | tail -6
V224HIP_NO_STAGGER=1 timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | tail -3
timeout -k 10 300 python3 bench.py --symbols 2000000 --steps 1 --warmup 1 --no-cpu --no-chain > $OUT/b.json 2>/dev/null
python3 -c "import json;d=json.load(open('$OUT/b.json'));print(d['value'], d['frames'])"
