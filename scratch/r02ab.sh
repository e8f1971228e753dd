#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02ab; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py -m gpu -q -x -k "chainback or framed or frames or dropin or api" 2>&1 | tail -15 || exit 1
timeout -k 10 300 python3 scratch/framed_time.py > $OUT/framed_time.txt 2>&1; cat $OUT/framed_time.txt
V224HIP_SERIAL_CHAINBACK=2 timeout -k 10 300 python3 scratch/framed_time.py > $OUT/framed_time_serial.txt 2>&1; cat $OUT/framed_time_serial.txt
