#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02au; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_dropin.py -m gpu -q -x > $OUT/t.log 2>&1; rc=$?; tail -4 $OUT/t.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | tail -7
V224HIP_CB_TOUCH=0 timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | head -3
sed -i 's|OUT=gpurun_out/r02at|OUT=gpurun_out/r02au/tr|' scratch/r02at.sh
bash scratch/r02at.sh 2>&1 | grep -E "k_chainback_par|k_init|per frame|k_l15|fused|move_start" 
