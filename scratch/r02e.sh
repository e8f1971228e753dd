#!/bin/bash
# round 2, fifth GPU call: framed batch with ring halves (tracebacks under the next frame's passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02e; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_dropin.py -m gpu -q -k "frames or decode or dropin or reference" > $OUT/gpu_tests.log 2>&1; rc=$?
tail -15 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scratch/framed_time.py > $OUT/framed_time.txt 2>&1 || { tail -20 $OUT/framed_time.txt; exit 1; }
cat $OUT/framed_time.txt
