#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
echo "with tracebacks"; V224HIP_NO_STAGGER=1 timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | tail -3
echo "passes only"; V224HIP_NO_STAGGER=1 V224HIP_FRAMES_NO_TB=1 timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | grep -v Assert | tail -8
