#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for w in 192 128 96 64 48; do V224HIP_CB_WARM=$w timeout -k 10 100 python3 scratch/cb_warm.py 2>&1 | tail -4; done
for w in 192 96 64; do echo "warm $w"; V224HIP_CB_WARM=$w timeout -k 10 200 python3 scratch/framed_time.py 2>&1 | tail -2; done
