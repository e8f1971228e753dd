#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for k in 0 1 2 3 4; do V224HIP_TEST_GAP=0 timeout -k 10 100 python scratch/pipe_pair.py $k 2>&1 | grep own; done
