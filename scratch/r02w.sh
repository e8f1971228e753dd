#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for g in 0 1 2 3 4 5 6 7 8; do V224HIP_TEST_GAP=$g timeout -k 10 100 python scratch/pipe_scan.py 2>&1 | grep gap; done
