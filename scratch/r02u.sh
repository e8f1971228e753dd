#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
V224HIP_VERBOSE=1 timeout -k 10 300 python scratch/place_cases.py 2>&1 | grep -E "spacer|create len" 
