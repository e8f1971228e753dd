#!/bin/bash
# round 3: one decoder of a pair at the top wave priority (option "prio"): each decoder's rate side by side
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03i; rm -rf $OUT; mkdir -p $OUT
for p in 0 1 0 1; do timeout -k 10 200 python3 scratch/prio_pair.py $p 2>&1 | tee -a $OUT/prio_pair.txt; done
