#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03k; rm -rf $OUT; mkdir -p $OUT
for p in 0 1 0 1; do timeout -k 10 200 python3 scratch/prio_two.py $p 2>&1 | tee -a $OUT/prio_two.txt; done
