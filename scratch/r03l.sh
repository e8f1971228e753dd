#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03l; rm -rf $OUT; mkdir -p $OUT
for lead in 0.25 0.5 0.75 3.5 6; do timeout -k 10 200 python3 scratch/prio_two.py 0 $lead 2>&1 | tee -a $OUT/lead.txt; done
