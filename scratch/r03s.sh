#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03s
for k in 0 1 2 3 4 5; do timeout -k 10 200 python3 scratch/queue_pair.py $k 2>&1 | tee -a gpurun_out/r03s/queue_pair.txt; done
