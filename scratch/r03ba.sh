#!/bin/bash
# round 3: what the placement probe sees with the last kernel (V224HIP_VERBOSE), six processes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ba; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3 4 5 6; do V224HIP_VERBOSE=1 timeout -k 10 200 python3 scratch/l15_variants.py default 2>$OUT/err$i.txt | tee -a $OUT/variants.txt; grep "placement" $OUT/err$i.txt | head -4 | tee -a $OUT/variants.txt; done
