#!/bin/bash
# round 3: shapes of the staggered priorities (reversed, two levels, from level 2 on) against the default
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03az; rm -rf $OUT; mkdir -p $OUT
for v in default p1r p1h p3 p3h default; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
