#!/bin/bash
# round 3: the neighbourhood of the build that pairs well (two-instruction decision packing + staggered wave priorities in level 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03aw; rm -rf $OUT; mkdir -p $OUT
for v in p1 dsign0 p1es1 p1ss8 p1ss16 p1la0 p1ds0 p1 default; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
