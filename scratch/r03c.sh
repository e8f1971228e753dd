#!/bin/bash
# A/B of lds15 builds (scratch/l15_variants.py): args = names under lib_alt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03c; mkdir -p $OUT
for v in default "$@" default; do
  timeout -k 10 200 python3 scratch/l15_variants.py $v >> $OUT/variants.txt 2>$OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  tail -1 $OUT/variants.txt
done
