#!/bin/bash
# round 3: k_acs_lds15 prologue: late input normalisation (L15_LATEADJ), progressive waits for the level-0 loads (L15_EARLYSTART),
# minima and their count loaded together -- builds: default (all), es0 (no progressive waits), adj0 (normalisation at stage 1), old (neither)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ab; rm -rf $OUT; mkdir -p $OUT
for v in old adj0 es0 default old default; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
timeout -k 10 900 python -m pytest tests/test_gpu_viterbi.py -x -q > $OUT/pytest_v.log 2>&1; tail -3 $OUT/pytest_v.log
