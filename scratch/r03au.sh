#!/bin/bash
# round 3: decision packing with v_perm's sign-replicating selectors (2 instructions per butterfly instead of 3): Viterbi tests,
# then lone / pair rates of the default build against lib_alt/dsign0 (L15_DECSIGN=0, the build before)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03au; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_viterbi.py -x -q > $OUT/pytest_v.log 2>&1; rc=$?; tail -3 $OUT/pytest_v.log
[ $rc -eq 0 ] || exit $rc
for v in default dsign0 default dsign0; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
