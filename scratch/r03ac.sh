#!/bin/bash
# round 3: k_acs_lds15 with the final defaults (late normalisation, joint minima/count load, no progressive waits) against the
# per-wave minima variant (L15_WAVEMIN=1); then the whole GPU suite on the default build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ac; rm -rf $OUT; mkdir -p $OUT
for v in default wmin es0 wmin default; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
