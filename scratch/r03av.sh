#!/bin/bash
# round 3: which builds of k_acs_lds15 with the two-instruction decision packing run a PAIR of launch chains staggered (21 us) rather
# than in lock-step (26 us): timing-only variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03av; rm -rf $OUT; mkdir -p $OUT
for v in default la0 es1 p1 p2 ms1 ss16 ss64 ds0 dsign0; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
