#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03o; rm -rf $OUT; mkdir -p $OUT
for kb in 0 88 0 88; do
  ISEE3DSP_FFT_LDS_KB=$kb V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/chain10M_lds$kb.json 2> $OUT/chain10M_lds$kb.err
  echo "FFT LDS floor $kb KiB: $(cut -c1-130 $OUT/chain10M_lds$kb.json)"; grep -E "v224hip progressive" $OUT/chain10M_lds$kb.err | head -3 | cut -c1-260
done
for kb in 0 88; do
  ISEE3DSP_FFT_LDS_KB=$kb timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/chain250k_lds$kb.json 2> $OUT/err.txt
  echo "250 kS/s, FFT LDS floor $kb KiB: $(cut -c1-130 $OUT/chain250k_lds$kb.json)"
done
