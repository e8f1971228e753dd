#!/bin/bash
# A/B on one box: k_acs_lds15 with the parity table (lib/) and with counted parities (lib_alt/), split 2 and split 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02k; rm -rf $OUT; mkdir -p $OUT
cp isee3-decoder_amd/lib/libviterbi224_hip.so $OUT/lib_tab.so
for v in tab notab tab notab; do
  if [ $v = tab ]; then cp $OUT/lib_tab.so isee3-decoder_amd/lib/libviterbi224_hip.so; else cp isee3-decoder_amd/lib_alt/libviterbi224_hip.so isee3-decoder_amd/lib/libviterbi224_hip.so; fi
  timeout -k 10 200 python3 bench.py --symbols 6000000 --steps 2 --warmup 1 --no-cpu --no-chain > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/b.json'));print('$v', d['value'], d['config']['split']['single_decoder']['value'], d['config']['split']['single_decoder']['split_avg_launch_ms'], d['roofline']['avg_launch_ms'])"
done
cp $OUT/lib_tab.so isee3-decoder_amd/lib/libviterbi224_hip.so; rm -f $OUT/lib_tab.so
