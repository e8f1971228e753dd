#!/bin/bash
# round 3: ablation of the first FFT pass (lib_alt/fft<mask>), spread of the two-decoder rate over fresh pairs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03f; rm -rf $OUT; mkdir -p $OUT
for v in default fft1 fft2 fft4 fft8 fft15; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -- python3 scratch/fft_time.py $v > $OUT/fft_$v.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  f=$(find $OUT/t_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(cat $OUT/fft_$v.txt)" | tee -a $OUT/fft_ablation.txt
  grep -E "k_fft_pass" $f | awk -F'","' '{printf "    %-48s calls %s avg %.1f us\n", substr($1,7,48), $2, $4/1000}' | tee -a $OUT/fft_ablation.txt
  rm -rf $OUT/t_$v
done
timeout -k 10 400 python3 scratch/pair_variance.py > $OUT/pair_variance.txt 2>&1; cat $OUT/pair_variance.txt
