#!/bin/bash
# round 3: first FFT pass with the LDS-transposed store (pass orders asc / desc), then the lds15 build variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03d; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_dsp.py -x -q -k "fft or pmd or icesync or stress or chain" > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -3 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for o in asc desc; do
  ISEE3DSP_FFT_ORDER=$o timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/chain10M_48s_$o.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
  cut -c1-140 $OUT/chain10M_48s_$o.json
  ISEE3DSP_FFT_ORDER=$o timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$o -- python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 --no-cpu > $OUT/trace_chain10M_$o.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
  f=$(find $OUT/trace_$o -name "*kernel_stats.csv" | head -1); cp $f $OUT/chain10M_kernel_stats_$o.csv; grep -E "k_fft_pass|k_mix|k_rotate" $OUT/chain10M_kernel_stats_$o.csv | cut -c1-50,230-400
  rm -rf $OUT/trace_$o
done
bash scratch/r03c.sh "$@"
