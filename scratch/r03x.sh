#!/bin/bash
# round 3: later FFT passes with walked twiddles (default) against value-by-value lookups (lib_alt/fftlookup)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03x; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -3 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for v in default fftlookup; do timeout -k 10 100 python3 scratch/fft_time.py $v 23 | tee -a $OUT/fft_walk.txt; done
for v in default fftlookup; do timeout -k 10 100 python3 scratch/fft_time.py $v 18 | tee -a $OUT/fft_walk.txt; done
for i in 1 2; do
timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "10M chain: $(cut -c1-130 $OUT/c.json)" | tee -a $OUT/fft_walk.txt
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 --no-cpu > $OUT/trace_chain10M.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $f $OUT/chain10M_kernel_stats.csv
python3 - $OUT/chain10M_kernel_stats.csv <<'PY' | tee -a $OUT/fft_walk.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), "%8.1f us" % (float(r['AverageNs']) / 1e3), r['Percentage'])
PY
rm -rf $OUT/trace
