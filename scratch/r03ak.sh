#!/bin/bash
# round 3: knobs of the search transform: 16 columns per workgroup (32 KiB tiles) against 32; grid of the exact-bin kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ak; rm -rf $OUT; mkdir -p $OUT
for v in "" "ISEE3DSP_FFT_F32_COLS=16" "ISEE3DSP_DFT_NB=256" "ISEE3DSP_DFT_NB=384" "ISEE3DSP_DFT_NB=512"; do
  env $v timeout -k 10 100 python3 scratch/fft_time.py default 23 | sed "s/^/[${v:-default}] /" | tee -a $OUT/ab.txt
done
ISEE3DSP_FFT_F32_COLS=16 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scratch/fft_time.py default 23 > $OUT/t.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY' | tee -a $OUT/ab.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), "%8.1f us avg %8.1f min %8.1f max" % (float(r['AverageNs']) / 1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf $OUT/trace
for i in 1 2; do for v in "" "ISEE3DSP_FFT_F32_COLS=16" "ISEE3DSP_DFT_NB=384"; do
  env $v timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); print('[${v:-default}] 10M chain', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.txt
done; done
