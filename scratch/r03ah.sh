#!/bin/bash
# round 3: the small kernels of a pmdemod block removed (per-workgroup sums, final sums on the host): DSP tests, transform timing with a kernel trace, chains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ah; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py -x -q > $OUT/pytest_dsp.log 2>&1; rc=$?; tail -5 $OUT/pytest_dsp.log
[ $rc -eq 0 ] || exit $rc
for lg in 23 18; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$lg -- python3 scratch/fft_time.py default $lg > $OUT/t$lg.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cat $OUT/t$lg.txt | tee -a $OUT/ab.txt
f=$(find $OUT/trace$lg -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY' | tee -a $OUT/ab.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), "%8.1f us avg %8.1f min %8.1f max" % (float(r['AverageNs']) / 1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf $OUT/trace$lg
done
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); print('10M chain', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.txt
  timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err || { tail -5 $OUT/c.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/c.json') if l.startswith('{')][-1]); print('250k chain', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.txt
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 --no-cpu > $OUT/trace_chain10M.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $f $OUT/chain10M_kernel_stats.csv
python3 - $OUT/chain10M_kernel_stats.csv <<'PY' | tee -a $OUT/ab.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), "%8.1f us" % (float(r['AverageNs']) / 1e3), r['Percentage'])
PY
rm -rf $OUT/trace
