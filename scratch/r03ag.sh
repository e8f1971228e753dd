#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ag; rm -rf $OUT; mkdir -p $OUT
for lg in 23 20; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$lg -- python3 scratch/fft_time.py default $lg > $OUT/t$lg.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cat $OUT/t$lg.txt
f=$(find $OUT/trace$lg -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), "%8.1f us avg %8.1f min %8.1f max" % (float(r['AverageNs']) / 1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf $OUT/trace$lg
done
