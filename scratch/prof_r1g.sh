#!/bin/bash
# final round-1 profiles: kernel-trace stats of `bench.py --split 1` (one decoder: what roofline.avg_launch_ms is measured on)
# and of the default `bench.py` (two decoders per stream), then PMC passes (one counter group per run) on a short run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/prof_r1g; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 bench.py --split 1 --no-cpu > $OUT/bench_split1_under_rocprof.json 2> $OUT/err1.txt || { echo "trace1 failed"; tail -5 $OUT/err1.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py --no-cpu > $OUT/bench_default_under_rocprof.json 2> $OUT/err2.txt || { echo "trace2 failed"; tail -5 $OUT/err2.txt; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT/trace1 -name "*kernel_stats.csv"); do head -4 $f | cut -c1-260; done
for f in $(find $OUT/trace2 -name "*kernel_stats.csv"); do head -4 $f | cut -c1-260; done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py --split 1 --symbols 30600 --steps 1 --warmup 0 --no-cpu > $OUT/pmc$i.log 2>&1 || { echo "pmc group $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" >> $OUT/pmc_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "k_acs_lds15<0" not in k: continue
    key = (k.split("(")[0][-40:], r["Counter_Name"])
    acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-42s %-24s avg/dispatch %16.1f  (dispatches %d)" % (k, c, s / n, n))
PY
    rm -f $(find $OUT/pmc$i -name "*counter_collection.csv") $(find $OUT/pmc$i -name "*kernel_trace.csv")
  else echo "group $i: no counter file" >> $OUT/pmc_summary.txt; fi
done
cat $OUT/pmc_summary.txt
