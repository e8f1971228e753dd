#!/bin/bash
# round 3: PMC look at the search transform's kernels at 2^23 (scratch/fft_time.py): instructions, busy and wait cycles per dispatch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ao; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 scratch/fft_time.py default 23 > $OUT/pmc$i.log 2>&1 || { echo "pmc group $i failed"; tail -3 $OUT/pmc$i.log; continue; }
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if not (k.startswith("k_dft_bins") or "k_fft_pass" in k): continue
    key = (k.split("(")[0][:46], r["Counter_Name"])
    acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-48s %-24s avg/dispatch %16.1f  (dispatches %d)" % (k, c, s / n, n))
PY
  rm -rf $OUT/pmc$i
done
cat $OUT/summary.txt
