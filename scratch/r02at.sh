#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02at; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 scratch/frames_trace.py > $OUT/run.txt 2>&1
cat $OUT/run.txt | tail -2
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last batch: everything after the last big gap (> 2 ms)
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if st[i] - max(en[:i][-50:]) > 2_000_000: cut = i
rows = rows[cut:]
print("kernels in the last batch:", len(rows), "span %.3f ms" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
byq = collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    names = collections.Counter(r["Kernel_Name"].split("(")[0][-30:] for r in rs)
    print("queue", q, "kernels", len(rs), "busy %.3f ms of %.3f ms span" % (busy / 1e6, span / 1e6))
    tot = collections.defaultdict(lambda: [0, 0])
    gaps = collections.defaultdict(lambda: [0, 0])
    for i, r in enumerate(rs):
        n = r["Kernel_Name"].split("(")[0][-30:]
        tot[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[n][1] += 1
        if i:
            g = int(r["Start_Timestamp"]) - int(rs[i - 1]["End_Timestamp"])
            gaps[n][0] += g; gaps[n][1] += 1
    for n in tot:
        print("    %-32s n %5d  avg %8.2f us   avg gap before %7.2f us" % (n, tot[n][1], tot[n][0] / tot[n][1] / 1e3, gaps[n][0] / max(1, gaps[n][1]) / 1e3))
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
