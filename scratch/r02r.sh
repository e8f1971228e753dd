#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02r; rm -rf $OUT; mkdir -p $OUT
export ISEE3DSP_NORMAL_PRIORITY=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --symbols 200000 --steps 1 --warmup 1 --no-cpu --chain-steps 2 > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
f=$(find $OUT/t -name "*kernel_trace.csv")
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
# last 40% of the trace = the chain part
cnt = collections.Counter()
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:30]
    cnt[(name, r.get("Queue_Id"), r.get("Stream_Id", ""))] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k)
PY
rm -f $f
