"""Summarise rocprofv3 --pmc counter_collection.csv files by kernel: dispatches and summed counter values.
usage: pmc_by_kernel.py out.json name=path.csv [name=path.csv ...]   (one csv per PMC pass)"""
import collections, csv, json, re, sys
out = collections.defaultdict(lambda: {"dispatches": 0})
for arg in sys.argv[2:]:
    _, path = arg.split("=", 1)
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r.get("Kernel_Name", "")).replace("void ", "").strip()
        c = r["Counter_Name"]
        out[k][c] = out[k].get(c, 0.0) + float(r["Counter_Value"])
        seen[k].add(r.get("Dispatch_Id", ""))
    for k, ids in seen.items():
        out[k]["dispatches"] = max(out[k]["dispatches"], len(ids))
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("WRITE_SIZE", 0) - 2 * kv[1].get("FETCH_SIZE", 0)):
    print("%-60s %s" % (k[:60], {a: round(b, 1) for a, b in v.items()}))
