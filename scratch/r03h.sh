#!/bin/bash
# (R03TAG=r03u bash scratch/r03h.sh writes to gpurun_out/r03u: the same call on a later build)
# round 3: whole GPU suite, default bench line, kernel traces (lone-decoder run and default run), PMC of k_acs_lds15
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/${R03TAG:-r03h}; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python3 -c "
import json; r=json.load(open('$OUT/bench_default.json')); ro=r['roofline']
print('value', r['value'], 'single', r['config']['split']['single_decoder']['value'], 'frac', ro['frac'], 'hbm', ro['hbm']['frac'], 'lone', ro['single_decoder']['frac'], ro['single_decoder']['hbm']['frac'])
print('chain', r['chain']['value'], r['chain']['roofline']['frac'] if r['chain'].get('roofline') else None, 'stress', r['stress']['value'], r['stress']['roofline']['frac'] if r['stress'].get('roofline') else None, r['stress']['config']['seams'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t1 -- python3 bench.py --split 1 --steps 1 --warmup 1 --no-cpu --no-chain --no-frames > $OUT/lone_under_rocprof.json 2> $OUT/t1.err || { tail -5 $OUT/t1.err; exit 1; }
cp $(find $OUT/t1 -name "*kernel_stats.csv" | head -1) $OUT/lone_kernel_stats.csv; rm -rf $OUT/t1
head -4 $OUT/lone_kernel_stats.csv | cut -c1-60,150-260
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu > $OUT/default_under_rocprof.json 2> $OUT/t2.err || { tail -5 $OUT/t2.err; exit 1; }
cp $(find $OUT/t2 -name "*kernel_stats.csv" | head -1) $OUT/default_kernel_stats.csv; rm -rf $OUT/t2
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py --split 1 --symbols 30600 --steps 1 --warmup 0 --no-cpu --no-chain --no-frames > $OUT/pmc$i.log 2>&1 || { echo "pmc group $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT/lds15_pmc_summary.txt <<'PY'
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
  rm -rf $OUT/pmc$i
done
cat $OUT/lds15_pmc_summary.txt
