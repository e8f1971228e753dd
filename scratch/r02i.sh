#!/bin/bash
# round 2: parity table in k_acs_lds15: Viterbi GPU tests, default bench, PMC VALU count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02i; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_dropin.py -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
tail -5 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/bench_default.json'));print(d['value'], d['config']['split']['single_decoder'], d['roofline']['avg_launch_ms']); c=d['chain']; print('chain', c['value'], c['ms_per_step'], c['host_capture']['value'], c['stage_engine_ms'], c['decoded_bits'])"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc -- python3 bench.py --split 1 --symbols 30600 --steps 1 --warmup 0 --no-cpu --no-chain > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -3 $OUT/pmc.log; exit 1; }
  f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
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
  rm -f $(find $OUT/pmc -name "*counter_collection.csv") $(find $OUT/pmc -name "*kernel_trace.csv")
done
cat $OUT/pmc_summary.txt
