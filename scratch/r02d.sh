#!/bin/bash
# round 2, fourth GPU call: trimmed k_acs_lds15 (branch-metric pairs by v_alignbit + op_sel, scalar-base loads): GPU
# tests, default bench, split over 3 / 4 decoders, PMC passes (one counter group per run), chain kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02d; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then echo "test run killed (rc $rc)"; tail -5 $OUT/gpu_tests.log; exit 1; fi
tail -30 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then echo "tests failed: not benchmarking a wrong kernel"; exit 1; fi
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
cut -c1-1500 $OUT/bench_default.json
for S in 3 4; do
  timeout -k 10 200 python3 bench.py --split $S --symbols 4000000 --steps 1 --warmup 1 --no-cpu --no-chain > $OUT/bench_split$S.json 2> $OUT/bench_split$S.err || { tail -5 $OUT/bench_split$S.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/bench_split$S.json'));print('split $S', d['value'], d['config']['split']['single_decoder'])"
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py --split 1 --symbols 30600 --steps 1 --warmup 0 --no-cpu --no-chain > $OUT/pmc$i.log 2>&1 || { echo "pmc group $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
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
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split1 -- python3 bench.py --split 1 --no-cpu --no-chain > $OUT/bench_split1_under_rocprof.json 2> $OUT/split1.err || { tail -5 $OUT/split1.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain10M -- python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 > $OUT/chain10M.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT/split1 $OUT/chain10M -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -c1-160; done
cut -c1-600 $OUT/chain10M.json
