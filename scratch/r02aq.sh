#!/bin/bash
# round 2, final build: default bench line, kernel stats (lone launches, chain at 250 kS/s and 10 MS/s), PMC passes of
# k_acs_lds15 (one counter group per run), framed batch, two-rank rehearsal of the bench contract on one device (gloo)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02aq; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
cut -c1-300 $OUT/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split1 -- python3 bench.py --split 1 --no-cpu --no-chain --no-frames > $OUT/bench_split1_under_rocprof.json 2> $OUT/split1.err || { tail -5 $OUT/split1.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain250k -- python3 bench.py --workload chain --no-cpu --steps 3 --warmup 1 > $OUT/chain250k.json 2> $OUT/chain250k.err || { tail -5 $OUT/chain250k.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain10M -- python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 2 --warmup 1 > $OUT/chain10M.json 2> $OUT/chain10M.err || { tail -5 $OUT/chain10M.err; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT/split1 $OUT/chain250k $OUT/chain10M -name "*kernel_stats.csv"); do echo $f; head -8 $f | cut -c1-200; done
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1 > $OUT/chain10M_48s_unprofiled.json 2>/dev/null
cut -c1-200 $OUT/chain10M_48s_unprofiled.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py --split 1 --symbols 30600 --steps 1 --warmup 0 --no-cpu --no-chain --no-frames > $OUT/pmc$i.log 2>&1 || { echo "pmc group $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
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
timeout -k 10 300 python3 scratch/framed_time.py > $OUT/framed_time.txt 2>&1; tail -6 $OUT/framed_time.txt
ISEE3_BENCH_ONE_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --symbols 2000000 --chain-steps 2 > $OUT/bench_2ranks_one_device.json 2> $OUT/bench_2ranks.err || { tail -20 $OUT/bench_2ranks.err; exit 1; }
cut -c1-400 $OUT/bench_2ranks_one_device.json
