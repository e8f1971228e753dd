#!/bin/bash
# HBM traffic of the 10 MS/s chain by kernel: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), one chain step of 12 s
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03pmc; rm -rf $OUT; mkdir -p $OUT
RATE=${1:-10000000}; SECS=${2:-12}
args=""
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --workload chain --chain-rate $RATE --chain-seconds $SECS --steps 1 --warmup 0 --no-cpu > $OUT/$c.json 2> $OUT/$c.err || { tail -5 $OUT/$c.err; exit 1; }
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  args="$args $c=$f"
done
python3 scratch/pmc_by_kernel.py $OUT/pmc_chain_${RATE}.json $args > $OUT/pmc_chain_${RATE}.txt
head -40 $OUT/pmc_chain_${RATE}.txt
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
