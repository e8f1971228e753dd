#!/bin/bash
# round 3: the 2-rank rehearsal (one GPU, gloo) of the whole default line incl. the stress record's gather-and-stitch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03w
ISEE3_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu > gpurun_out/r03w/bench_2ranks_one_device.json 2> gpurun_out/r03w/err.txt || { tail -8 gpurun_out/r03w/err.txt; exit 1; }
python3 -c "
import json; r=json.load(open('gpurun_out/r03w/bench_2ranks_one_device.json'))
print(r['n_gpus'], r['ranks_seen'], r['value'], r['ms_per_step_per_rank'], r['chain']['value'], r['stress']['value'], r['stress']['ms_per_step_per_rank'], r['stress']['config']['seams'], r['stress']['check'], r['stress']['capture_generated_in_s'])"
timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 > gpurun_out/r03w/bench_default.json 2> gpurun_out/r03w/err1.txt || { tail -8 gpurun_out/r03w/err1.txt; exit 1; }
python3 -c "
import json; r=json.load(open('gpurun_out/r03w/bench_default.json')); print(r['value'], r['chain']['value'], r['stress']['value'], r['stress']['capture_generated_in_s'], r['cpu_baseline']['value'])"
