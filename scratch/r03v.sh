#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03v
V224HIP_VERBOSE=1 timeout -k 10 300 python3 scratch/pair_matrix.py 5 2> gpurun_out/r03v/create.err | tee gpurun_out/r03v/pair_matrix.txt
grep "create len" gpurun_out/r03v/create.err | cut -c1-200
for m in progressive whole; do ISEE3_CHAIN_MODE=$m V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --steps 2 --warmup 1 --no-cpu 2>&1 >/dev/null | grep "isee3chain: last symbol" | head -3 | sed "s/^/$m: /"; done | tee gpurun_out/r03v/front_end_alone.txt
