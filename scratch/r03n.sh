#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03n
V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1 --no-cpu > gpurun_out/r03n/chain10M_verbose.json 2> gpurun_out/r03n/chain10M_verbose.err
grep -E "progressive|isee3chain/vdecode" gpurun_out/r03n/chain10M_verbose.err | head -12
cut -c1-160 gpurun_out/r03n/chain10M_verbose.json
