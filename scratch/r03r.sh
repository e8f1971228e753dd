#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03r
for gm in 0.0 -1.0 -2.0 -3.0; do
V224HIP_PROG_GAMMA=$gm V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > gpurun_out/r03r/c.json 2> gpurun_out/r03r/c.err
echo "250k gamma $gm: $(cut -c1-130 gpurun_out/r03r/c.json)"; grep -E "v224hip progressive" gpurun_out/r03r/c.err | sed -n 3,4p | cut -c1-300
V224HIP_PROG_GAMMA=$gm V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > gpurun_out/r03r/c.json 2> gpurun_out/r03r/c.err
echo "10M gamma $gm: $(cut -c1-130 gpurun_out/r03r/c.json)"; grep -E "v224hip progressive" gpurun_out/r03r/c.err | sed -n 3,4p | cut -c1-300
done 2>&1 | tee gpurun_out/r03r/gamma.txt
