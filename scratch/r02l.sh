#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02l; rm -rf $OUT; mkdir -p $OUT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2
for mode in 1 0 1 0; do
  ISEE3_CHAIN_SHARE=$mode timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 5 --warmup 2 > $OUT/c.json 2>/dev/null
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('share=$mode', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
done
