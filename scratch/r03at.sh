#!/bin/bash
# round 3: chains at a time per GPU in the stress record (configs[4]): 1, 2 (default), 3, 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03at; rm -rf $OUT; mkdir -p $OUT
for c in 2 1 3 4 2 3; do
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --symbols 300000 --no-cpu --no-frames --segment-concurrency $c > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$OUT/b.json') if l.startswith('{')][-1]); s=d['stress']; print('concurrency $c: stress', s['value'], s['ms_per_step'], s['config']['seams'], s['check'])" | tee -a $OUT/ab.txt
done
