#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02ac; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 2 > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print(d['value'], d['frames']); print(d['chain']['value'])"
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
