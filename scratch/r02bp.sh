#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bp; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_viterbi.py -m gpu -q -x > $OUT/t.log 2>&1; rc=$?; tail -3 $OUT/t.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 100 python3 scratch/pair_frames.py 2>&1 | head -4
timeout -k 10 300 python3 bench.py --no-cpu --no-chain --no-frames --steps 2 --warmup 1 > $OUT/b.json 2>/dev/null
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('split', d['value'], 'single', d['config']['split']['single_decoder']['value'], 'launch', d['roofline']['avg_launch_ms'], 'pair', d['roofline']['two_decoders']['launch_pair_ms'])"
