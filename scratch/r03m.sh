#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03m
timeout -k 10 300 python3 scratch/pair_modes.py 2>&1 | tee gpurun_out/r03m/pair_modes.txt
