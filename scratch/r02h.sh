#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r02h
timeout -k 10 500 python scratch/converge.py > gpurun_out/r02h/converge.txt 2>&1; cat gpurun_out/r02h/converge.txt
