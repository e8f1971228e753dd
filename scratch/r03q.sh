#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r03q
timeout -k 10 900 python3 -X faulthandler scratch/stability.py ${SOAK_ITERS:-2000} > gpurun_out/r03q/stability.txt 2>&1; tail -5 gpurun_out/r03q/stability.txt
