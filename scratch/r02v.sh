#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r02v
AMD_LOG_LEVEL=3 timeout -k 10 300 python scratch/queue_cases.py W 2>&1 | grep -E "rocdevice.cpp|hipStreamCreate|hipStreamDestroy|^case" | grep -E "hardware queue|Created SWq|acquireQueue|releaseQueue|hipStreamCreate|hipStreamDestroy|^case|selected" | sed -e 's/\x1b\[[0-9;]*m//g' | cut -c1-20,60-230 > gpurun_out/r02v/queues.txt
cat gpurun_out/r02v/queues.txt | head -120
