#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for k in 0 1 2 3 4 5 6 7 8; do ISEE3DSP_NORMAL_PRIORITY=1 timeout -k 10 100 python scratch/queue_scan.py $k 2>&1 | grep dummy; done
for k in 0 1 2 3 4; do timeout -k 10 100 python scratch/queue_scan.py $k 2>&1 | grep dummy; done
for k in 0 2 4 6; do GPU_MAX_HW_QUEUES=4 ISEE3DSP_NORMAL_PRIORITY=1 timeout -k 10 100 python scratch/queue_scan.py $k 2>&1 | grep dummy; done
