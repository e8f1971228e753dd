#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for k in 0 1 2 3 4 5 6 7; do timeout -k 10 100 python scratch/queue_scan.py $k 2>&1 | grep dummy; done
echo "--- GPU_MAX_HW_QUEUES=4"
for k in 0 1 2 3 4 5; do GPU_MAX_HW_QUEUES=4 timeout -k 10 100 python scratch/queue_scan.py $k 2>&1 | grep dummy; done
echo "--- GPU_MAX_HW_QUEUES=2"
for k in 0 1 2 3; do GPU_MAX_HW_QUEUES=2 timeout -k 10 100 python scratch/queue_scan.py $k 2>&1 | grep dummy; done
