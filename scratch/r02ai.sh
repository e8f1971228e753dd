#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02ai; rm -rf $OUT; mkdir -p $OUT
LABEL="no background decoder      " BG=0 timeout -k 10 120 python3 scratch/starve.py 2>&1 | tail -2
LABEL="background decoder         " timeout -k 10 120 python3 scratch/starve.py 2>&1 | tail -2
LABEL="background low             " BG_PRIO=low timeout -k 10 120 python3 scratch/starve.py 2>&1 | tail -2
LABEL="both decoders low          " BG_PRIO=low CHAIN_DEC_PRIO=low timeout -k 10 120 python3 scratch/starve.py 2>&1 | tail -2
LABEL="front end high             " ISEE3DSP_HIGH_PRIORITY=1 timeout -k 10 120 python3 scratch/starve.py 2>&1 | tail -2
LABEL="both low + front end high  " BG_PRIO=low CHAIN_DEC_PRIO=low ISEE3DSP_HIGH_PRIORITY=1 timeout -k 10 120 python3 scratch/starve.py 2>&1 | tail -2
