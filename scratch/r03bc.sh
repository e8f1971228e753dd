#!/bin/bash
# round 3: the whole GPU suite on the last commit of the round
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03bc; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; exit $rc
