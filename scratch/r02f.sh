#!/bin/bash
# round 2, sixth GPU call: generalised FFT sources + icesync correlator: full GPU tests, smoke
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02f; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then echo "test run killed (rc $rc)"; tail -5 $OUT/gpu_tests.log; exit 1; fi
tail -30 $OUT/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
