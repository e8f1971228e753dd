#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02bl; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dsp.py tests/test_gpu_dropin.py -m gpu -q -x > $OUT/t.log 2>&1; rc=$?; tail -3 $OUT/t.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 100 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { local label="$1"; shift
  timeout -k 10 300 python3 bench.py --workload chain --no-cpu "$@" > $OUT/c.json 2> $OUT/c.err || { echo "$label: failed"; tail -3 $OUT/c.err; return; }
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('%-22s' % '$label', c['value'], c['ms_per_step'], (c.get('host_capture') or {}).get('value'), {k:v for k,v in (c.get('stage_engine_ms') or {}).items() if k!='what'}, c.get('check'), (c.get('config') or {}).get('seams'))"
}
run "250k 60 s" --steps 4 --warmup 2
run "10M 48 s" --chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1
run "10M 12 s" --chain-rate 10000000 --chain-seconds 12 --steps 3 --warmup 1
run "10M 27 s, 8 segments" --chain-rate 10000000 --chain-seconds 27 --chain-segments 8 --steps 2 --warmup 1
run "2.5M 30 s" --chain-rate 2500000 --chain-seconds 30 --steps 3 --warmup 1
