#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02ao; rm -rf $OUT; mkdir -p $OUT
run() { local label="$1"; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 $EXTRA > $OUT/c.json 2> $OUT/c.err || { echo "$label: failed"; tail -5 $OUT/c.err; return; }
  python3 -c "import json;c=json.load(open('$OUT/c.json'));print('%-34s' % '$label', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'}, c['check'], c['host_capture']['identical_output'])"
}
run "shared front-end stream" X=1
run "symdemod on the null stream" ISEE3_CHAIN_SY_NULL=1
run "shared front-end stream" X=1
run "symdemod on the null stream" ISEE3_CHAIN_SY_NULL=1
timeout -k 10 300 python3 bench.py --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 3 > $OUT/b.json 2> $OUT/b.err
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('bench then chain, shared', d['value'], d['chain']['value'])"
ISEE3_CHAIN_SY_NULL=1 timeout -k 10 300 python3 bench.py --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 3 > $OUT/b.json 2> $OUT/b.err
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('bench then chain, sy null', d['value'], d['chain']['value'])"
EXTRA="--chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1"
run "10M 48 s shared" X=1
run "10M 48 s sy null" ISEE3_CHAIN_SY_NULL=1
