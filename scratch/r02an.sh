#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02br; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; rc=$?; tail -5 $OUT/gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/bench_default.json'));print('default bench', d['value'], d['config']['split']['single_decoder']['value'], d['frames']['decoders_2'], d['frames']['decoders_3']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'}, c['cpu_baseline']['value'])"
for a in "--chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1" "--chain-rate 10000000 --chain-seconds 12 --steps 3 --warmup 1"; do
timeout -k 10 200 python3 bench.py --workload chain --no-cpu $a > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain $a', c['value'], c['ms_per_step'], c['host_capture']['value'])"
done
