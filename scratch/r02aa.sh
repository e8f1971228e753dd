#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02aa; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --split 1 --symbols 2000000 --steps 1 --warmup 1 --no-cpu --chain-steps 4 > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('split 1 bench', d['value']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --chain-steps 4 > $OUT/bench_default.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/bench_default.json'));print('default bench', d['value'], d['config']['split']['single_decoder']); c=d['chain']; print('   chain', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 4 --warmup 2 > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain only', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1 > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain 10M 48 s', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 3 --warmup 1 > $OUT/c.json 2>/dev/null
python3 -c "import json;c=json.load(open('$OUT/c.json'));print('chain 10M 12 s', c['value'], c['ms_per_step'], c['host_capture']['value'], {k:v for k,v in c['stage_engine_ms'].items() if k!='what'})"
timeout -k 10 600 python -m pytest tests/test_gpu_dsp.py -m gpu -q -x 2>&1 | tail -3
