#!/bin/bash
# round 2, seventh GPU call: long stream blocks shared between two decoders (chain + vdecode pipe mode)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02g; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then echo "test run killed (rc $rc)"; tail -5 $OUT/gpu_tests.log; exit 1; fi
tail -30 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/bench_default.json'));print(d['value'], d['config']['split']['single_decoder']['value']); c=d['chain']; print('chain', c['value'], c['ms_per_step'], c['host_capture']['value'], c['stage_engine_ms'], c['decoded_bits'])"
ISEE3_CHAIN_SHARE=0 timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 3 --warmup 1 > $OUT/chain250k_noshare.json 2>/dev/null; python3 -c "import json;c=json.load(open('$OUT/chain250k_noshare.json'));print('250k no share', c['value'], c['ms_per_step'], c['stage_engine_ms'])"
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 12 --steps 3 --warmup 1 > $OUT/chain10M.json 2>/dev/null; python3 -c "import json;c=json.load(open('$OUT/chain10M.json'));print('10M', c['value'], c['ms_per_step'], c['host_capture']['value'], c['stage_engine_ms'])"
ISEE3_CHAIN_SHARE=2 timeout -k 10 200 python3 bench.py --workload chain --no-cpu --steps 3 --warmup 1 > $OUT/chain250k_share2.json 2>/dev/null; python3 -c "import json;c=json.load(open('$OUT/chain250k_share2.json'));print('250k share always', c['value'], c['ms_per_step'], c['stage_engine_ms'])"
timeout -k 10 200 python3 bench.py --workload chain --no-cpu --chain-rate 10000000 --chain-seconds 48 --steps 2 --warmup 1 > $OUT/chain10M_48s.json 2>/dev/null; python3 -c "import json;c=json.load(open('$OUT/chain10M_48s.json'));print('10M 48 s', c['value'], c['ms_per_step'], c['host_capture']['value'], c['stage_engine_ms'], c['decoded_bits'])"
