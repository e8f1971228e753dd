#!/bin/bash
# round 3: PMC traffic of the chain at both rates, then the default bench line with the stress record
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
bash scratch/r03_pmc_chain.sh 10000000 12 || exit 1
mkdir -p gpurun_out/r03e; cp gpurun_out/r03pmc/pmc_chain_10000000.* gpurun_out/r03e/
bash scratch/r03_pmc_chain.sh 250000 60 || exit 1
cp gpurun_out/r03pmc/pmc_chain_250000.* gpurun_out/r03e/
cd $R
timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.err || { tail -5 gpurun_out/r03e/bench_default.err; exit 1; }
python3 -c "
import json; r=json.load(open('gpurun_out/r03e/bench_default.json')); print(r['value'], r['chain']['value'], json.dumps(r.get('stress'))[:1500])"
