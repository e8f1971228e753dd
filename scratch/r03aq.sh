#!/bin/bash
# round 3, build with the search transform and the late normalisation: scratch/r03h.sh (whole GPU suite, default bench, traces,
# PMC of k_acs_lds15), the driver's bench form, and the chain's PMC passes at both sample rates
export R03TAG=r03aq
bash scratch/r03h.sh || exit 1
cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03aq/bench_driver_form.json 2> gpurun_out/r03aq/bench_driver_form.err ) 2>&1 | grep real
python3 -c "
import json; r=json.load(open('gpurun_out/r03aq/bench_driver_form.json')); print('driver form:', r['value'], r['ms_per_step'], r['chain']['value'], r['stress']['value'])"
bash scratch/r03_pmc_chain.sh 10000000 12 > gpurun_out/r03aq/pmc10M.log 2>&1 || { tail -5 gpurun_out/r03aq/pmc10M.log; exit 1; }
cp gpurun_out/r03pmc/pmc_chain_10000000.* gpurun_out/r03aq/
bash scratch/r03_pmc_chain.sh 250000 60 > gpurun_out/r03aq/pmc250k.log 2>&1 || { tail -5 gpurun_out/r03aq/pmc250k.log; exit 1; }
cp gpurun_out/r03pmc/pmc_chain_250000.* gpurun_out/r03aq/
head -12 gpurun_out/r03aq/pmc_chain_10000000.txt
