#!/bin/bash
# round 3: validation call of the build with the two-instruction decision packing + staggered level-3 priorities:
# scratch/r03h.sh (whole GPU suite, default bench, traces, PMC of k_acs_lds15) and the driver's bench form
export R03TAG=r03ax
bash scratch/r03h.sh || exit 1
cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03ax/bench_driver_form.json 2> gpurun_out/r03ax/bench_driver_form.err ) 2>&1 | grep real
python3 -c "
import json; r=json.load(open('gpurun_out/r03ax/bench_driver_form.json')); print('driver form:', r['value'], r['ms_per_step'], r['chain']['value'], r['stress']['value'])"
