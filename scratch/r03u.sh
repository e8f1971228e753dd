#!/bin/bash
# the headline profile call (scratch/r03h.sh) on the last build of the round, plus the driver's own bench form
export R03TAG=r03u
bash scratch/r03h.sh || exit 1
cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03u/bench_driver_form.json 2> gpurun_out/r03u/bench_driver_form.err ) 2>&1 | grep real
python3 -c "
import json; r=json.load(open('gpurun_out/r03u/bench_driver_form.json')); print('driver form:', r['value'], r['ms_per_step'], r['chain']['value'], r['stress']['value'])"
