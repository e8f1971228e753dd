#!/bin/bash
# round 3: the default bench line of the last build with the re-issued PMC constants, and l15_variants once more (a fourth box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03ay; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python3 -c "
import json; r=json.load(open('$OUT/bench_default.json')); ro=r['roofline']
print('value', r['value'], 'single', r['config']['split']['single_decoder']['value'], 'frac', ro['frac'], 'hbm', ro['hbm']['frac'], 'lone', ro['single_decoder']['frac'], ro['launch_pair_ms'])
print('chain', r['chain']['value'], 'stress', r['stress']['value'], r['stress']['config']['seams'], 'cpu', r['cpu_baseline']['value'])"
for v in default dsign0; do timeout -k 10 200 python3 scratch/l15_variants.py $v 2>$OUT/err.txt | tee -a $OUT/variants.txt || { tail -5 $OUT/err.txt; exit 1; }; done
