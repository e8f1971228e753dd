#!/bin/bash
# repeat of the default bench line (box-to-box variation) + split-1 kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r02j; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 > $OUT/bench_default$i.json 2> $OUT/bench_default$i.err || { tail -20 $OUT/bench_default$i.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/bench_default$i.json'));print(d['value'], d['config']['split']['single_decoder'], d['roofline']['avg_launch_ms'], d['roofline']['frac']); c=d['chain']; print('chain', c['value'], c['ms_per_step'], c['host_capture']['value'], c['stage_engine_ms'], c['decoded_bits'])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split1 -- python3 bench.py --split 1 --no-cpu --no-chain > $OUT/bench_split1_under_rocprof.json 2> $OUT/split1.err || { tail -5 $OUT/split1.err; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
for f in $(find $OUT/split1 -name "*kernel_stats.csv"); do head -4 $f | cut -c1-200; done
