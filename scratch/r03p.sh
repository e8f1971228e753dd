#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r03p; rm -rf $OUT; mkdir -p $OUT
ISEE3DSP_FFT_SMALL=1 ISEE3DSP_FFT_ORDER=asc timeout -k 10 300 python -m pytest tests/test_gpu_dsp.py -x -q -k "fft or pmd or stress or icesync" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for cfg in "0 desc" "1 asc" "1 desc" "0 desc" "1 asc"; do set -- $cfg
  ISEE3DSP_FFT_SMALL=$1 ISEE3DSP_FFT_ORDER=$2 V224HIP_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload chain --chain-rate 10000000 --chain-seconds 48 --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err
  echo "small $1 order $2: $(cut -c1-130 $OUT/c.json)" | tee -a $OUT/small.txt; grep -E "v224hip progressive" $OUT/c.err | sed -n 2,3p | cut -c1-250 | tee -a $OUT/small.txt
done
for v in "0 desc" "1 asc"; do set -- $v
  ISEE3DSP_FFT_SMALL=$1 ISEE3DSP_FFT_ORDER=$2 timeout -k 10 100 python3 scratch/fft_time.py default 23 | tee -a $OUT/small.txt
  ISEE3DSP_FFT_SMALL=$1 ISEE3DSP_FFT_ORDER=$2 timeout -k 10 300 python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu > $OUT/c.json 2> $OUT/c.err; echo "250k small $1 $2: $(cut -c1-130 $OUT/c.json)" | tee -a $OUT/small.txt
done
