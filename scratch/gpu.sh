#!/bin/bash
# local wrapper: rebuild everything in-tree (the .so files travel with the snapshot), then run a script on the GPU box
# usage: scratch/gpu.sh <timeout-seconds> <script under scratch/> [args]
set -e
cd "$(dirname "$0")/.."
make -s -C isee3-decoder_amd all
make -s -C oracle all
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "bash scratch/$*"
