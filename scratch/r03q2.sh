#!/bin/bash
export SOAK_ITERS=4000
bash scratch/r03q.sh
