#!/bin/bash
# on the GPU box: Deep1B shape, matrix-core rounds forced (IVFADC_LB_EVERYWHERE) with the production build and build variants, and the exact tables
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export IVFADC_LB_EVERYWHERE=1
tools/ab_lib.sh "$@" -- --config deep1b --steps 10 --warmup 2 2>&1 | tee gpurun_out/deep_lb_ab.txt
unset IVFADC_LB_EVERYWHERE
timeout -k 10 300 python bench.py --config deep1b --steps 10 --warmup 2 --no-cpu-baseline --no-sweep 2>/dev/null | grep -oE "\"value\": [0-9.]+|scan_ms_per_launch\": [0-9.]+" | tr '\n' ' ' | tee -a gpurun_out/deep_lb_ab.txt; echo " <- exact tables (default)" | tee -a gpurun_out/deep_lb_ab.txt
