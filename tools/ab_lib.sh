#!/bin/bash
# A/B builds of the library on the GPU box: tools/ab_lib.sh <suffixA> <suffixB> [...] -- <bench args>
# (variants built with tools/build_variant.sh; the library in use is restored from the first suffix at the end)
cd $GRAFT_REPO_ROOT/ivfadc.jl_amd/csrc
V=()
while [ "$1" != "--" ]; do V+=("$1"); shift; done
shift
for rep in 1 2; do for v in "${V[@]}"; do cp libivfadc_hip_$v.so libivfadc_hip.so; touch libivfadc_hip.so; (cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --no-cpu-baseline --no-sweep "$@" 2>&1 | grep -oE "\"value\": [0-9.]+|scan_ms_per_launch\": [0-9.]+|coarse_ms_per_launch\": [0-9.]+" | tr '\n' ' '; echo " <- $v"); done; done
cp libivfadc_hip_${V[0]}.so libivfadc_hip.so
