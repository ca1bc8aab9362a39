#!/bin/bash
# A/B two builds of the library on the GPU box: tools/ab_lib.sh <suffixA> <suffixB> -- <bench args>
cd $GRAFT_REPO_ROOT/ivfadc.jl_amd/csrc
A=$1; B=$2; shift 3
for rep in 1 2; do for v in $A $B; do cp libivfadc_hip_$v.so libivfadc_hip.so; touch libivfadc_hip.so; (cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -oE "\"value\": [0-9.]+|scan_ms_per_launch\": [0-9.]+|coarse_ms_per_launch\": [0-9.]+" | tr '\n' ' '; echo " <- $v"); done; done
