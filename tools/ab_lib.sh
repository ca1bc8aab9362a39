#!/bin/bash
# A/B builds of the library on the GPU box: tools/ab_lib.sh <suffixA> <suffixB> [...] -- <bench args>
# (variants built with tools/build_variant.sh).  The production library is backed up first and restored on ANY exit.
cd $GRAFT_REPO_ROOT/ivfadc.jl_amd/csrc || exit 1
V=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do V+=("$1"); shift; done
if [ $# -eq 0 ] || [ ${#V[@]} -eq 0 ]; then echo "usage: tools/ab_lib.sh <suffix> [<suffix> ...] -- <bench args>"; exit 2; fi
shift
cp -p libivfadc_hip.so libivfadc_hip.so.ab_backup || exit 1
trap 'cp -p libivfadc_hip.so.ab_backup libivfadc_hip.so; rm -f libivfadc_hip.so.ab_backup' EXIT
for rep in 1 2; do for v in "${V[@]}"; do cp -p libivfadc_hip_$v.so libivfadc_hip.so; touch libivfadc_hip.so; (cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --no-cpu-baseline --no-sweep "$@" 2>&1 | grep -oE "\"value\": [0-9.]+|scan_ms_per_launch\": [0-9.]+|coarse_ms_per_launch\": [0-9.]+" | tr '\n' ' '; echo " <- $v"); done; done
