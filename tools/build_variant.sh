#!/bin/bash
# tools/build_variant.sh <suffix> [extra hipcc flags]: builds csrc/libivfadc_hip_<suffix>.so (for tools/ab_lib.sh)
cd "$(dirname "$0")/../ivfadc.jl_amd/csrc"
sfx=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 "$@" -o libivfadc_hip_$sfx.so ivfadc_hip.hip 2>&1 | grep -E "error|warning: loop" | head
ls -la libivfadc_hip_$sfx.so
