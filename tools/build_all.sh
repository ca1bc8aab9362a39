#!/bin/bash
# tools/build_all.sh: production library + the diagnostic variant (-DIVFADC_DEBUG, csrc/libivfadc_hip_dbg.so); fails loudly
set -e
cd "$(dirname "$0")/../ivfadc.jl_amd/csrc"
FL="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17"
/opt/rocm/bin/hipcc $FL -o libivfadc_hip.so.tmp ivfadc_hip.hip 2>&1 | grep -E "error" -A4 || true
test -s libivfadc_hip.so.tmp && mv libivfadc_hip.so.tmp libivfadc_hip.so
if [ "$1" = "dbg" ]; then
  /opt/rocm/bin/hipcc $FL -DIVFADC_DEBUG -o libivfadc_hip_dbg.so.tmp ivfadc_hip.hip 2>&1 | grep -E "error" -A4 || true
  test -s libivfadc_hip_dbg.so.tmp && mv libivfadc_hip_dbg.so.tmp libivfadc_hip_dbg.so
fi
ls -la --time-style=+%T libivfadc_hip*.so
