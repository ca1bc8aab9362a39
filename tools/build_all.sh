#!/bin/bash
# tools/build_all.sh: production library + (with "dbg") the diagnostic variant (-DIVFADC_DEBUG, csrc/libivfadc_hip_dbg.so).
# Fails loudly: a compile error ends the script with hipcc's status and leaves the previous library untouched.
set -euo pipefail
cd "$(dirname "$0")/../ivfadc.jl_amd/csrc"
FL="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17"
build() {   # $1 = output, rest = extra flags
  local out="$1"; shift
  rm -f "$out.tmp"
  if ! /opt/rocm/bin/hipcc $FL "$@" -o "$out.tmp" ivfadc_hip.hip 2>build.log; then
    grep -E "error" -A4 build.log || cat build.log
    rm -f "$out.tmp"
    echo "build_all.sh: hipcc failed for $out (previous library left in place)" >&2
    exit 1
  fi
  mv "$out.tmp" "$out"
}
build libivfadc_hip.so
if [ "${1:-}" = "dbg" ]; then
  build libivfadc_hip_dbg.so -DIVFADC_DEBUG
fi
ls -la --time-style=+%T libivfadc_hip*.so
