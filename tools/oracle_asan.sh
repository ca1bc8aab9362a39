#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (CPU only; GPU ASan is not available on this pool): copies the oracle,
# the tests and the package to a scratch directory, rebuilds liboracle.so with the sanitizers and runs the oracle and
# golden-fixture tests against it.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cp -r "$root/oracle" "$root/tests" "$root/ivfadc.jl_amd" "$root/ivfadc_jl_amd.py" "$tmp/"
cd "$tmp/oracle"
gcc -O1 -g -std=c11 -ffp-contract=off -fopenmp -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o liboracle.so ivfadc_oracle.c -lm
cd "$tmp"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle.py tests/test_golden.py -q -m "not gpu" -p no:cacheprovider
rm -rf "$tmp"
