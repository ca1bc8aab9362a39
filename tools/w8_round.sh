#!/bin/bash
# tools/w8_round.sh <tag>: parity tests of the eight-wave kernel, the three SIFT1B regimes, then the W8_PROF variant's per-wave counters
tag=${1:-w8}
timeout -k 10 600 python3 -m pytest tests/test_gpu_wg8.py -x -q -m gpu 2>&1 | tail -3 || exit 1
tools/sift1b_ab.sh $tag || exit 1
if [ -f ivfadc.jl_amd/csrc/libivfadc_hip_prof.so ]; then
  cp -p ivfadc.jl_amd/csrc/libivfadc_hip.so /tmp/keep_prod.so
  cp -p ivfadc.jl_amd/csrc/libivfadc_hip_prof.so ivfadc.jl_amd/csrc/libivfadc_hip.so
  for extra in "" "--w 1" "--nq 2048"; do
    timeout -k 10 300 python3 bench.py --config sift1b --table-mode ${TM:-6} --single-mode --no-cpu-baseline --steps 10 --warmup 3 --windows 1 $extra 2>&1 | grep w8prof | head -1
  done
  cp -p /tmp/keep_prod.so ivfadc.jl_amd/csrc/libivfadc_hip.so
fi
