#!/bin/bash
# tools/w8_quick.sh [variant suffixes...]: the eight-wave kernel's parity tests, the three SIFT1B regimes per library, then the W8_PROF counters
# (if csrc/libivfadc_hip_prof.so exists)
timeout -k 10 600 python3 -m pytest tests/test_gpu_wg8.py -x -q -m gpu 2>&1 | tail -2 || exit 1
bash tools/ab3.sh "$@" || exit 1
if [ -f ivfadc.jl_amd/csrc/libivfadc_hip_prof.so ]; then
  cp -p ivfadc.jl_amd/csrc/libivfadc_hip_prof.so ivfadc.jl_amd/csrc/libivfadc_hip.so
  for extra in "--w 1" "" "--nq 2048"; do
    timeout -k 10 300 python3 bench.py --config sift1b --table-mode ${TM:-6} --single-mode --no-cpu-baseline --steps 10 --warmup 3 --windows 1 $extra 2>&1 | grep -E "w8prof" | tail -1
  done
fi
