#!/bin/bash
# on the GPU box: per-phase cycle stamps of the query-major kernel on a configuration (diagnostic build copied over the library of the box's snapshot)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
cp -p ivfadc.jl_amd/csrc/libivfadc_hip.so /tmp/lib_backup.so
cp ivfadc.jl_amd/csrc/libivfadc_hip_dbg.so ivfadc.jl_amd/csrc/libivfadc_hip.so
IVFADC_DEBUG_STAMPS=1 timeout -k 10 600 python bench.py --config ${1:-deep1b} --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/stamps_${1:-deep1b}.json 2> gpurun_out/stamps_${1:-deep1b}.txt
cp -p /tmp/lib_backup.so ivfadc.jl_amd/csrc/libivfadc_hip.so
grep "ivfadc stamps" gpurun_out/stamps_${1:-deep1b}.txt | tail -6
