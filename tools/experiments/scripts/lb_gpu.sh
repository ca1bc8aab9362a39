#!/bin/bash
# on the GPU box: LB parity tests, probe (production library), then the phase stamps of the diagnostic library
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core" > gpurun_out/lb_test.log 2>&1; tail -2 gpurun_out/lb_test.log
timeout -k 10 300 python tools/lb_probe.py > gpurun_out/lb_probe.txt 2>&1; grep table_mode gpurun_out/lb_probe.txt
cd ivfadc.jl_amd/csrc && cp libivfadc_hip_dbg.so libivfadc_hip.so && cd ../..
IVFADC_DEBUG_STAMPS=1 timeout -k 10 300 python tools/lb_probe.py 4096 8 > gpurun_out/lb_stamps.txt 2>&1; grep -v "^\[bench" gpurun_out/lb_stamps.txt | sed -n 2,5p
