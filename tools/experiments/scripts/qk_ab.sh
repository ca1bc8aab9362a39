#!/bin/bash
# on the GPU box: A/B of library variants (tools/build_variant.sh) around the quarter-key bound of the LDS selectors
# base = without, qk = quarter keys from every selector, qk2 = from the LDS selectors (K > 64) only
set -e
cd $GRAFT_REPO_ROOT
{
echo "== sift1m single-mode K=10"; tools/ab_lib.sh base qk2 -- --steps 300 --warmup 30 --single-mode --no-other-configs --no-host-to-host
echo "== sift1m two lanes K=10"; tools/ab_lib.sh base qk2 -- --steps 300 --warmup 30 --no-other-configs --no-host-to-host
echo "== sift1m single K=100"; tools/ab_lib.sh base qk qk2 -- --steps 100 --warmup 10 --single-mode --no-other-configs --no-host-to-host --K 100
echo "== sift1m single K=1000"; tools/ab_lib.sh base qk2 -- --steps 30 --warmup 5 --single-mode --no-other-configs --no-host-to-host --K 1000
echo "== deep1b K=100"; tools/ab_lib.sh base qk2 -- --config deep1b --steps 10 --warmup 2 --no-other-configs --no-host-to-host --K 100
echo "== hd K=100"; tools/ab_lib.sh base qk2 -- --config hd --steps 10 --warmup 2 --no-other-configs --no-host-to-host --K 100
echo "== sift1b w=8 K=100"; tools/ab_lib.sh base qk2 -- --config sift1b --steps 5 --warmup 2 --no-other-configs --no-host-to-host --K 100
echo "== sift1b w=8"; tools/ab_lib.sh base qk2 -- --config sift1b --steps 8 --warmup 2 --no-other-configs --no-host-to-host
} > gpurun_out/qk2_ab.txt 2>&1
cat gpurun_out/qk2_ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/qk2_gpu_tests.txt 2>&1 || { tail -30 gpurun_out/qk2_gpu_tests.txt; exit 1; }
tail -3 gpurun_out/qk2_gpu_tests.txt
