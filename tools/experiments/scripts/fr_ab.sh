#!/bin/bash
# on the GPU box: A/B of the "closest cell alone first" rounds of the query-major kernel (variants from tools/build_variant.sh: base, fr)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/fr_gpu_tests.txt 2>&1 || { tail -30 gpurun_out/fr_gpu_tests.txt; exit 1; }
tail -2 gpurun_out/fr_gpu_tests.txt
{
echo "== sift1m two lanes"; tools/ab_lib.sh base fr -- --steps 300 --warmup 30 --no-other-configs --no-host-to-host
echo "== sift1m single lane (riders)"; tools/ab_lib.sh base fr -- --steps 300 --warmup 30 --single-mode --no-other-configs --no-host-to-host
echo "== sift1m lowrank two lanes"; tools/ab_lib.sh base fr -- --steps 300 --warmup 30 --no-other-configs --no-host-to-host --data lowrank
echo "== sift1m w=32 two lanes"; tools/ab_lib.sh base fr -- --steps 200 --warmup 20 --no-other-configs --no-host-to-host --w 32
echo "== deep1b"; tools/ab_lib.sh base fr -- --config deep1b --steps 10 --warmup 2 --no-other-configs --no-host-to-host
echo "== deep1b w=1"; tools/ab_lib.sh base fr -- --config deep1b --w 1 --steps 20 --warmup 2 --no-other-configs --no-host-to-host
echo "== hd"; tools/ab_lib.sh base fr -- --config hd --steps 10 --warmup 2 --no-other-configs --no-host-to-host
} > gpurun_out/fr_ab.txt 2>&1
cat gpurun_out/fr_ab.txt
