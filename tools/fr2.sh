#!/bin/bash
# on the GPU box: more shapes for the A/B of tools/fr_ab.sh (variants base, fr)
cd $GRAFT_REPO_ROOT
{
echo "== sift1m two lanes"; tools/ab_lib.sh base fr -- --steps 300 --warmup 30 --no-other-configs --no-host-to-host
echo "== sift1m lowrank two lanes"; tools/ab_lib.sh base fr -- --steps 300 --warmup 30 --no-other-configs --no-host-to-host --data lowrank
echo "== sift1m single lane (riders)"; tools/ab_lib.sh base fr -- --steps 300 --warmup 30 --single-mode --no-other-configs --no-host-to-host
echo "== sift1m w=32 two lanes"; tools/ab_lib.sh base fr -- --steps 200 --warmup 20 --no-other-configs --no-host-to-host --w 32
echo "== sift1m w=1 two lanes"; tools/ab_lib.sh base fr -- --steps 200 --warmup 20 --no-other-configs --no-host-to-host --w 1
echo "== sift1m K=100 single"; tools/ab_lib.sh base fr -- --steps 100 --warmup 10 --single-mode --no-other-configs --no-host-to-host --K 100
echo "== deep1b"; tools/ab_lib.sh base fr -- --config deep1b --steps 10 --warmup 2 --no-other-configs --no-host-to-host
echo "== deep1b w=3"; tools/ab_lib.sh base fr -- --config deep1b --w 3 --steps 20 --warmup 2 --no-other-configs --no-host-to-host
} > gpurun_out/fr3_ab.txt 2>&1
cat gpurun_out/fr3_ab.txt
