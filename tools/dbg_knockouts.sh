#!/bin/bash
# tools/dbg_knockouts.sh <tag> <bench args...>: the diagnostic library (libivfadc_hip_dbg.so) in place of the production one, one bench run
# per IVFADC_DEBUG_FLAGS value (0 = everything, 1 = candidates dropped, 2 = table build only, 4 = no table build, 5 = fast path without build)
cd $GRAFT_REPO_ROOT/ivfadc.jl_amd/csrc || exit 1
tag=$1; shift
cp -p libivfadc_hip.so libivfadc_hip.so.ab_backup || exit 1
trap 'cp -p libivfadc_hip.so.ab_backup libivfadc_hip.so; rm -f libivfadc_hip.so.ab_backup' EXIT
cp -p libivfadc_hip_dbg.so libivfadc_hip.so; touch libivfadc_hip.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$tag
for f in 0 1 2 4 5; do
  IVFADC_DEBUG_FLAGS=$f timeout -k 10 300 python3 bench.py --single-mode --no-cpu-baseline --steps 10 --warmup 3 --windows 1 "$@" 2>gpurun_out/$tag/f$f.err | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags $f: ms/step', d['ms_per_step'], 'scan', d['roofline']['scan_ms_per_launch'])"
done
