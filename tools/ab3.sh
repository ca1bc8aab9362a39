#!/bin/bash
# tools/ab3.sh <variant suffixes...>: each variant library in turn on the three SIFT1B regimes (scan ms per launch)
cd $GRAFT_REPO_ROOT/ivfadc.jl_amd/csrc || exit 1
cp -p libivfadc_hip.so /tmp/ab3_backup.so || exit 1
trap 'cp -p /tmp/ab3_backup.so libivfadc_hip.so' EXIT
for v in prod "$@"; do
  if [ $v = prod ]; then cp -p /tmp/ab3_backup.so libivfadc_hip.so; else cp -p libivfadc_hip_$v.so libivfadc_hip.so; fi
  touch libivfadc_hip.so
  line="$v:"
  for extra in "" "--w 1" "--nq 2048"; do
    r=$(cd $GRAFT_REPO_ROOT; timeout -k 10 300 python3 bench.py --config sift1b --table-mode ${TM:-6} --single-mode --no-cpu-baseline --steps 10 --warmup 3 --windows 1 $extra 2>/dev/null | grep -oE "scan_ms_per_launch\": ?[0-9.]+" | head -1 | grep -oE "[0-9.]+$")
    line="$line  $r"
  done
  echo "$line"
done
