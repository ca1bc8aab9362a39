cd $GRAFT_REPO_ROOT/ivfadc.jl_amd/csrc
for w in 1 4; do cp libivfadc_hip_w$w.so libivfadc_hip.so; touch libivfadc_hip.so; echo "MINW=$w"; (cd $GRAFT_REPO_ROOT; for pg in 1 2; do IVFADC_FORCE_PG=$pg timeout 300 python bench.py --no-cpu-baseline --steps 100 2>&1 | grep -oE "\"value\": [0-9.]+|scan_ms_per_launch\": [0-9.]+" | tr '\n' ' '; echo " pg=$pg"; done); done
