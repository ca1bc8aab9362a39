#!/bin/bash
# unprofiled ivfadc_search_batches rates under a few settings (GPU box): tools/batches_matrix.sh > gpurun_out/...
# CROWD=n: the process already owns n torch streams with work on them (hardware-queue sharing); the stream probe can be switched off for comparison
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python3 tools/batches_trace.py 150 16 2>&1 | grep -A1 'ivfadc_search_batches,'; }
for crowd in 0 3 6; do
run CROWD=$crowd
run CROWD=$crowd IVFADC_NO_STREAM_PROBE=1
run CROWD=$crowd IVFADC_NO_PIPELINE=1
run CROWD=$crowd IVFADC_HOST_LEGACY=1
done
