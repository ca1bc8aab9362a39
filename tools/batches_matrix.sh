#!/bin/bash
# unprofiled ivfadc_search_batches rates under a few settings (GPU box): tools/batches_matrix.sh > gpurun_out/...
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python3 tools/batches_trace.py 150 16 2>&1 | grep -A1 'ivfadc_search_batches,'; }
for crowd in 0 3 6; do
run CROWD=$crowd
run CROWD=$crowd IVFADC_COPY_PRIO=-1
run CROWD=$crowd IVFADC_COPY_PRIO=-1 IVFADC_LANE2_PRIO=1
run CROWD=$crowd IVFADC_COPY_PRIO=-1 IVFADC_LANE2_PRIO=-1
run CROWD=$crowd IVFADC_COPY_PRIO=1 IVFADC_LANE2_PRIO=-1
done
