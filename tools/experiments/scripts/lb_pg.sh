#!/bin/bash
# on the GPU box: LB rounds with 1..4 probes per round (IVFADC_FORCE_PG), production library
for pg in 4 3 2; do echo "PG=$pg"; IVFADC_FORCE_PG=$pg timeout -k 10 300 python tools/lb_probe.py 2>&1 | grep "table_mode=0"; done
