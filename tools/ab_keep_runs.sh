#!/bin/bash
# config 4: a batch that keeps its anchors as runs (round 4) against one that expands them on the host (CPECAN_KEEP_RUNS=0): host stages
for v in 0 1; do echo "== CPECAN_KEEP_RUNS=$v"; CPECAN_KEEP_RUNS=$v CPECAN_TRACE_HOST=1 timeout -k 10 300 python tools/e2e_stages.py 50000 4 2>&1 | grep -v hipMalloc | grep "add \|cpecan upload" | sed -n 5,12p | cut -c1-200; done
