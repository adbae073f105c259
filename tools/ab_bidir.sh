#!/bin/bash
# forward and backward sweeps side by side (CPECAN_BIDIR=1) against the library's own choice: config A and the shards of config B
CPECAN_BIDIR=1 CPECAN_TRACE_HOST=1 python bench.py --config A --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs 2>&1 | grep "cpecan class" | head -2
bash tools/ab_kstats.sh "--config A" CPECAN_BIDIR=0:default CPECAN_BIDIR=1:default
for p in 1250 2500; do bash tools/ab_kstats.sh "--config B --pairs $p" CPECAN_BIDIR=0:default CPECAN_BIDIR=1:default CPECAN_SPLIT=1:default; done
