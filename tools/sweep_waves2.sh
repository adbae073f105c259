#!/bin/bash
# diagnostic: throughput with exactly 2 pairs per wave at each residency (pairs = 2 * 256 * cap)
for c in "$@"; do
  CPECAN_MAX_WAVES_PER_CU=$c timeout -k 10 200 python bench.py --steps 2 --warmup 1 --pairs $((512*c)) --no-cpu-baseline 2>/dev/null > /tmp/sw.json || exit 1
  python - "$c" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print("cap", sys.argv[1], "waves", d["e2e"]["waves"], "cells/s %.3e" % d["value"], "ms", round(d["ms_per_step"], 1))
PY
done
