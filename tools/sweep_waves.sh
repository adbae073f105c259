#!/bin/bash
# diagnostic: throughput of config B (5376 pairs) as a function of resident waves per CU
for c in "$@"; do
  CPECAN_MAX_WAVES_PER_CU=$c timeout -k 10 200 python bench.py --steps 2 --warmup 1 --pairs 5376 --no-cpu-baseline 2>/dev/null > /tmp/sw.json || exit 1
  python - "$c" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print("cap", sys.argv[1], "waves", d["e2e"]["waves"], "cells/s %.3e" % d["value"], "ms", round(d["ms_per_step"], 1))
PY
done
