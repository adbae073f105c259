#!/bin/bash
# diagnostic: kernel time of config B (4096 pairs, 8 waves/CU) with parts of the sweep disabled (CPECAN_DEBUG_SKIP bits:
# 2 no traceback, 4 no ring stores, 8 no refresh work, 16 no F prefetch loads, 32 no candidates).  Results are wrong with
# any bit set; timing only.
for skip in "$@"; do
  CPECAN_DEBUG_SKIP=$skip timeout -k 10 200 python bench.py --steps 3 --warmup 1 --pairs 4096 --no-cpu-baseline 2>/dev/null > /tmp/sw.json || { echo "skip $skip failed"; tail -3 /tmp/sw.json; continue; }
  python - "$skip" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print("skip", sys.argv[1], "waves", d["e2e"]["waves"], "kernel ms", round(d["ms_per_step"], 2), flush=True)
PY
done
