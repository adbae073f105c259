#!/bin/bash
# diagnostic: kernel time with / without the traceback, at 1 and 7 waves per CU
for cap in 1 8; do for skip in 0 2; do
  CPECAN_MAX_WAVES_PER_CU=$cap CPECAN_DEBUG_SKIP=$skip timeout -k 10 200 python bench.py --steps 2 --warmup 1 --pairs 1792 --no-cpu-baseline 2>/dev/null > /tmp/sw.json || exit 1
  python - "$cap" "$skip" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print("cap", sys.argv[1], "skip", sys.argv[2], "waves", d["e2e"]["waves"], "ms", round(d["ms_per_step"], 1))
PY
done; done
