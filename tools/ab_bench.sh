#!/bin/bash
# usage (on the GPU box): tools/ab_bench.sh pairs tagA tagB ...  -> kernel ms of bench.py for each variant, twice, interleaved
pairs=$1; shift
for rep in 1 2; do for tag in "$@"; do
  CPECAN_LIB=$PWD/build_ab/$tag.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --pairs $pairs --no-cpu-baseline --no-other-configs --no-e2e 2>/dev/null > /tmp/sw.json || { echo "$tag failed"; exit 1; }
  python - "$tag" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print(sys.argv[1], "waves", d["e2e"]["waves"], "kernel ms %.2f" % d["ms_per_step"], "cells/s %.3e" % d["value"], flush=True)
PY
done; done
