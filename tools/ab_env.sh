#!/bin/bash
# usage (on the GPU box): tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...  -> kernel ms of bench.py under each environment
# setting, twice, interleaved (variants are only comparable within one gpurun call: boxes differ by several % in clock)
args=$1; shift
for rep in 1 2; do for kv in "$@"; do
  env $kv timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs $args 2>/dev/null > /tmp/sw.json || { echo "$kv failed"; exit 1; }
  python - "$kv" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print(sys.argv[1], "waves", d["e2e"]["waves"], "kernel ms %.3f" % d["ms_per_step"], "cells/s %.3e" % d["value"], flush=True)
PY
done; done
