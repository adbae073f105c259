#!/bin/bash
# timing build without the symbol fetch on the forward chain (results invalid) on the under-subscribed launches
for rep in 1 2; do for tag in base nosym; do
  for args in "--config A" "--config B --pairs 1250"; do
  CPECAN_LIB=$PWD/build_ab/$tag.so timeout -k 10 200 python bench.py --steps 6 --warmup 2 $args --no-cpu-baseline --no-other-configs --no-e2e 2>/dev/null > /tmp/sw.json || { echo "$tag failed"; exit 1; }
  python - "$tag" "$args" <<'PY'
import sys, json
d = json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "kernel ms %.3f" % d["ms_per_step"], flush=True)
PY
done; done; done
