#!/bin/bash
# config 4 end to end by the number of batches in flight
for rep in 1 2; do for dp in 4 5 6; do
  timeout -k 10 300 python bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --e2e-batches 17 --e2e-depth $dp 2>/dev/null > /tmp/e4.json || { echo "$dp failed"; exit 1; }
  python - "$dp" <<'PY'
import sys, json
d = json.loads(open('/tmp/e4.json').read().strip().splitlines()[-1])
e = d["e2e"]
print("depth", sys.argv[1], "kernel ms %.2f" % d["ms_per_step"], "e2e cells/s %.3e" % d["value_e2e"], "steady ms %.1f total ms %.1f" % (1e3 * e["pipelined_steady_s_per_batch"], 1e3 * e["pipelined_total_s_per_batch"]), "device GB %.1f" % (e["device_bytes"] / 1e9), flush=True)
PY
done; done
