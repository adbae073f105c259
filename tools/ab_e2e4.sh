#!/bin/bash
# config 4 end to end under environment settings: tools/ab_e2e4.sh VAR=a VAR=b ...
for rep in 1 2; do for kv in "$@"; do
  env $kv timeout -k 10 300 python bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --e2e-batches 13 2>/dev/null > /tmp/e4.json || { echo "$kv failed"; exit 1; }
  python - "$kv" <<'PY'
import sys, json
d = json.loads(open('/tmp/e4.json').read().strip().splitlines()[-1])
e = d["e2e"]
print(sys.argv[1], "kernel ms %.2f" % d["ms_per_step"], "e2e cells/s %.3e" % d["value_e2e"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, (list, dict))}, flush=True)
PY
done; done
