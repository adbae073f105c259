#!/bin/bash
# usage (on the GPU box): tools/ab_kstats.sh "<bench args>" tagA tagB ...  -> per-kernel average ms (rocprofv3 --kernel-trace
# --stats) of bench.py with build_ab/<tag>.so, each twice, interleaved.  A tag "VAR=x:tag" also sets an environment variable.
R=${GRAFT_REPO_ROOT:-$(pwd)}
args=$1; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for t in "$@"; do
  kv=""; tag=$t
  case $t in *:*) kv=${t%%:*}; tag=${t##*:};; esac
  lib=$R/build_ab/$tag.so; [ "$tag" = default ] && lib=$R/cpecan_amd/libcpecan_hip.so
  rm -rf /tmp/ks_$tag
  [ -n "$kv" ] && export $kv
  CPECAN_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs $args > /tmp/ks_$tag.log 2>&1 || { echo "$t failed"; tail -3 /tmp/ks_$tag.log; exit 1; }
  [ -n "$kv" ] && unset ${kv%%=*}
  python3 - "$t" /tmp/ks_$tag <<'PY'
import csv, glob, sys, re
tag, out = sys.argv[1:3]
st = glob.glob(out + '/*/*_kernel_stats.csv')[0]
parts = []
tot = 0.0
for r in csv.DictReader(open(st)):
    if 'pairhmm' in r['Name']:
        m = re.search(r"<([^>]*)>", r['Name'])
        ms = float(r['AverageNs']) / 1e6
        tot += ms
        parts.append("<%s> %.3f" % (m.group(1) if m else '?', ms))
print(tag, "sum %.3f ms |" % tot, " | ".join(parts), flush=True)
PY
done; done
