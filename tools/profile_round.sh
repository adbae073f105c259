#!/bin/bash
# Round profile of the bench command (config B): kernel-trace stats, then HBM traffic counters in separate passes.
# usage: bash tools/profile_round.sh <tag>     (writes gpurun_out/<tag>/...)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-prof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json.log 2>&1
grep '^{' $OUT/bench.json.log > $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for name in ('fetch', 'write'):
    f = glob.glob(out + '/' + name + '/*/*_counter_collection.csv')[0]
    tot = 0.0
    n = 0
    for r in csv.DictReader(open(f)):
        if 'pairhmm' in r['Kernel_Name']:
            tot += float(r['Counter_Value']); n += 1
    res[name + '_kb_per_launch'] = tot
st = glob.glob(out + '/stats/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(st)):
    if 'pairhmm' in r['Name']:
        res['kernel'] = r['Name']; res['calls'] = int(r['Calls']); res['avg_ms'] = float(r['AverageNs']) / 1e6
# MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a
# streaming read -> doubled; WRITE_SIZE is exact for streaming stores.
res['hbm_bytes_per_launch'] = (2 * res['fetch_kb_per_launch'] + res['write_kb_per_launch']) * 1024
json.dump(res, open(out + '/traffic.json', 'w'), indent=1)
print(json.dumps(res))
PY
