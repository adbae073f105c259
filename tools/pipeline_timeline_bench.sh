#!/bin/bash
# diagnostic: kernel + copy timeline (rocprofv3 --kernel-trace --memory-copy-trace) of bench.py's end-to-end pipeline for one
# config: when does each sweep / table build / copy start and end.  usage: bash tools/pipeline_timeline_bench.sh <config> [depth]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-4}
DEPTH=${2:-0}
OUT=$R/gpurun_out/timeline_bench_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --e2e-batches 9 --e2e-depth $DEPTH > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for r in csv.DictReader(open(glob.glob(out + '/t/*/*_kernel_trace.csv')[0])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]))
mc = glob.glob(out + '/t/*/*_memory_copy_trace.csv')
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
t0 = rows[0][0]
big = [r for r in rows if (r[1] - r[0]) > 1e6]
for s, e, n in big[-70:]:
    print("%9.2f -> %9.2f ms (%7.2f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
