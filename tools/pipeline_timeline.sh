#!/bin/bash
# diagnostic: kernel timeline (rocprofv3 --kernel-trace) of tools/e2e_stages.py: when does each sweep start and end,
# what runs beside it.  usage: bash tools/pipeline_timeline.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $R/tools/e2e_stages.py > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for r in csv.DictReader(open(glob.glob(out + '/t/*/*_kernel_trace.csv')[0])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
mc = glob.glob(out + '/t/*/*_memory_copy_trace.csv')
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') ))
rows.sort()
t0 = rows[0][0]
big = [r for r in rows if (r[1] - r[0]) > 1e6 or 'pairhmm' in r[2]]
for s, e, n in big[-60:]:
    print("%9.2f -> %9.2f ms (%7.2f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
