#!/bin/bash
# diagnostic: effective shader clock during the sweep kernel and during the logAdd microbenchmark
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_clock
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e > $OUT.a.log 2>&1
timeout -k 10 100 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $R/tools/logadd_rate > $OUT.b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for d in ('a', 'b'):
    cc = glob.glob(out + '/' + d + '/*/*_counter_collection.csv')[0]
    kt = glob.glob(out + '/' + d + '/*/*_kernel_trace.csv')[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r['Dispatch_Id']] = (r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e9)
    for r in csv.DictReader(open(cc)):
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        name, s = dur.get(r['Dispatch_Id'], ('?', 0))
        if s > 0.01 and ('pairhmm' in name or 'Pdi' in name or ' k<' in name or name.startswith('void k')):
            print(d, name[:40], 'dur %.1f ms' % (s * 1e3), 'clock %.2f GHz' % (float(r['Counter_Value']) / 8 / s / 1e9))
PY
