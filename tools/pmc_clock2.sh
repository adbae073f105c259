#!/bin/bash
# diagnostic: effective shader clock (GRBM_GUI_ACTIVE per XCD / kernel time) of the DP kernels for a python tool run.
# usage: [env] tools/pmc_clock2.sh <tag> -- <script> <args>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift 2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/$1 ${@:2} > $OUT.a.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
cc = glob.glob(out + '/a/*/*_counter_collection.csv')[0]
kt = glob.glob(out + '/a/*/*_kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e9)
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    name, s = dur.get(r['Dispatch_Id'], ('?', 0))
    if s > 0.002 and 'pairhmm' in name:
        print(name[:48], 'dur %.2f ms' % (s * 1e3), 'clock %.2f GHz' % (float(r['Counter_Value']) / 8 / s / 1e9))
PY
