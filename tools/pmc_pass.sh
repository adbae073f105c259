#!/bin/bash
# diagnostic: SQ counters of the sweep kernel on 4000 config-B pairs (separate rocprofv3 passes, no tracing domains)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc}
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 1 --warmup 0 --pairs 4000 --no-cpu-baseline --no-e2e > $OUT.$n.log 2>&1
}
mkdir -p $OUT
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for d in ('a', 'b'):
    f = glob.glob(out + '/' + d + '/*/*_counter_collection.csv')[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'pairhmm' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    kt = glob.glob(out + '/' + d + '/*/*_kernel_trace.csv')[0]
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in csv.DictReader(open(kt)) if 'pairhmm' in r['Kernel_Name']]
    print(d, 'kernel ms', dur, {k: '%.4g' % v for k, v in sorted(agg.items())})
PY
