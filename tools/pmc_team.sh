#!/bin/bash
# diagnostic: SQ counters of the team kernel on tools/sparse_bench.py 4000 2000 400 (one rocprofv3 --pmc pass, no trace domains)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_team
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 $R/tools/sparse_bench.py 4000 2000 400 > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -- python3 $R/tools/sparse_bench.py 4000 2000 400 > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for d in ('a', 'b'):
    f = glob.glob(out + '/' + d + '/*/*_counter_collection.csv')[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'pairhmm' in r['Kernel_Name']:
            key = (r['Kernel_Name'][:40], r['Counter_Name'])
            agg[key] = agg.get(key, 0) + float(r['Counter_Value'])
    for k in sorted(agg):
        print(d, k[0], k[1], '%.4g' % agg[k])
PY
