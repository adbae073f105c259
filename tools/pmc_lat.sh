#!/bin/bash
# diagnostic: average LDS / VMEM latency of the sweep kernel = SQ_INST_LEVEL_x / SQ_INSTS_x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_lat
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
run() { n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 1 --warmup 0 --pairs 4000 --no-cpu-baseline --no-e2e > $OUT.$n.log 2>&1 || { tail -5 $OUT.$n.log; return 1; }
}
run a SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES && \
run b SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES && \
run c SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for d in ('a', 'b', 'c'):
    fs = glob.glob(out + '/' + d + '/*/*_counter_collection.csv')
    if not fs: continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if 'pairhmm' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    print(d, {k: '%.4g' % v for k, v in sorted(agg.items())})
PY
