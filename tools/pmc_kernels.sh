#!/bin/bash
# diagnostic: per-KERNEL (forward launch / item launch of a split class) times and SQ counters of one bench.py launch.
# usage: [ENV=..] tools/pmc_kernels.sh <outtag> <bench args...>     -> gpurun_out/<outtag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-e2e --no-other-configs $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 3 --warmup 1 > $OUT/stats.log 2>&1
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $B --steps 1 --warmup 0 > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; return 1; }
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU && \
run b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT && \
run c SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_IFETCH
python3 - $OUT > $R/gpurun_out/$TAG.txt <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
def short(n):
    m = re.search(r"cpecan_pairhmm_\w+<([^>]*)>", n)
    return (m.group(0) if m else n)[:70]
st = glob.glob(out + '/stats/*/*_kernel_stats.csv')
if st:
    for r in csv.DictReader(open(st[0])):
        if 'pairhmm' in r['Name']:
            print('stats', short(r['Name']), 'calls', r['Calls'], 'avg ms %.3f' % (float(r['AverageNs']) / 1e6))
for d in 'abc':
    fs = glob.glob(out + '/' + d + '/*/*_counter_collection.csv')
    if not fs: continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if 'pairhmm' in r['Kernel_Name']:
            k = short(r['Kernel_Name'])
            agg.setdefault(k, {})
            agg[k][r['Counter_Name']] = agg[k].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    for k, v in agg.items():
        print(d, k, {c: '%.4g' % x for c, x in sorted(v.items())})
PY
cat $R/gpurun_out/$TAG.txt
