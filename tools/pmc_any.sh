#!/bin/bash
# diagnostic: SQ counter passes over any python tool.  usage: tools/pmc_any.sh <outtag> <kernel-substring> -- <python script and args>
# (environment is inherited: FWD_SCALE_ONLY=forward FWD_SCALE_EXPANSION=40 tools/pmc_any.sh fwd12 pairhmm -- tools/fwd_scale.py 12)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; KSUB=$2; shift 3
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/$SCRIPT $ARGS > $OUT.$n.log 2>&1 || { tail -5 $OUT.$n.log; return 1; }
}
SCRIPT=$1; shift; ARGS="$@"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU && \
run b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT && \
run c SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 && \
run d SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_CYCLES
python3 - $OUT $KSUB <<'PY'
import csv, glob, sys
out, ksub = sys.argv[1], sys.argv[2]
for d in 'abcd':
    fs = glob.glob(out + '/' + d + '/*/*_counter_collection.csv')
    if not fs: continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if ksub in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    kt = glob.glob(out + '/' + d + '/*/*_kernel_trace.csv')[0]
    dur = [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 2) for r in csv.DictReader(open(kt)) if ksub in r['Kernel_Name']]
    print(d, 'kernel ms', dur, {k: '%.4g' % v for k, v in sorted(agg.items())})
PY
