#!/bin/bash
# Round profile of the bench command: the bench line, rocprofv3 kernel-trace stats, HBM traffic counters and the vector /
# LDS instruction counters, each in a pass of its own (rocprofv3 --pmc never together with other trace domains).
# usage: bash tools/profile_round.sh <tag> [config]     (writes gpurun_out/<tag>/...; config B by default)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-prof}
CFG=${2:-B}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-other-configs > $OUT/bench.json.log 2>&1
grep '^{' $OUT/bench.json.log > $OUT/bench.json
B="python3 $R/bench.py --config $CFG --no-cpu-baseline --no-e2e --no-other-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 3 --warmup 1 > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 1 --warmup 0 > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/valu -- $B --steps 1 --warmup 0 > $OUT/valu.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/lds -- $B --steps 1 --warmup 0 > $OUT/lds.log 2>&1
python3 - $OUT $R $CFG <<'PY'
import csv, glob, json, sys
out, root, cfg = sys.argv[1:4]
sys.path.insert(0, root)
import bench
line = json.load(open(out + '/bench.json'))
pairs = line['config']['pairs_total']
cells = line['config']['cells_total']
def counters(name):
    f = glob.glob(out + '/' + name + '/*/*_counter_collection.csv')[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'pairhmm' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return agg
res = {'config': cfg, 'pairs': pairs, 'cells': cells, 'kernel_source_hash': bench.kernel_source_hash()}
st = glob.glob(out + '/stats/*/*_kernel_stats.csv')[0]
ks = []
for r in csv.DictReader(open(st)):
    if 'pairhmm' in r['Name']:
        ks.append({'kernel': r['Name'], 'calls': int(r['Calls']), 'avg_ms': float(r['AverageNs']) / 1e6})
res['kernels'] = ks
# MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a
# streaming read -> doubled; WRITE_SIZE is exact for streaming stores.  One launch each (--steps 1 --warmup 0).
fetch, write = counters('fetch'), counters('write')
res['fetch_kb_per_launch'] = fetch.get('FETCH_SIZE', 0.0)
res['write_kb_per_launch'] = write.get('WRITE_SIZE', 0.0)
res['hbm_bytes_per_launch'] = (2 * res['fetch_kb_per_launch'] + res['write_kb_per_launch']) * 1024
json.dump(res, open(out + '/traffic.json', 'w'), indent=1)
# vector-instruction and LDS ceilings from the counters of ONE launch, priced with the issue rates measured on this chip
# (profiles/r02_valu_rate.txt, >= 2 waves per SIMD: fp64 add / mul / fma 4.3 cycles, 32-bit 2.4; MI355X_MICROARCH.md:
# 256 CUs x 4 SIMDs at 2.4 GHz, one LDS array per CU)
v, l = counters('valu'), counters('lds')
fp64 = v.get('SQ_INSTS_VALU_FMA_F64', 0) + v.get('SQ_INSTS_VALU_ADD_F64', 0) + v.get('SQ_INSTS_VALU_MUL_F64', 0)
valu = v.get('SQ_INSTS_VALU', 0)
valu_cycles = fp64 * 4.3 + max(0.0, valu - fp64) * 2.4
clock, simds, cus = 2.4e9, 1024, 256
comp = {'config': cfg, 'pairs': pairs, 'cells': cells, 'kernel_source_hash': bench.kernel_source_hash(),
        'source': 'rocprofv3 --pmc SQ_INSTS_VALU* / SQ_LDS_IDX_ACTIVE on one launch; issue rates from profiles/r02_valu_rate.txt',
        'valu_insts': valu, 'valu_fp64_insts': fp64, 'salu_insts': v.get('SQ_INSTS_SALU', 0), 'lds_insts': v.get('SQ_INSTS_LDS', 0),
        'valu_cycles_per_64_cells': valu_cycles / (cells / 64.0),
        'valu_floor_ms': valu_cycles / simds / clock * 1e3,
        'ceiling_cells_per_s': cells / (valu_cycles / simds / clock),
        'lds_array_cycles': l.get('SQ_LDS_IDX_ACTIVE', 0), 'lds_bank_conflict_cycles': l.get('SQ_LDS_BANK_CONFLICT', 0),
        'lds_floor_ms': l.get('SQ_LDS_IDX_ACTIVE', 0) / cus / clock * 1e3,
        'lds_ceiling_cells_per_s': cells / (l.get('SQ_LDS_IDX_ACTIVE', 1) / cus / clock),
        'sq_wait_any_frac': l.get('SQ_WAIT_ANY', 0) / max(1.0, v.get('SQ_WAVE_CYCLES', 1)),
        'sq_active_valu_frac_of_wave': v.get('SQ_ACTIVE_INST_VALU', 0) / max(1.0, v.get('SQ_WAVE_CYCLES', 1)),
        'waves': l.get('SQ_WAVES', 0)}
json.dump(comp, open(out + '/compute.json', 'w'), indent=1)
print(json.dumps(res))
print(json.dumps(comp))
PY
