#!/bin/bash
# diagnostic: rocprofv3 kernel stats of the cpecan_realign command line on the realign_bench input (one batch of N cigars).
# usage (on the GPU box): bash tools/realign_prof.sh [N]   -> gpurun_out/realign_prof/stats.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-50000}
OUT=$R/gpurun_out/realign_prof
rm -rf $OUT/prof; mkdir -p $OUT
D=$(python3 - $N <<PY
import sys
sys.path.insert(0, "$R/tools")
import realign_bench
d, lines, bases = realign_bench.generate(int(sys.argv[1]), 1000000)
print(d)
PY
)
cd /tmp && export TMPDIR=/tmp
cp $D/in.cigar $OUT/in.cigar
# the binary reads the cigars from stdin: rocprofv3 passes its stdin through
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $R/cpecan_amd/cpecan_realign --batch $N $D/seqs.fa < $D/in.cigar > $OUT/out.cigar 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/prof/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
with open(sys.argv[1] + '/stats.txt', 'w') as o:
    for r in rows:
        line = "%-70s calls %5s total %9.3f ms avg %9.3f ms" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6)
        print(line); o.write(line + "\n")
PY
