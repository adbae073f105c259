#!/bin/bash
# copies the judged artifacts of gpurun_out/<tag> (written by tools/profile_round.sh) into profiles/<tag>_*
set -e
tag=$1
src=gpurun_out/$tag
cp $src/bench.json profiles/${tag}_bench.json
cp $src/traffic.json profiles/${tag}_traffic.json
cp $src/compute.json profiles/${tag}_compute.json
cp "$(ls -t $src/stats/*/*_kernel_stats.csv | head -1)" profiles/${tag}_kernel_stats.csv   # the newest run: gpurun merges into what earlier calls left
python3 - $src profiles/$tag <<'PY'
import csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
kt = newest(src + '/stats/*/*_kernel_trace.csv')
rows = list(csv.DictReader(open(kt)))
keep = [r for r in rows if 'pairhmm' in r['Kernel_Name']]
w = csv.DictWriter(open(dst + '_kernel_trace.csv', 'w', newline=''), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for name in ('fetch', 'write', 'valu', 'lds'):
    f = newest(src + '/' + name + '/*/*_counter_collection.csv')
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if 'pairhmm' in r['Kernel_Name']]
    w = csv.DictWriter(open(dst + '_pmc_' + name + '.csv', 'w', newline=''), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
ls -la profiles/${tag}_*
