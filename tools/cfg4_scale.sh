#!/bin/bash
# config 4 at several batch sizes: is the launch bound by the longest regions' chains (time flat) or by throughput (time ~ pairs)?
for p in 50000 25000 12500 6000; do
  timeout -k 10 300 python bench.py --config 4 --pairs $p --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pairs', $p, 'cells %.3e'%d['config']['cells_total'], 'ms %.2f'%d['ms_per_step'], 'cells/s %.3e'%d['value'])"
done
