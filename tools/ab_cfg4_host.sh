for kv in "X=1" "OMP_WAIT_POLICY=passive" "CPECAN_THREADS=12" "CPECAN_THREADS=8" "OMP_WAIT_POLICY=passive CPECAN_THREADS=12"; do
  env $kv timeout -k 10 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --e2e-batches 13 > /tmp/b4.json 2>/dev/null
  python -c "
import json,sys; d=json.loads(open('/tmp/b4.json').read().strip().splitlines()[-1]); e=d['e2e']; print('$kv', 'depth', e['pipeline_depth'], 'kernel ms %.1f' % d['ms_per_step'], 'e2e %.3e' % d['value_e2e'], 'steady ms %.1f' % (1e3*e['pipelined_steady_s_per_batch']), 'plan_upload ms %.1f' % (1e3*e['plan_upload_s']))"
done
