set -e
for t in "" 60 ; do
  for cfg in "A" "B --pairs 1250"; do
    echo "== CPECAN_TEAM=$t config $cfg"
    if [ -z "$t" ]; then python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
    else CPECAN_TEAM=$t CPECAN_TRACE_HOST=1 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-other-configs 2>&1 | grep -E "cpecan class|ms_per_step" | tail -3 | cut -c1-400
    fi
  done
done
