# A/B of a library variant (build_ab/<tag>.so, tools/ab_build.sh) against build_ab/base.so in ONE call: config A forced to the
# one-launch form, the 1 kb / 2 kb shapes of tools/split_forms.py, config B.  usage: bash tools/prio_ab.sh <tag>
for lib in build_ab/base.so build_ab/$1.so; do
  echo "== $lib"
  CPECAN_LIB=$lib timeout -k 10 200 python tools/split_forms.py 1000 1000 3 2>&1 | grep "expansion  50"
  CPECAN_LIB=$lib timeout -k 10 200 python tools/split_forms.py 1000 1000 2>&1 | grep "expansion  50\|expansion 100"
  CPECAN_LIB=$lib timeout -k 10 200 python tools/split_forms.py 2500 1000 2>&1 | grep "expansion  50"
  CPECAN_LIB=$lib timeout -k 10 200 python tools/split_forms.py 1250 2000 2>&1 | grep "expansion 100"
  for i in 1 2; do CPECAN_LIB=$lib timeout -k 10 200 python bench.py --config B --no-cpu-baseline --no-e2e --steps 5 --warmup 2 2>&1 | grep -o "\"ms_per_step\": [0-9.]*"; done
done
