# timing-only A/B of the one-launch form's device-scope ring accesses (build_ab/{base,nold,nost,nocoh}.so from tools/ab_build.sh
# with -DCPK_COH_LD=0 / -DCPK_COH_ST=0): which side of the hand-off costs what.  Results of the variants are undefined.
for lib in base nold nost nocoh; do
  echo "== $lib"
  CPECAN_LIB=build_ab/$lib.so timeout -k 10 200 python tools/split_forms.py 1000 1000 3 2>&1 | grep "expansion  50"
  CPECAN_LIB=build_ab/$lib.so timeout -k 10 200 python tools/split_forms.py 1000 1000 2>&1 | grep "expansion  50"
  CPECAN_LIB=build_ab/$lib.so timeout -k 10 200 python tools/split_forms.py 1250 2000 2>&1 | grep "expansion 100"
  CPECAN_LIB=build_ab/$lib.so timeout -k 10 200 python bench.py --config B --no-cpu-baseline --no-e2e --steps 5 --warmup 2 2>&1 | grep -o "\"ms_per_step\": [0-9.]*"
done
