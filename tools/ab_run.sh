#!/bin/bash
# usage (on the GPU box): tools/ab_run.sh "caps" tagA tagB ...   -> fwd_scale for each variant, twice, interleaved
caps=$1; shift
for rep in 1 2; do for tag in "$@"; do
  echo "== $tag (rep $rep)"
  CPECAN_LIB=$PWD/build_ab/$tag.so timeout -k 10 200 python -u tools/fwd_scale.py $caps || exit 1
done; done
