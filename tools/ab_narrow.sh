#!/bin/bash
# usage (GPU box): tools/ab_narrow.sh tagA tagB ... -> config-5-like match/expect timing per variant, interleaved twice
for rep in 1 2; do for tag in "$@"; do
  echo "== $tag"; CPECAN_LIB=$PWD/build_ab/$tag.so timeout -k 10 300 python -u tools/narrow_bench.py 5000 2>&1 | tail -2
done; done
