#!/bin/bash
# the one-launch form with the forward waves at s_setprio 3 (build_ab/prio3.so) against none (prio0.so): strong-scaled shards of config B
for p in 1250 2500 5000; do echo "== $p pairs"; bash tools/ab_bench.sh $p prio0 prio3; done
