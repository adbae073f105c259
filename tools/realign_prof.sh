#!/bin/bash
# Kernel-level profile of one cpecan_realign run on the files tools/realign_bench.py left behind (pass their directory).
set -e
d=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/realign_prof -o realign -- $GRAFT_REPO_ROOT/cpecan_amd/cpecan_realign --batch 50000 $d/seqs.fa < $d/in.cigar > /dev/null
