#!/bin/bash
# config 4: the packed classes' long regions as a split class (cpecan_kernels.hip, "cut"): from how many diagonals?
bash tools/ab_env.sh "--config 4" CPECAN_PACKED_SPLIT=0 CPECAN_PACKED_SPLIT=1 CPECAN_X=auto CPECAN_PACKED_SPLIT_FROM=2000 CPECAN_PACKED_SPLIT_FROM=3000 CPECAN_PACKED_SPLIT_FROM=4000 CPECAN_PACKED_SPLIT_FROM=6000 CPECAN_PACKED_SPLIT_FROM=8000
