#!/bin/bash
export CPECAN_BIDIR=1
bash tools/ab_kstats.sh "--config A" base noemit nopasses
