#!/bin/bash
# timing builds of the traceback's F-row prefetch (results invalid) on the under-subscribed launches: per-kernel ms
bash tools/ab_kstats.sh "--config A" base nopf hotrows
bash tools/ab_kstats.sh "--config B --pairs 1250" base nopf hotrows
