#!/usr/bin/env python3
"""Kernel times of the MEA + left-shift consumers on a batch of config-4-like pairs (indel emitter, narrow bands): run it
under rocprofv3 --kernel-trace --stats to see cpecan_post_mea / cpecan_post_left_shift beside the DP kernels.
Usage: python tools/mea_bench.py [pairs]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from cpecan_amd import api, workload
from bench import model_and_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = dict(workload.CONFIGS["4"])
torch.zeros(1, device="cuda")
sm, params, mtype = model_and_params(api, cfg)
problems = workload.config_problems("4", np.arange(n))
prepared, n_prepared, _keep = api.Batch.prepare_problems(problems)
with api.Batch(sm, params, emit=api.EMIT_INDEL, device=0) as b:
    b.set_post(api.POST_MEA | api.POST_LEFT_SHIFT, 0.5)
    b.add_prepared(prepared, n_prepared)
    b.upload()
    for _ in range(3):
        t0 = time.perf_counter()
        b.run()
        b.download()
        dt = time.perf_counter() - t0
        st = b.stats()
        print("pairs %d cells %d: sweep kernels %.2f ms, run+download %.1f ms" % (n, st.cells, st.kernelMs, dt * 1e3), flush=True)
