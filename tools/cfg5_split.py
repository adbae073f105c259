#!/usr/bin/env python3
"""Where config 5's time goes: the same regions under the forward-only, match and expectation emitters (kernel ms from the
library's own events).  Usage: python tools/cfg5_split.py [pairs] [config]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from cpecan_amd import api, workload
from bench import model_and_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
name = sys.argv[2] if len(sys.argv) > 2 else "5"
cfg = dict(workload.CONFIGS[name])
torch.zeros(1, device="cuda")
sm, params, mtype = model_and_params(api, cfg)
problems = workload.config_problems(name, np.arange(n))
prepared, n_prepared, _keep = api.Batch.prepare_problems(problems)
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for label, emit in (("forward", api.EMIT_FORWARD), ("match", api.EMIT_MATCH), ("indel", api.EMIT_INDEL), ("expect", api.EMIT_EXPECT)):
    if only and label not in only:
        continue
    with api.Batch(sm, params, emit=emit, device=0) as b:
        b.add_prepared(prepared, n_prepared)
        b.upload()
        ms = []
        for _ in range(4):
            b.run()
            b.download()
            ms.append(b.stats().kernelMs)
        st = b.stats()
        print("%-8s cells %d  kernel ms %s  -> %.3g cells/s  form %d waves %d" %
              (label, st.cells, " ".join("%.2f" % m for m in ms), st.cells / (min(ms) * 1e-3), st.launchForm, st.wavesPerLaunch), flush=True)
