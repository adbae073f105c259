"""diagnostic: wall time of every stage of a batch of one bench config, batch after batch (serial), then two in flight.
usage: python tools/e2e_stages.py [pairs] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpecan_amd import api, workload

import bench
name = sys.argv[2] if len(sys.argv) > 2 else "B"
cfg = workload.CONFIGS[name]
n = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n_pairs"]
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sm, p, _ = bench.model_and_params(api, cfg)
probs = workload.config_problems(name, range(n))
# E2E_TRIPLES=1: the anchors of the realign configuration as one triple per column (round 3) instead of as runs
as_runs = bool(cfg.get("realign")) and os.environ.get("E2E_TRIPLES") != "1"
arr, cnt, keep = (api.Batch.prepare_problems_runs if as_runs else api.Batch.prepare_problems)(probs)
T = time.perf_counter


def stages(b=None):
    t = [T()]
    b = api.Batch(sm, p); t.append(T())
    b.add_prepared(arr, cnt); t.append(T())
    b.upload(); t.append(T())
    b.run(); t.append(T())
    b.download(); t.append(T())
    st = b.stats()
    b.close(); t.append(T())
    d = [1e3 * (t[i + 1] - t[i]) for i in range(len(t) - 1)]
    print("create %.1f add %.1f upload %.1f (h2d %.1f) run-launch %.1f download %.1f (kernel %.1f d2h %.1f) close %.1f | total %.1f ms" % (
        d[0], d[1], d[2], st.h2dMs, d[3], d[4], st.kernelMs, st.d2hMs, d[5], 1e3 * (t[-1] - t[0])), flush=True)


print("serial:")
for _ in range(5):
    stages()
print("two in flight:")


def start():
    t0 = T()
    b = api.Batch(sm, p)
    b.add_prepared(arr, cnt)
    t1 = T()
    b.upload()
    t2 = T()
    b.run()
    return b, 1e3 * (t1 - t0), 1e3 * (t2 - t1)


t0 = T()
prev, a0, u0 = start()
for k in range(1, 8):
    cur, a, u = start()
    t1 = T()
    prev.download()
    t2 = T()
    prev.close()
    t3 = T()
    print("batch %d: add %.1f upload %.1f | download(prev) %.1f close(prev) %.1f | elapsed %.1f ms" % (k, a, u, 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (T() - t0)), flush=True)
    prev = cur
prev.download(); prev.close()
print("8 batches in %.1f ms = %.1f ms per batch" % (1e3 * (T() - t0), 1e3 * (T() - t0) / 8))
