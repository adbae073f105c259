"""config-4 style end-to-end rate with P producer threads, each running its own pipeline of batches (depth D) on one GPU:
is the single producer's host share (packing + planning, ~28 ms per batch) what bounds bench.py's value_e2e?
usage: python tools/e2e_producers.py [producers] [depth] [batches_per_producer] [config]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpecan_amd import api, workload
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 12
name = sys.argv[4] if len(sys.argv) > 4 else "4"
cfg = workload.CONFIGS[name]
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sm, p, _ = bench.model_and_params(api, cfg)
probs = workload.config_problems(name, range(cfg["n_pairs"]))
as_runs = bool(cfg.get("realign"))
arr, cnt, keep = (api.Batch.prepare_problems_runs if as_runs else api.Batch.prepare_problems)(probs)


def make():
    b = api.Batch(sm, p)
    b.add_prepared(arr, cnt)
    b.upload()
    return b


cells = None
# warm-up: P * (D + 1) batches alive at once, twice
for _ in range(2):
    ws = []
    for _k in range(P * (D + 1)):
        b = make(); b.run(); b.download_begin(); ws.append(b)
    for b in ws:
        b.download_end()
    cells = ws[0].stats().cells
    for b in ws:
        b.close()
done_t = []
lock = threading.Lock()


def producer():
    inflight = []
    for _ in range(NB):
        b = make(); b.run(); b.download_begin(); inflight.append(b)
        if len(inflight) >= D:
            old = inflight.pop(0); old.download_end()
            with lock:
                done_t.append(time.perf_counter())
            old.close()
    for old in inflight:
        old.download_end()
        with lock:
            done_t.append(time.perf_counter())
        old.close()


t0 = time.perf_counter()
ths = [threading.Thread(target=producer) for _ in range(P)]
for t in ths:
    t.start()
for t in ths:
    t.join()
done_t.sort()
n = len(done_t)
steady = (done_t[-1] - done_t[P * D - 1]) / (n - P * D)   # from the moment every pipeline is full
print("%d producers x depth %d, %d batches: %.1f ms per batch steady (%.3e cells/s), %.1f ms per batch with fill and drain"
      % (P, D, n, 1e3 * steady, cells / steady, 1e3 * (done_t[-1] - t0) / n), flush=True)
