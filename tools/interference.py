"""diagnostic: what slows a config-B sweep when other work runs beside it (the pipelined case is 25-30 % slower than the
kernel alone): a stream of pageable D2H copies, a table build + upload of another batch, or busy host cores?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cpecan_amd import api, workload

torch.zeros(1, device="cuda"); torch.cuda.synchronize()
cfg = workload.CONFIGS["B"]
sm = api.stateMachine5_construct(api.fiveState)
p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=100, splitMatrixBiggerThanThis=10 ** 15)
probs = workload.config_problems("B", range(10000))
arr, cnt, keep = api.Batch.prepare_problems(probs)
b = api.Batch(sm, p); b.add_prepared(arr, cnt); b.upload()
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()


def kernel_ms():
    b.run(s1.cuda_stream); s1.synchronize(); b.download(); return b.stats().kernelMs


print("alone: %.1f %.1f ms" % (kernel_ms(), kernel_ms()))
stop = False
big = torch.empty(283 * 1024 * 1024 // 4, dtype=torch.int32, device="cuda")
host = torch.empty_like(big, device="cpu")  # pageable


def copier():
    with torch.cuda.stream(s2):
        while not stop:
            host.copy_(big, non_blocking=False)


def pinned_copier():
    hp = torch.empty_like(big, device="cpu").pin_memory()
    with torch.cuda.stream(s2):
        while not stop:
            hp.copy_(big, non_blocking=True); s2.synchronize()


def uploader():
    while not stop:
        b2 = api.Batch(sm, p); b2.add_prepared(arr, cnt); b2.upload(); b2.close()


def packer():
    while not stop:
        b2 = api.Batch(sm, p); b2.add_prepared(arr, cnt); b2.close()


def _unused_uploader_notable():
    os.environ["CPECAN_DIAG_SKIP_TABLE"] = "1"
    try:
        uploader()
    finally:
        del os.environ["CPECAN_DIAG_SKIP_TABLE"]


def burner():
    a = np.random.rand(600, 600)
    while not stop:
        a @ a


for name, fn, n in (("pageable D2H copies beside it", copier, 1), ("pinned D2H copies beside it", pinned_copier, 1),
                    ("another batch being packed (host only) beside it", packer, 1),

                    ("another batch being packed / planned / uploaded beside it", uploader, 1), ("16 busy host threads", burner, 16)):
    stop = False
    ts = [threading.Thread(target=fn) for _ in range(n)]
    for t in ts: t.start()
    time.sleep(0.3)
    r = [kernel_ms() for _ in range(3)]
    stop = True
    for t in ts: t.join()
    print("%s: %s ms" % (name, " ".join("%.1f" % v for v in r)), flush=True)
