"""diagnostic: the pipelined / two-thread batch scenario of tests/test_gpu_parity.py, repeated, with mismatch counts."""
import os, sys, random, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import test_gpu_parity as T
from cpecan_amd import api
from cpecan_amd.workload import make_pair

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(907)
pkw = dict(diagonalExpansion=20, splitMatrixBiggerThanThis=900)
jobs = [T._fuzz_problems(rng, 60, 20) for _ in range(5)]
jobs.append(([make_pair(3, i, 2000, 100) for i in range(3)], [(False, False)] * 3))
serial = [T._run_batch(0, probs, raggeds, **pkw)[0] for probs, raggeds in jobs]
p = api.pairwiseAlignmentBandingParameters_construct(**pkw)


def start(k):
    probs, raggeds = jobs[k]
    b = api.Batch(T._sm(0), p)
    b.add_many([(sx, sy, a, rl, rr) for (sx, sy, a), (rl, rr) in zip(probs, raggeds)])
    b.upload(); b.run()
    return b


def finish(k, b, into):
    b.download()
    into[k] = [b.result(i) for i in range(len(jobs[k][0]))]
    b.close()


def bad(got):
    n = 0
    for k, (g, w) in enumerate(zip(got, serial)):
        if g is None:
            n += 1000
            continue
        for i, (a, c) in enumerate(zip(g, w)):
            if not np.array_equal(a, c):
                n += 1
                print("   mismatch job %d problem %d: got %d triples, want %d; lens %d x %d, ragged %s" % (
                    k, i, len(a), len(c), len(jobs[k][0][i][0]), len(jobs[k][0][i][1]), jobs[k][1][i]), flush=True)
    return n


tot_p = tot_t = 0
for rep in range(reps):
    piped = [None] * len(jobs)
    prev = start(0)
    for k in range(1, len(jobs)):
        cur = start(k)
        finish(k - 1, prev, piped)
        prev = cur
    finish(len(jobs) - 1, prev, piped)
    tot_p += bad(piped)
    threaded = [None] * len(jobs)

    def worker(t):
        for k in range(t, len(jobs), 2):
            finish(k, start(k), threaded)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    tot_t += bad(threaded)
print("reps %d: pipelined mismatches %d, two-thread mismatches %d" % (reps, tot_p, tot_t))
