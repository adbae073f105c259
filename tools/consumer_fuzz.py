#!/usr/bin/env python3
"""Differential fuzz of the wave-per-problem consumers against the lane-per-problem kernels (CPECAN_POST_LANES=1) and,
every tenth round, the CPU oracle: ordered filter and MEA chain on random lists of awkward sizes (0, 1, 63-65, 127-129
pairs; sequences longer than the 2048 LDS column counters; columns of more than 64 pairs; exact ties).
Usage: python tools/consumer_fuzz.py [rounds] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

from cpecan_amd import api
import oracle_binding as ob

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)


def tup(a):
    return [tuple(int(v) for v in r) for r in np.asarray(a).reshape(-1, 3)]


def both(fn):
    os.environ.pop("CPECAN_POST_LANES", None)
    wave = fn()
    os.environ["CPECAN_POST_LANES"] = "1"
    lanes = fn()
    os.environ.pop("CPECAN_POST_LANES", None)
    return wave, lanes


sizes = [0, 1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 500, 1500, 4000]
for r in range(rounds):
    n = rng.choice(sizes) if rng.random() < 0.6 else rng.randrange(0, 800)
    shape = rng.random()
    if shape < 0.2:
        lX, lY = rng.randrange(1, 6), rng.randrange(max(1, n // 3), n + 40)  # few, very long columns
    elif shape < 0.35:
        lX, lY = rng.randrange(2100, 5000), rng.randrange(2100, 5000)  # beyond the LDS column counters
    else:
        lX = rng.randrange(1, max(2, n)) if n else rng.randrange(1, 50)
        lY = rng.randrange(1, max(2, n)) if n else rng.randrange(1, 50)
    cells = set()
    if rng.random() < 0.7:  # a noisy diagonal first
        for i in range(min(lX, lY, n)):
            cells.add((i, min(lY - 1, max(0, i + rng.randrange(-3, 4)))))
    tries = 0
    while len(cells) < min(n, lX * lY) and tries < 20 * n + 100:
        cells.add((rng.randrange(lX), rng.randrange(lY)))
        tries += 1
    cells = list(cells)
    if rng.random() < 0.5:
        cells.sort(key=lambda c: (c[0] + c[1], -(c[0] - c[1])))
    else:
        rng.shuffle(cells)
    coarse = rng.random() < 0.4
    pairs = [((rng.randrange(0, 6) * 2000000) if coarse else rng.randrange(-1000, 10000001), x, y) for x, y in cells]
    gamma = rng.choice([0.0, 0.1, 0.5, 0.85])
    w, l = both(lambda: tup(api.filterPairwiseAlignmentToMakePairsOrdered(pairs, "A" * lX, "A" * lY, gamma)))
    assert w == l, ("ordered", r, n, lX, lY, gamma)
    if r % 10 == 0:
        assert w == tup(ob.filter_pairs_ordered(pairs, lX, lY, gamma)), ("ordered vs oracle", r)
    # the MEA chain wants non-negative weights and the emitters' list order
    mp = sorted(((abs(p[0]) + 1, p[1], p[2]) for p in pairs), key=lambda t: (t[1] + t[2], -(t[1] - t[2])))
    heavy = rng.random() < 0.5
    gx = [(rng.randrange(1, 9000000 if heavy else 200000), rng.randrange(lX), rng.randrange(lY)) for _ in range(rng.randrange(0, 2 * lX + 1))]
    gy = [(rng.randrange(1, 9000000 if heavy else 200000), rng.randrange(lX), rng.randrange(lY)) for _ in range(rng.randrange(0, 2 * lY + 1))]
    gg = float(np.float32(rng.choice([0.0, 0.25, 0.5, 1.0])))
    if mp:
        def mea():
            a, s = api.getMaximalExpectedAccuracyPairwiseAlignment(mp, gx, gy, lX, lY, gapGamma=gg)
            return tup(a), s
        w, l = both(mea)
        assert w == l, ("mea", r, n, lX, lY, gg)
        if r % 10 == 0:
            oa, os_ = ob.mea_alignment(mp, gx, gy, lX, lY, gg)
            assert w == (tup(oa), os_), ("mea vs oracle", r)
    if r % 20 == 0:
        print("round %d ok (n %d, %d x %d)" % (r, len(pairs), lX, lY), flush=True)
print("consumer fuzz ok: %d rounds" % rounds)
