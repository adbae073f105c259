"""Replays one round of tools/soak_forms.py (same generator) and bisects a mismatch between CPECAN_SPLIT=0 and another form over
the knobs that select code paths.  usage: python tools/soak_repro.py <round> <seed> [split]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cpecan_amd import api
from cpecan_amd.workload import make_pair

target, seed0 = int(sys.argv[1]), int(sys.argv[2])
split = sys.argv[3] if len(sys.argv) > 3 else "1"
rng = random.Random(seed0)
torch.zeros(1, device="cuda")
for rd in range(target + 1):
    n = rng.choice((20, 60, 150, 400))
    five = rng.random() < 0.6
    E = rng.choice((8, 20, 50, 100, 120))
    every = rng.choice((20, 50, 50, 150, 400))
    thr = rng.choice((0.01, 0.01, 0.2, 0.0))
    seed = rng.randrange(1 << 30)
    probs = []
    for i in range(n):
        L = int(150 * (20 ** rng.random()))
        if rd == target:
            sx, sy, a = make_pair(seed, i, L, E, anchor_every=every)
            probs.append((sx, sy, a, rng.random() < 0.2, rng.random() < 0.2))
        else:
            rng.random(); rng.random()
print("round %d: %d pairs, %s-state, E=%d, anchors every %d, threshold %g" % (target, n, 5 if five else 3, E, every, thr))
sm = api.stateMachine5_construct() if five else api.stateMachine3_construct()
p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=E, threshold=thr)
arr, cnt, keep = api.Batch.prepare_problems(probs)


def run(env):
    for k in ("CPECAN_SPLIT", "CPECAN_TABLE_WAVE", "CPECAN_TRACE3", "CPECAN_FUSED3", "CPECAN_FWD3", "CPECAN_ABS", "CPECAN_DENSE", "CPECAN_ABS_WINDOWS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with api.Batch(sm, p) as b:
        b.add_prepared(arr, cnt)
        b.upload(); b.run(); b.download()
        st = b.stats()
        return [b.result(i).copy() for i in range(n)], st


ref, st0 = run({"CPECAN_SPLIT": "0"})
for extra in ({}, {"CPECAN_ABS_WINDOWS": "0"}, {"CPECAN_TABLE_WAVE": "0"}, {"CPECAN_TRACE3": "0", "CPECAN_FUSED3": "0"}, {"CPECAN_FWD3": "0"}, {"CPECAN_ABS": "0"},
              {"CPECAN_TRACE3": "0", "CPECAN_FUSED3": "0", "CPECAN_FWD3": "0", "CPECAN_TABLE_WAVE": "0"}):
    env = {"CPECAN_SPLIT": split}
    env.update(extra)
    got, st = run(env)
    bad = [i for i in range(n) if not np.array_equal(ref[i], got[i])]
    print(env, "form", st.launchForm, "mismatching problems:", bad[:10])
    for i in bad[:2]:
        a, b2 = ref[i], got[i]
        if a.shape != b2.shape:
            print("   problem %d: %d vs %d triples" % (i, len(a), len(b2)))
            continue
        rows = np.nonzero((a != b2).any(axis=1))[0]
        print("   problem %d (lX %d lY %d): %d rows differ, first at %d: ref %s got %s; last at %d" % (
            i, len(probs[i][0]), len(probs[i][1]), len(rows), rows[0], a[rows[0]].tolist(), b2[rows[0]].tolist(), rows[-1]))
        print("   rows:", rows[:12].tolist(), "ref", a[rows[:4]].tolist(), "got", b2[rows[:4]].tolist())
