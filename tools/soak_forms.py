"""soak: random batches (20-400 pairs of 150-3000 bp, expansions 8-120, three- or five-state, anchor spacing 20-400, thresholds
0.01 / 0.2 / 0) run under every launch form (CPECAN_SPLIT=0 one wave per region, 1 two launches, 2 one launch) and, for
three-state, CPECAN_DENSE=1; the lists must be identical triple for triple.  usage: python tools/soak_forms.py [rounds] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cpecan_amd import api
from cpecan_amd.workload import make_pair

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.zeros(1, device="cuda")
total = 0
for rd in range(rounds):
    n = rng.choice((20, 60, 150, 400))
    five = rng.random() < 0.6
    E = rng.choice((8, 20, 50, 100, 120))
    every = rng.choice((20, 50, 50, 150, 400))
    thr = rng.choice((0.01, 0.01, 0.2, 0.0))
    seed = rng.randrange(1 << 30)
    probs = []
    for i in range(n):
        L = int(150 * (20 ** rng.random()))
        sx, sy, a = make_pair(seed, i, L, E, anchor_every=every)
        probs.append((sx, sy, a, rng.random() < 0.2, rng.random() < 0.2))
    sm = api.stateMachine5_construct() if five else api.stateMachine3_construct()
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=E, threshold=thr)
    arr, cnt, keep = api.Batch.prepare_problems(probs)
    forms = [("0", None), ("1", None), ("2", None)] + ([("1", "1"), ("2", "1"), ("0", "1")] if not five else [])
    ref = None
    for split, dense in forms:
        os.environ["CPECAN_SPLIT"] = split
        if dense is None:
            os.environ.pop("CPECAN_DENSE", None)
        else:
            os.environ["CPECAN_DENSE"] = dense
        with api.Batch(sm, p) as b:
            b.add_prepared(arr, cnt)
            b.upload(); b.run(); b.download()
            got = [b.result(i) for i in range(n)]
        if ref is None:
            ref = got
        else:
            for i in range(n):
                if not np.array_equal(ref[i], got[i]):
                    print("MISMATCH round %d problem %d split=%s dense=%s (%d vs %d triples)" % (rd, i, split, dense, len(ref[i]), len(got[i])))
                    sys.exit(1)
    total += sum(len(r) for r in ref)
    print("round %d: %d pairs, %s-state, E=%d, anchors every %d, threshold %g: %d forms agree (%d triples)"
          % (rd, n, 5 if five else 3, E, every, thr, len(forms), sum(len(r) for r in ref)), flush=True)
print("soak ok: %d rounds, %d triples compared under every form" % (rounds, total))
