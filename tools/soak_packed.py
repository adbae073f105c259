"""soak of the packed kernel's launch forms: random batches of narrow-band problems -- realign-style (anchors on every matching
column, expansion 2-10, lengths 50-6000, ragged ends, split rectangles) and random sparse anchors with short traceback schedules
-- run as whole regions (CPECAN_PACKED_SPLIT=0), as a split class (=1), with the automatic cut (unset, CPECAN_PACKED_SPLIT_FROM
random) and through the sweep kernel (CPECAN_PACKED=0); the lists must be identical triple for triple, and every 16th problem
equal to the oracle's.  usage: python tools/soak_packed.py [rounds] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from cpecan_amd import api
from cpecan_amd.workload import make_realign_batch
import oracle_binding as ob
from parity import assert_pairs_match

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.zeros(1, device="cuda")
KNOBS = ("CPECAN_PACKED", "CPECAN_PACKED_SPLIT", "CPECAN_PACKED_SPLIT_FROM")


def rand_seq(n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def evolve(s):
    out = []
    for ch in s:
        r = rng.random()
        if r < 0.05:
            out.append(rng.choice("ACGT"))
        elif r < 0.08:
            continue
        else:
            out.append(ch)
        if rng.random() < 0.03:
            out.append(rng.choice("ACGT"))
    return "".join(out)


total = 0
for rd in range(rounds):
    mtype = rng.choice((0, 1, 2, 3))
    style = rng.choice(("realign", "realign", "sparse"))
    thr = rng.choice((0.01, 0.01, 0.2, 0.0))
    if style == "realign":
        E = rng.choice((2, 4, 4, 6, 10))
        n = rng.choice((70, 200, 600))
        hi = rng.choice((800, 3000, 6000))
        probs = make_realign_batch(rng.randrange(1 << 20), n, 50, hi, E)
        kw = dict(diagonalExpansion=E, splitMatrixBiggerThanThis=rng.choice((10, 10, 400, 10 ** 12)), threshold=thr,
                  minDiagsBetweenTraceBack=rng.choice((1000, 1000, 300)), traceBackDiagonals=rng.choice((40, 40, 12)))
    else:
        E = rng.choice((2, 6, 14, 26))
        n = rng.choice((70, 150))
        probs = []
        for _ in range(n):
            sx = rand_seq(rng.randrange(1, 1200))
            sy = evolve(sx) or "C"
            anchors, x, y = [], -1, -1
            while True:
                x += rng.randrange(1, 5)
                y += rng.randrange(1, 5)
                if x >= len(sx) or y >= len(sy):
                    break
                anchors.append((x, y, E))
            probs.append((sx, sy, anchors))
        tbd = rng.randrange(3, 40)
        kw = dict(diagonalExpansion=E, splitMatrixBiggerThanThis=rng.choice((10, 50, 10 ** 12)), threshold=thr,
                  minDiagsBetweenTraceBack=tbd + rng.randrange(20, 300), traceBackDiagonals=tbd)
    raggeds = [(rng.random() < 0.5, rng.random() < 0.5) for _ in probs]
    sm = api.stateMachine5_construct(mtype) if mtype in (0, 1) else api.stateMachine3_construct(mtype)
    p = api.pairwiseAlignmentBandingParameters_construct(**kw)
    forms = [{"CPECAN_PACKED": "2", "CPECAN_PACKED_SPLIT": "0"}, {"CPECAN_PACKED": "2", "CPECAN_PACKED_SPLIT": "1"},
             {"CPECAN_PACKED": "2", "CPECAN_PACKED_SPLIT_FROM": str(rng.choice((100, 500, 1500, 4000)))}, {"CPECAN_PACKED": "2"},
             {"CPECAN_PACKED": "0"}]
    ref = None
    for env in forms:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        with api.Batch(sm, p) as b:
            for (sx, sy, a), (rl, rr) in zip(probs, raggeds):
                b.add(sx, sy, a, rl, rr)
            b.upload(); b.run(); b.download()
            got = [b.result(i).copy() for i in range(len(probs))]
        if ref is None:
            ref = got
        else:
            for i in range(len(probs)):
                if not np.array_equal(ref[i], got[i]):
                    print("MISMATCH round %d problem %d under %s (%d vs %d triples)" % (rd, i, env, len(ref[i]), len(got[i])))
                    sys.exit(1)
    om, op = ob.model(mtype), ob.params(**kw)
    for i in range(0, len(probs), 16):
        sx, sy, a = probs[i]
        assert_pairs_match(ref[i], ob.aligned_pairs(om, sx, sy, a, op, *raggeds[i]), threshold=op.threshold)
    total += sum(len(r) for r in ref)
    print("round %d: %s, model %d, %d problems, E=%d, threshold %g, %s: %d forms agree (%d triples), oracle ok"
          % (rd, style, mtype, len(probs), E, thr, {k: v for k, v in kw.items() if k != "diagonalExpansion"}, len(forms),
             sum(len(r) for r in ref)), flush=True)
print("soak ok: %d rounds, %d triples compared under every form" % (rounds, total))
