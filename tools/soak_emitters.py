"""soak of the indel and expectation emitters over the kernels that carry them: random batches of narrow, medium and wide bands
(expansions 2-60 with dense or sparse anchors, or no anchors at all on pairs of up to 700 bp) run through the library's own
choice of kernels, with the packed kernel off (CPECAN_PACKED=0), with the team kernel forced from 100 cells (CPECAN_TEAM=100) or
off (CPECAN_TEAM=0), and -- expectations -- with the second pass everywhere (CPECAN_EXP_INSWEEP=0).  The three lists of the indel
emitter must be identical under every setting; the expectation counts within 1e-8 of one another (the order of the sums differs)
and, every other round, within 1e-5 of the oracle's.  usage: python tools/soak_emitters.py [rounds] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from cpecan_amd import api
from cpecan_amd.workload import make_pair
import oracle_binding as ob

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.zeros(1, device="cuda")
KNOBS = ("CPECAN_PACKED", "CPECAN_TEAM", "CPECAN_EXP_INSWEEP", "CPECAN_EXP_ONE_GROUP")
SETTINGS = [{}, {"CPECAN_PACKED": "0"}, {"CPECAN_TEAM": "100"}, {"CPECAN_TEAM": "0"}, {"CPECAN_PACKED": "2"}]


def setenv(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)


for rd in range(rounds):
    mtype = rng.choice((0, 1, 2, 3))
    style = rng.choice(("narrow", "medium", "wide", "unanchored"))
    n = rng.choice((8, 30, 70))
    if style == "unanchored":
        probs = [make_pair(rng.randrange(1 << 20), i, rng.randrange(40, 700), 0)[:2] + ((),) for i in range(min(n, 12))]
        E = 20
    else:
        E = {"narrow": rng.choice((2, 4, 10)), "medium": rng.choice((20, 40)), "wide": 60}[style]
        every = {"narrow": rng.choice((1, 3, 50)), "medium": rng.choice((20, 50, 150)), "wide": rng.choice((150, 400))}[style]
        probs = [make_pair(rng.randrange(1 << 20), i, rng.randrange(60, 1800), E, anchor_every=max(every, 2)) for i in range(n)]
    raggeds = [(rng.random() < 0.3, rng.random() < 0.3) for _ in probs]
    tbd = rng.choice((40, 40, 12))
    kw = dict(diagonalExpansion=E, traceBackDiagonals=tbd, minDiagsBetweenTraceBack=rng.choice((1000, 300, tbd + 60)),
              threshold=rng.choice((0.01, 0.2)))
    sm = api.stateMachine5_construct(mtype) if mtype in (0, 1) else api.stateMachine3_construct(mtype)
    p = api.pairwiseAlignmentBandingParameters_construct(**kw)
    # ---- the three lists of the indel emitter
    ref = None
    for env in SETTINGS:
        setenv(env)
        with api.Batch(sm, p, emit=api.EMIT_INDEL) as b:
            for (sx, sy, a), (rl, rr) in zip(probs, raggeds):
                b.add(sx, sy, a, rl, rr)
            b.upload(); b.run(); b.download()
            got = [[b.result(i, l).copy() for l in range(3)] for i in range(len(probs))]
        if ref is None:
            ref = got
        else:
            for i in range(len(probs)):
                for l in range(3):
                    if not np.array_equal(ref[i][l], got[i][l]):
                        print("MISMATCH round %d problem %d list %d under %s" % (rd, i, l, env))
                        sys.exit(1)
    # ---- expectation counts
    accs = []
    for env in SETTINGS + [{"CPECAN_EXP_INSWEEP": "0"}, {"CPECAN_EXP_ONE_GROUP": "0"}]:
        setenv(env)
        acc = api.hmm_constructEmpty(0.0, mtype)
        with api.Batch(sm, p, emit=api.EMIT_EXPECT) as b:
            for (sx, sy, a), (rl, rr) in zip(probs, raggeds):
                b.add(sx, sy, a, rl, rr)
            b.upload(); b.run(); b.download()
            b.expectations(acc)
        S = acc.stateNumber
        accs.append(np.array(list(acc.transitions)[:S * S] + list(acc.emissions)[:S * 16] + [acc.likelihood]))
    for k, v in enumerate(accs[1:]):
        try:
            np.testing.assert_allclose(v, accs[0], rtol=1e-6, atol=1e-12)  # (fp32 events summed in different orders)
            np.testing.assert_allclose(v[-1], accs[0][-1], rtol=1e-11)
        except AssertionError:
            print("MISMATCH round %d (%s, model %d, %d problems, E=%d, %s): expectation counts under %s against the default"
                  % (rd, style, mtype, len(probs), E, kw, (SETTINGS + [{"CPECAN_EXP_INSWEEP": "0"}, {"CPECAN_EXP_ONE_GROUP": "0"}])[k + 1]), flush=True)
            raise
    note = ""
    if rd % 2 == 0:
        oacc = ob.hmm(mtype, 0.0)
        om, op = ob.model(mtype), ob.params(**kw)
        for (sx, sy, a), (rl, rr) in zip(probs, raggeds):
            ob.expectations(om, oacc, sx, sy, a, op, rl, rr)
        want = np.array(list(oacc.T)[:S * S] + list(oacc.E)[:S * 16] + [oacc.likelihood])
        np.testing.assert_allclose(accs[0], want, rtol=1e-5, atol=1e-12)
        note = ", oracle ok"
    print("round %d: %s, model %d, %d problems, E=%d, %s: indel lists equal under %d settings, counts under %d%s"
          % (rd, style, mtype, len(probs), E, {k: v for k, v in kw.items() if k != "diagonalExpansion"}, len(SETTINGS), len(accs), note), flush=True)
print("soak ok: %d rounds" % rounds)
