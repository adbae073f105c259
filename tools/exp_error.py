"""How far the expectation counts of the HIP path are from the oracle's (the gate is 1e-5 relative, north_star): config-5 style
pairs (1 kb, band 10) and a wider band, every model type; prints the largest relative deviation over the transition and
emission counts and of the likelihood.  usage: python tools/exp_error.py [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_binding as ob
from cpecan_amd import api
from cpecan_amd.workload import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for mtype in (0, 1, 2, 3):
    for E, L in ((10, 1000), (40, 600)):
        probs = [make_pair(5 + mtype, i, L, E) for i in range(n if E == 10 else n // 4)]
        kw = dict(diagonalExpansion=E)
        sm = api.stateMachine5_construct(mtype) if mtype in (0, 1) else api.stateMachine3_construct(mtype)
        acc = api.hmm_constructEmpty(0.0, mtype)
        with api.Batch(sm, api.pairwiseAlignmentBandingParameters_construct(**kw), emit=api.EMIT_EXPECT) as b:
            for sx, sy, a in probs:
                b.add(sx, sy, a, False, False)
            b.upload(); b.run(); b.download()
            b.expectations(acc)
        oacc = ob.hmm(mtype, 0.0)
        om, op = ob.model(mtype), ob.params(**kw)
        for sx, sy, a in probs:
            ob.expectations(om, oacc, sx, sy, a, op, False, False)
        S = acc.stateNumber
        t = np.array(list(acc.transitions)[:S * S]); to = np.array(list(oacc.T)[:S * S])
        e = np.array(list(acc.emissions)[:S * 16]); eo = np.array(list(oacc.E)[:S * 16])
        mt, me = to > 0, eo > 0
        print("model %d band %3d, %4d pairs: transitions max rel %.2e, emissions max rel %.2e, likelihood rel %.2e"
              % (mtype, E, len(probs), np.max(np.abs(t[mt] - to[mt]) / to[mt]), np.max(np.abs(e[me] - eo[me]) / eo[me]),
                 abs(acc.likelihood - oacc.likelihood) / abs(oacc.likelihood)), flush=True)
