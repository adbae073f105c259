"""Run as a child process by tests/test_gpu_exact.py with CPECAN_LIB pointing at the diagnostic library
(`make -C cpecan_amd/csrc exact`: logAdd as the reference's operations one for one, impl/pairwiseAligner.c:287-307).
Every log-space value of the sweeps' debug buffers, every emitted triple, every forward probability must EQUAL the
oracle's -- tolerance zero.  Prints one line per case and exits non-zero on the first difference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from cpecan_amd import api  # noqa: E402
from cpecan_amd.workload import make_pair  # noqa: E402
import oracle_binding as ob  # noqa: E402


def sm(mtype):
    return api.stateMachine5_construct(mtype) if mtype < 2 else api.stateMachine3_construct(mtype)


def main():
    assert "exact" in os.path.basename(api.LIB_PATH), api.LIB_PATH
    cases = [
        ("tiny5", 0, ("AGCG", "AGTTCG", ()), dict(threshold=0.2), False),
        ("tiny3", 2, ("AGCG", "AGTTCG", ()), dict(threshold=0.2), False),
        ("A_1kb", 2, make_pair(2, 0, 1000, 50), dict(diagonalExpansion=50), False),
        ("B_2kb", 0, make_pair(3, 0, 2000, 100), dict(diagonalExpansion=100), False),
        ("short_tracebacks", 0, make_pair(3, 1, 600, 10), dict(diagonalExpansion=10, minDiagsBetweenTraceBack=50,
                                                            traceBackDiagonals=7), False),
        ("asym_ragged", 1, make_pair(3, 2, 1500, 30), dict(diagonalExpansion=30), True),
    ]
    for name, mtype, (sx, sy, a), pkw, ragged in cases:
        p = api.pairwiseAlignmentBandingParameters_construct(**pkw)
        want, tr = ob.aligned_pairs_traced(ob.model(mtype), sx, sy, a, ob.params(**pkw), ragged, ragged)
        with api.Batch(sm(mtype), p, debug=True) as b:
            b.add(sx, sy, a, ragged, ragged)
            b.upload()
            b.run()
            b.download()
            got = b.result(0)
            fb, tot = b.debug_fetch(0, tr["n_cells"], tr["n_diagonals"])
        want = np.asarray(want, dtype=np.int64).reshape(-1, 3)
        assert np.array_equal(np.asarray(got, dtype=np.int64), want), (name, "triples differ")
        m = ~np.isnan(tr["fb_match"])
        assert np.array_equal(fb[m], tr["fb_match"][m]), (name, "F.match + B.match differs",
                                                          float(np.nanmax(np.abs(fb[m] - tr["fb_match"][m]))))
        mt = ~np.isnan(tr["total_used"])
        assert mt.any() and np.array_equal(tot[mt], tr["total_used"][mt]), (name, "totals differ")
        print("exact: %s: %d triples, %d cells, %d totals equal" % (name, len(want), int(m.sum()), int(mt.sum())), flush=True)
    # the production forms of the same kernels (split classes, no debug buffers): a batch of config-B-shaped pairs
    probs = [make_pair(5, i, 2000, 100) for i in range(24)]
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=100)
    with api.Batch(sm(0), p) as b:
        b.add_many(probs)
        b.upload()
        b.run()
        b.download()
        for i, (sx, sy, a) in enumerate(probs):
            want = np.asarray(ob.aligned_pairs(ob.model(0), sx, sy, a, ob.params(diagonalExpansion=100)), dtype=np.int64).reshape(-1, 3)
            assert np.array_equal(np.asarray(b.result(i), dtype=np.int64), want), ("batch", i)
    print("exact: batch of %d config-B pairs: every triple equal" % len(probs), flush=True)
    # forward probabilities (pure logAdd arithmetic)
    for mtype in (0, 2):
        sx, sy, a = make_pair(6, mtype, 700, 40)
        p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=40)
        got = api.computeForwardProbability(sx, sy, a, p, sm(mtype), False, False)
        want = ob.forward_prob(ob.model(mtype), sx, sy, a, ob.params(diagonalExpansion=40), False, False)
        assert got == want, ("forward", mtype, got, want)
    print("exact: forward probabilities equal", flush=True)


if __name__ == "__main__":
    main()
