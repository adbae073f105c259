"""Parity gate shared by the GPU tests, smoke() and bench.py.

Posterior scores are floor(p * 1e7) of p = exp(log-space fp64); the north star asks for 1e-5 relative.
Two triple lists match when (a) they hold the same (x, y) set except for pairs whose probability is within
1e-5 relative of the emission threshold, (b) common pairs differ by at most max(1, 1e-5 * score) and
(c) the common pairs appear in the same order."""
import numpy as np

PROB_1 = 10000000
REL_TOL = 1e-5


def assert_pairs_match(got, want, threshold=0.01, rel=REL_TOL, check_order=True):
    got = np.asarray(got, dtype=np.int64).reshape(-1, 3)
    want = np.asarray(want, dtype=np.int64).reshape(-1, 3)
    g = {(int(x), int(y)): int(s) for s, x, y in got}
    w = {(int(x), int(y)): int(s) for s, x, y in want}
    assert len(g) == len(got), "duplicate coordinates in the HIP output"
    edge = threshold * PROB_1
    slack = max(2.0, rel * edge * 2)
    for k in set(g) ^ set(w):
        s = g.get(k, w.get(k))
        assert abs(s - edge) <= slack, "pair %s (score %d) present on one side only" % (k, s)
    worst = 0.0
    for k in set(g) & set(w):
        tol = max(1.0, rel * w[k])
        d = abs(g[k] - w[k])
        assert d <= tol, "pair %s: hip %d vs oracle %d" % (k, g[k], w[k])
        worst = max(worst, d / max(1.0, w[k]))
    if check_order:
        common = set(g) & set(w)
        go = [(int(x), int(y)) for _, x, y in got if (int(x), int(y)) in common]
        wo = [(int(x), int(y)) for _, x, y in want if (int(x), int(y)) in common]
        assert go == wo, "list order differs from the reference order"
    return worst
