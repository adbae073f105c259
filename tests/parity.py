"""Parity gate shared by the GPU tests, smoke() and bench.py.

Posterior scores are floor(p * 1e7) of p = exp(log-space fp64); the north star asks for 1e-5 relative.
Two triple lists match when (a) they hold the same (x, y) set except for pairs whose probability is within
1e-5 relative of the emission threshold, (b) common pairs differ by at most max(1, 1e-5 * score) and
(c) the common pairs appear in the same order."""
import numpy as np

PROB_1 = 10000000
REL_TOL = 1e-5
# Log-space values (forward probabilities, per-cell F+B sums, totals): the shipped library fuses the three multiply-adds
# of the logAdd cubic and adds Q(d) = P(d) - d to the larger operand instead of P(d) to the smaller one
# (cpk_device_common.inl), so a value differs from the oracle's by the roundings of a few thousand additions at
# magnitudes of 1e2..1e4: observed < 1e-10, gate 1e-9 absolute (i.e. ~1e-12 relative; the north star asks for 1e-5).
# The diagnostic build `make -C cpecan_amd/csrc EXACT=1` is bit-identical to the oracle (CPECAN_EXPECT_BIT_EXACT=1
# tightens the gate to equality for it).
import os
LOG_TOL = 0.0 if os.environ.get("CPECAN_EXPECT_BIT_EXACT") == "1" else 1e-9


def assert_log_close(got, want, what="log value"):
    if want == float("-inf") or got == float("-inf"):
        assert got == want, "%s: hip %r vs oracle %r" % (what, got, want)
        return
    assert abs(got - want) <= LOG_TOL * max(1.0, abs(want)), "%s: hip %.17g vs oracle %.17g" % (what, got, want)


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
