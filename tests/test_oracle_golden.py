"""Pins the CPU oracle (oracle/) against every fixed fixture the reference's own tests hold for
the hot path and against the known answers SURVEY.md section 8c recorded from the reference.
CPU only."""
import json
import math
import os
import random

import numpy as np
import pytest

import oracle_binding as ob

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_fixtures.json")))
NEG_INF = float("-inf")


def test_bands_golden():
    g = GOLD["test_bands"]
    got = ob.band(g["anchors"], g["lX"], g["lY"], g["expansion"])
    assert got == [tuple(d) for d in g["diagonals"]]


def test_diagonal_validity():
    g = GOLD["test_diagonal"]
    for d in g["valid"]:
        assert ob.lib().orc_diagonal_valid(*d) == 1
    for d in g["invalid"]:
        assert ob.lib().orc_diagonal_valid(*d) == 0


def test_symbol_golden():
    g = GOLD["test_symbol"]
    assert [ob.symbol(ch) for ch in g["string"]] == g["symbols"]


def test_split_points_golden():
    g = GOLD["test_getSplitPoints"]
    for case in g["cases"]:
        got = ob.split_points(case["anchors"], case["lX"], case["lY"], g["matrixSize"], case["raggedLeft"],
                              case["raggedRight"])
        assert got == [tuple(r) for r in case["expect"]], case


def test_logadd_tolerance_and_exactness():
    # tests/pairwiseAlignerTest.c:134-144 -- +-0.001 in linear space
    rng = random.Random(7)
    for _ in range(20000):
        i, j = rng.random(), rng.random()
        if i == 0 or j == 0:
            continue
        l = math.exp(ob.log_add(math.log(i), math.log(j)))
        assert abs(l - (i + j)) < 0.001
    # structural properties of impl/pairwiseAligner.c:303-307
    assert ob.log_add(NEG_INF, -3.0) == -3.0 and ob.log_add(-3.0, NEG_INF) == -3.0
    assert ob.log_add(NEG_INF, NEG_INF) == NEG_INF
    assert ob.log_add(0.0, -7.5) == 0.0 and ob.log_add(-7.5, 0.0) == 0.0
    assert ob.log_add(1.25, -2.0) == ob.log_add(-2.0, 1.25)
    # tie: P(0) + y with the float32-valued constant term
    assert ob.log_add(2.0, 2.0) == float(np.float32(0.693203116424741)) + 2.0
    # the polynomial is evaluated with float32-valued coefficients in double Horner form
    c = [float(np.float32(v)) for v in (-0.014532321752540, 0.139942324101744, 0.495635523139337, 0.692140569840976)]
    d = 1.75
    assert ob.log_add(d, 0.0) == ((c[0] * d + c[1]) * d + c[2]) * d + c[3]


def _cells(m, fill):
    return np.full(m.S, fill, dtype=np.float64)


def _ptr(a):
    import ctypes as C
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def test_cell_forward_equals_backward():
    # tests/pairwiseAlignerTest.c:155-182
    import ctypes as C
    m = ob.model(ob.FIVE_STATE)
    L = ob.lib()
    S = m.S
    middleF = np.array([m.start[i] for i in range(S)])
    lowerF, upperF, currentF = _cells(m, NEG_INF), _cells(m, NEG_INF), _cells(m, NEG_INF)
    middleB, lowerB, upperB = _cells(m, NEG_INF), _cells(m, NEG_INF), _cells(m, NEG_INF)
    currentB = np.array([m.end[i] for i in range(S)])
    cX, cY = 0, 3
    L.orc_cell_forward(C.byref(m), _ptr(lowerF), None, None, _ptr(middleF), cX, cY)
    L.orc_cell_forward(C.byref(m), _ptr(upperF), _ptr(middleF), None, None, cX, cY)
    L.orc_cell_forward(C.byref(m), _ptr(currentF), _ptr(lowerF), _ptr(middleF), _ptr(upperF), cX, cY)
    L.orc_cell_backward(C.byref(m), _ptr(currentB), _ptr(lowerB), _ptr(middleB), _ptr(upperB), cX, cY)
    L.orc_cell_backward(C.byref(m), _ptr(upperB), _ptr(middleB), None, None, cX, cY)
    L.orc_cell_backward(C.byref(m), _ptr(lowerB), None, None, _ptr(middleB), cX, cY)

    def dot2(cell, prior):
        t = cell[0] + prior[0]
        for i in range(1, S):
            t = ob.log_add(t, cell[i] + prior[i])
        return t

    f = dot2(currentF, [m.end[i] for i in range(S)])
    b = dot2(middleB, [m.start[i] for i in range(S)])
    assert abs(f - b) < 1e-5


def test_known_answer_dp_pair_set():
    # tests/pairwiseAlignerTest.c:242-324 (pair set), scores from SURVEY 8c
    g = GOLD["test_diagonalDPCalculations"]
    m = ob.model(ob.FIVE_STATE)
    p = ob.params(threshold=g["threshold"], diagonalExpansion=2)
    pairs = ob.aligned_pairs(m, g["sX"], g["sY"], (), p)
    assert sorted((int(x), int(y)) for _, x, y in pairs) == sorted(tuple(q) for q in g["pairs"])
    assert len(pairs) == 4


def test_known_answer_total_probability_constant_over_diagonals():
    # tests/pairwiseAlignerTest.c:291-296: the per-diagonal total equals the forward probability (0.01)
    g = GOLD["test_diagonalDPCalculations"]
    m = ob.model(ob.FIVE_STATE)
    # one refresh per diagonal: minDiags etc. untouched, sequences are tiny -> single traceback; the trace
    # records the total used per emitted diagonal (refreshed every 10th)
    pairs, tr = ob.aligned_pairs_traced(m, g["sX"], g["sY"], (), ob.params(threshold=g["threshold"]))
    fwd = ob.forward_prob(m, g["sX"], g["sY"])
    used = tr["total_used"][1:]
    assert np.all(np.abs(used - fwd) < g["per_diagonal_total_tolerance"])


def test_survey_known_answers():
    g = GOLD["survey_known_answers"]
    m5, m3 = ob.model(ob.FIVE_STATE), ob.model(ob.THREE_STATE)
    pairs = ob.aligned_pairs(m5, g["sX"], g["sY"], (), ob.params(threshold=0.2))
    assert pairs.tolist() == g["aligned_pairs_5state_threshold_0.2"]
    tol = g["tolerance"]["forward_prob"]
    assert abs(ob.forward_prob(m5, g["sX"], g["sY"]) - g["forward_prob_5state"]) <= tol
    assert abs(ob.forward_prob(m3, g["sX"], g["sY"]) - g["forward_prob_3state"]) <= tol
    assert abs(ob.forward_prob(m3, g["sX"], g["sY"], ragged_left=True, ragged_right=True)
               - g["forward_prob_3state_ragged_both"]) <= tol
    h = ob.hmm(ob.FIVE_STATE, 0.0)
    ob.expectations(m5, h, g["sX"], g["sY"])
    e = g["expectations_5state_pseudo0"]
    et = g["tolerance"]["expectations"]
    assert abs(h.likelihood - e["likelihood"]) <= et
    assert abs(h.T[0 * 5 + 0] - e["T_M_M"]) <= et
    assert abs(h.T[0 * 5 + 1] - e["T_M_sX"]) <= et
    assert abs(h.E[0] - e["E_M_A_A"]) <= et
    # likelihood is added once per diagonal (impl/pairwiseAligner.c:743): 10 diagonals x forward prob
    assert abs(h.likelihood - 10 * g["forward_prob_5state"]) < 1e-6


def test_hmm_normalise_golden():
    # tests/pairwiseAlignerTest.c:997-1073
    for t in (ob.FIVE_STATE, ob.FIVE_STATE_ASYM, ob.THREE_STATE, ob.THREE_STATE_ASYM):
        h = ob.hmm(t, 0.0)
        S = h.S
        for f in range(S):
            for to in range(S):
                h.T[f * S + to] += f * S + to
        for s in range(S):
            for i in range(16):
                h.E[s * 16 + i] += s * 16 + i
        ob.lib().orc_hmm_normalise(h)
        for f in range(S):
            z = f * S * S + (S * (S - 1)) // 2
            for to in range(S):
                assert h.T[f * S + to] == (f * S + to) / z
        for s in range(S):
            z = 16 * 16 * s + (16 * 15) // 2
            for i in range(16):
                assert h.E[s * 16 + i] == (s * 16 + i) / z


# ---- randomised property tests mirroring the reference's (tests/pairwiseAlignerTest.c:326-438,649-674) ----

_ALPHABET = "AaCcGgTt" * 11 + "N"


def _rand_seq(rng, n):
    return "".join(rng.choice(_ALPHABET) for _ in range(n))


def _evolve(rng, s):
    s = [rng.choice(_ALPHABET) if rng.random() > 0.8 else ch for ch in s]
    s = "".join(s)
    while rng.random() > 0.2:
        a = _rand_seq(rng, rng.randrange(2, 4))
        b = _rand_seq(rng, rng.randrange(0, 10))
        s = s.replace(a, b)
    return s


def _rand_anchors(rng, lX, lY):
    out, x, y = [], -1, -1
    while True:
        x += rng.randrange(1, 20)
        y += rng.randrange(1, 20)
        e = 2 * rng.randrange(0, 5)
        if x >= lX or y >= lY:
            return out
        out.append((x, y, e))


def _check_pairs(pairs, lX, lY):
    seen = set()
    for score, x, y in pairs.tolist():
        assert 0 < score <= ob.PROB_1
        assert 0 <= x < lX and 0 <= y < lY
        assert (x, y) not in seen
        seen.add((x, y))


def test_banded_driver_random_property():
    rng = random.Random(11)
    m = ob.model(ob.FIVE_STATE)
    for _ in range(100):
        sx = _rand_seq(rng, rng.randrange(0, 100))
        sy = _evolve(rng, sx)
        tbd = rng.randrange(1, 10)
        p = ob.params(traceBackDiagonals=tbd, minDiagsBetweenTraceBack=tbd + rng.randrange(2, 10),
                      diagonalExpansion=2 * rng.randrange(0, 10), dynamicAnchorExpansion=int(rng.random() > 0.5),
                      splitMatrixBiggerThanThis=10 ** 12)
        pairs, tr = ob.aligned_pairs_traced(m, sx, sy, _rand_anchors(rng, len(sx), len(sy)), p)
        _check_pairs(pairs, len(sx), len(sy))
        if len(sx) + len(sy) > 0:
            # every diagonal 1..N emitted exactly once (impl/pairwiseAligner.c:868)
            assert not np.any(np.isnan(tr["total_used"][1:]))


def test_traceback_schedule_invariance():
    """Posteriors from many short traceback segments agree with a single full traceback to ~logAdd noise."""
    rng = random.Random(5)
    m = ob.model(ob.FIVE_STATE)
    sx = _rand_seq(rng, 300)
    sy = _evolve(rng, sx)
    big = ob.params(diagonalExpansion=1000, splitMatrixBiggerThanThis=10 ** 12)
    small = ob.params(diagonalExpansion=1000, splitMatrixBiggerThanThis=10 ** 12, traceBackDiagonals=20,
                      minDiagsBetweenTraceBack=50)
    a = {(x, y): s for s, x, y in ob.aligned_pairs(m, sx, sy, (), big).tolist()}
    b = {(x, y): s for s, x, y in ob.aligned_pairs(m, sx, sy, (), small).tolist()}
    common = set(a) & set(b)
    assert len(common) > 100
    diffs = sorted(abs(a[k] - b[k]) for k in common)
    # a 20-diagonal warm-up from the end-state prior is itself an approximation in the reference
    assert diffs[len(diffs) // 2] < 0.002 * ob.PROB_1
    assert diffs[-1] < 0.1 * ob.PROB_1


def test_forward_probability_properties():
    # tests/pairwiseAlignerTest.c:1157-1188
    rng = random.Random(3)
    m = ob.model(ob.THREE_STATE)
    for _ in range(100):
        sx = _rand_seq(rng, rng.randrange(10, 100))
        sy = _evolve(rng, sx)
        rl, rr = rng.random() > 0.5, rng.random() > 0.5
        lp = ob.forward_prob(m, sx, sy, (), None, rl, rr)
        lpi = ob.forward_prob(m, sx, sx, (), None, rl, rr)
        assert NEG_INF < lp <= 0.0
        assert lp <= lpi


@pytest.mark.parametrize("mtype", [ob.FIVE_STATE, ob.THREE_STATE_ASYM, ob.THREE_STATE])
def test_em_likelihood_monotone(mtype):
    # tests/pairwiseAlignerTest.c:1091-1155 (10 tests instead of 100 to keep the CPU suite short)
    rng = random.Random(100 + mtype)
    for _ in range(10):
        sx = _rand_seq(rng, rng.randrange(10, 100))
        sy = _evolve(rng, sx)
        h = ob.hmm(mtype, 0.0)
        S = h.S
        for i in range(S * S):
            h.T[i] = rng.random()
        for i in range(S * 16):
            h.E[i] = rng.random()
        ob.lib().orc_hmm_normalise(h)
        m = ob.model_from_hmm(h)
        prev = NEG_INF
        for _it in range(10):
            acc = ob.hmm(mtype, 1e-12)
            ob.expectations(m, acc, sx, sy)
            ob.lib().orc_hmm_normalise(acc)
            assert prev <= acc.likelihood * 0.95
            prev = acc.likelihood
            m = ob.model_from_hmm(acc)


def test_split_regions_cover_and_order():
    """With a tiny split threshold the per-region lists are concatenated in region order and each
    region's list is (segment desc, diagonal asc, xmy desc); coordinates stay unique and in range."""
    rng = random.Random(9)
    m = ob.model(ob.FIVE_STATE)
    sx = _rand_seq(rng, 400)
    sy = _evolve(rng, sx)
    n = min(len(sx), len(sy))
    anchors = [(i, i, 4) for i in range(5, n - 5, 37)]
    p = ob.params(diagonalExpansion=4, splitMatrixBiggerThanThis=10)
    pairs = ob.aligned_pairs(m, sx, sy, anchors, p, True, True)
    _check_pairs(pairs, len(sx), len(sy))
    assert len(pairs) > 0


# ---- consumers of the posterior lists ----
def test_left_shift_alignment_golden():
    """tests/pairwiseAlignerTest.c:944-995 (test_leftShiftAlignment)."""
    fx = GOLD["test_leftShiftAlignment"]
    pairs = [(fx["score"], x, y) for x, y in zip(fx["alignedX"], fx["alignedY"])]
    got = ob.left_shift_alignment(pairs, fx["seqX"], fx["seqY"])
    assert len(got) == len(fx["shiftedX"])
    assert [int(v) for v in got[:, 1]] == fx["shiftedX"]
    assert [int(v) for v in got[:, 2]] == fx["shiftedY"]


def test_reweight_by_hand():
    """impl/pairwiseAligner.c:1519-1558 worked by hand: every base keeps PROB_1 minus the listed mass (floored at 0),
    a pair loses gapGamma times the unaligned mass of its two bases; the result is truncated towards zero."""
    P = 10000000
    pairs = [(6000000, 0, 0), (3000000, 0, 1)]
    got = ob.reweight_aligned_pairs(pairs, 1, 2, 0.5)
    assert [int(v) for v in got[:, 0]] == [6000000 - (1000000 + 4000000) // 2, 3000000 - (1000000 + 7000000) // 2]
    # over-subscribed base: mass floored at zero (:1529-1533); gapGamma <= 0 is the identity (:1551)
    got = ob.reweight_aligned_pairs([(9000000, 0, 0), (9000000, 0, 1)], 1, 2, 1.0)
    assert [int(v) for v in got[:, 0]] == [9000000 - (0 + 1000000), 9000000 - (0 + 1000000)]
    assert (ob.reweight_aligned_pairs(pairs, 1, 2, 0.0) == np.array(pairs)).all()
    assert abs(ob.score_by_posterior(1, 2, pairs) - 100.0 * 2 * 9000000 / (3 * P)) < 1e-12
    assert abs(ob.score_by_posterior_ignoring_gaps(pairs) - 100.0 * 9000000 / (2 * P)) < 1e-12


def test_mea_alignment_properties():
    """getMaximalExpectedAccuracyPairwiseAlignment (:1628-1724): the result is a strictly increasing chain of input pairs;
    with no gap mass and gapGamma 0 it is the heaviest chain (checked against an O(n^2) DP on small inputs)."""
    import random
    rng = random.Random(5)
    for _ in range(50):
        lX, lY = rng.randrange(3, 12), rng.randrange(3, 12)
        cells = sorted({(rng.randrange(lX), rng.randrange(lY)) for _ in range(rng.randrange(1, 14))},
                       key=lambda c: (c[0] + c[1], c[1] - c[0]))
        pairs = [(rng.randrange(1, 10000000), x, y) for x, y in cells]
        got, score = ob.mea_alignment(pairs, [], [], lX, lY, 0.0)
        xs = [int(v) for v in got[:, 1]]
        ys = [int(v) for v in got[:, 2]]
        assert all(a < b for a, b in zip(xs, xs[1:])) and all(a < b for a, b in zip(ys, ys[1:]))
        assert {tuple(int(v) for v in r) for r in got} <= set(pairs)
        best = [0] * len(pairs)
        for i, (w, x, y) in enumerate(pairs):
            best[i] = w + max([best[j] for j in range(len(pairs)) if pairs[j][1] < x and pairs[j][2] < y and j < i] + [0])
        assert score == max(best)
        assert sum(int(v) for v in got[:, 0]) == score


def _heaviest_chain(pairs, gamma):
    """Independent O(n^2) statement of what pairwiseAlignColumns maximises (multipleAligner.c:358-492)."""
    g = float(np.float32(gamma))
    cand = sorted((x, y, w / 1e7) for w, x, y in pairs if w / 1e7 >= g and w > 0)
    best = []
    for i, (x, y, w) in enumerate(cand):
        best.append(w + max([best[j] for j in range(i) if cand[j][0] < x and cand[j][1] < y] + [0.0]))
    return max(best + [0.0])


def test_filter_pairs_ordered_by_hand():
    """filterPairwiseAlignmentToMakePairsOrdered (multipleAligner.c:945-972), jitter-free: hand-worked cases, ties included."""
    P = 10000000
    # the diagonal chain (0,0),(1,1),(2,2) outweighs the single heavy off-diagonal pair; output is in reverse input order
    pairs = [(int(.6 * P), 0, 0), (int(.9 * P), 0, 2), (int(.5 * P), 1, 1), (int(.4 * P), 2, 2), (int(.05 * P), 2, 0)]
    got = ob.filter_pairs_ordered(pairs, 3, 3, 0.1)
    assert [tuple(int(v) for v in r) for r in got] == [pairs[3], pairs[2], pairs[0]]
    # raise matchGamma above the chain's weakest links: only the heavy pair is a candidate
    got = ob.filter_pairs_ordered(pairs, 3, 3, 0.85)
    assert [tuple(int(v) for v in r) for r in got] == [pairs[1]]
    # equal-weight alternatives: the entry with the smaller y is the one the frontier keeps (:418-427)
    pairs = [(int(.5 * P), 0, 1), (int(.5 * P), 1, 0)]
    got = ob.filter_pairs_ordered(pairs, 2, 2, 0.0)
    assert [tuple(int(v) for v in r) for r in got] == [pairs[1]]
    # non-positive weights never enter (:393), even with matchGamma 0; an empty list stays empty
    assert len(ob.filter_pairs_ordered([(0, 0, 0), (-5, 1, 1)], 2, 2, 0.0)) == 0
    assert len(ob.filter_pairs_ordered([], 4, 4, 0.0)) == 0


def test_filter_pairs_ordered_properties():
    """As the reference's test of pairwiseAlignColumns (tests/multipleAlignerTest.c:124-148, checkAlignment): the kept
    pairs are consistent (strictly increasing in both coordinates once sorted) and drawn from the input; and the chain is
    the heaviest one among pairs of weight >= matchGamma."""
    import random
    rng = random.Random(11)
    for trial in range(200):
        lX, lY = rng.randrange(1, 25), rng.randrange(1, 25)
        cells = list({(rng.randrange(lX), rng.randrange(lY)) for _ in range(rng.randrange(0, 60))})
        rng.shuffle(cells)
        coarse = trial % 2 == 0  # coarse weights make exact ties common
        pairs = [((rng.randrange(0, 11) * 1000000) if coarse else rng.randrange(-1000, 10000001), x, y) for x, y in cells]
        gamma = rng.choice([0.0, 0.1, 0.5, 0.85])
        got = [tuple(int(v) for v in r) for r in ob.filter_pairs_ordered(pairs, lX, lY, gamma)]
        assert set(got) <= set(pairs)
        index = {p: i for i, p in enumerate(pairs)}
        assert [index[p] for p in got] == sorted((index[p] for p in got), reverse=True)
        chain = sorted(got, key=lambda p: p[1])
        assert all(a[1] < b[1] and a[2] < b[2] for a, b in zip(chain, chain[1:]))
        assert all(w / 1e7 >= float(np.float32(gamma)) and w > 0 for w, _, _ in got)
        assert abs(sum(w for w, _, _ in got) / 1e7 - _heaviest_chain(pairs, gamma)) < 1e-9


def test_identity_scores_by_hand():
    """scoreByIdentity / scoreByIdentityIgnoringGaps (:1562-1580): N never matches, case is ignored."""
    sx, sy = "ACgTN", "aCCTN"
    pairs = [(1, 0, 0), (1, 1, 1), (1, 2, 2), (1, 3, 3), (1, 4, 4)]
    assert ob.score_by_identity(sx, sy, pairs) == 100.0 * 2 * 3 / 10
    assert ob.score_by_identity_ignoring_gaps(sx, sy, pairs) == 100.0 * 3 / 5
    assert ob.score_by_identity("", "", []) == 0.0


# ---- what the reference's tests hold beyond geometry ----
import reference_cases as rc  # noqa: E402


def test_getAlignedPairsWithRaggedEnds_exact_outcome_oracle():
    """tests/pairwiseAlignerTest.c:676-715: a 100-base core inside 300 bases, ragged on both sides, ordered filter at 0.2
    => exactly the 100 pairs (x, x + 100).  150 seeded trials; pins the ragged start / end priors of the five-state model
    (stateMachine.c:407-447) and the jitter-free ordered filter with an exact outcome."""
    om, op = ob.model(ob.FIVE_STATE), ob.params()
    for trial in range(150):
        sx, sy = rc.ragged_ends_trial(trial)
        pairs = ob.aligned_pairs(om, sx, sy, (), op, True, True)
        out = ob.filter_pairs_ordered(pairs, len(sx), len(sy), 0.2)
        assert len(out) == 100, (trial, len(out))
        assert all(int(y) == int(x) + 100 for _, x, y in out), trial
        assert all(0 < int(s) <= 10000000 for s, _, _ in out)


def test_trained_hmm_text_of_the_reference_loads_into_the_documented_model():
    """cPecanEmTest.py:112-113 writes this trained five-state-asymmetric HMM as text and loads it back: the file layout is
    `type, S*S transitions, likelihood` / `S*16 emissions` (stateMachine.c:133-202).  The model built from it follows
    stateMachine5_loadAsymmetric (stateMachine.c:529-575) term for term; here: no long/short swap is triggered."""
    mtype, T, lik, E = rc.trained_hmm_numbers()
    assert mtype == ob.FIVE_STATE_ASYM and len(T) == 25 and len(E) == 80 and lik == -83964693614.2
    h = ob.hmm(ob.FIVE_STATE_ASYM, 0.0)
    for i, v in enumerate(T):
        h.T[i] = v
    for i, v in enumerate(E):
        h.E[i] = v
    m = ob.model_from_hmm(h)
    tr = {(m.tr[i].frm, m.tr[i].to): m.tr[i].tP for i in range(m.nTransitions)}
    M, SX, SY, LX, LY = 0, 1, 2, 3, 4
    for (f, t) in [(M, M), (SX, M), (SY, M), (LX, M), (LY, M), (M, SX), (SX, SX), (M, LX), (LX, LX), (M, SY), (SY, SY), (M, LY), (LY, LY)]:
        assert tr[(f, t)] == math.log(T[f * 5 + t]), (f, t)
    for x in range(4):
        for y in range(4):
            assert m.matchEm[x * 5 + y] == math.log(E[x * 4 + y])
    gx = [sum(E[s * 16 + x * 4 + y] for s in (SX, LX) for y in range(4)) for x in range(4)]
    for x in range(4):
        assert abs(m.gapXEm[x] - math.log(gx[x] / sum(gx))) < 1e-15


def test_encode_human_chimp_full_length_oracle():
    """tests/pairwiseAlignerLongTest.c: the ~57 kb human / chimp ENCODE fragments, the only real sequences in the
    reference tree, aligned at full length (115 k anti-diagonals, ~115 traceback segments) and scored against the embedded
    reference alignment the way the test logs it (:100-108).  The reference asserts nothing but uniqueness; the 0.99
    bars below are this repo's."""
    sx, sy, anchors, true_pairs = rc.encode_human_chimp()
    assert len(sx) == 57553 and len(sy) == 57344 and len(true_pairs) == 56835
    p = ob.params(diagonalExpansion=20)
    pairs = ob.aligned_pairs(ob.model(ob.FIVE_STATE), sx, sy, anchors, p)
    assert len({(int(x), int(y)) for _, x, y in pairs}) == len(pairs)          # the reference's own assertion (:75)
    out = ob.filter_pairs_ordered(pairs, len(sx), len(sy), 0.5)
    sens, spec = rc.sensitivity_specificity(out, true_pairs)
    assert sens > 0.99 and spec > 0.99, (sens, spec)


@pytest.mark.parametrize("species,n_x,n_y,n_true,sens_bar,spec_bar",
                         [("mouse", 57553, 32750, 20406, 0.75, 0.85), ("dog", 57553, 54187, 39241, 0.90, 0.94)])
def test_encode_human_mouse_and_dog_oracle(species, n_x, n_y, n_true, sens_bar, spec_bar):
    """tests/pairwiseAlignerLongTest.c:128-134: the human / mouse and human / dog ENCODE pairs at full length.  Divergent
    sequences (67 % / 75 % identity), anchors with stretches of up to 8 kb of one sequence between them: banded
    parallelograms alternate with full rectangles of up to 1947 x 1329 cells.  The reference asserts uniqueness of the
    pairs (:75) and logs sensitivity / specificity against the embedded alignment; the bars are this repo's."""
    sx, sy, anchors, true_pairs = rc.encode_human_other(species)
    assert (len(sx), len(sy), len(true_pairs)) == (n_x, n_y, n_true)
    pairs = ob.aligned_pairs(ob.model(ob.FIVE_STATE), sx, sy, anchors, ob.params(diagonalExpansion=20))
    assert len({(int(x), int(y)) for _, x, y in pairs}) == len(pairs)
    out = ob.filter_pairs_ordered(pairs, len(sx), len(sy), 0.5)
    sens, spec = rc.sensitivity_specificity(out, true_pairs)
    assert sens > sens_bar and spec > spec_bar, (sens, spec)

