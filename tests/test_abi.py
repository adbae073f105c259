"""CPU-side checks of the product library: it loads, exports every symbol include/cpecan_hip.h declares,
its integer host logic (band, split points, models, hmm I/O) agrees with the reference fixtures and with the
oracle, and compute calls fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import json
import os
import random
import re

import numpy as np
import pytest

import oracle_binding as ob
from cpecan_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cpecan_hip.h")).read()
    declared = set(re.findall(r"\b(cpecan_[a-z_0-9]+)\s*\(", header))
    L = api.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(api.EXPORTS)
    # the realign front end (include/cpecan_realign.h)
    from cpecan_amd import realign
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cpecan_realign.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(cpecan_[a-z_0-9]+)\s*\(", header))
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(realign.EXPORTS)
    # the reference-named layer: every function include/cpecan_dropin.h declares is exported too
    dropin = open(os.path.join(ROOT, "include", "cpecan_dropin.h")).read()
    dropin = re.sub(r"/\*.*?\*/", "", dropin, flags=re.S)
    dropin = re.sub(r"^#define.*$", "", dropin, flags=re.M)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z_0-9]*)\s*\([^;{]*\)\s*;", dropin))
    names -= {"void", "double", "sizeof"}
    names = {n for n in names if not n.startswith("(")}
    assert {"getAlignedPairsUsingAnchors", "getExpectationsUsingAnchors", "computeForwardProbability",
            "stateMachine5_construct", "hmm_loadFromFile", "band_construct", "stList_append",
            "getPosteriorProbsWithBanding", "logAdd", "diagonal_construct", "band_constructDynamic", "hmm_randomise",
            "hmm_jsonParse", "diagonalCalculationPosteriorMatchProbs", "symbolString_construct"} <= names
    assert C.c_char_p.in_dll(L, "PAIRWISE_ALIGNMENT_EXCEPTION_ID").value == b"PAIRWISE_ALIGNMENT_EXCEPTION"
    for name in sorted(names):
        assert hasattr(L, name), name


def test_band_golden_and_vs_oracle():
    g = GOLD["test_bands"]
    assert api.band_construct(g["anchors"], g["lX"], g["lY"], g["expansion"]) == [tuple(d) for d in g["diagonals"]]
    rng = random.Random(1)
    for _ in range(300):
        lX, lY = rng.randrange(0, 120), rng.randrange(0, 120)
        anchors, x, y = [], -1, -1
        while True:
            x += rng.randrange(1, 20)
            y += rng.randrange(1, 20)
            if x >= lX or y >= lY:
                break
            anchors.append((x, y, 2 * rng.randrange(0, 6)))
        e = 2 * rng.randrange(0, 11)
        for dyn in (False, True):
            assert api.band_construct(anchors, lX, lY, e, dyn) == ob.band(anchors, lX, lY, e, dyn)


def test_band_rejects_bad_anchors():
    with pytest.raises(api.CpecanError):
        api.band_construct([(3, 3, 0), (2, 5, 0)], 10, 10, 2)
    with pytest.raises(api.CpecanError):
        api.band_construct([(3, 3, 0)], 10, 10, 3)  # odd expansion, pairwiseAligner.c:187


def test_split_points_golden_and_vs_oracle():
    g = GOLD["test_getSplitPoints"]
    for case in g["cases"]:
        got = api.getSplitPoints(case["anchors"], case["lX"], case["lY"], g["matrixSize"], case["raggedLeft"],
                                 case["raggedRight"])
        assert got == [tuple(r) for r in case["expect"]]
    rng = random.Random(2)
    for _ in range(200):
        lX, lY = rng.randrange(1, 5000), rng.randrange(1, 5000)
        anchors, x, y = [], -1, -1
        while True:
            x += rng.randrange(1, 700)
            y += rng.randrange(1, 700)
            if x >= lX or y >= lY:
                break
            anchors.append((x, y, 0))
        mx = rng.choice([10, 1000, 250000])
        for rl in (0, 1):
            for rr in (0, 1):
                assert api.getSplitPoints(anchors, lX, lY, mx, rl, rr) == ob.split_points(anchors, lX, lY, mx, rl, rr)


def _oracle_model_as_dict(m):
    return {(t.block, t.frm, t.to): t.tP for t in list(m.tr)[:m.nTransitions]}


def test_default_models_match_oracle_constants():
    for t in (api.fiveState, api.threeState):
        pm = api.stateMachine5_construct(t) if t == api.fiveState else api.stateMachine3_construct(t)
        om = ob.model(t)
        tr = _oracle_model_as_dict(om)
        assert tr[(1, 0, 0)] == pm.matchContinue
        assert tr[(0, 0, 1)] == pm.gapShortOpenX and tr[(0, 1, 1)] == pm.gapShortExtendX
        assert tr[(2, 0, 2)] == pm.gapShortOpenY and tr[(2, 2, 2)] == pm.gapShortExtendY
        assert tr[(1, 1, 0)] == pm.matchFromShortGapX and tr[(1, 2, 0)] == pm.matchFromShortGapY
        if t == api.fiveState:
            assert tr[(0, 0, 3)] == pm.gapLongOpenX and tr[(0, 3, 3)] == pm.gapLongExtendX
            assert tr[(1, 3, 0)] == pm.matchFromLongGapX and tr[(1, 4, 0)] == pm.matchFromLongGapY
        else:
            assert tr[(0, 2, 1)] == pm.gapShortSwitchToX and tr[(2, 1, 2)] == pm.gapShortSwitchToY
        for x in range(4):
            assert om.gapXEm[x] == pm.emissionGapX[x] and om.gapYEm[x] == pm.emissionGapY[x]
            for y in range(4):
                assert om.matchEm[x * 5 + y] == pm.emissionMatch[x * 4 + y]


@pytest.mark.parametrize("mtype", [0, 1, 2, 3])
def test_model_from_hmm_matches_oracle(mtype):
    rng = random.Random(40 + mtype)
    for _ in range(20):
        ph, oh = api.hmm_constructEmpty(0.0, mtype), ob.hmm(mtype, 0.0)
        S = ph.stateNumber
        for i in range(S * S):
            ph.transitions[i] = oh.T[i] = rng.random()
        for i in range(S * 16):
            ph.emissions[i] = oh.E[i] = rng.random()
        api.hmm_normalise(ph)
        ob.lib().orc_hmm_normalise(oh)
        assert list(ph.transitions)[:S * S] == list(oh.T)[:S * S]
        pm, om = api.hmm_getStateMachine(ph), ob.model_from_hmm(oh)
        tr = _oracle_model_as_dict(om)
        assert tr[(1, 0, 0)] == pm.matchContinue
        assert tr[(0, 0, 1)] == pm.gapShortOpenX and tr[(2, 0, 2)] == pm.gapShortOpenY
        assert tr[(0, 1, 1)] == pm.gapShortExtendX and tr[(2, 2, 2)] == pm.gapShortExtendY
        assert tr[(1, 1, 0)] == pm.matchFromShortGapX and tr[(1, 2, 0)] == pm.matchFromShortGapY
        if S == 5:
            assert tr[(0, 0, 3)] == pm.gapLongOpenX and tr[(2, 0, 4)] == pm.gapLongOpenY
            assert tr[(0, 3, 3)] == pm.gapLongExtendX and tr[(2, 4, 4)] == pm.gapLongExtendY
            assert tr[(1, 3, 0)] == pm.matchFromLongGapX and tr[(1, 4, 0)] == pm.matchFromLongGapY
        else:
            assert tr[(0, 2, 1)] == pm.gapShortSwitchToX and tr[(2, 1, 2)] == pm.gapShortSwitchToY
        for x in range(4):
            assert om.gapXEm[x] == pm.emissionGapX[x] and om.gapYEm[x] == pm.emissionGapY[x]
            for y in range(4):
                assert om.matchEm[x * 5 + y] == pm.emissionMatch[x * 4 + y]


@pytest.mark.parametrize("mtype", [0, 1, 2, 3])
def test_hmm_write_load_normalise(tmp_path, mtype):
    # tests/pairwiseAlignerTest.c:997-1073
    h = api.hmm_constructEmpty(0.0, mtype)
    S = h.stateNumber
    for i in range(S * S):
        h.transitions[i] += i
    for i in range(S * 16):
        h.emissions[i] += i
    path = str(tmp_path / "t.hmm")
    api.hmm_write(h, path)
    g = api.hmm_loadFromFile(path)
    assert g.type == mtype and g.stateNumber == S
    assert list(g.transitions)[:S * S] == [float(i) for i in range(S * S)]
    assert list(g.emissions)[:S * 16] == [float(i) for i in range(S * 16)]
    api.hmm_normalise(g)
    for f in range(S):
        z = f * S * S + (S * (S - 1)) // 2
        for to in range(S):
            assert g.transitions[f * S + to] == (f * S + to) / z


def test_parameter_defaults():
    p = api.pairwiseAlignmentBandingParameters_construct()
    assert (p.threshold, p.minDiagsBetweenTraceBack, p.traceBackDiagonals, p.diagonalExpansion,
            p.splitMatrixBiggerThanThis, p.dynamicAnchorExpansion) == (0.01, 1000, 40, 20, 9000000, 0)


def test_compute_fails_loudly_without_gpu():
    if api.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(api.CpecanError):
        api.getAlignedPairsUsingAnchors(api.stateMachine5_construct(), "ACGT", "ACGT", (),
                                        api.pairwiseAlignmentBandingParameters_construct())
    # adding problems and planning are host-only; binding the GPU (upload) is where it must fail
    b = api.Batch(api.stateMachine5_construct())
    b.add("ACGT", "ACGT")
    with pytest.raises(api.CpecanError, match="no usable HIP device"):
        b.upload()
    with pytest.raises(api.CpecanError):
        b.run()
    e = api.Batch(api.stateMachine5_construct())
    with pytest.raises(api.CpecanError):
        e.upload()  # even an empty batch refuses to pretend it ran


def test_planning_cell_counts_match_oracle_band():
    """Host planning (bands, split regions) runs without a GPU; its cell counts equal the oracle's."""
    from cpecan_amd.workload import make_batch
    probs = make_batch(3, 6, 700, 30)
    kw = dict(diagonalExpansion=30, splitMatrixBiggerThanThis=2000)
    b = api.Batch(api.stateMachine5_construct(), api.pairwiseAlignmentBandingParameters_construct(**kw))
    for sx, sy, a in probs:
        b.add(sx, sy, a, True, False)
    try:
        b.upload()
    except api.CpecanError:
        pass  # no GPU here: planning has run, the device step refused
    st = b.stats()
    want = sum(ob.band_cells(sx, sy, a, ob.params(**kw), True, False) for sx, sy, a in probs)
    assert st.cells == want and st.problems == 6 and st.regions >= 6


def test_planning_survives_dense_traceback_schedules():
    """minDiagsBetweenTraceBack just above traceBackDiagonals + 1: a traceback point nearly every diagonal (the segment
    table is sized by that spacing, not by minDiagsBetweenTraceBack alone)."""
    from cpecan_amd.workload import make_batch
    probs = make_batch(5, 3, 700, 10)
    kw = dict(diagonalExpansion=10, traceBackDiagonals=3, minDiagsBetweenTraceBack=5)
    b = api.Batch(api.stateMachine5_construct(), api.pairwiseAlignmentBandingParameters_construct(**kw))
    for sx, sy, a in probs:
        b.add(sx, sy, a)
    try:
        b.upload()
    except api.CpecanError:
        pass
    assert b.stats().cells == sum(ob.band_cells(sx, sy, a, ob.params(**kw)) for sx, sy, a in probs)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cpecan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".inl", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_binding" not in text and "liboracle" not in text and "cpecan_oracle" not in text, f


def test_anchors_from_alignment():
    """convertPairwiseForwardStrandAlignmentToAnchorPairs (impl/pairwiseAligner.c:979-1003) and the exact-match filter of
    cPecanRealign.c:277-281, worked by hand: host-side integer code, no GPU."""
    #      x: 0123 4567     y: 01 234567 (two bases inserted in Y after 2 columns, one base of X skipped later)
    ops = [(api.OP_MATCH, 2), (api.OP_INDEL_X, 0), (api.OP_INDEL_Y, 2), (api.OP_MATCH, 3), (api.OP_INDEL_X, 1), (api.OP_MATCH, 2)]
    got = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(ops, 0, 0, 0, 4)
    assert got.tolist() == [[0, 0, 4], [1, 1, 4], [2, 4, 4], [3, 5, 4], [4, 6, 4], [6, 7, 4], [7, 8, 4]]
    # trim removes `trim` columns at both ends of every match operation (:987)
    got = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(ops, 10, 20, 1, 0)
    assert got.tolist() == [[13, 25, 0]]
    # the filter keeps equal letters (case-insensitive), never N
    sx, sy = "ACgTNAGT", "AcxxGANAG"
    got = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(ops, 0, 0, 0, 2, sx, sy)
    want = [(x, y) for x, y in [(0, 0), (1, 1), (2, 4), (3, 5), (4, 6), (6, 7), (7, 8)]
            if sx[x].upper() == sy[y].upper() and sx[x].upper() != "N"]
    assert [tuple(r[:2]) for r in got.tolist()] == want and want == [(0, 0), (1, 1), (2, 4)]
    with pytest.raises(api.CpecanError):
        api.convertPairwiseForwardStrandAlignmentToAnchorPairs([(7, 1)], 0, 0, 0, 0)


def test_add_many_matches_one_by_one_planning():
    """cpecan_batch_add_many (two parallel passes over the problems) lays the batch out exactly as repeated
    cpecan_batch_add does: same regions, same cell counts per problem, same errors."""
    import random as _r
    rng = _r.Random(9)
    probs = []
    for i in range(150):
        lX, lY = rng.randrange(0, 400), rng.randrange(0, 400)
        sx = "".join(rng.choice("ACGTNacgt") for _ in range(lX))
        sy = "".join(rng.choice("ACGTNacgt") for _ in range(lY))
        anchors, x, y = [], -1, -1
        while rng.random() < 0.8:
            x += rng.randrange(1, 90)
            y += rng.randrange(1, 90)
            if x >= lX or y >= lY:
                break
            anchors.append((x, y, 10))
        probs.append((sx, sy, anchors, rng.random() < 0.5, rng.random() < 0.5))
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=10, splitMatrixBiggerThanThis=2500)
    with api.Batch(api.stateMachine5_construct(), p) as one, api.Batch(api.stateMachine5_construct(), p) as many:
        for pr in probs:
            one.add(*pr)
        assert many.add_many(probs[:100]) == 0
        assert many.add_many(probs[100:]) == 100
        assert many.add_many([]) == 150
        L = api.lib()
        for b in (one, many):  # the upload fails for want of a GPU, after the planning has been checked by the host code
            if L.cpecan_device_count() <= 0:
                with pytest.raises(api.CpecanError):
                    b.upload()
        s1, s2 = one.stats(), many.stats()
        assert (s1.problems, s1.regions) == (s2.problems, s2.regions) or L.cpecan_device_count() <= 0
    with api.Batch(api.stateMachine5_construct(), p) as b:
        with pytest.raises(api.CpecanError):
            b.add_many([("ACGT", "ACGT", [(1, 1, 0)]), ("ACGT", "ACGT", [(2, 2, 0), (1, 3, 0)])])  # anchors not increasing
        assert b.add_many([("ACGT", "ACGT", [(1, 1, 0)])]) == 0  # the failed call left the batch untouched


def test_filter_to_remove_overlap():
    """filterToRemoveOverlap (impl/pairwiseAligner.c:1095-1135) as tests/pairwiseAlignerTest.c:496-553 tests it: random
    subsets of a grid, sorted; the survivors are strictly increasing in both coordinates and are exactly the pairs that
    no OTHER pair overlaps (the reference's own check compares every pair with itself too, which only an empty result
    satisfies); the product against the oracle's two-pass restatement, expansions carried through."""
    rng = random.Random(12)
    some_kept = 0
    for trial in range(150):
        lX, lY = rng.randrange(0, 40), rng.randrange(0, 40)
        accept = rng.random() ** 3  # mostly sparse grids: dense ones keep nothing
        e = 2 * rng.randrange(0, 5)
        pairs = [(x, y, e) for x in range(lX) for y in range(lY) if rng.random() < accept]
        if trial % 2 and lX and lY:  # a handful of cells: the case in which pairs do survive
            pairs = sorted({(rng.randrange(lX), rng.randrange(lY), e) for _ in range(rng.randrange(0, 7))})
        got = [tuple(int(v) for v in r) for r in api.filterToRemoveOverlap(pairs)]
        want = [tuple(int(v) for v in r) for r in ob.filter_to_remove_overlap(pairs)]
        assert got == want
        assert all(a[0] < b[0] and a[1] < b[1] for a, b in zip(got, got[1:]))
        brute = [p for i, p in enumerate(pairs)
                 if not any(j != i and ((q[0] <= p[0] and q[1] >= p[1]) or (q[0] >= p[0] and q[1] <= p[1]))
                            for j, q in enumerate(pairs))]
        assert got == brute
        some_kept += bool(got)
    assert some_kept > 30
    assert api.filterToRemoveOverlap([]).shape == (0, 3)
    with pytest.raises(api.CpecanError):
        api.filterToRemoveOverlap([(5, 5, 0), (4, 9, 0)])  # not sorted (asserted in the reference, :1124)


def test_hmm_loadFromFile_reads_the_reference_trained_hmm_text():
    """The on-disk layout is the reference's (stateMachine.c:133-202), not merely what this build's writer emits: the
    trained five-state-asymmetric HMM that cPecanEmTest.py:112-113 holds as text (space separated, 12 significant digits)
    loads to exactly the numbers in the text, and hmm_getStateMachine follows stateMachine5_loadAsymmetric (:529-575)."""
    import math
    import reference_cases as rc
    mtype, T, lik, E = rc.trained_hmm_numbers()
    h = api.hmm_loadFromFile(rc.TRAINED_HMM)
    assert h.type == api.fiveStateAsymmetric == mtype and h.stateNumber == 5
    assert list(h.transitions) == T and list(h.emissions) == E and h.likelihood == lik
    m = api.hmm_getStateMachine(h)
    assert m.matchContinue == math.log(T[0])
    assert m.gapShortOpenX == math.log(T[1]) and m.gapShortOpenY == math.log(T[2])
    assert m.gapLongOpenX == math.log(T[3]) and m.gapLongOpenY == math.log(T[4])
    assert m.matchFromShortGapX == math.log(T[5]) and m.gapShortExtendX == math.log(T[6])
    assert m.matchFromShortGapY == math.log(T[10]) and m.gapShortExtendY == math.log(T[12])
    assert m.matchFromLongGapX == math.log(T[15]) and m.gapLongExtendX == math.log(T[18])
    assert m.matchFromLongGapY == math.log(T[20]) and m.gapLongExtendY == math.log(T[24])
    for x in range(4):
        for y in range(4):
            assert m.emissionMatch[x * 4 + y] == math.log(E[x * 4 + y])
    # the oracle builds the same model from the same numbers
    oh = ob.hmm(ob.FIVE_STATE_ASYM, 0.0)
    for i, v in enumerate(T):
        oh.T[i] = v
    for i, v in enumerate(E):
        oh.E[i] = v
    om = ob.model_from_hmm(oh)
    for x in range(4):
        assert abs(m.emissionGapX[x] - om.gapXEm[x]) < 1e-15 and abs(m.emissionGapY[x] - om.gapYEm[x]) < 1e-15


def test_planning_is_the_same_for_runs_triples_and_both_band_walks(monkeypatch):
    """Host planning needs no GPU (cpecan_batch_upload plans, then fails for want of a device here): the band cells,
    regions and diagonals of a batch are the same whether the anchors come as runs (cpecan_batch_add_many_runs) or as one
    triple per column, and whether planning walks the runs of diagonal-neighbour anchors in closed form or every diagonal
    (CPECAN_FAST_WALK=0) -- expansions 2 to 10, default and short traceback schedules, split rectangles, anchors from the
    first column on.  (On the GPU box tests/test_gpu_parity.py compares the lists as well; tools/asan_cpu.sh runs this under
    AddressSanitizer.)"""
    import random
    from cpecan_amd import workload
    if api.device_count() > 0:
        pytest.skip("a GPU is present: the upload succeeds (covered by the GPU tests)")
    rng = random.Random(5)
    for E, seed, kw in ((4, 21, {}), (2, 22, dict(minDiagsBetweenTraceBack=150, traceBackDiagonals=21)),
                        (10, 23, dict(minDiagsBetweenTraceBack=64, traceBackDiagonals=40)), (6, 24, dict(splitMatrixBiggerThanThis=10 ** 12))):
        probs = workload.make_realign_batch(seed, 120, 30, 4000, expansion=E)
        same = "".join(rng.choice("ACGT") for _ in range(3000))
        probs.append((same, same, np.array([(i, i, E) for i in range(len(same))], dtype=np.int64)))
        problems = [(sx, sy, a, True, True) for sx, sy, a in probs]
        pkw = dict(diagonalExpansion=E, splitMatrixBiggerThanThis=10)
        pkw.update(kw)
        p = api.pairwiseAlignmentBandingParameters_construct(**pkw)
        seen = set()
        for runs in (False, True):
            for walk in ("0", "1"):
                monkeypatch.setenv("CPECAN_FAST_WALK", walk)
                with api.Batch(api.stateMachine5_construct(), p) as b:
                    (b.add_many_runs if runs else b.add_many)(problems)
                    with pytest.raises(api.CpecanError):
                        b.upload()
                    st = b.stats()
                    seen.add((st.problems, st.regions, st.cells, st.diagonals))
        assert len(seen) == 1 and next(iter(seen))[2] > 0, seen
