"""GPU parity tests of the list consumers: reweightAlignedPairs2, posterior scores, the MEA chain and
leftShiftAlignment, device kernels against the oracle's restatements.  Integer / order-defined arithmetic: bit-exact."""
import json
import os
import random

import numpy as np
import pytest

import oracle_binding as ob
from cpecan_amd import api
from test_gpu_parity import _evolve, _rand_anchors, _rand_seq, _sm

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_fixtures.json")))


def _problems(rng, n, lo=20, hi=160):
    out = []
    for _ in range(n):
        sx = _rand_seq(rng, rng.randrange(lo, hi))
        sy = _evolve(rng, sx)
        if not sy:
            sy = "A"
        out.append((sx, sy, _rand_anchors(rng, len(sx), len(sy)) if rng.random() > 0.5 else []))
    return out


def test_left_shift_golden_through_the_gpu():
    """tests/pairwiseAlignerTest.c:944-995 (test_leftShiftAlignment)."""
    fx = GOLD["test_leftShiftAlignment"]
    pairs = [(fx["score"], x, y) for x, y in zip(fx["alignedX"], fx["alignedY"])]
    got = api.leftShiftAlignment(pairs, fx["seqX"], fx["seqY"])
    assert got[:, 1].tolist() == fx["shiftedX"] and got[:, 2].tolist() == fx["shiftedY"]
    assert got[:, 0].tolist() == [1] * 9
    # empty input: the boundary loop alone, score 1 (:1754)
    got = api.leftShiftAlignment([], "ACGT", "TTGT")
    assert got.tolist() == [[1, 2, 2], [1, 3, 3]]


def test_reweight_and_scores_standalone_vs_oracle():
    rng = random.Random(11)
    p = api.pairwiseAlignmentBandingParameters_construct()
    for sx, sy, a in _problems(rng, 12):
        pairs = api.getAlignedPairsUsingAnchors(_sm(0), sx, sy, a, p)
        for gamma in (0.5, float(np.float32(0.85)), 2.0, 0.0, -1.0):
            got = api.reweightAlignedPairs2(pairs, len(sx), len(sy), gamma)
            want = ob.reweight_aligned_pairs(pairs, len(sx), len(sy), gamma)
            assert (got.astype(np.int64) == want).all()
        assert api.scoreByPosteriorProbability(len(sx), len(sy), pairs) == ob.score_by_posterior(len(sx), len(sy), pairs)
        if len(pairs):
            assert api.scoreByPosteriorProbabilityIgnoringGaps(pairs) == ob.score_by_posterior_ignoring_gaps(pairs)


def test_batch_reweight_matches_standalone_and_oracle():
    rng = random.Random(12)
    probs = _problems(rng, 40)
    p = api.pairwiseAlignmentBandingParameters_construct(splitMatrixBiggerThanThis=400)  # split regions too
    gamma = float(np.float32(0.5))

    def run(flags):
        with api.Batch(_sm(0), p) as b:
            b.set_post(flags, gamma)
            for sx, sy, a in probs:
                b.add(sx, sy, a, True, True)
            b.upload()
            b.run()
            b.download()
            return [b.result(i) for i in range(len(probs))], [b.scores(i) for i in range(len(probs))]

    plain, plain_scores = run(0)
    rew, rew_scores = run(api.POST_REWEIGHT)
    for i, (sx, sy, a) in enumerate(probs):
        want = ob.reweight_aligned_pairs(plain[i], len(sx), len(sy), gamma)
        assert (rew[i].astype(np.int64) == want).all()
        assert plain_scores[i][0] == ob.score_by_posterior(len(sx), len(sy), plain[i])
        assert rew_scores[i][0] == ob.score_by_posterior(len(sx), len(sy), want)
        if len(want):
            assert rew_scores[i][1] == ob.score_by_posterior_ignoring_gaps(want)


def test_mea_and_left_shift_vs_oracle():
    rng = random.Random(13)
    probs = _problems(rng, 30)
    p = api.pairwiseAlignmentBandingParameters_construct()
    gamma = float(np.float32(0.5))
    with api.Batch(_sm(0), p, emit=api.EMIT_INDEL) as b:
        b.set_post(api.POST_MEA | api.POST_LEFT_SHIFT, gamma)
        for sx, sy, a in probs:
            b.add(sx, sy, a)
        b.upload()
        b.run()
        b.download()
        for i, (sx, sy, a) in enumerate(probs):
            m, gx, gy = b.result(i, 0), b.result(i, 1), b.result(i, 2)
            mea, score = ob.mea_alignment(m, gx, gy, len(sx), len(sy), gamma)
            want = ob.left_shift_alignment(mea, sx, sy)
            got = b.result(i, 3)
            assert (got.astype(np.int64) == want).all(), i
            assert b.scores(i)[2] == score
            # the stages on their own
            mea_gpu, score_gpu = api.getMaximalExpectedAccuracyPairwiseAlignment(m, gx, gy, len(sx), len(sy), gapGamma=gamma)
            assert (mea_gpu.astype(np.int64) == mea).all() and score_gpu == score
            assert (api.leftShiftAlignment(mea, sx, sy).astype(np.int64) == want).all()
    # getShiftedMEAAlignment end to end on one problem
    sx, sy, a = probs[0]
    got, score = api.getShiftedMEAAlignment(sx, sy, a, p, _sm(0), gapGamma=gamma)
    om, op = ob.model(0), ob.params()
    m, gx, gy = ob.aligned_pairs_with_indels(om, sx, sy, a, op)
    mea, _ = ob.mea_alignment(m, gx, gy, len(sx), len(sy), gamma)
    want = ob.left_shift_alignment(mea, sx, sy)
    assert got[:, 1:].tolist() == want[:, 1:].tolist()  # scores may differ by one unit of 1e-7 (device exp vs libm)


def test_mea_wave_and_lane_kernels_agree_on_long_walks(monkeypatch):
    """getMaximalExpectedAccuracyPairwiseAlignment with one wave per problem (default) against one lane per problem
    (CPECAN_POST_LANES=1) and the oracle, on lists built so that the walk back from a pair runs over hundreds of earlier
    pairs (few records): beyond the 64 pairs the wave keeps in registers, with exact ties in the weights."""
    rng = random.Random(41)
    cases = []
    for lX, lY, n_pairs, coarse in ((400, 380, 900, False), (300, 320, 700, True), (70, 60, 64, False), (1500, 1500, 2600, False)):
        cells = set()
        for i in range(min(lX, lY)):  # a noisy diagonal, then scattered pairs; list order = anti-diagonal, as the emitters give it
            cells.add((i, min(lY - 1, max(0, i + rng.randrange(-2, 3)))))
        while len(cells) < n_pairs:
            cells.add((rng.randrange(lX), rng.randrange(lY)))
        cells = sorted(cells, key=lambda c: (c[0] + c[1], -(c[0] - c[1])))
        pairs = [((rng.randrange(1, 6) * 2000000) if coarse else rng.randrange(1, 10000001), x, y) for x, y in cells]
        # heavy gap mass: extending the chain soon stops setting records, so later walks run far back
        gx = [(rng.randrange(1000000, 9000000), x, rng.randrange(lY)) for x in range(lX) for _ in range(2)]
        gy = [(rng.randrange(1000000, 9000000), rng.randrange(lX), y) for y in range(lY) for _ in range(2)]
        cases.append((pairs, gx, gy, lX, lY))
    gamma = float(np.float32(0.5))
    for pairs, gx, gy, lX, lY in cases:
        want, want_score = ob.mea_alignment(pairs, gx, gy, lX, lY, gamma)
        for lanes in (None, "1"):
            if lanes:
                monkeypatch.setenv("CPECAN_POST_LANES", lanes)
            else:
                monkeypatch.delenv("CPECAN_POST_LANES", raising=False)
            got, score = api.getMaximalExpectedAccuracyPairwiseAlignment(pairs, gx, gy, lX, lY, gapGamma=gamma)
            assert (np.asarray(got).reshape(-1, 3).astype(np.int64) == want).all() and score == want_score, (lX, lanes)


def test_consumer_argument_errors():
    p = api.pairwiseAlignmentBandingParameters_construct()
    with api.Batch(_sm(0), p) as b:
        with pytest.raises(api.CpecanError):
            b.set_post(api.POST_MEA, 0.5)  # needs the gap lists (EMIT_INDEL)
        with pytest.raises(api.CpecanError):
            b.set_post(api.POST_LEFT_SHIFT, 0.5)
    with pytest.raises(api.CpecanError):
        api.reweightAlignedPairs2([(5, 3, 0)], 2, 2, 0.5)  # x outside the sequence


def test_realign_flow_end_to_end():
    """The cPecanRealign loop (cPecanRealign.c:509-553) as one batch: anchors from an existing alignment's operations
    with the exact-match filter, posteriors with expansion 4 / split at 10 / ragged ends, reweighting on the device --
    against the same steps through the oracle."""
    rng = random.Random(17)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=4, splitMatrixBiggerThanThis=10)
    op = ob.params(diagonalExpansion=4, splitMatrixBiggerThanThis=10)
    gamma = float(np.float32(0.2))
    cases = []
    for _ in range(30):
        # a "true" alignment as a list of operations, and two sequences that follow it with some substitutions
        ops, sx, sy = [], [], []
        for _ in range(rng.randrange(3, 25)):
            kind = rng.choice([api.OP_MATCH] * 4 + [api.OP_INDEL_X, api.OP_INDEL_Y])
            n = rng.randrange(1, 30) if kind == api.OP_MATCH else rng.randrange(1, 4)
            ops.append((kind, n))
            for _ in range(n):
                b = rng.choice("ACGT")
                if kind != api.OP_INDEL_Y:
                    sx.append(b)
                if kind != api.OP_INDEL_X:
                    sy.append(b if kind != api.OP_MATCH or rng.random() > 0.1 else rng.choice("ACGTN"))
        sx, sy = "".join(sx), "".join(sy)
        if not sx or not sy:
            continue
        anchors = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(ops, 0, 0, 0, 4, sx, sy)
        assert all(sx[x].upper() == sy[y].upper() != "N" for x, y, _ in anchors.tolist())
        cases.append((sx, sy, [tuple(a) for a in anchors.tolist()]))
    with api.Batch(_sm(0), p) as b:
        b.set_post(api.POST_REWEIGHT, gamma)
        for sx, sy, a in cases:
            b.add(sx, sy, a, True, True)
        b.upload()
        b.run()
        b.download()
        for i, (sx, sy, a) in enumerate(cases):
            want = ob.aligned_pairs(ob.model(0), sx, sy, a, op, True, True)
            got = b.result(i)
            assert got[:, 1:].tolist() == want[:, 1:].tolist()
            # scores may differ by one unit before reweighting (device exp vs libm): reweight what the GPU produced
            plain = api.getAlignedPairsUsingAnchors(_sm(0), sx, sy, a, p, True, True)
            assert (got.astype(np.int64) == ob.reweight_aligned_pairs(plain, len(sx), len(sy), gamma)).all()


def _as_tuples(a):
    return [tuple(int(v) for v in r) for r in np.asarray(a).reshape(-1, 3)]


def test_filter_pairs_ordered_vs_oracle():
    """filterPairwiseAlignmentToMakePairsOrdered (impl/multipleAligner.c:945-972): the device's Fenwick-tree chain against
    the oracle's frontier restatement of pairwiseAlignColumns, on random lists with many exact ties and on real posterior
    lists; identical lists in identical order."""
    rng = random.Random(23)
    for trial in range(120):
        lX, lY = rng.randrange(1, 60), rng.randrange(1, 60)
        cells = list({(rng.randrange(lX), rng.randrange(lY)) for _ in range(rng.randrange(0, 200))})
        rng.shuffle(cells)
        coarse = trial % 2 == 0
        pairs = [((rng.randrange(0, 11) * 1000000) if coarse else rng.randrange(-1000, 10000001), x, y) for x, y in cells]
        gamma = rng.choice([0.0, 0.1, 0.5, 0.85])
        got = api.filterPairwiseAlignmentToMakePairsOrdered(pairs, "A" * lX, "A" * lY, gamma)
        want = ob.filter_pairs_ordered(pairs, lX, lY, gamma)
        assert _as_tuples(got) == _as_tuples(want)
    p = api.pairwiseAlignmentBandingParameters_construct(threshold=0.001)
    for sx, sy, a in _problems(rng, 10, 60, 300):
        pairs = api.getAlignedPairsUsingAnchors(_sm(0), sx, sy, a, p)
        for gamma in (0.0, 0.3, 0.85):
            got = api.filterPairwiseAlignmentToMakePairsOrdered(pairs, sx, sy, gamma)
            want = ob.filter_pairs_ordered(pairs, len(sx), len(sy), gamma)
            assert _as_tuples(got) == _as_tuples(want)
            chain = sorted(_as_tuples(got), key=lambda t: t[1])
            assert all(u[1] < v[1] and u[2] < v[2] for u, v in zip(chain, chain[1:]))  # checkAlignment's property


def test_filter_pairs_ordered_wave_and_lane_kernels_agree(monkeypatch):
    """One wave per problem (default) against one lane per problem (CPECAN_POST_LANES=1) and the oracle: columns of more
    than 64 pairs, a list longer than the 2047 chain slots the wave keeps in LDS, lists in shuffled order."""
    rng = random.Random(37)
    cases = []
    lX, lY = 3, 200  # every cell of three columns: 200 pairs a column
    cases.append(([(rng.randrange(1, 10000001), x, y) for x in range(lX) for y in range(lY)], lX, lY))
    lX = lY = 3000  # a noisy diagonal and scattered pairs: ~5000 in all
    cells = {(i, min(lY - 1, max(0, i + rng.randrange(-3, 4)))) for i in range(lX) for _ in range(2)}
    cells |= {(rng.randrange(lX), rng.randrange(lY)) for _ in range(1500)}
    cells = list(cells)
    rng.shuffle(cells)
    cases.append(([(rng.randrange(1, 10000001), x, y) for x, y in cells], lX, lY))
    cases.append(([(rng.randrange(0, 3) * 5000000, x, y) for x, y in cells[:2500]], lX, lY))  # many exact ties
    for pairs, lX, lY in cases:
        for gamma in (0.0, 0.4):
            want = _as_tuples(ob.filter_pairs_ordered(pairs, lX, lY, gamma))
            monkeypatch.delenv("CPECAN_POST_LANES", raising=False)
            assert _as_tuples(api.filterPairwiseAlignmentToMakePairsOrdered(pairs, "A" * lX, "A" * lY, gamma)) == want
            monkeypatch.setenv("CPECAN_POST_LANES", "1")
            assert _as_tuples(api.filterPairwiseAlignmentToMakePairsOrdered(pairs, "A" * lX, "A" * lY, gamma)) == want


def test_identity_scores_vs_oracle():
    rng = random.Random(29)
    p = api.pairwiseAlignmentBandingParameters_construct()
    for sx, sy, a in _problems(rng, 10):
        sx = "".join(c.lower() if rng.random() < 0.2 else ("N" if rng.random() < 0.05 else c) for c in sx)
        pairs = api.getAlignedPairsUsingAnchors(_sm(0), sx, sy, a, p)
        assert api.scoreByIdentity(sx, sy, len(sx), len(sy), pairs) == ob.score_by_identity(sx, sy, pairs)
        if len(pairs):
            assert api.scoreByIdentityIgnoringGaps(sx, sy, pairs) == ob.score_by_identity_ignoring_gaps(sx, sy, pairs)
    assert api.scoreByIdentity("ACGT", "ACGT", 4, 4, []) == 0.0


def test_batch_realign_step_reweight_then_ordered():
    """cPecanRealign.c:552-563 as one consumer stage of the batch: reweightAlignedPairs2, then
    filterPairwiseAlignmentToMakePairsOrdered, then the four scores of the ordered list."""
    rng = random.Random(31)
    probs = _problems(rng, 40, 30, 220)
    p = api.pairwiseAlignmentBandingParameters_construct(splitMatrixBiggerThanThis=900)
    gap_gamma, match_gamma = float(np.float32(0.5)), 0.6
    with api.Batch(_sm(0), p) as b:
        b.set_post(api.POST_REWEIGHT | api.POST_ORDERED, gap_gamma, match_gamma)
        for sx, sy, a in probs:
            b.add(sx, sy, a, True, True)
        b.upload()
        b.run()
        b.download()
        for i, (sx, sy, a) in enumerate(probs):
            reweighted = b.result(i)  # list 0, as the device reweighted it
            want = ob.filter_pairs_ordered(reweighted, len(sx), len(sy), match_gamma)
            got = b.result(i, 3)
            assert _as_tuples(got) == _as_tuples(want)
            by_post, by_post_ig, _ = b.scores(i)
            by_id, by_id_ig = b.identity_scores(i)
            assert by_post == ob.score_by_posterior(len(sx), len(sy), want)
            assert by_id == ob.score_by_identity(sx, sy, want)
            if len(want):
                assert by_post_ig == ob.score_by_posterior_ignoring_gaps(want)
                assert by_id_ig == ob.score_by_identity_ignoring_gaps(sx, sy, want)
            else:
                assert np.isnan(by_post_ig) and np.isnan(by_id_ig)  # 0 / 0, as in the reference
    with api.Batch(_sm(0), p, api.EMIT_INDEL) as b:
        with pytest.raises(api.CpecanError):
            b.set_post(api.POST_ORDERED | api.POST_MEA, 0.5)
