"""GPU tests of the realign front end (include/cpecan_realign.h; cPecanRealign.c:509-600), one batch per call.

What the reference's own tests of cPecanRealign check (cPecanRealignTest.py:20-113) is repeated on synthetic cigars:
--rescoreOriginalAlignment gives the input alignment back, the default mode keeps the coordinates, every rescoring mode
gives a score in [0, 100].  On top of that the loop is compared, cigar by cigar, with the same steps made one at a time
through the single-problem GPU entry points and the cigar helpers (each of which has its own parity test)."""
import os
import random
import subprocess

import numpy as np
import pytest

from cpecan_amd import api
from cpecan_amd.realign import Cigar, Realigner, realign_options

pytestmark = pytest.mark.gpu

M, DX, IY = api.OP_MATCH, api.OP_INDEL_X, api.OP_INDEL_Y
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIFT = int(os.environ.get("CPECAN_SEED_SHIFT", "0"))  # soak runs: other random worlds
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a"}


def _revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def _world(rng, n_seqs=6, n_cigars=40):
    """Sequences by name and cigars over parts of them: a 'true' alignment as operations (first and last a match, no two
    neighbours of one type), the aligned stretch of Y derived from X's with substitutions, either strand."""
    seqs, cigars = {}, []
    flank = lambda: "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 30)))
    for k in range(n_cigars):
        ops, last = [], None
        n_ops = 2 * rng.randrange(0, 8) + 1
        for i in range(n_ops):
            t = M if i % 2 == 0 else rng.choice([DX, IY])
            ops.append((t, rng.randrange(1, 40) if t == M else rng.choice([1, 1, 2, 3, 6, 15])))
        sx, sy = [], []
        for t, n in ops:
            for _ in range(n):
                b = rng.choice("ACGT")
                if t != IY:
                    sx.append(b if rng.random() > 0.03 else b.lower())
                if t != DX:
                    sy.append(b if t != M or rng.random() > 0.12 else rng.choice("ACGTN"))
        sx, sy = "".join(sx), "".join(sy)
        strand1, strand2 = rng.random() < 0.75, rng.random() < 0.6
        lx, ly = flank(), flank()
        name1, name2 = "X%d" % k, "Y%d" % k
        seqs[name1] = lx + (sx if strand1 else _revcomp(sx)) + flank()
        seqs[name2] = ly + (sy if strand2 else _revcomp(sy)) + flank()
        a = (len(lx), len(lx) + len(sx)) if strand1 else (len(lx) + len(sx), len(lx))
        b = (len(ly), len(ly) + len(sy)) if strand2 else (len(ly) + len(sy), len(ly))
        cigars.append(Cigar(name1, a[0], a[1], strand1, name2, b[0], b[1], strand2, float(rng.randrange(1, 5000)), ops))
    return seqs, cigars


def _sub(seqs, c):
    x = seqs[c.contig1][min(c.start1, c.end1):max(c.start1, c.end1)]
    y = seqs[c.contig2][min(c.start2, c.end2):max(c.start2, c.end2)]
    return (x if c.strand1 else _revcomp(x)), (y if c.strand2 else _revcomp(y))


def _realigner(seqs, **opts):
    r = Realigner(options=realign_options(**opts))
    for name, s in seqs.items():
        r.add_sequence(name + " a description", s)
    return r


def _stepwise(seqs, c, o):
    """One cigar through cPecanRealign.c:511-581 with one GPU call per step."""
    sx, sy = _sub(seqs, c)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=o.params.diagonalExpansion,
                                                         splitMatrixBiggerThanThis=o.params.splitMatrixBiggerThanThis)
    anchors = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(c.ops, 0, 0, o.constraintDiagonalTrim,
                                                                     o.params.diagonalExpansion, sx, sy)
    pairs = api.getAlignedPairsUsingAnchors(api.stateMachine5_construct(), sx, sy, [tuple(a) for a in anchors.tolist()], p,
                                            True, True)
    pairs = api.reweightAlignedPairs2(pairs, len(sx), len(sy), float(np.float32(o.gapGamma)))
    final = api.filterPairwiseAlignmentToMakePairsOrdered(pairs, sx, sy, o.matchGamma)
    xy = sorted((int(x), int(y)) for _, x, y in final.tolist())
    out = Cigar.from_aligned_pairs(c.contig1, c.contig2, c.score, len(sx), len(sy), xy)
    for side in ("1", "2"):  # rebasePairwiseAlignmentCoordinates (:232) back to the input's frame
        strand = getattr(c, "strand" + side)
        lo = min(getattr(c, "start" + side), getattr(c, "end" + side))
        s, e = getattr(out, "start" + side) + lo, getattr(out, "end" + side) + lo
        if not strand:
            s, e = e, s
        setattr(out, "start" + side, s)
        setattr(out, "end" + side, e)
        setattr(out, "strand" + side, strand)
    return out, final, (sx, sy)


def test_rescore_original_alignment_returns_the_input():
    """cPecanRealignTest.py:20-32 (testCPecanRealignDummy) and :76-103 (scores between 0 and 100)."""
    rng = random.Random(41 + SHIFT)
    seqs, cigars = _world(rng)
    with _realigner(seqs, rescoreOriginalAlignment=1) as r:
        assert r.realign(cigars) == cigars
    for flag in ("rescoreByIdentity", "rescoreByPosteriorProb", "rescoreByIdentityIgnoringGaps",
                 "rescoreByPosteriorProbIgnoringGaps"):
        with _realigner(seqs, rescoreOriginalAlignment=1, **{flag: 1}) as r:
            got = r.realign(cigars)
        for a, b in zip(got, cigars):
            assert a.same_coordinates(b) and a.ops == b.ops
            assert 0.0 <= a.score <= 100.0
        if flag == "rescoreByIdentityIgnoringGaps":  # every aligned column of the input that is a true match
            for a, c in zip(got, cigars):
                sx, sy = _sub(seqs, c)
                cols = [(int(x), int(y)) for x, y, _ in
                        api.convertPairwiseForwardStrandAlignmentToAnchorPairs(c.ops, 0, 0, 0, 4).tolist()]
                same = sum(sx[x].upper() == sy[y].upper() != "N" for x, y in cols)
                assert a.score == 100.0 * same / len(cols)


def test_realign_keeps_coordinates_and_matches_the_stepwise_flow():
    """cPecanRealignTest.py:34-45 (testCPecanRealign: same coordinates), then every cigar against the stepwise flow."""
    rng = random.Random(43 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=60)
    o = realign_options()
    with _realigner(seqs) as r:
        got = r.realign(cigars)
    assert len(got) == len(cigars)
    changed = 0
    for a, c in zip(got, cigars):
        assert a.same_coordinates(c) and a.score == c.score
        want, _, _ = _stepwise(seqs, c, o)
        assert a == want
        assert Cigar.parse(a.format()) == a  # consistent (checkPairwiseAlignment, :582)
        changed += a.ops != c.ops
    assert changed > 5  # the realignment does move indels
    # the four rescoring modes, scored on the final pairs (:556-564)
    score_fns = {
        "rescoreByPosteriorProb": lambda f, sx, sy: api.scoreByPosteriorProbability(len(sx), len(sy), f),
        "rescoreByPosteriorProbIgnoringGaps": lambda f, sx, sy: api.scoreByPosteriorProbabilityIgnoringGaps(f),
        "rescoreByIdentity": lambda f, sx, sy: api.scoreByIdentity(sx, sy, len(sx), len(sy), f),
        "rescoreByIdentityIgnoringGaps": lambda f, sx, sy: api.scoreByIdentityIgnoringGaps(sx, sy, f),
    }
    for flag, fn in score_fns.items():
        with _realigner(seqs, **{flag: 1}) as r:
            rescored = r.realign(cigars[:20])
        for a, c in zip(rescored, cigars):
            _, final, (sx, sy) = _stepwise(seqs, c, o)
            if len(final):
                assert a.score == fn(final, sx, sy)
                assert 0.0 <= a.score <= 100.0


def _oracle_flow(seqs, c, o):
    """One cigar through cPecanRealign.c:511-581 WITHOUT the library: anchors and cigar operations restated here in
    Python, posteriors / reweighting / ordered chain from the CPU oracle (tests/oracle_binding.py)."""
    import oracle_binding as orc
    sx, sy = _sub(seqs, c)
    trim, expansion = o.constraintDiagonalTrim, o.params.diagonalExpansion
    anchors, x, y = [], 0, 0
    for t, n in c.ops:  # aligned columns of the input, `trim` columns cut from both ends of a block, mismatches dropped (:277-281)
        if t == M:
            for l in range(trim, n - trim):
                a, b = sx[x + l].upper(), sy[y + l].upper()
                if a == b and a != "N":
                    anchors.append((x + l, y + l, expansion))
        x += n if t != IY else 0
        y += n if t != DX else 0
    m = orc.model(orc.FIVE_STATE)
    p = orc.params(diagonalExpansion=expansion, splitMatrixBiggerThanThis=o.params.splitMatrixBiggerThanThis)
    pairs = orc.aligned_pairs(m, sx, sy, anchors, p, True, True)
    pairs = orc.reweight_aligned_pairs(pairs, len(sx), len(sy), float(np.float32(o.gapGamma)))
    final = orc.filter_pairs_ordered(pairs, len(sx), len(sy), o.matchGamma)
    xy = sorted((int(px), int(py)) for _, px, py in np.asarray(final).reshape(-1, 3).tolist())
    ops, cx, cy, i = [], 0, 0, 0  # maximal diagonal runs; what lies between them is an X-indel, then a Y-indel (:49-96)
    while i < len(xy):
        x0, y0 = xy[i]
        run = 1
        while i + run < len(xy) and xy[i + run] == (x0 + run, y0 + run):
            run += 1
        if x0 > cx:
            ops.append((DX, x0 - cx))
        if y0 > cy:
            ops.append((IY, y0 - cy))
        ops.append((M, run))
        cx, cy, i = x0 + run, y0 + run, i + run
    if len(sx) > cx:
        ops.append((DX, len(sx) - cx))
    if len(sy) > cy:
        ops.append((IY, len(sy) - cy))
    return ops


def test_realign_matches_the_oracle_flow():
    """The batched realignment against a flow that shares no code with the library: Python for the cigar <-> columns
    conversions, the CPU oracle for posteriors, reweighting and the ordered chain.  Coordinates are the input's."""
    rng = random.Random(47 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=50)
    o = realign_options()
    with _realigner(seqs) as r:
        got = r.realign(cigars)
    for a, c in zip(got, cigars):
        assert a.same_coordinates(c)
        assert [tuple(op) for op in a.ops] == _oracle_flow(seqs, c, o), c.format()


def test_realign_options_change_the_flow_consistently():
    rng = random.Random(47 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=25)
    for opts in (dict(gapGamma=0.0, matchGamma=0.5), dict(constraintDiagonalTrim=2, diagonalExpansion=8),
                 dict(splitMatrixBiggerThanThis=400, gapGamma=0.9, matchGamma=0.0)):
        o = realign_options(**opts)
        with _realigner(seqs, **opts) as r:
            got = r.realign(cigars)
        for a, c in zip(got, cigars):
            assert a == _stepwise(seqs, c, o)[0]


def test_split_indels_longer_than_this():
    """--splitIndelsLongerThanThis (:584-591): the realigned cigar cut by splitPairwiseAlignment."""
    rng = random.Random(53 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=40)
    with _realigner(seqs) as r:
        whole = r.realign(cigars)
    with _realigner(seqs, splitIndelsLongerThanThis=4) as r:
        pieces = r.realign(cigars)
    want = [p for c in whole for p in c.split(4)]
    assert pieces == want and len(pieces) > len(whole)


def test_expectations_mode_sums_the_single_problem_expectations():
    """--outputExpectations (:530-534, :497): the batch's counts against getExpectationsUsingAnchors cigar by cigar."""
    rng = random.Random(59 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=30)
    o = realign_options()
    acc = api.hmm_constructEmpty(0.000000000001, api.fiveState)
    with _realigner(seqs) as r:
        r.expectations(cigars, acc)
    want = api.hmm_constructEmpty(0.000000000001, api.fiveState)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=4, splitMatrixBiggerThanThis=10)
    sm = api.stateMachine5_construct()
    for c in cigars:
        sx, sy = _sub(seqs, c)
        anchors = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(c.ops, 0, 0, 0, 4, sx, sy)
        api.getExpectationsUsingAnchors(sm, want, sx, sy, [tuple(a) for a in anchors.tolist()], p, True, True)
    np.testing.assert_allclose(list(acc.transitions), list(want.transitions), rtol=1e-9)
    np.testing.assert_allclose(list(acc.emissions), list(want.emissions), rtol=1e-9)
    np.testing.assert_allclose(acc.likelihood, want.likelihood, rtol=1e-9)


def test_command_line_end_to_end(tmp_path):
    """The cpecan_realign binary: fasta files + cigars on stdin -> cigars on stdout, identical to the library call; the
    posterior-probability files of the last cigar; the expectations file loads back as an HMM (cPecanEm's loop)."""
    exe = os.path.join(ROOT, "cpecan_amd", "cpecan_realign")
    assert os.path.exists(exe), "build the command line with make -C cpecan_amd/csrc"
    rng = random.Random(61 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=30)
    names = sorted(seqs)
    fa1, fa2 = tmp_path / "a.fa", tmp_path / "b.fa"
    for path, part in ((fa1, names[::2]), (fa2, names[1::2])):
        with open(path, "w") as f:
            for n in part:
                f.write(">%s some words\n" % n)
                for i in range(0, len(seqs[n]), 37):
                    f.write(seqs[n][i:i + 37] + "\n")
    text = "".join(c.format() + "\n" for c in cigars)
    post, allp = tmp_path / "post.tsv", tmp_path / "all.tsv"

    def run(*args):
        res = subprocess.run([exe, *args, str(fa1), str(fa2)], input=text, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        return [line for line in res.stdout.split("\n") if line]

    with _realigner(seqs, rescoreByPosteriorProb=1) as r:
        want = [c.format() for c in r.realign(cigars)]
    # a batch size that does not divide the input: several batches, same output
    got = run("--rescoreByPosteriorProb", "--batch", "7", "--outputPosteriorProbs", str(post), "--outputAllPosteriorProbs",
              str(allp))
    assert got == want
    _, final, (sx, sy) = _stepwise(seqs, cigars[-1], realign_options())
    c = cigars[-1]
    rows = [line.split("\t") for line in open(post).read().split("\n") if line]
    assert len(rows) == len(final)
    for (x, y, pr), (w, fx, fy) in zip(rows, final.tolist()):
        tx = min(c.start1, c.end1) + (fx if c.strand1 else len(sx) - 1 - fx)  # transformCoordinate (:283)
        ty = min(c.start2, c.end2) + (fy if c.strand2 else len(sy) - 1 - fy)
        assert (int(x), int(y)) == (tx, ty) and pr == "%f" % (w / 1e7)
    assert len(open(allp).read().split("\n")) - 1 >= len(rows)
    assert run("-x") == [c.format() for c in cigars]
    hmm_path = tmp_path / "expectations.hmm"
    assert run("--outputExpectations", str(hmm_path)) == []
    loaded = api.hmm_loadFromFile(str(hmm_path))
    acc = api.hmm_constructEmpty(0.000000000001, api.fiveState)
    with _realigner(seqs) as r:
        r.expectations(cigars, acc)
    np.testing.assert_allclose(list(loaded.transitions), list(acc.transitions), rtol=1e-6, atol=1e-5)  # %f text
    # --loadHmm: the expectations, normalised as cPecanEm.py does between iterations, are the next iteration's model
    api.hmm_normalise(loaded)
    model_path = tmp_path / "model.hmm"
    api.hmm_write(loaded, str(model_path))
    res = subprocess.run([exe, "--loadHmm", str(model_path), str(fa1), str(fa2)], input=text, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0 and len([l for l in res.stdout.split("\n") if l]) == len(cigars)
    bad = subprocess.run([exe, str(fa1)], input=text, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "no sequence named" in bad.stderr


def test_several_devices_from_one_process(tmp_path):
    """VERDICT r3 item 6 (SURVEY 8e from the C side): cpecan_realigner_set_devices / cpecan_realign --devices deal the cigars
    of a call over a device list from ONE process -- here two logical shards on device 0, the one GPU of the box.  The
    realigned cigars are those of a single batch, cigar for cigar, in input order; the expectation counts are the sum over
    the shards (106 doubles added on the host, cPecanEm.py:184-188).  And the reference's own fan-out -- one process per
    shard of the cigar file, the shards' expectation files summed -- gives the same file as one run."""
    from cpecan_amd.realign import shard_bounds
    exe = os.path.join(ROOT, "cpecan_amd", "cpecan_realign")
    rng = random.Random(67 + SHIFT)
    seqs, cigars = _world(rng, n_cigars=90)
    with _realigner(seqs) as r:
        want = r.realign(cigars)
        acc1 = api.hmm_constructEmpty(0.000000000001, api.fiveState)
        r.expectations(cigars, acc1)
        for devices in ([0, 0], [0, 0, 0]):
            r.set_devices(devices)
            assert r.realign(cigars) == want
            accN = api.hmm_constructEmpty(0.000000000001, api.fiveState)
            r.expectations(cigars, accN)
            # (not to the last bit: a shard is a smaller batch, whose size classes may hand a region to another kernel -- the
            # packed kernel forms an event as exp(x - total), the sweep kernel against a window reference -- and an event's
            # exponent is an fp32 value either way: ~1e-7 per event against north_star's 1e-5; measured here 2e-8)
            np.testing.assert_allclose(list(accN.transitions), list(acc1.transitions), rtol=1e-6)
            np.testing.assert_allclose(list(accN.emissions), list(acc1.emissions), rtol=1e-6, atol=1e-300)
            np.testing.assert_allclose(accN.likelihood, acc1.likelihood, rtol=1e-9)
        r.set_devices([0, 0])
        assert r.realign(cigars[:1]) == want[:1]  # fewer cigars than shards
    # the command line
    fa = tmp_path / "all.fa"
    with open(fa, "w") as f:
        for n in sorted(seqs):
            f.write(">%s\n%s\n" % (n, seqs[n]))

    def run(text, *args):
        res = subprocess.run([exe, *args, str(fa)], input=text, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        return [line for line in res.stdout.split("\n") if line]

    text = "".join(c.format() + "\n" for c in cigars)
    assert run(text, "--devices", "0,0") == [c.format() for c in want]
    assert run(text, "--devices", "0-0,0", "--batch", "25") == [c.format() for c in want]
    one, multi = tmp_path / "one.hmm", tmp_path / "multi.hmm"
    assert run(text, "--outputExpectations", str(one)) == []
    assert run(text, "--outputExpectations", str(multi), "--devices", "0,0") == []
    whole, both = api.hmm_loadFromFile(str(one)), api.hmm_loadFromFile(str(multi))
    np.testing.assert_allclose(list(both.transitions), list(whole.transitions), rtol=1e-6, atol=2e-6)  # %f text
    np.testing.assert_allclose(list(both.emissions), list(whole.emissions), rtol=1e-6, atol=2e-6)
    # one process per shard of the file, files summed (cPecanEm.py:168-188); the pseudo count 1e-12 of every file vanishes in %f
    lo, mid, hi = shard_bounds(cigars, 2)
    parts = []
    for k, (a, b) in enumerate(((lo, mid), (mid, hi))):
        path = tmp_path / ("shard%d.hmm" % k)
        assert run("".join(c.format() + "\n" for c in cigars[a:b]), "--outputExpectations", str(path)) == []
        parts.append(api.hmm_loadFromFile(str(path)))
    summed_t = [x + y for x, y in zip(parts[0].transitions, parts[1].transitions)]
    summed_e = [x + y for x, y in zip(parts[0].emissions, parts[1].emissions)]
    np.testing.assert_allclose(summed_t, list(whole.transitions), rtol=1e-6, atol=3e-6)
    np.testing.assert_allclose(summed_e, list(whole.emissions), rtol=1e-6, atol=3e-6)
    np.testing.assert_allclose(parts[0].likelihood + parts[1].likelihood, whole.likelihood, rtol=1e-6, atol=3e-6)
    bad = subprocess.run([exe, "--devices", "0,x", str(fa)], input=text, capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0


def test_degenerate_cigars():
    """Empty alignments, alignments of indels only, a matchGamma nothing reaches, one-base sequences: no pair survives and
    the cigar falls back to the unaligned ends (convertAlignedPairsToPairwiseAlignment's end pair, :56-87)."""
    seqs = {"a": "ACGTACGTAC", "b": "ACGTTCGTAC", "n": "NNNNNNNNNN", "one": "G"}
    cigars = [
        Cigar("a", 3, 3, True, "b", 5, 5, True, 7.0, []),                      # nothing aligned at all
        Cigar("a", 0, 5, True, "b", 0, 3, True, 7.0, [(DX, 5), (IY, 3)]),      # indels only: no anchors, full matrix
        Cigar("a", 0, 10, True, "n", 0, 10, True, 7.0, [(M, 10)]),             # every anchor filtered (N never matches)
        Cigar("one", 0, 1, True, "one", 1, 0, False, 7.0, [(M, 1)]),           # one base against its reverse complement
        Cigar("a", 0, 10, True, "b", 0, 10, True, 7.0, [(M, 10)]),
    ]
    o = realign_options()
    with _realigner(seqs) as r:
        got = r.realign(cigars)
    assert len(got) == len(cigars)
    for a, c in zip(got, cigars):
        assert a.same_coordinates(c)
        assert a == _stepwise(seqs, c, o)[0]
    assert got[0].ops == []
    assert got[4].ops == [(M, 10)]
    with _realigner(seqs, matchGamma=1.5) as r:  # a posterior never reaches 1.5
        none = r.realign(cigars)
    assert none[4].ops == [(DX, 10), (IY, 10)] and none[1].ops == [(DX, 5), (IY, 3)]
    with _realigner(seqs, rescoreOriginalAlignment=1, rescoreByPosteriorProb=1, constraintDiagonalTrim=2) as r:
        trimmed = r.realign(cigars[4:])
    # -x with a trim: only the untrimmed columns are anchors, the rest of the match run comes back as indels (:548)
    assert trimmed[0].ops == [(DX, 2), (IY, 2), (M, 6), (DX, 2), (IY, 2)] and 0.0 <= trimmed[0].score <= 100.0
    with _realigner(seqs, splitIndelsLongerThanThis=0) as r:
        assert r.realign(cigars[:2]) == []  # nothing but indels: no piece survives the split
    acc = api.hmm_constructEmpty(0.0, api.fiveState)
    with _realigner(seqs) as r:
        r.expectations(cigars, acc)
    assert np.isfinite(list(acc.transitions)).all() and acc.likelihood != 0.0
    with pytest.raises(api.CpecanError):
        with _realigner(seqs) as r:
            r.realign([Cigar("a", 0, 11, True, "b", 0, 11, True, 1.0, [(M, 11)])])  # beyond the end of the sequence
    with pytest.raises(api.CpecanError):
        with _realigner(seqs) as r:
            r.realign([Cigar("a", 0, 5, True, "zz", 0, 5, True, 1.0, [(M, 5)])])  # unknown sequence


def test_nothing_but_empty_problems():
    """Batches and realign calls whose every problem is empty (or that hold no problem at all): nothing to launch, empty
    lists, the cigar of an empty alignment comes back as it went in."""
    p = api.pairwiseAlignmentBandingParameters_construct()
    with api.Batch(api.stateMachine5_construct(), p) as b:
        b.set_post(api.POST_REWEIGHT | api.POST_ORDERED, 0.5)
        b.add("", "", ())
        b.upload()
        b.run()
        b.download()
        assert b.result(0).shape == (0, 3) and b.result(0, 3).shape == (0, 3)
        assert b.scores(0)[0] == 0.0 and np.isnan(b.scores(0)[1])
    with api.Batch(api.stateMachine5_construct(), p) as b:
        b.upload()
        b.run()
        b.download()
    empty = Cigar("a", 2, 2, True, "b", 1, 1, True, 3.0, [])
    with Realigner() as r:
        r.add_sequence("a", "ACGT")
        r.add_sequence("b", "ACGT")
        assert r.realign([empty]) == [empty]
        acc = api.hmm_constructEmpty(0.0, api.fiveState)
        r.expectations([empty], acc)
        assert acc.likelihood == 0.0
