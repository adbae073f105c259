"""Host logic of the realign front end (include/cpecan_realign.h) that needs no GPU: the cigar text
format, convertAlignedPairsToPairwiseAlignment (cPecanRealign.c:49-96) and splitPairwiseAlignment (:117-230) against
column-by-column restatements written here, the fasta reader, and the loud failure of the realign loop without a GPU."""
import random

import pytest

from cpecan_amd import api
from cpecan_amd.realign import Cigar, Realigner, realign_options

M, DX, IY = api.OP_MATCH, api.OP_INDEL_X, api.OP_INDEL_Y
CH = {M: "M", DX: "D", IY: "I"}


def _rle(columns):
    ops = []
    for c in columns:
        if ops and ops[-1][0] == c:
            ops[-1][1] += 1
        else:
            ops.append([c, 1])
    return [(t, n) for t, n in ops]


def _random_ops(rng, n_ops, ends_with_match=True):
    """Operations without two neighbours of one type, so that a run-length code of the columns gives them back."""
    ops, last = [], None
    for i in range(n_ops):
        choices = [t for t in (M, M, M, DX, IY) if t != last]
        if ends_with_match and i in (0, n_ops - 1) and last != M:
            choices = [M]
        t = rng.choice(choices)
        ops.append((t, rng.randrange(1, 12) if t == M else rng.choice([1, 1, 2, 3, 8, 20])))
        last = t
    return ops


def _cigar(rng, ops, strand1=True, strand2=True, score=42.0):
    span1 = sum(n for t, n in ops if t != IY)
    span2 = sum(n for t, n in ops if t != DX)
    s1, s2 = rng.randrange(0, 50), rng.randrange(0, 50)
    a = (s1, s1 + span1) if strand1 else (s1 + span1, s1)
    b = (s2, s2 + span2) if strand2 else (s2 + span2, s2)
    return Cigar("target.1", a[0], a[1], strand1, "query|x", b[0], b[1], strand2, score, ops)


def test_cigar_text_round_trip_and_layout():
    rng = random.Random(1)
    c = Cigar("chrT", 10, 25, True, "readQ", 45, 27, False, 87.5, [(M, 5), (DX, 2), (M, 3), (IY, 5), (M, 5)])
    text = c.format()
    # the query (contig2) comes first; D = bases of contig1 only, I = bases of contig2 only; %f score
    assert text == "cigar: readQ 45 27 - chrT 10 25 + 87.500000 M 5 D 2 M 3 I 5 M 5"
    assert Cigar.parse(text) == c
    assert Cigar.parse("  cigar:  readQ 45 27 -  chrT 10 25 + 87.5   M 5 D 2 M 3 I 5 M 5 \n") == c
    for _ in range(200):
        c = _cigar(rng, _random_ops(rng, rng.randrange(1, 12), False), rng.random() < 0.5, rng.random() < 0.5,
                   round(rng.uniform(0, 1000), 3))
        assert Cigar.parse(c.format()) == c
    empty = Cigar("a", 3, 3, True, "b", 7, 7, True, 0.0, [])
    assert Cigar.parse(empty.format()) == empty


@pytest.mark.parametrize("line", [
    "", "vulgar: q 0 5 + t 0 5 + 1 M 5", "cigar: q 0 5 + t 0 5 + 1 M 4",  # operations do not add up
    "cigar: q 0 5 + t 0 5 + 1 M", "cigar: q 0 5 + t 0 5 + 1 X 5", "cigar: q 0 5 * t 0 5 + 1 M 5",
    "cigar: q 0 5 + t 0 5 + score M 5", "cigar: q 0 5 + t 5 0 + 1 M 5",  # '+' strand running backwards
    "cigar: q 0 5 +", "cigar: q 0 5 + t 0 5 + 1 M -5 M 10",
])
def test_cigar_parse_rejects(line):
    with pytest.raises(api.CpecanError):
        Cigar.parse(line)


def test_cigar_from_aligned_pairs_against_columns():
    """convertAlignedPairsToPairwiseAlignment: every pair is a match column, the bases skipped before it become a D run
    then an I run, and what is left behind the last pair becomes trailing indels (the "end matched pair", :56)."""
    rng = random.Random(2)
    for _ in range(300):
        l1, l2 = rng.randrange(0, 40), rng.randrange(0, 40)
        xy, x, y = [], -1, -1
        while True:
            x += rng.choice([1, 1, 1, 2, 5])
            y += rng.choice([1, 1, 1, 2, 5])
            if x >= l1 or y >= l2:
                break
            xy.append((x, y))
        columns, px, py = [], -1, -1
        for x, y in xy + [(l1, l2)]:
            columns += [DX] * (x - px - 1) + [IY] * (y - py - 1) + [M]
            px, py = x, y
        got = Cigar.from_aligned_pairs("t", "q", 5.0, l1, l2, xy)
        assert got.ops == _rle(columns[:-1])
        assert (got.start1, got.end1, got.strand1, got.start2, got.end2, got.strand2) == (0, l1, True, 0, l2, True)
        assert got.score == 5.0 and got.contig1 == "t" and got.contig2 == "q"
    with pytest.raises(api.CpecanError):
        Cigar.from_aligned_pairs("t", "q", 0.0, 3, 3, [(3, 0)])


def _split_by_columns(c, max_len):
    """splitPairwiseAlignment restated on alignment columns: interior runs of indel columns longer than max_len are cut
    out, runs at either end are dropped, each piece starts at its first and ends behind its last match column."""
    cols = [t for t, n in c.ops for _ in range(n)]
    pieces, cur, i = [], [], 0
    p1, p2 = c.start1, c.start2
    d1, d2 = (1 if c.strand1 else -1), (1 if c.strand2 else -1)
    pos = []  # position before each column
    for t in cols:
        pos.append((p1, p2))
        p1 += d1 if t != IY else 0
        p2 += d2 if t != DX else 0
    pos.append((p1, p2))
    while i < len(cols):
        if cols[i] == M:
            cur.append(i)
            i += 1
            continue
        j = i
        while j < len(cols) and cols[j] != M:
            j += 1
        if cur and j < len(cols) and j - i <= max_len:
            cur.extend(range(i, j))
        elif cur and j - i > max_len or (cur and j == len(cols)):
            pieces.append(cur)
            cur = []
        i = j
    if cur:
        pieces.append(cur)
    out = []
    for piece in pieces:
        first, last = piece[0], piece[-1]
        out.append(Cigar(c.contig1, pos[first][0], pos[last + 1][0], c.strand1, c.contig2, pos[first][1], pos[last + 1][1],
                         c.strand2, c.score, _rle([cols[k] for k in piece])))
    return out


def test_cigar_split_against_columns():
    rng = random.Random(3)
    cut_some = 0
    for _ in range(400):
        c = _cigar(rng, _random_ops(rng, rng.randrange(1, 14), ends_with_match=rng.random() < 0.5), rng.random() < 0.5,
                   rng.random() < 0.5)
        max_len = rng.choice([0, 1, 2, 3, 7, 25])
        got = c.split(max_len)
        want = _split_by_columns(c, max_len)
        assert got == want, (c, max_len)
        cut_some += len(got) > 1
        for piece in got:  # checkPairwiseAlignment on every piece (cPecanRealign.c:226-228): parse accepts only consistent ones
            assert Cigar.parse(piece.format()) == piece
            assert piece.ops[0][0] == M and piece.ops[-1][0] == M
    assert cut_some > 50
    # an alignment of indels only has no piece at all
    assert _cigar(rng, [(DX, 4), (IY, 2)]).split(10) == []


def test_realign_options_defaults_and_checks():
    o = realign_options()
    assert (o.params.diagonalExpansion, o.params.splitMatrixBiggerThanThis, o.constraintDiagonalTrim) == (4, 10, 0)
    assert (o.gapGamma, round(o.matchGamma, 6), o.splitIndelsLongerThanThis) == (0.5, 0.85, -1)
    assert not (o.rescoreOriginalAlignment or o.rescoreByIdentity or o.rescoreByPosteriorProb)
    with pytest.raises(api.CpecanError):
        Realigner(options=realign_options(diagonalExpansion=3))  # must be even (cPecanRealign.c:427)
    with pytest.raises(api.CpecanError):
        Realigner(options=realign_options(gapGamma=-1.0))


def test_fasta_reader_and_missing_inputs(tmp_path):
    fa = tmp_path / "s.fa"
    fa.write_text(">seqA some description\nACGT\nAC GT\n\n>seqB\nTTTT\n>seqA again but shorter\nAC\n")
    with Realigner() as r:
        assert r.read_fasta(str(fa)) == 3
        with pytest.raises(api.CpecanError):
            r.read_fasta(str(tmp_path / "absent.fa"))
        with pytest.raises(api.CpecanError):
            r.add_sequence("   ", "ACGT")
        assert r.realign([]) == []
        if api.lib().cpecan_device_count() <= 0:
            c = Cigar("seqA", 0, 8, True, "seqB", 0, 4, True, 1.0, [(M, 4), (DX, 4)])
            with pytest.raises(api.CpecanError) as e:
                r.realign([c])  # no GPU: loud failure, never a CPU path
            assert "no usable HIP device" in str(e.value) or "HIP" in str(e.value)


def test_shard_bounds_match_the_python_deal():
    """cpecan_realign_shard_bounds (the C side's deal of cigars over devices, cpecan_realigner_set_devices) cuts where
    cpecan_amd.dist.cost_balanced_bounds -- the deal of the torch.distributed path -- cuts: contiguous shards of about
    equal band cells, in input order (the reference's own fan-out is one process per shard of the file,
    cPecanEm.py:168-188)."""
    from cpecan_amd import dist as cdist
    from cpecan_amd.realign import shard_bounds
    rng = random.Random(404)
    for n, world in ((0, 3), (1, 2), (5, 8), (200, 2), (777, 8), (1000, 5)):
        cigars = []
        for i in range(n):
            lx, ly = rng.randrange(1, 5000), rng.randrange(1, 5000)
            s1, s2 = rng.randrange(0, 1000), rng.randrange(0, 1000)
            plus1, plus2 = rng.random() < 0.5, rng.random() < 0.5
            cigars.append(Cigar("a", s1 if plus1 else s1 + lx, s1 + lx if plus1 else s1, plus1,
                                "b", s2 if plus2 else s2 + ly, s2 + ly if plus2 else s2, plus2, 0.0, []))
        got = shard_bounds(cigars, world)
        assert got[0] == 0 and got[-1] == n and all(a <= b for a, b in zip(got, got[1:]))
        costs = [cdist.cigar_cost(c) for c in cigars]
        for rank in range(world):
            lo, hi = cdist.cost_balanced_bounds(costs, rank, world)
            assert (got[rank], got[rank + 1]) == (lo, hi), (n, world, rank)
        if n >= 200:  # balanced: no shard is more than one cigar's cost away from its share
            total, worst = sum(costs), max(costs)
            for k in range(world):
                assert abs(sum(costs[got[k]:got[k + 1]]) - total / world) <= 2 * worst


def test_anchor_runs_are_the_anchor_list_run_length_coded():
    """cpecan_anchor_runs_from_alignment (what the realign front end now hands the batch, cpecan_batch_add_many_runs) gives
    the runs of the list convertPairwiseForwardStrandAlignmentToAnchorPairs + the exact-match filter give
    (pairwiseAligner.c:979-1003, cPecanRealign.c:277-281, :525-529): expanded again, the same anchors in the same order."""
    import numpy as np
    rng = random.Random(808)
    for trial in range(60):
        ops = _random_ops(rng, rng.randrange(1, 14))
        lx = sum(n for t, n in ops if t != IY)
        ly = sum(n for t, n in ops if t != DX)
        sx = "".join(rng.choice("ACGTNacgt") for _ in range(lx))
        # Y: X's bases along the match columns with some substitutions, random bases in its own insertions
        sy, x = [], 0
        for t, n in ops:
            if t == M:
                sy += [c if rng.random() < 0.8 else rng.choice("ACGT") for c in sx[x:x + n]]
            elif t == IY:
                sy += [rng.choice("ACGT") for _ in range(n)]
            if t != IY:
                x += n
        sy = "".join(sy)
        assert len(sy) == ly
        for trim in (0, 1, 3):
            for filt in (False, True):
                a = api.convertPairwiseForwardStrandAlignmentToAnchorPairs(ops, 0, 0, trim, 4, sx if filt else None, sy if filt else None)
                runs = api.anchor_runs_from_alignment(ops, 0, 0, trim, 4, sx if filt else None, sy if filt else None)
                assert np.array_equal(runs, api.anchor_runs(a))
                back = [(x0 + i, y0 + i, e) for x0, y0, n, e in runs.tolist() for i in range(n)]
                assert back == [tuple(t) for t in a.tolist()]
                for (x0, y0, n, e), (x1, y1, n1, e1) in zip(runs.tolist(), runs.tolist()[1:]):
                    assert (x1, y1) != (x0 + n, y0 + n)  # maximal: two runs never abut on one diagonal
