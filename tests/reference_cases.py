"""Inputs of the reference's own tests that pin more than geometry, shared by the oracle (CPU) and
HIP (GPU) suites.  Data: tests/golden/ (made by tests/golden/make_reference_fixtures.py)."""
import gzip
import json
import os
import random

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAINED_HMM = os.path.join(GOLDEN, "trained_hmm_cPecanEmTest.txt")

# impl/randomSequences.c:13-16 (getRandomChar): 11 x "AaCcGgTt" and one N
RANDOM_CHAR_ALPHABET = "AaCcGgTt" * 11 + "N"


def random_sequence(rng, n):
    return "".join(RANDOM_CHAR_ALPHABET[rng.randrange(len(RANDOM_CHAR_ALPHABET))] for _ in range(n))


def ragged_ends_trial(trial, core=100, flank=100):
    """tests/pairwiseAlignerTest.c:676-715 (test_getAlignedPairsWithRaggedEnds): X = a random core, Y = random prefix + X +
    random suffix; aligned with ragged ends on both sides and put through the ordered filter at gamma 0.2 the result is
    EXACTLY the `core` pairs (x, x + flank).  Own seeded generator over the reference's alphabet."""
    rng = random.Random(7000 + trial)
    sx = random_sequence(rng, core)
    return sx, random_sequence(rng, flank) + sx + random_sequence(rng, flank)


def trained_hmm_numbers():
    """(type, transitions[25], likelihood, emissions[80]) as Python parses the text: the layout of stateMachine.c:133-202."""
    l1, l2 = open(TRAINED_HMM).read().split("\n")[:2]
    a = l1.split()
    return int(a[0]), [float(v) for v in a[1:26]], float(a[26]), [float(v) for v in l2.split()]


def _anchors_from_rows(a, b, anchor_every, flank, equal_bases):
    """(anchors[n,3], true_pairs) from two rows of an alignment: one column every `anchor_every` among the columns that
    lie more than `flank` columns inside a gapless run of aligned columns (equal_bases: of equal, non-N bases)."""
    av = np.frombuffer(a.encode(), dtype=np.uint8)
    bv = np.frombuffer(b.encode(), dtype=np.uint8)
    gap = ord("-")
    ina, inb = av != gap, bv != gap
    x = np.cumsum(ina) - 1          # index of the base in column l (where the row has one)
    y = np.cumsum(inb) - 1
    both = ina & inb
    true_pairs = set(zip(x[both].tolist(), y[both].tolist()))
    # consider only columns where at least one of the two rows has a base (the other rows of the multiple alignment
    # contribute all-gap columns for this pair)
    keep = ina | inb
    eq = ((both & (av == bv) & (av != ord("N"))) if equal_bases else both)[keep]
    xs, ys = x[keep], y[keep]
    run = np.zeros(len(eq), dtype=np.int64)   # length of the run of such columns ending here
    c = 0
    for i, e in enumerate(eq):
        c = c + 1 if e else 0
        run[i] = c
    ahead = np.zeros(len(eq), dtype=np.int64)  # ... and starting here
    c = 0
    for i in range(len(eq) - 1, -1, -1):
        c = c + 1 if eq[i] else 0
        ahead[i] = c
    inside = np.nonzero((run > flank) & (ahead > flank))[0]
    chosen, last = [], -10 ** 9
    for i in inside:
        if i - last >= anchor_every:
            chosen.append(i)
            last = i
    anchors = np.array([(xs[i], ys[i], 0) for i in chosen], dtype=np.int64).reshape(-1, 3)
    return anchors, true_pairs


def encode_human_chimp(anchor_every=10, flank=14):
    """tests/pairwiseAlignerLongTest.c: the ~57 kb human / chimp ENCODE fragments with the reference alignment the test
    scores against.  Returns (sX, sY, anchors[n,3], true_pairs set).  The reference finds its anchors with lastz (absent
    here); these come from the embedded alignment the way lastz + constraintDiagonalTrim would leave them: columns of
    equal bases at least `flank` columns inside a gapless run of equal bases, one every `anchor_every` columns."""
    d = json.load(gzip.open(os.path.join(GOLDEN, "encode_human_chimp.json.gz")))
    anchors, true_pairs = _anchors_from_rows(d["humanAlign"].upper(), d["chimpAlign"].upper(), anchor_every, flank, True)
    return d["humanSeq"], d["chimpSeq"], anchors, true_pairs


def encode_human_other(species, anchor_every=10, flank=14):
    """The same test file's human / mouse and human / dog pairs (test_pairwiseAligner_LongHumanMouse / _LongHumanDog,
    :128-134): 67 % / 75 % identical over the aligned columns, so runs of equal bases are too short to anchor on; a lastz
    block is gapless with mismatches inside, trimmed by constraintDiagonalTrim at both ends -- here: columns more than
    `flank` inside a gapless run of ALIGNED columns of the embedded alignment, one every `anchor_every`.  The stretches
    without anchors are up to 1947 x 1329 (mouse) and 1189 x 1145 (dog) bases: full DP rectangles between banded ones."""
    assert species in ("mouse", "dog")
    d = json.load(gzip.open(os.path.join(GOLDEN, "encode_human_chimp.json.gz")))
    e = json.load(gzip.open(os.path.join(GOLDEN, "encode_mouse_dog.json.gz")))
    anchors, true_pairs = _anchors_from_rows(d["humanAlign"].upper(), e[species + "Align"].upper(), anchor_every, flank, False)
    return d["humanSeq"], e[species + "Seq"], anchors, true_pairs


def sensitivity_specificity(pairs, true_pairs):
    """As the reference's long test logs them (tests/pairwiseAlignerLongTest.c:100-108)."""
    got = {(int(x), int(y)) for _, x, y in np.asarray(pairs).reshape(-1, 3)}
    inter = len(got & true_pairs)
    return inter / max(1, len(true_pairs)), inter / max(1, len(got))
