#!/usr/bin/env python3
"""Generates the data fixtures that come out of the reference's own tests (run in the build container, where
/root/reference exists; the GPU box only sees the committed outputs).  DATA ONLY: string constants that the reference's
tests hold as inputs / expected values -- no source text is copied.

  trained_hmm_cPecanEmTest.txt   the two lines of the trained five-state-asymmetric HMM that cPecanEmTest.py:112-113
                                 writes to a file and loads back (a nanopore-trained model): pins the on-disk layout
                                 hmm_loadFromFile must read (type, S*S transitions, likelihood / S*16 emissions).
  encode_human_chimp.json.gz     tests/pairwiseAlignerLongTest.c:14-38: the ~57 kb human and chimp ENCODE fragments and the
                                 two rows of the reference multiple alignment the test scores against (humanSeq, chimpSeq,
                                 humanAlign, chimpAlign string constants).
  encode_mouse_dog.json.gz       the same file's mouse (~33 kb) and dog (~54 kb) fragments and their rows of the same
                                 multiple alignment (mouseSeq, mouseAlign, dogSeq, dogAlign; test_pairwiseAligner_LongHumanMouse
                                 / _LongHumanDog, :128-134): 67 % / 75 % identity over the aligned columns, unaligned
                                 stretches of up to 8 kb -- the human row is the one in encode_human_chimp.json.gz.
"""
import gzip
import json
import os
import re

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def c_string_constant(text, name):
    """The value of `char *name = "..." "..." ;` (adjacent literals concatenated)."""
    m = re.search(r"\*\s*" + name + r"\s*=\s*((?:\s*\"[^\"]*\")+)\s*;", text)
    assert m, name
    return "".join(re.findall(r"\"([^\"]*)\"", m.group(1)))


def main():
    py = open(os.path.join(REF, "cPecanEmTest.py")).read()
    lines = [ln.replace("\\n", "") for ln in re.findall(r"fH\.write\(\"([^\"]*)\"\)", py[py.index("def testHMMToBlast"):])]
    assert len(lines) == 2 and len(lines[0].split()) == 1 + 25 + 1 and len(lines[1].split()) == 80
    with open(os.path.join(OUT, "trained_hmm_cPecanEmTest.txt"), "w") as f:
        f.write(lines[0] + "\n" + lines[1] + "\n")
    c = open(os.path.join(REF, "tests", "pairwiseAlignerLongTest.c")).read()
    d = {"source": "tests/pairwiseAlignerLongTest.c:14-38 (string constants), scored as in :40-122"}
    for name in ("humanSeq", "chimpSeq", "humanAlign", "chimpAlign"):
        d[name] = c_string_constant(c, name)
    assert len(d["humanAlign"]) == len(d["chimpAlign"])
    assert d["humanAlign"].replace("-", "").upper() == d["humanSeq"].upper()
    assert d["chimpAlign"].replace("-", "").upper() == d["chimpSeq"].upper()
    with gzip.GzipFile(os.path.join(OUT, "encode_human_chimp.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(d).encode())
    e = {"source": "tests/pairwiseAlignerLongTest.c (string constants), scored as in :40-122 by :128-134"}
    for name in ("mouseSeq", "mouseAlign", "dogSeq", "dogAlign"):
        e[name] = c_string_constant(c, name)
    for sp in ("mouse", "dog"):
        assert len(e[sp + "Align"]) == len(d["humanAlign"])
        assert e[sp + "Align"].replace("-", "").upper() == e[sp + "Seq"].upper()
    with gzip.GzipFile(os.path.join(OUT, "encode_mouse_dog.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(e).encode())
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
