"""Kernel time of a batch that mixes band-100 pairs (config B's shape) with a few unanchored rectangles, against the two
parts run on their own: the size classes keep the wide regions from dictating LDS, occupancy and per-wave scratch of the
rest.  Usage: python tools/mixed_bench.py [pairs] [rectangles] [rectangle_bp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpecan_amd import api
from cpecan_amd.workload import make_pair


def run(problems, label):
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=100)
    with api.Batch(api.stateMachine5_construct(), p) as b:
        for sx, sy, a in problems:
            b.add(sx, sy, a)
        b.upload()
        b.run()
        b.run()
        b.download()
        st = b.stats()
        print("%-28s %6d regions %.3e cells  kernel %8.2f ms  %5d waves  %6.1f GB device" %
              (label, st.regions, st.cells, st.kernelMs, st.wavesPerLaunch, st.deviceBytes / 2 ** 30), flush=True)
        return st.kernelMs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    base = [make_pair(1, i % 256, 2000, 100) for i in range(n)]
    wide = [make_pair(2, i, L, 0)[:2] + ((),) for i in range(k)]
    a = run(base, "banded pairs alone")
    b = run(wide, "rectangles alone")
    c = run(base + wide, "both in one batch")
    print("one batch / (sum of the parts) = %.2f; / max of the parts = %.2f" % (c / (a + b), c / max(a, b)))


if __name__ == "__main__":
    main()
