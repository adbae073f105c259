"""Throughput with SPARSE anchors (lastz-style: an anchor every few hundred bases, expansion 20): between two anchors the
band is the whole rectangle they span, so diagonals are 21 cells wide at the anchors and hundreds in between.
Usage: python tools/sparse_bench.py [pairs] [length] [anchor_every]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpecan_amd import api
from cpecan_amd.workload import make_pair


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    for every in ([int(sys.argv[3])] if len(sys.argv) > 3 else [50, 150, 400, 1000]):
        p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=20)
        probs = [make_pair(5, i % 256, L, 20, anchor_every=every) for i in range(n)]
        with api.Batch(api.stateMachine5_construct(), p) as b:
            for sx, sy, a in probs:
                b.add(sx, sy, a)
            b.upload()
            b.run()
            b.run()
            b.download()
            st = b.stats()
            print("anchor every %4d bp: %.3e cells (%.0f per diagonal)  kernel %8.2f ms = %.2e cells/s  %5d waves  %5.1f GB" %
                  (every, st.cells, st.cells / st.diagonals, st.kernelMs, st.cells / st.kernelMs * 1e3, st.wavesPerLaunch,
                   st.deviceBytes / 2 ** 30), flush=True)


if __name__ == "__main__":
    main()
