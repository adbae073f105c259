"""Throughput on UNANCHORED pairs (the multiple aligner's getAlignedPairs below anchorMatrixBiggerThanThis: the band is the
whole matrix, diagonals up to min(lX, lY) + 1 cells).  Usage: python tools/unanchored_bench.py [pairs] [length ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpecan_amd import api
from cpecan_amd.workload import make_pair


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    for L in ([int(v) for v in sys.argv[2:]] or [100, 200, 300, 400, 500]):
        p = api.pairwiseAlignmentBandingParameters_construct()
        probs = [make_pair(9, i % 256, L, 0)[:2] for i in range(n)]
        with api.Batch(api.stateMachine5_construct(), p) as b:
            b.add_many([(sx, sy, ()) for sx, sy in probs])
            b.upload()
            b.run()
            b.run()
            b.download()
            st = b.stats()
            print("%4d x %4d unanchored: %.3e cells  kernel %8.2f ms = %.2e cells/s  %5d waves  %5.1f GB" %
                  (L, L, st.cells, st.kernelMs, st.cells / st.kernelMs * 1e3, st.wavesPerLaunch, st.deviceBytes / 2 ** 30), flush=True)


if __name__ == "__main__":
    main()
