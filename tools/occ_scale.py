"""diagnostic: the split-class sweeps (forward launch + item launch) against resident waves per CU on a band narrow enough
for 12 waves per CU to fit the LDS.  usage: python tools/occ_scale.py <pairs> <expansion> cap...   (needs a GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpecan_amd import api, workload

n, E = int(sys.argv[1]), int(sys.argv[2])
caps = [int(c) for c in sys.argv[3:]]
cfg = dict(workload.CONFIGS["B"])
sM = api.stateMachine5_construct(api.fiveState)
p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=E, splitMatrixBiggerThanThis=10 ** 15)
probs = workload.make_batch(cfg["seed"], n, cfg["length"], E)
for cap in caps:
    os.environ["CPECAN_MAX_WAVES_PER_CU"] = str(cap)
    with api.Batch(sM, p) as b:
        b.add_many([(sx, sy, a, False, False) for sx, sy, a in probs])
        b.upload()
        best = 1e30
        for _ in range(3):
            b.run()
            b.download()
            best = min(best, b.stats().kernelMs)
        st = b.stats()
        print("cap %2d waves %5d cells %.3e ms %7.2f cells/s %.3e" % (cap, st.wavesPerLaunch, st.cells, best, st.cells / best * 1e3), flush=True)
