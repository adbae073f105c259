import time, sys, os
sys.path.insert(0, os.getcwd())
from cpecan_amd import api
from cpecan_amd.workload import make_pair
p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=20)
sm = api.stateMachine5_construct()
# usage: python tools/call_latency.py   (CPECAN_SPLIT=0: one wave per region; unset: tracebacks as queue items when the
# launch leaves wave slots idle, as a lone pair does)
for L, E in ((100, 20), (1000, 20), (2000, 100), (10000, 100)):
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=E)
    sx, sy, a = make_pair(3, 0, L, E)
    api.getAlignedPairsUsingAnchors(sm, sx, sy, a, p)
    t = time.time()
    n = 50
    for _ in range(n):
        api.getAlignedPairsUsingAnchors(sm, sx, sy, a, p)
    print("single call, %d bp: %.2f ms" % (L, 1e3 * (time.time() - t) / n))
    t = time.time()
    for _ in range(n):
        with api.Batch(sm, p) as b:
            b.add(sx, sy, a)
            t0 = time.time(); b.upload(); t1 = time.time(); b.run(); b.download(); t2 = time.time()
    print("   last batch: upload %.2f ms, run+download %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
