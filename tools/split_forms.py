"""diagnostic: kernel time of one batch of five-state pairs under the three launch forms (CPECAN_SPLIT=0 one wave per
region, 1 forward launch + traceback items, 2 one launch) for several band widths.  usage: python tools/split_forms.py [pairs] [bp] [states: 5 | 3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpecan_amd import api
from cpecan_amd.workload import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sm = api.stateMachine3_construct() if (len(sys.argv) > 3 and sys.argv[3] == "3") else api.stateMachine5_construct()
for E in (30, 50, 70, 100):
    probs = [make_pair(11, i, L, E) + (False, False) for i in range(n)]
    arr, cnt, keep = api.Batch.prepare_problems(probs)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=E)
    line = []
    for form in ("0", "1", "2", None):
        if form is None:
            os.environ.pop("CPECAN_SPLIT", None)
        else:
            os.environ["CPECAN_SPLIT"] = form
        with api.Batch(sm, p) as b:
            b.add_prepared(arr, cnt)
            b.upload()
            for _ in range(2):
                b.run()
            b.download()
            ms = []
            for _ in range(3):
                b.run()
                b.download()
                ms.append(b.stats().kernelMs)
            st = b.stats()
        line.append("%s: %.2f ms" % ("default" if form is None else "split=" + form, sorted(ms)[1]))
    print("expansion %3d (%.0f cells per diagonal, %d pairs of %d bp): %s" % (E, st.cells / st.diagonals, n, L, ", ".join(line)), flush=True)
