import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from cpecan_amd import api
from cpecan_amd.workload import make_pair
import oracle_binding as ob
from parity import assert_pairs_match
sx, sy, _ = make_pair(77, 0, 3000, 0)
p = api.pairwiseAlignmentBandingParameters_construct()
t = time.time()
with api.Batch(api.stateMachine5_construct(), p) as b:
    b.add(sx, sy, ())
    b.upload(); b.run(); b.download()
    st = b.stats(); got = b.result(0)
print("3000 x 3000 unanchored: %d cells, kernel %.1f ms, %.2f GB device, %d pairs, wall %.2f s" % (st.cells, st.kernelMs, st.deviceBytes / 2**30, len(got), time.time() - t))
t = time.time()
want = ob.aligned_pairs(ob.model(0), sx, sy, (), ob.params())
print("oracle %.1f s" % (time.time() - t))
print("worst score difference", assert_pairs_match(got, want, threshold=0.01))
