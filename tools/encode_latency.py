"""The reference's three real sequence pairs (tests/pairwiseAlignerLongTest.c: human against chimp, mouse, dog; ~57 kb, one
alignment per call, the caller's shape of the reference's own long test): wall time of one getAlignedPairsUsingAnchors call
through the HIP path (CPECAN_SPLIT=0: one wave for the region; unset: its traceback segments as queue items), and of the
same call in the CPU oracle.  usage: python tools/encode_latency.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import reference_cases as rc
import oracle_binding as ob
from cpecan_amd import api

sm, om = api.stateMachine5_construct(), ob.model(ob.FIVE_STATE)
for name in ("chimp", "mouse", "dog"):
    sx, sy, anchors, _ = rc.encode_human_chimp() if name == "chimp" else rc.encode_human_other(name)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=20)
    t = time.time()
    want = ob.aligned_pairs(om, sx, sy, anchors, ob.params(diagonalExpansion=20))
    t_cpu = time.time() - t
    line = []
    for form in ("0", None):
        if form is None:
            os.environ.pop("CPECAN_SPLIT", None)
        else:
            os.environ["CPECAN_SPLIT"] = form
        api.getAlignedPairsUsingAnchors(sm, sx, sy, anchors, p)
        ts = []
        for _ in range(5):
            t = time.time()
            got = api.getAlignedPairsUsingAnchors(sm, sx, sy, anchors, p)
            ts.append(time.time() - t)
        with api.Batch(sm, p) as b:
            b.add(sx, sy, anchors)
            b.upload(); b.run(); b.download()
            st = b.stats()
        line.append("%s %.1f ms (kernels %.1f ms)" % ("one wave" if form == "0" else "default", 1e3 * sorted(ts)[2], st.kernelMs))
    print("human/%s: %d x %d bp, %d anchors, %d cells, %d pairs: oracle (1 core) %.2f s; HIP %s"
          % (name, len(sx), len(sy), len(anchors), st.cells, len(got), t_cpu, ", ".join(line)), flush=True)
