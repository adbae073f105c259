"""Debug aid (GPU box): runs a few problems through the HIP path with the debug buffers on and reports
where fb = F.match + B.match and the per-diagonal totals first differ from the oracle's trace."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import oracle_binding as ob  # noqa: E402
from cpecan_amd import api  # noqa: E402
from cpecan_amd.workload import make_pair  # noqa: E402


def run_case(name, mtype, sx, sy, anchors, **pkw):
    okw = dict(pkw)
    p = api.pairwiseAlignmentBandingParameters_construct(**pkw)
    sm = api.stateMachine5_construct(mtype) if mtype in (0, 1) else api.stateMachine3_construct(mtype)
    want, tr = ob.aligned_pairs_traced(ob.model(mtype), sx, sy, anchors, ob.params(**okw))
    with api.Batch(sm, p, debug=True) as b:
        b.add(sx, sy, anchors)
        b.upload()
        b.run()
        b.download()
        got = b.result(0)
        st = b.stats()
        fb, tot = b.debug_fetch(0, tr["n_cells"], tr["n_diagonals"])
    ofb, otot = tr["fb_match"], tr["total_used"]
    m = ~np.isnan(ofb)
    dfb = np.abs(fb[m] - ofb[m])
    dfb[np.isnan(dfb)] = np.inf
    both_inf = np.isinf(fb[m]) & np.isinf(ofb[m]) & (fb[m] == ofb[m])
    dfb[both_inf] = 0
    mt = ~np.isnan(otot)
    dtot = np.abs(tot[mt] - otot[mt])
    dtot[np.isnan(dtot)] = np.inf
    print("%-28s cells=%d segs=%d | pairs hip=%d oracle=%d | max|dfb|=%.3e max|dtot|=%.3e | kernel %.3f ms" % (
        name, tr["n_cells"], tr["n_tracebacks"], len(got), len(want), dfb.max() if dfb.size else 0,
        dtot.max() if dtot.size else 0, st.kernelMs))
    if dtot.size and dtot.max() > 0:
        bad = np.nonzero(mt)[0][np.argmax(dtot > 0)]
        print("   first total mismatch at diagonal", bad, tot[bad], otot[bad])
    if dfb.size and dfb.max() > 0:
        idx = np.nonzero(m)[0][np.argmax(dfb > 0)]
        d = np.searchsorted(tr["cell_offset"], idx, side="right") - 1
        print("   first fb mismatch at cell", idx, "diagonal", d, "k", idx - tr["cell_offset"][d], fb[idx], ofb[idx])
    same = len(got) == len(want) and np.array_equal(got.astype(np.int64), want)
    print("   triples identical:", same)
    return same


if __name__ == "__main__":
    ok = True
    ok &= run_case("AGCG/AGTTCG 5-state", 0, "AGCG", "AGTTCG", (), threshold=0.2)
    ok &= run_case("AGCG/AGTTCG 3-state", 2, "AGCG", "AGTTCG", (), threshold=0.2)
    sx, sy, a = make_pair(1, 0, 200, 0)
    ok &= run_case("200bp no anchors 5-state", 0, sx, sy, (), diagonalExpansion=20)
    sx, sy, a = make_pair(2, 0, 1000, 50)
    ok &= run_case("1kb E=50 3-state", 2, sx, sy, a, diagonalExpansion=50)
    sx, sy, a = make_pair(3, 0, 2000, 100)
    ok &= run_case("2kb E=100 5-state", 0, sx, sy, a, diagonalExpansion=100)
    sx, sy, a = make_pair(3, 1, 600, 10)
    ok &= run_case("600bp short tracebacks", 0, sx, sy, a, diagonalExpansion=10, minDiagsBetweenTraceBack=50,
                   traceBackDiagonals=7)
    print("ALL IDENTICAL" if ok else "DIFFERENCES FOUND")
