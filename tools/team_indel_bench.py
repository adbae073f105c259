"""kernel ms of the indel or (TEAM_BENCH_EMIT=expect) the expectation emitter on wide bands (unanchored pairs: diagonals of L + 1
cells) with the team kernel and with one wave per region (CPECAN_TEAM=0).  usage: python tools/team_indel_bench.py [pairs] [length ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpecan_amd import api
from cpecan_amd.workload import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lengths = [int(v) for v in sys.argv[2:]] or [500, 700, 900]
torch.zeros(1, device="cuda")
sm, p = api.stateMachine5_construct(), api.pairwiseAlignmentBandingParameters_construct()
expect = os.environ.get("TEAM_BENCH_EMIT") == "expect"
for L in lengths:
    probs = [make_pair(77, i, L, 0)[:2] + ((),) for i in range(n)]
    for team in (None, "0"):
        if team is None:
            os.environ.pop("CPECAN_TEAM", None)
        else:
            os.environ["CPECAN_TEAM"] = team
        ms = []
        for rep in range(3):
            with api.Batch(sm, p, emit=api.EMIT_EXPECT if expect else api.EMIT_INDEL) as b:
                for sx, sy, a in probs:
                    b.add(sx, sy, a, False, False)
                b.upload(); b.run(); b.download()
                st = b.stats()
                ms.append(st.kernelMs)
        print("%d unanchored pairs of %d bp, %s emitter, %s: kernel ms %.2f (%.3e cells/s)"
              % (n, L, "expectation" if expect else "indel", "team kernel" if team is None else "one wave per region", min(ms), st.cells / (min(ms) * 1e-3)), flush=True)
