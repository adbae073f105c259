"""diagnostic: throughput on narrow-band workloads (BASELINE configs 4 and 5 in miniature).  usage: python tools/narrow_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpecan_amd import api, workload

def run(name, probs, emit, raggeds, **pkw):
    sM = api.stateMachine5_construct(api.fiveState)
    p = api.pairwiseAlignmentBandingParameters_construct(**pkw)
    with api.Batch(sM, p, emit=emit) as b:
        for sx, sy, a in probs:
            b.add(sx, sy, a, *raggeds)
        t0 = time.time(); b.upload(); t1 = time.time()
        best = 1e30
        for _ in range(3):
            b.run(); b.download(); best = min(best, b.stats().kernelMs)
        st = b.stats()
        print("%s: problems %d regions %d cells %.3e diags %.3e avgW %.1f kernel ms %.2f cells/s %.3e diags/s %.3e upload %.2fs waves %d" % (
            name, st.problems, st.regions, st.cells, st.diagonals, st.cells / st.diagonals, best, st.cells / best * 1e3,
            st.diagonals / best * 1e3, t1 - t0, st.wavesPerLaunch), flush=True)

n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
maxlen = int(os.environ.get("NARROW_MAXLEN", "5000"))
run("config4-like (realign, E=4, split at 10)", workload.make_realign_batch(4, n4, 100, maxlen, 4), api.EMIT_MATCH, (True, True),
    diagonalExpansion=4, splitMatrixBiggerThanThis=10)
if len(sys.argv) > 2:
    run("config4-like, expectations (EM E-step on realign bands)", workload.make_realign_batch(4, n4, 100, maxlen, 4), api.EMIT_EXPECT,
        (True, True), diagonalExpansion=4, splitMatrixBiggerThanThis=10)
    sys.exit(0)
run("config5-like (E=10, expectations)", workload.make_batch(5, 2 * n4, 1000, 10), api.EMIT_EXPECT, (False, False), diagonalExpansion=10)
run("config5-like (E=10, match)", workload.make_batch(5, 2 * n4, 1000, 10), api.EMIT_MATCH, (False, False), diagonalExpansion=10)
