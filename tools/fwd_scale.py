"""diagnostic: kernel time per band cell vs resident waves per CU, forward-only and full emitters.
usage: python tools/fwd_scale.py [caps...]   (needs a GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpecan_amd import api, workload

caps = [int(c) for c in sys.argv[1:]] or [1, 2, 4, 6, 8, 10]
cfg = dict(workload.CONFIGS["B"])
if os.environ.get("FWD_SCALE_EXPANSION"):
    cfg["expansion"] = int(os.environ["FWD_SCALE_EXPANSION"])
only = os.environ.get("FWD_SCALE_ONLY", "")
sM = api.stateMachine5_construct(api.fiveState)
p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=cfg["expansion"], splitMatrixBiggerThanThis=10 ** 15)
for emit, name in ((api.EMIT_FORWARD, "forward"), (api.EMIT_MATCH, "match")):
    if only and name != only:
        continue
    for cap in caps:
        n = 2 * 256 * cap
        os.environ["CPECAN_MAX_WAVES_PER_CU"] = str(cap)
        probs = workload.make_batch(cfg["seed"], n, cfg["length"], cfg["expansion"])
        with api.Batch(sM, p, emit=emit) as b:
            for pr in probs:
                b.add(*pr)
            b.upload()
            best = 1e30
            for _ in range(3):
                b.run()
                b.download()
                best = min(best, b.stats().kernelMs)
            st = b.stats()
            cyc = best * 1e-3 * 2.36e9 * cap * 256 / (st.cells / 64.0)
            simd = best * 1e-3 * 2.36e9 * 1024 / st.diagonals
            print("%s cap %2d waves %5d cells %.3e diags %.3e ms %7.2f cells/s %.3e  wave-cycles per 64 cells %.0f  SIMD-cycles per diagonal %.0f" % (
                name, cap, st.wavesPerLaunch, st.cells, st.diagonals, best, st.cells / best * 1e3, cyc, simd), flush=True)
