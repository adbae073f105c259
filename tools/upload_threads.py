"""Host planning + upload time of a config-B batch (10 000 pairs x 2 kb, expansion 100) for several host thread counts
(CPECAN_THREADS); the first upload of the process (HIP start-up) is made on a small batch and not reported."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from cpecan_amd import api
    from cpecan_amd.workload import make_pair
    n = int(sys.argv[2])
    probs = [make_pair(seed=1, index=i, length=2000, expansion=100) for i in range(min(n, 256))]
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=100)
    with api.Batch(api.stateMachine5_construct(), p) as b:
        b.add(*probs[0])
        b.upload()
    for rep in range(2):
        with api.Batch(api.stateMachine5_construct(), p) as b:
            for i in range(n):
                b.add(*probs[i % len(probs)])
            t = time.time()
            b.upload()
            print("threads=%s upload %.3f s" % (os.environ.get("CPECAN_THREADS", "default"), time.time() - t), flush=True)
else:
    for thr in ("4", "16", "64", "256"):
        env = dict(os.environ, CPECAN_THREADS=thr)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", "10000"], env=env, check=True)
