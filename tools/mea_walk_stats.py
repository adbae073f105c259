import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cpecan_amd import api, workload
from bench import model_and_params
cfg = dict(workload.CONFIGS["4"])
torch.zeros(1, device="cuda")
sm, params, mtype = model_and_params(api, cfg)
probs = workload.config_problems("4", np.arange(40))
for pi in (3, 17, 31):
    sx, sy, a = probs[pi][:3]
    m, gx, gy = api.getAlignedPairsWithIndelsUsingAnchors(sm, sx, sy, a, params, True, True)
    m = np.asarray(m).reshape(-1, 3).astype(np.int64); gx = np.asarray(gx).reshape(-1, 3).astype(np.int64); gy = np.asarray(gy).reshape(-1, 3).astype(np.int64)
    lX, lY = len(sx), len(sy)
    cx = np.zeros(lX, np.int64); cy = np.zeros(lY, np.int64)
    np.add.at(cx, gx[:, 1], gx[:, 0]); np.add.at(cy, gy[:, 2], gy[:, 0])
    cx = np.cumsum(cx); cy = np.cumsum(cy)
    n = len(m)
    def gm(c, s, l): return 0 if l == 0 else int(c[s + l - 1]) - (int(c[s - 1]) if s > 0 else 0)
    g = np.float32(0.5)
    bs = np.zeros(n + 1); rec = np.zeros(n + 1, bool); top = 0.0; walks = []
    for i in range(n + 1):
        w, x, y = (0, lX, lY) if i == n else m[i]
        score = float(np.float32(w) + np.float32(gm(cx, 0, x) + gm(cy, 0, y)) * g)
        L = 0
        for j in range(i - 1, -1, -1):
            L += 1
            x2, y2 = m[j, 1], m[j, 2]
            if x2 < x and y2 < y:
                gg = np.float32(gm(cx, x2 + 1, x - x2 - 1) + gm(cy, y2 + 1, y - y2 - 1)) * g
                sc = int((float(w) + bs[j]) + float(gg))
                if sc > score: score = float(sc)
                if rec[j]: break
        walks.append(L)
        bs[i] = score
        tail = np.float32((gm(cx, x + 1, lX - x - 1) if x < lX else 0) + (gm(cy, y + 1, lY - y - 1) if y < lY else 0)) * g
        sc = score + float(tail)
        if sc >= top: top = sc; rec[i] = True
    walks = np.array(walks)
    print("problem", pi, "lX", lX, "pairs", n, "gx", len(gx), "records", int(rec.sum()), "walk mean %.1f median %d p90 %d max %d" % (walks.mean(), np.median(walks), np.percentile(walks, 90), walks.max()), flush=True)
