"""Seeded synthetic workloads for tests and bench.py.

Counter-based splitmix64, so every pair's sequences depend only on (seed, pair index):
any rank can regenerate exactly its shard.  X is uniform ACGT; Y is X with 5 % substitutions,
2 % single-base deletions and 2 % single-base insertions; anchors are, for every 50th aligned
column of the true alignment, the first aligned column at or after it whose two bases are equal
(so anchors stay ~50 apart and the band is 101-~155 cells wide at expansion 100, the geometry
BASELINE.md quotes), as (x, y, expansion) triples.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(seed, counters):
    """splitmix64 output for state seed + (counter+1)*golden, vectorised over counters."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(counters, dtype=np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _unit(u):
    return (u >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def make_pair(seed, index, length, expansion, anchor_every=50, sub=0.05, dele=0.02, ins=0.02):
    """Returns (sx: bytes, sy: bytes, anchors: int64[n,3])."""
    pair_seed = int(splitmix64(seed, [index])[0])
    L = int(length)
    u = splitmix64(pair_seed, np.arange(4 * L, dtype=np.uint64))
    xb = (u[:L] & np.uint64(3)).astype(np.int64)
    r = _unit(u[L:2 * L])
    alt = (xb + 1 + (u[2 * L:3 * L] % np.uint64(3)).astype(np.int64)) % 4
    insb = (u[3 * L:4 * L] & np.uint64(3)).astype(np.int64)
    is_sub = r < sub
    is_del = (r >= sub) & (r < sub + dele)
    is_ins = (r >= sub + dele) & (r < sub + dele + ins)
    # number of Y bases contributed by column i: inserted base (if any) + the aligned base (unless deleted)
    contrib = is_ins.astype(np.int64) + (~is_del).astype(np.int64)
    ystart = np.concatenate(([0], np.cumsum(contrib)))[:-1]
    ylen = int(contrib.sum())
    y = np.zeros(ylen, dtype=np.int64)
    ins_pos = ystart[is_ins]
    y[ins_pos] = insb[is_ins]
    keep = ~is_del
    ypos = ystart + is_ins.astype(np.int64)
    yb = np.where(is_sub, alt, xb)
    y[ypos[keep]] = yb[keep]
    # anchors
    cols = np.nonzero(keep)[0]
    same = xb[cols] == yb[cols]
    # for each slot (every anchor_every-th aligned column) take the first equal-base column at or after it
    nxt = np.where(same, np.arange(len(cols)), len(cols))
    nxt = np.minimum.accumulate(nxt[::-1])[::-1]
    slots = np.arange(anchor_every // 2, len(cols), anchor_every)
    chosen = np.unique(nxt[slots])
    chosen = chosen[chosen < len(cols)]
    pick = cols[chosen]
    anchors = np.stack([pick, ypos[pick], np.full_like(pick, expansion)], axis=1).astype(np.int64)
    return _BASES[xb].tobytes(), _BASES[y].tobytes(), anchors


def make_batch(seed, n_pairs, length, expansion, first=0, **kw):
    return [make_pair(seed, first + i, length, expansion, **kw) for i in range(n_pairs)]


def make_realign_batch(seed, n_pairs, min_len=100, max_len=5000, expansion=4, first=0):
    """BASELINE config 4 (cPecanRealign mode): lengths log-uniform in [min_len, max_len]; anchors = every
    aligned column of the true alignment whose bases are equal (cPecanRealign.c:525-529 keeps exact matches only)."""
    out = []
    for i in range(n_pairs):
        u = _unit(splitmix64(seed ^ 0xC0FFEE, [first + i]))[0]
        length = int(round(np.exp(np.log(min_len) + u * (np.log(max_len) - np.log(min_len)))))
        out.append(make_pair(seed, first + i, length, expansion, anchor_every=1))
    return out


# The BASELINE.json configs this repo measures.
CONFIGS = {
    "plumbing": dict(seed=1, n_pairs=1, length=200, expansion=0, model="fiveState", anchors=False),
    "A": dict(seed=2, n_pairs=1000, length=1000, expansion=50, model="threeState", anchors=True),
    "B": dict(seed=3, n_pairs=10000, length=2000, expansion=100, model="fiveState", anchors=True),
    # configs 4 and 5 of BASELINE.json are parity / multi-GPU cases, not bench lines (see tests/test_gpu_parity.py):
    #  4: make_realign_batch(seed=4, 50000 pairs, 100-5000 bp, expansion 4), ragged ends, splitMatrixBiggerThanThis=10
    #  5: make_batch(seed=5, 100000 pairs, 1000 bp, expansion 10), expectation emitter + all-reduce of the counts
}
