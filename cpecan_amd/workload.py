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


def realign_lengths(seed, indices, min_len=100, max_len=5000):
    """Sequence length of pair i of a realign-mode batch: log-uniform in [min_len, max_len], a function of (seed, i)."""
    u = _unit(splitmix64(seed ^ 0xC0FFEE, np.asarray(indices, dtype=np.uint64)))
    return np.rint(np.exp(np.log(min_len) + u * (np.log(max_len) - np.log(min_len)))).astype(np.int64)


def make_realign_batch(seed, n_pairs, min_len=100, max_len=5000, expansion=4, first=0, indices=None):
    """BASELINE config 4 (cPecanRealign mode): lengths log-uniform in [min_len, max_len]; anchors = every
    aligned column of the true alignment whose bases are equal (cPecanRealign.c:525-529 keeps exact matches only)."""
    idx = np.arange(first, first + n_pairs) if indices is None else np.asarray(indices)
    lengths = realign_lengths(seed, idx, min_len, max_len)
    return [make_pair(seed, int(i), int(L), expansion, anchor_every=1) for i, L in zip(idx, lengths)]


def config_problems(name, indices):
    """The pairs `indices` of a BASELINE config as (sX, sY, anchors, raggedLeft, raggedRight) -- any rank can make any
    subset (strong scaling deals pairs out by cost, see pair_costs)."""
    cfg = CONFIGS[name]
    idx = [int(i) for i in indices]
    if cfg.get("realign"):
        probs = make_realign_batch(cfg["seed"], 0, cfg["min_len"], cfg["max_len"], cfg["expansion"], indices=idx)
    else:
        probs = [make_pair(cfg["seed"], i, cfg["length"], cfg["expansion"]) for i in idx]
    rg = bool(cfg.get("ragged"))
    return [(sx, sy, (a if cfg["anchors"] else ()), rg, rg) for sx, sy, a in probs]


def pair_costs(name, n_pairs):
    """Band cells a pair costs, near enough to balance ranks: diagonals (2 L) times band width; a function of the
    pair's length alone, so no sequence has to be generated to partition a batch."""
    cfg = CONFIGS[name]
    if cfg.get("realign"):
        L = realign_lengths(cfg["seed"], np.arange(n_pairs), cfg["min_len"], cfg["max_len"])
    else:
        L = np.full(n_pairs, cfg["length"], dtype=np.int64)
    width = (cfg["expansion"] + 1) if cfg["anchors"] else L
    return 2.0 * L * width


# The BASELINE.json configs this repo measures.
CONFIGS = {
    "plumbing": dict(seed=1, n_pairs=1, length=200, expansion=0, model="fiveState", anchors=False),
    "A": dict(seed=2, n_pairs=1000, length=1000, expansion=50, model="threeState", anchors=True),
    "B": dict(seed=3, n_pairs=10000, length=2000, expansion=100, model="fiveState", anchors=True),
    # cPecanRealign mode (cPecanRealign.c:355-357, :525-537): expansion 4, split at gaps of 10, ragged ends (1, 1)
    "4": dict(seed=4, n_pairs=50000, realign=True, min_len=100, max_len=5000, length=0, expansion=4, model="fiveState",
              anchors=True, ragged=True, split=10),
    # cPecanEm expectation step (pairwiseAligner.c:735-746; cPecanRealign.c:493,530-534): expectation emitter
    "5": dict(seed=5, n_pairs=100000, length=1000, expansion=10, model="fiveState", anchors=True, emit="expect"),
}
