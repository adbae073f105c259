"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under cpecan_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

FIVE_STATE, FIVE_STATE_ASYM, THREE_STATE, THREE_STATE_ASYM = 0, 1, 2, 3
PROB_1 = 10000000


class Diagonal(C.Structure):
    _fields_ = [("xay", C.c_int64), ("xmyL", C.c_int64), ("xmyR", C.c_int64)]


class Transition(C.Structure):
    _fields_ = [("block", C.c_int32), ("frm", C.c_int32), ("to", C.c_int32), ("tP", C.c_double)]


class Model(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("S", C.c_int32),
        ("matchState", C.c_int32), ("gapXState", C.c_int32), ("gapYState", C.c_int32),
        ("nTransitions", C.c_int32),
        ("tr", Transition * 16),
        ("matchEm", C.c_double * 25), ("gapXEm", C.c_double * 5), ("gapYEm", C.c_double * 5),
        ("start", C.c_double * 5), ("raggedStart", C.c_double * 5),
        ("end", C.c_double * 5), ("raggedEnd", C.c_double * 5),
    ]


class Hmm(C.Structure):
    _fields_ = [("type", C.c_int32), ("S", C.c_int32), ("T", C.c_double * 25), ("E", C.c_double * 80),
                ("likelihood", C.c_double)]


class Params(C.Structure):
    _fields_ = [
        ("threshold", C.c_double),
        ("minDiagsBetweenTraceBack", C.c_int64),
        ("traceBackDiagonals", C.c_int64),
        ("diagonalExpansion", C.c_int64),
        ("splitMatrixBiggerThanThis", C.c_int64),
        ("dynamicAnchorExpansion", C.c_int32),
    ]


class Trace(C.Structure):
    _fields_ = [
        ("nDiagonals", C.c_int64), ("nCells", C.c_int64), ("nTracebacks", C.c_int64),
        ("cellOffset", C.POINTER(C.c_int64)), ("totalUsed", C.POINTER(C.c_double)),
        ("fbMatch", C.POINTER(C.c_double)), ("forward", C.POINTER(C.c_double)),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(ORACLE_DIR, f))
                                              for f in ("cpecan_oracle.c", "cpecan_oracle.h", "Makefile"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i64p = C.POINTER(C.c_int64)
        L.orc_logAdd.restype = C.c_double
        L.orc_logAdd.argtypes = [C.c_double, C.c_double]
        L.orc_symbol.restype = C.c_int32
        L.orc_symbol.argtypes = [C.c_char]
        L.orc_diagonal_valid.restype = C.c_int
        L.orc_diagonal_valid.argtypes = [C.c_int64] * 3
        L.orc_params_default.argtypes = [C.POINTER(Params)]
        L.orc_band.restype = C.c_int
        L.orc_band.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(Diagonal)]
        L.orc_split_points.restype = C.c_int64
        L.orc_split_points.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, i64p]
        L.orc_model_default.argtypes = [C.POINTER(Model), C.c_int32]
        L.orc_model_from_hmm.restype = C.c_int
        L.orc_model_from_hmm.argtypes = [C.POINTER(Model), C.POINTER(Hmm)]
        L.orc_hmm_init.argtypes = [C.POINTER(Hmm), C.c_int32, C.c_double]
        L.orc_hmm_normalise.argtypes = [C.POINTER(Hmm)]
        dp = C.POINTER(C.c_double)
        L.orc_cell_forward.argtypes = [C.POINTER(Model), dp, dp, dp, dp, C.c_int32, C.c_int32]
        L.orc_cell_backward.argtypes = [C.POINTER(Model), dp, dp, dp, dp, C.c_int32, C.c_int32]
        common = [C.POINTER(Model), C.c_char_p, C.c_char_p, i64p, C.c_int64, C.POINTER(Params), C.c_int, C.c_int]
        L.orc_aligned_pairs.restype = C.c_int64
        L.orc_aligned_pairs.argtypes = common + [C.POINTER(i64p)]
        L.orc_aligned_pairs_traced.restype = C.c_int64
        L.orc_aligned_pairs_traced.argtypes = common + [C.POINTER(i64p), C.POINTER(Trace)]
        L.orc_aligned_pairs_with_indels.argtypes = common + [C.POINTER(i64p), i64p] * 3
        L.orc_expectations.argtypes = [C.POINTER(Model), C.POINTER(Hmm), C.c_char_p, C.c_char_p, i64p, C.c_int64,
                                       C.POINTER(Params), C.c_int, C.c_int]
        L.orc_forward_prob.restype = C.c_double
        L.orc_forward_prob.argtypes = common
        L.orc_band_cells.restype = C.c_int64
        L.orc_band_cells.argtypes = [C.c_char_p, C.c_char_p, i64p, C.c_int64, C.POINTER(Params), C.c_int, C.c_int]
        L.orc_batch_aligned_pairs.restype = C.c_int64
        L.orc_batch_aligned_pairs.argtypes = [C.POINTER(Model), C.c_char_p, i64p, i64p, i64p, C.c_int64,
                                              C.POINTER(Params), C.c_int, C.c_int, C.c_int, i64p]
        L.orc_batch_expectations.restype = C.c_int64
        L.orc_batch_expectations.argtypes = [C.POINTER(Model), C.POINTER(Hmm), C.c_char_p, i64p, i64p, i64p, C.c_int64,
                                             C.POINTER(Params), C.c_int, C.c_int, C.c_int]
        L.orc_reweight_aligned_pairs.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_double]
        L.orc_score_by_posterior.restype = C.c_double
        L.orc_score_by_posterior.argtypes = [C.c_int64, C.c_int64, i64p, C.c_int64]
        L.orc_score_by_posterior_ignoring_gaps.restype = C.c_double
        L.orc_score_by_posterior_ignoring_gaps.argtypes = [i64p, C.c_int64]
        L.orc_mea_alignment.restype = C.c_int64
        L.orc_mea_alignment.argtypes = [i64p, C.c_int64, i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                                        i64p, dp]
        L.orc_left_shift_alignment.restype = C.c_int64
        L.orc_left_shift_alignment.argtypes = [i64p, C.c_int64, C.c_char_p, C.c_char_p, i64p]
        L.orc_score_by_identity.restype = C.c_double
        L.orc_score_by_identity.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, i64p, C.c_int64]
        L.orc_score_by_identity_ignoring_gaps.restype = C.c_double
        L.orc_score_by_identity_ignoring_gaps.argtypes = [C.c_char_p, C.c_char_p, i64p, C.c_int64]
        L.orc_filter_pairs_ordered.restype = C.c_int64
        L.orc_filter_pairs_ordered.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_double, i64p]
        L.orc_filter_to_remove_overlap.restype = C.c_int64
        L.orc_filter_to_remove_overlap.argtypes = [i64p, C.c_int64, i64p]
        L.orc_trace_free.argtypes = [C.POINTER(Trace)]
        L.orc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _i64(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int64).reshape(-1))
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def _anchors(anchors):
    """anchors: iterable of (x, y) or (x, y, expansion) -> flat int64 triples."""
    rows = []
    for t in anchors:
        t = tuple(int(v) for v in t)
        rows.append(t if len(t) == 3 else (t[0], t[1], 0))
    arr = np.array(rows, dtype=np.int64).reshape(-1)
    if arr.size == 0:
        arr = np.zeros(3, dtype=np.int64)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_int64)), len(rows)


def params(**kw):
    p = Params()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def model(mtype=FIVE_STATE):
    m = Model()
    lib().orc_model_default(C.byref(m), mtype)
    return m


def hmm(mtype, pseudo=0.0):
    h = Hmm()
    lib().orc_hmm_init(C.byref(h), mtype, pseudo)
    return h


def model_from_hmm(h):
    m = Model()
    if lib().orc_model_from_hmm(C.byref(m), C.byref(h)) != 0:
        raise ValueError("bad hmm type")
    return m


def log_add(x, y):
    return lib().orc_logAdd(x, y)


def symbol(ch):
    return lib().orc_symbol(ch.encode() if isinstance(ch, str) else ch)


def band(anchors, lX, lY, expansion, dynamic=False):
    arr, ptr, n = _anchors(anchors)
    out = (Diagonal * (lX + lY + 1))()
    rc = lib().orc_band(ptr, n, lX, lY, expansion, int(dynamic), out)
    if rc != 0:
        raise ValueError("invalid diagonal")
    return [(d.xay, d.xmyL, d.xmyR) for d in out]


def split_points(anchors, lX, lY, max_matrix, ragged_left, ragged_right):
    arr, ptr, n = _anchors(anchors)
    out = np.zeros(4 * (n + 2), dtype=np.int64)
    cnt = lib().orc_split_points(ptr, n, lX, lY, max_matrix, int(ragged_left), int(ragged_right),
                                 out.ctypes.data_as(C.POINTER(C.c_int64)))
    return [tuple(int(v) for v in out[4 * i:4 * i + 4]) for i in range(cnt)]


def _take(ptr, n):
    if n == 0:
        out = np.zeros((0, 3), dtype=np.int64)
    else:
        out = np.ctypeslib.as_array(ptr, shape=(n * 3,)).copy().reshape(n, 3)
    lib().orc_free(C.cast(ptr, C.c_void_p))
    return out


def _b(s):
    return s.encode() if isinstance(s, str) else s


def aligned_pairs(m, sx, sy, anchors=(), p=None, ragged_left=False, ragged_right=False):
    p = p or params()
    arr, ptr, n = _anchors(anchors)
    out = C.POINTER(C.c_int64)()
    cnt = lib().orc_aligned_pairs(C.byref(m), _b(sx), _b(sy), ptr, n, C.byref(p), int(ragged_left), int(ragged_right),
                                  C.byref(out))
    return _take(out, cnt)


def aligned_pairs_traced(m, sx, sy, anchors=(), p=None, ragged_left=False, ragged_right=False):
    p = p or params()
    arr, ptr, n = _anchors(anchors)
    out = C.POINTER(C.c_int64)()
    tr = Trace()
    cnt = lib().orc_aligned_pairs_traced(C.byref(m), _b(sx), _b(sy), ptr, n, C.byref(p), int(ragged_left),
                                         int(ragged_right), C.byref(out), C.byref(tr))
    pairs = _take(out, cnt)
    info = {}
    if tr.nDiagonals > 0:
        nd, nc = tr.nDiagonals, tr.nCells
        info = dict(
            n_diagonals=nd, n_cells=nc, n_tracebacks=tr.nTracebacks,
            cell_offset=np.ctypeslib.as_array(tr.cellOffset, shape=(nd + 1,)).copy(),
            total_used=np.ctypeslib.as_array(tr.totalUsed, shape=(nd,)).copy(),
            fb_match=np.ctypeslib.as_array(tr.fbMatch, shape=(nc,)).copy(),
            forward=np.ctypeslib.as_array(tr.forward, shape=(nc * m.S,)).copy().reshape(nc, m.S),
        )
        lib().orc_trace_free(C.byref(tr))
    return pairs, info


def aligned_pairs_with_indels(m, sx, sy, anchors=(), p=None, ragged_left=False, ragged_right=False):
    p = p or params()
    arr, ptr, n = _anchors(anchors)
    outs = [C.POINTER(C.c_int64)() for _ in range(3)]
    cnts = [C.c_int64() for _ in range(3)]
    lib().orc_aligned_pairs_with_indels(C.byref(m), _b(sx), _b(sy), ptr, n, C.byref(p), int(ragged_left),
                                        int(ragged_right), C.byref(outs[0]), C.byref(cnts[0]), C.byref(outs[1]),
                                        C.byref(cnts[1]), C.byref(outs[2]), C.byref(cnts[2]))
    return tuple(_take(o, c.value) for o, c in zip(outs, cnts))


def expectations(m, acc, sx, sy, anchors=(), p=None, ragged_left=False, ragged_right=False):
    p = p or params()
    arr, ptr, n = _anchors(anchors)
    lib().orc_expectations(C.byref(m), C.byref(acc), _b(sx), _b(sy), ptr, n, C.byref(p), int(ragged_left),
                           int(ragged_right))
    return acc


def forward_prob(m, sx, sy, anchors=(), p=None, ragged_left=False, ragged_right=False):
    p = p or params()
    arr, ptr, n = _anchors(anchors)
    return lib().orc_forward_prob(C.byref(m), _b(sx), _b(sy), ptr, n, C.byref(p), int(ragged_left), int(ragged_right))


def band_cells(sx, sy, anchors=(), p=None, ragged_left=False, ragged_right=False):
    p = p or params()
    arr, ptr, n = _anchors(anchors)
    return lib().orc_band_cells(_b(sx), _b(sy), ptr, n, C.byref(p), int(ragged_left), int(ragged_right))


def _pack_problems(problems):
    """(sx, sy, anchors[, ...]) tuples as one NUL-separated sequence blob + offsets, and one anchor array + offsets."""
    blob = bytearray()
    seq_off = []
    anchor_rows = []
    anchor_off = [0]
    for pr in problems:
        sx, sy, anchors = pr[0], pr[1], pr[2]
        seq_off.append(len(blob)); blob += _b(sx) + b"\0"
        seq_off.append(len(blob)); blob += _b(sy) + b"\0"
        a = np.asarray(anchors, dtype=np.int64).reshape(-1, 3)
        anchor_rows.append(a)
        anchor_off.append(anchor_off[-1] + len(a))
    aa = np.concatenate(anchor_rows) if anchor_rows else np.zeros((0, 3), dtype=np.int64)
    if aa.size == 0:
        aa = np.zeros((1, 3), dtype=np.int64)
    return bytes(blob), _i64(seq_off), _i64(aa), _i64(anchor_off)


def batch_aligned_pairs(m, problems, p=None, ragged_left=False, ragged_right=False, threads=1):
    """problems: list of (sx, sy, anchors[n,3][, ...]). Returns (pairs emitted, band cells)."""
    p = p or params()
    blob, (so, sop), (aa, aap), (ao, aop) = _pack_problems(problems)
    cells = C.c_int64()
    n = lib().orc_batch_aligned_pairs(C.byref(m), blob, sop, aap, aop, len(problems), C.byref(p),
                                      int(ragged_left), int(ragged_right), int(threads), C.byref(cells))
    return n, cells.value


def batch_expectations(m, problems, p, acc, ragged_left=False, ragged_right=False, threads=1):
    """orc_expectations over a batch with OpenMP over problems; counts are added to acc. Returns the band cells."""
    blob, (so, sop), (aa, aap), (ao, aop) = _pack_problems(problems)
    return lib().orc_batch_expectations(C.byref(m), C.byref(acc), blob, sop, aap, aop, len(problems), C.byref(p),
                                        int(ragged_left), int(ragged_right), int(threads))


# ---- consumers of the posterior lists ----
def _triples(t):
    a = np.ascontiguousarray(np.asarray(t, dtype=np.int64).reshape(-1, 3))
    return a, a.ctypes.data_as(C.POINTER(C.c_int64)), len(a)


def reweight_aligned_pairs(pairs, lX, lY, gap_gamma):
    a, ptr, n = _triples(pairs)
    a = a.copy()
    lib().orc_reweight_aligned_pairs(a.ctypes.data_as(C.POINTER(C.c_int64)), n, lX, lY, float(gap_gamma))
    return a


def score_by_posterior(lX, lY, pairs):
    a, ptr, n = _triples(pairs)
    return lib().orc_score_by_posterior(lX, lY, ptr, n)


def score_by_posterior_ignoring_gaps(pairs):
    a, ptr, n = _triples(pairs)
    return lib().orc_score_by_posterior_ignoring_gaps(ptr, n)


def mea_alignment(pairs, gap_x, gap_y, lX, lY, gap_gamma):
    """Returns (alignment int64[n,3], alignmentScore)."""
    a, pa, n = _triples(pairs)
    gx, pgx, ngx = _triples(gap_x)
    gy, pgy, ngy = _triples(gap_y)
    out = np.zeros((max(n, 1), 3), dtype=np.int64)
    score = C.c_double()
    cnt = lib().orc_mea_alignment(pa, n, pgx, ngx, pgy, ngy, lX, lY, C.c_float(gap_gamma),
                                  out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(score))
    return out[:cnt].copy(), score.value


def left_shift_alignment(pairs, sx, sy):
    a, pa, n = _triples(pairs)
    out = np.zeros((n + min(len(sx), len(sy)) + 1, 3), dtype=np.int64)
    cnt = lib().orc_left_shift_alignment(pa, n, _b(sx), _b(sy), out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:cnt].copy()


def score_by_identity(sx, sy, triples):
    a, pa, n = _triples(triples)
    return lib().orc_score_by_identity(_b(sx), _b(sy), len(sx), len(sy), pa, n)


def score_by_identity_ignoring_gaps(sx, sy, triples):
    a, pa, n = _triples(triples)
    return lib().orc_score_by_identity_ignoring_gaps(_b(sx), _b(sy), pa, n)


def filter_pairs_ordered(pairs, lX, lY, match_gamma):
    """filterPairwiseAlignmentToMakePairsOrdered without the reference's random jitter; matchGamma is the float of
    cPecanRealign.c:355 widened to double, as the reference passes it."""
    a, pa, n = _triples(pairs)
    out = np.zeros((max(n, 1), 3), dtype=np.int64)
    cnt = lib().orc_filter_pairs_ordered(pa, n, lX, lY, float(np.float32(match_gamma)),
                                         out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:cnt].copy()


def filter_to_remove_overlap(pairs):
    a, pa, n = _triples(pairs)
    out = np.zeros((max(n, 1), 3), dtype=np.int64)
    cnt = lib().orc_filter_to_remove_overlap(pa, n, out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:cnt].copy()
