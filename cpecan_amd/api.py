"""ctypes binding of libcpecan_hip.so (include/cpecan_hip.h) and a host-side mirror of the reference's
operator interface for the hot path.

Names follow the reference (inc/pairwiseAligner.h, inc/stateMachine.h): ``stateMachine5_construct``,
``pairwiseAlignmentBandingParameters_construct``, ``getAlignedPairsUsingAnchors`` ... so the parity tests read
like tests/pairwiseAlignerTest.c.  All DP work happens in the HIP library; if it is missing or no GPU is
present every compute call raises -- there is no CPU fallback here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPECAN_LIB") or os.path.join(_HERE, "libcpecan_hip.so")  # CPECAN_LIB: A/B builds of the same ABI (tools/)

fiveState, fiveStateAsymmetric, threeState, threeStateAsymmetric = 0, 1, 2, 3  # inc/stateMachine.h:28-33
EMIT_MATCH, EMIT_INDEL, EMIT_EXPECT, EMIT_FORWARD = 0, 1, 2, 3
PAIR_ALIGNMENT_PROB_1 = 10000000  # inc/pairwiseAligner.h:26


class CpecanError(RuntimeError):
    pass


class StateMachine(C.Structure):
    """cpecan_model: the flattened StateMachine5/StateMachine3 (impl/stateMachine.c:377-399, 631-646)."""
    _fields_ = [
        ("type", C.c_int32), ("reserved", C.c_int32),
        ("matchContinue", C.c_double),
        ("matchFromShortGapX", C.c_double), ("matchFromShortGapY", C.c_double),
        ("matchFromLongGapX", C.c_double), ("matchFromLongGapY", C.c_double),
        ("gapShortOpenX", C.c_double), ("gapShortOpenY", C.c_double),
        ("gapShortExtendX", C.c_double), ("gapShortExtendY", C.c_double),
        ("gapShortSwitchToX", C.c_double), ("gapShortSwitchToY", C.c_double),
        ("gapLongOpenX", C.c_double), ("gapLongOpenY", C.c_double),
        ("gapLongExtendX", C.c_double), ("gapLongExtendY", C.c_double),
        ("gapLongSwitchToX", C.c_double), ("gapLongSwitchToY", C.c_double),
        ("emissionMatch", C.c_double * 16), ("emissionGapX", C.c_double * 4), ("emissionGapY", C.c_double * 4),
    ]

    @property
    def stateNumber(self):
        return 5 if self.type in (fiveState, fiveStateAsymmetric) else 3


class Hmm(C.Structure):
    """cpecan_hmm (inc/stateMachine.h:61-67)."""
    _fields_ = [("type", C.c_int32), ("stateNumber", C.c_int32), ("transitions", C.c_double * 25),
                ("emissions", C.c_double * 80), ("likelihood", C.c_double)]


class PairwiseAlignmentParameters(C.Structure):
    """cpecan_params: the fields of PairwiseAlignmentParameters the DP reads (inc/pairwiseAligner.h:28-41)."""
    _fields_ = [
        ("threshold", C.c_double),
        ("minDiagsBetweenTraceBack", C.c_int64),
        ("traceBackDiagonals", C.c_int64),
        ("diagonalExpansion", C.c_int64),
        ("splitMatrixBiggerThanThis", C.c_int64),
        ("dynamicAnchorExpansion", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Problem(C.Structure):
    """cpecan_problem: one element of cpecan_batch_add_many."""
    _fields_ = [("sX", C.c_char_p), ("lX", C.c_int64), ("sY", C.c_char_p), ("lY", C.c_int64),
                ("anchors", C.POINTER(C.c_int64)), ("nAnchors", C.c_int64),
                ("raggedLeft", C.c_int32), ("raggedRight", C.c_int32)]


class ProblemRuns(C.Structure):
    """cpecan_problem_runs: one element of cpecan_batch_add_many_runs (anchors as (x, y, length, expansion) runs)."""
    _fields_ = [("sX", C.c_char_p), ("lX", C.c_int64), ("sY", C.c_char_p), ("lY", C.c_int64),
                ("runs", C.POINTER(C.c_int64)), ("nRuns", C.c_int64),
                ("raggedLeft", C.c_int32), ("raggedRight", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("problems", C.c_int64), ("regions", C.c_int64), ("cells", C.c_int64), ("diagonals", C.c_int64),
        ("pairs", C.c_int64), ("deviceBytes", C.c_int64),
        ("kernelMs", C.c_double), ("h2dMs", C.c_double), ("d2hMs", C.c_double),
        ("launches", C.c_int32), ("wavesPerLaunch", C.c_int32), ("launchForm", C.c_int32), ("reserved", C.c_int32),
    ]


# Every symbol include/cpecan_hip.h declares (checked by tests/test_abi.py).
EXPORTS = [
    "cpecan_model_default", "cpecan_model_from_hmm", "cpecan_hmm_init", "cpecan_hmm_normalise", "cpecan_hmm_write",
    "cpecan_hmm_load", "cpecan_params_default", "cpecan_band", "cpecan_split_points", "cpecan_device_count", "cpecan_current_device",
    "cpecan_last_error", "cpecan_batch_create", "cpecan_batch_destroy", "cpecan_batch_add", "cpecan_batch_upload",
    "cpecan_batch_run", "cpecan_batch_download", "cpecan_batch_download_begin", "cpecan_batch_download_end", "cpecan_batch_result", "cpecan_batch_expectations",
    "cpecan_batch_stats", "cpecan_batch_set_debug", "cpecan_batch_debug_fetch", "cpecan_batch_forward_prob",
    "cpecan_get_aligned_pairs_using_anchors", "cpecan_get_aligned_pairs_with_indels_using_anchors",
    "cpecan_compute_forward_probability", "cpecan_free",
    "cpecan_batch_set_post", "cpecan_batch_scores", "cpecan_reweight_aligned_pairs", "cpecan_posterior_scores",
    "cpecan_mea_alignment", "cpecan_left_shift_alignment", "cpecan_get_shifted_mea_alignment",
    "cpecan_anchors_from_alignment", "cpecan_batch_set_match_gamma", "cpecan_batch_identity_scores",
    "cpecan_identity_scores", "cpecan_filter_pairs_ordered", "cpecan_batch_add_many",
    "cpecan_filter_to_remove_overlap", "cpecan_cache_trim", "cpecan_ref_cells",
    "cpecan_batch_add_many_runs", "cpecan_anchor_runs", "cpecan_anchor_runs_from_alignment",
]
OP_MATCH, OP_INDEL_X, OP_INDEL_Y = 0, 1, 2
POST_REWEIGHT, POST_MEA, POST_LEFT_SHIFT, POST_ORDERED = 1, 2, 4, 8

_lib = None


def lib():
    """Loads the HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CpecanError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                          "`make -C cpecan_amd/csrc`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    i64p, i32p, dp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    vp = C.c_void_p
    L.cpecan_model_default.argtypes = [C.POINTER(StateMachine), C.c_int32]
    L.cpecan_model_from_hmm.argtypes = [C.POINTER(StateMachine), C.POINTER(Hmm)]
    L.cpecan_hmm_init.argtypes = [C.POINTER(Hmm), C.c_int32, C.c_double]
    L.cpecan_hmm_normalise.argtypes = [C.POINTER(Hmm)]
    L.cpecan_hmm_write.argtypes = [C.POINTER(Hmm), C.c_char_p]
    L.cpecan_hmm_load.argtypes = [C.POINTER(Hmm), C.c_char_p]
    L.cpecan_params_default.argtypes = [C.POINTER(PairwiseAlignmentParameters)]
    L.cpecan_band.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, i64p]
    L.cpecan_split_points.restype = C.c_int64
    L.cpecan_split_points.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, i64p]
    L.cpecan_device_count.restype = C.c_int
    L.cpecan_cache_trim.restype = C.c_int64
    L.cpecan_cache_trim.argtypes = [C.c_int]
    L.cpecan_last_error.restype = C.c_char_p
    L.cpecan_batch_create.argtypes = [C.POINTER(vp), C.POINTER(StateMachine), C.POINTER(PairwiseAlignmentParameters),
                                      C.c_int, C.c_int]
    L.cpecan_batch_destroy.argtypes = [vp]
    L.cpecan_batch_destroy.restype = None
    L.cpecan_batch_add.restype = C.c_int64
    L.cpecan_batch_add.argtypes = [vp, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, i64p, C.c_int64, C.c_int, C.c_int]
    L.cpecan_batch_upload.argtypes = [vp]
    L.cpecan_batch_run.argtypes = [vp, vp]
    L.cpecan_batch_download.argtypes = [vp]
    L.cpecan_batch_download_begin.argtypes = [vp]
    L.cpecan_batch_download_end.argtypes = [vp]
    L.cpecan_batch_result.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(i32p), i64p]
    L.cpecan_batch_expectations.argtypes = [vp, C.POINTER(Hmm)]
    L.cpecan_batch_stats.argtypes = [vp, C.POINTER(Stats)]
    L.cpecan_batch_set_debug.argtypes = [vp, C.c_int]
    L.cpecan_batch_debug_fetch.argtypes = [vp, C.c_int64, dp, C.c_int64, dp, C.c_int64]
    L.cpecan_get_aligned_pairs_using_anchors.argtypes = [
        C.POINTER(StateMachine), C.c_char_p, C.c_char_p, i64p, C.c_int64, C.POINTER(PairwiseAlignmentParameters),
        C.c_int, C.c_int, C.POINTER(i32p), i64p]
    L.cpecan_batch_forward_prob.argtypes = [vp, C.c_int64, dp]
    L.cpecan_get_aligned_pairs_with_indels_using_anchors.argtypes = [
        C.POINTER(StateMachine), C.c_char_p, C.c_char_p, i64p, C.c_int64, C.POINTER(PairwiseAlignmentParameters),
        C.c_int, C.c_int, C.POINTER(i32p), i64p, C.POINTER(i32p), i64p, C.POINTER(i32p), i64p]
    L.cpecan_compute_forward_probability.argtypes = [
        C.POINTER(StateMachine), C.c_char_p, C.c_char_p, i64p, C.c_int64, C.POINTER(PairwiseAlignmentParameters),
        C.c_int, C.c_int, dp]
    L.cpecan_free.argtypes = [vp]
    L.cpecan_free.restype = None
    L.cpecan_anchors_from_alignment.restype = C.c_int64
    L.cpecan_anchors_from_alignment.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_char_p,
                                                C.c_int64, C.c_char_p, C.c_int64, i64p]
    L.cpecan_anchor_runs_from_alignment.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_char_p,
                                                    C.c_int64, C.c_char_p, C.c_int64, i64p, C.c_int64]
    L.cpecan_anchor_runs_from_alignment.restype = C.c_int64
    L.cpecan_batch_set_post.argtypes = [vp, C.c_int, C.c_double]
    L.cpecan_batch_scores.argtypes = [vp, C.c_int64, dp, dp, dp]
    L.cpecan_filter_to_remove_overlap.argtypes = [i64p, C.c_int64, i64p]
    L.cpecan_filter_to_remove_overlap.restype = C.c_int64
    L.cpecan_batch_add_many.argtypes = [vp, C.POINTER(Problem), C.c_int64]
    L.cpecan_batch_add_many.restype = C.c_int64
    L.cpecan_batch_add_many_runs.argtypes = [vp, C.POINTER(ProblemRuns), C.c_int64]
    L.cpecan_batch_add_many_runs.restype = C.c_int64
    L.cpecan_anchor_runs.argtypes = [i64p, C.c_int64, i64p, C.c_int64]
    L.cpecan_anchor_runs.restype = C.c_int64
    L.cpecan_batch_set_match_gamma.argtypes = [vp, C.c_float]
    L.cpecan_batch_identity_scores.argtypes = [vp, C.c_int64, dp, dp]
    L.cpecan_identity_scores.argtypes = [i32p, C.c_int64, C.c_char_p, C.c_char_p, dp, dp]
    L.cpecan_filter_pairs_ordered.argtypes = [i32p, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.POINTER(i32p), i64p]
    L.cpecan_reweight_aligned_pairs.argtypes = [i32p, C.c_int64, C.c_int64, C.c_int64, C.c_double]
    L.cpecan_posterior_scores.argtypes = [i32p, C.c_int64, C.c_int64, C.c_int64, dp, dp]
    L.cpecan_mea_alignment.argtypes = [i32p, C.c_int64, i32p, C.c_int64, i32p, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                                       C.POINTER(i32p), i64p, dp]
    L.cpecan_left_shift_alignment.argtypes = [i32p, C.c_int64, C.c_char_p, C.c_char_p, C.POINTER(i32p), i64p]
    L.cpecan_get_shifted_mea_alignment.argtypes = [
        C.POINTER(StateMachine), C.c_char_p, C.c_char_p, i64p, C.c_int64, C.POINTER(PairwiseAlignmentParameters),
        C.c_float, C.c_int, C.c_int, C.POINTER(i32p), i64p, dp]
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        msg = lib().cpecan_last_error()
        raise CpecanError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
    return rc


def _bytes(s):
    return s.encode() if isinstance(s, str) else bytes(s)


def _anchor_array(anchorPairs):
    """anchorPairs: sequence of (x, y) or (x, y, expansion) -- the stIntTuple list of the reference -- or int64[n,3]."""
    if isinstance(anchorPairs, np.ndarray):
        a = np.ascontiguousarray(anchorPairs, dtype=np.int64).reshape(-1, 3)
    else:
        rows = [(int(t[0]), int(t[1]), int(t[2]) if len(t) > 2 else 0) for t in anchorPairs]
        a = np.array(rows, dtype=np.int64).reshape(-1, 3)
    n = a.shape[0]
    if n == 0:
        a = np.zeros((1, 3), dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64)), n


def anchor_runs(anchorPairs):
    """cpecan_anchor_runs: the anchors as int64[nRuns, 4] quadruples (x, y, length, expansion)."""
    a, ptr, n = _anchor_array(anchorPairs)
    if n == 0:
        return np.zeros((0, 4), dtype=np.int64)
    cnt = _check(lib().cpecan_anchor_runs(ptr, n, None, 0), "cpecan_anchor_runs")
    out = np.zeros((max(1, cnt), 4), dtype=np.int64)
    _check(lib().cpecan_anchor_runs(ptr, n, out.ctypes.data_as(C.POINTER(C.c_int64)), cnt), "cpecan_anchor_runs")
    return out[:cnt]


# ---- model / parameter constructors, named as in the reference ----
def stateMachine5_construct(type=fiveState):  # impl/stateMachine.c:482
    m = StateMachine()
    _check(lib().cpecan_model_default(C.byref(m), type), "stateMachine5_construct")
    if m.stateNumber != 5:
        raise CpecanError("Wrong type for five state %i" % type)
    return m


def stateMachine3_construct(type=threeState):  # impl/stateMachine.c:716
    m = StateMachine()
    _check(lib().cpecan_model_default(C.byref(m), type), "stateMachine3_construct")
    if m.stateNumber != 3:
        raise CpecanError("Tried to create a three state state-machine with the wrong type")
    return m


def hmm_constructEmpty(pseudoExpectation, type):  # impl/stateMachine.c:23
    h = Hmm()
    _check(lib().cpecan_hmm_init(C.byref(h), type, pseudoExpectation), "hmm_constructEmpty")
    return h


def hmm_normalise(hmm):  # impl/stateMachine.c:88
    _check(lib().cpecan_hmm_normalise(C.byref(hmm)), "hmm_normalise")


def hmm_getStateMachine(hmm):  # impl/stateMachine.c:797
    m = StateMachine()
    _check(lib().cpecan_model_from_hmm(C.byref(m), C.byref(hmm)), "hmm_getStateMachine")
    return m


def hmm_write(hmm, path):  # impl/stateMachine.c:133
    _check(lib().cpecan_hmm_write(C.byref(hmm), _bytes(path)), "hmm_write")


def hmm_loadFromFile(path):  # impl/stateMachine.c:145
    h = Hmm()
    _check(lib().cpecan_hmm_load(C.byref(h), _bytes(path)), "hmm_loadFromFile")
    return h


def pairwiseAlignmentBandingParameters_construct(**overrides):  # impl/pairwiseAligner.c:1334
    p = PairwiseAlignmentParameters()
    _check(lib().cpecan_params_default(C.byref(p)), "pairwiseAlignmentBandingParameters_construct")
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def band_construct(anchorPairs, lX, lY, expansion, dynamic=False):  # impl/pairwiseAligner.c:128-234
    a, ptr, n = _anchor_array(anchorPairs)
    out = np.zeros(3 * (lX + lY + 1), dtype=np.int64)
    _check(lib().cpecan_band(ptr, n, lX, lY, expansion, int(dynamic), out.ctypes.data_as(C.POINTER(C.c_int64))),
           "band_construct")
    return [tuple(int(v) for v in out[3 * i:3 * i + 3]) for i in range(lX + lY + 1)]


def getSplitPoints(anchorPairs, lX, lY, maxMatrixSize, raggedLeft, raggedRight):  # impl/pairwiseAligner.c:1230
    a, ptr, n = _anchor_array(anchorPairs)
    out = np.zeros(4 * (n + 2), dtype=np.int64)
    cnt = _check(lib().cpecan_split_points(ptr, n, lX, lY, maxMatrixSize, int(raggedLeft), int(raggedRight),
                                           out.ctypes.data_as(C.POINTER(C.c_int64))), "getSplitPoints")
    return [tuple(int(v) for v in out[4 * i:4 * i + 4]) for i in range(cnt)]


def device_count():
    return lib().cpecan_device_count()


def cache_trim(device=-1):
    """Idle device blocks of `device` (-1: every device) and idle host blocks go back to the driver; bytes released."""
    return int(lib().cpecan_cache_trim(device))


class Batch:
    """N independent alignment problems on one GPU (cpecan_batch_*)."""

    def __init__(self, sM, p=None, emit=EMIT_MATCH, device=0, debug=False):
        self._h = C.c_void_p()
        self._p = p or pairwiseAlignmentBandingParameters_construct()
        self._sM = sM
        _check(lib().cpecan_batch_create(C.byref(self._h), C.byref(sM), C.byref(self._p), emit, device),
               "cpecan_batch_create")
        if debug:
            _check(lib().cpecan_batch_set_debug(self._h, 1), "cpecan_batch_set_debug")
        self.n = 0

    def close(self):
        if self._h:
            lib().cpecan_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add(self, sX, sY, anchorPairs=(), raggedLeft=False, raggedRight=False):
        sx, sy = _bytes(sX), _bytes(sY)
        a, ptr, n = _anchor_array(anchorPairs)
        idx = _check(lib().cpecan_batch_add(self._h, sx, len(sx), sy, len(sy), ptr, n, int(raggedLeft),
                                            int(raggedRight)), "cpecan_batch_add")
        self.n += 1
        return idx

    @staticmethod
    def prepare_problems(problems):
        """The C-ABI form of a list of problems -- (sX, sY, anchorPairs[, raggedLeft, raggedRight]) -- as one
        cpecan_problem array: what a C caller holds anyway.  Returns (array, n, keepalive)."""
        problems = list(problems)
        arr = (Problem * max(1, len(problems)))()
        keep = []
        for i, pr in enumerate(problems):
            sx, sy = _bytes(pr[0]), _bytes(pr[1])
            a, ptr, n = _anchor_array(pr[2] if len(pr) > 2 else ())
            keep.append((sx, sy, a))
            arr[i].sX, arr[i].lX, arr[i].sY, arr[i].lY = sx, len(sx), sy, len(sy)
            arr[i].anchors, arr[i].nAnchors = ptr, n
            arr[i].raggedLeft = int(pr[3]) if len(pr) > 3 else 0
            arr[i].raggedRight = int(pr[4]) if len(pr) > 4 else 0
        return arr, len(problems), keep

    def add_prepared(self, arr, n):
        if isinstance(arr, C.Array) and arr._type_ is ProblemRuns:
            first = _check(lib().cpecan_batch_add_many_runs(self._h, arr, n), "cpecan_batch_add_many_runs")
        else:
            first = _check(lib().cpecan_batch_add_many(self._h, arr, n), "cpecan_batch_add_many")
        self.n += n
        return first

    @staticmethod
    def prepare_problems_runs(problems):
        """The run form of the same list (cpecan_problem_runs): the anchors of every problem as (x, y, length, expansion)
        quadruples, one per run of diagonal neighbours (cpecan_anchor_runs) -- what a realign-style caller holds before it
        expands the match operations of a cigar into one anchor per column.  Returns (array, n, keepalive)."""
        problems = list(problems)
        arr = (ProblemRuns * max(1, len(problems)))()
        keep = []
        for i, pr in enumerate(problems):
            sx, sy = _bytes(pr[0]), _bytes(pr[1])
            runs = anchor_runs(pr[2] if len(pr) > 2 else ())
            keep.append((sx, sy, runs))
            arr[i].sX, arr[i].lX, arr[i].sY, arr[i].lY = sx, len(sx), sy, len(sy)
            arr[i].runs, arr[i].nRuns = runs.ctypes.data_as(C.POINTER(C.c_int64)), (runs.shape[0] if runs.size else 0)
            arr[i].raggedLeft = int(pr[3]) if len(pr) > 3 else 0
            arr[i].raggedRight = int(pr[4]) if len(pr) > 4 else 0
        return arr, len(problems), keep

    def add_many_runs(self, problems):
        arr, n, _keep = self.prepare_problems_runs(problems)
        return self.add_prepared(arr, n)

    def add_many(self, problems):
        """problems: iterable of (sX, sY, anchorPairs[, raggedLeft, raggedRight]); cut, converted and copied in parallel
        by cpecan_batch_add_many.  Returns the index of the first."""
        arr, n, _keep = self.prepare_problems(problems)
        return self.add_prepared(arr, n)

    def upload(self):
        _check(lib().cpecan_batch_upload(self._h), "cpecan_batch_upload")

    def run(self, stream=None):
        _check(lib().cpecan_batch_run(self._h, C.c_void_p(stream or 0)), "cpecan_batch_run")

    def download(self):
        _check(lib().cpecan_batch_download(self._h), "cpecan_batch_download")

    def download_begin(self):
        """download() on a helper thread of the batch's own; download_end() waits for it."""
        _check(lib().cpecan_batch_download_begin(self._h), "cpecan_batch_download_begin")

    def download_end(self):
        _check(lib().cpecan_batch_download_end(self._h), "cpecan_batch_download_end")

    def result(self, problem, which=0):
        ptr = C.POINTER(C.c_int32)()
        n = C.c_int64()
        _check(lib().cpecan_batch_result(self._h, problem, which, C.byref(ptr), C.byref(n)), "cpecan_batch_result")
        if n.value == 0:
            return np.zeros((0, 3), dtype=np.int32)
        return np.ctypeslib.as_array(ptr, shape=(n.value * 3,)).copy().reshape(n.value, 3)

    def set_post(self, flags, gapGamma=0.0, matchGamma=None):
        """Consumers applied on the device by download(): POST_REWEIGHT (reweightAlignedPairs2), POST_MEA
        (getMaximalExpectedAccuracyPairwiseAlignment, result list 3), POST_MEA | POST_LEFT_SHIFT (getShiftedMEAAlignment),
        POST_ORDERED (filterPairwiseAlignmentToMakePairsOrdered with matchGamma, result list 3)."""
        _check(lib().cpecan_batch_set_post(self._h, int(flags), float(gapGamma)), "cpecan_batch_set_post")
        if matchGamma is not None:
            _check(lib().cpecan_batch_set_match_gamma(self._h, C.c_float(matchGamma)), "cpecan_batch_set_match_gamma")

    def identity_scores(self, problem):
        """(scoreByIdentity, scoreByIdentityIgnoringGaps) of a problem's final list."""
        a, b = C.c_double(), C.c_double()
        _check(lib().cpecan_batch_identity_scores(self._h, problem, C.byref(a), C.byref(b)),
               "cpecan_batch_identity_scores")
        return a.value, b.value

    def scores(self, problem):
        """(scoreByPosteriorProbability, scoreByPosteriorProbabilityIgnoringGaps, MEA alignment score) of a problem."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _check(lib().cpecan_batch_scores(self._h, problem, C.byref(a), C.byref(b), C.byref(c)), "cpecan_batch_scores")
        return a.value, b.value, c.value

    def expectations(self, hmm):
        """Adds the batch's expectation counts into hmm (EMIT_EXPECT), like getExpectationsUsingAnchors on each problem."""
        _check(lib().cpecan_batch_expectations(self._h, C.byref(hmm)), "cpecan_batch_expectations")
        return hmm

    def forward_prob(self, problem):
        v = C.c_double()
        _check(lib().cpecan_batch_forward_prob(self._h, problem, C.byref(v)), "cpecan_batch_forward_prob")
        return v.value

    def stats(self):
        s = Stats()
        _check(lib().cpecan_batch_stats(self._h, C.byref(s)), "cpecan_batch_stats")
        return s

    def debug_fetch(self, problem, cells, diagonals):
        fb = np.zeros(cells, dtype=np.float64)
        tot = np.zeros(diagonals, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _check(lib().cpecan_batch_debug_fetch(self._h, problem, fb.ctypes.data_as(dp), cells, tot.ctypes.data_as(dp),
                                              diagonals), "cpecan_batch_debug_fetch")
        return fb, tot


def getAlignedPairsUsingAnchors(sM, sX, sY, anchorPairs, p, alignmentHasRaggedLeftEnd=False,
                                alignmentHasRaggedRightEnd=False):
    """impl/pairwiseAligner.c:1431: list of (score, x, y), in the reference's list order."""
    a, ptr, n = _anchor_array(anchorPairs)
    out = C.POINTER(C.c_int32)()
    cnt = C.c_int64()
    _check(lib().cpecan_get_aligned_pairs_using_anchors(C.byref(sM), _bytes(sX), _bytes(sY), ptr, n, C.byref(p),
                                                        int(alignmentHasRaggedLeftEnd), int(alignmentHasRaggedRightEnd),
                                                        C.byref(out), C.byref(cnt)), "getAlignedPairsUsingAnchors")
    res = np.ctypeslib.as_array(out, shape=(max(cnt.value, 1) * 3,))[:cnt.value * 3].copy().reshape(cnt.value, 3)
    lib().cpecan_free(C.cast(out, C.c_void_p))
    return res


def _take_list(ptr, n):
    res = np.ctypeslib.as_array(ptr, shape=(max(n, 1) * 3,))[:n * 3].copy().reshape(n, 3)
    lib().cpecan_free(C.cast(ptr, C.c_void_p))
    return res


def getAlignedPairsWithIndelsUsingAnchors(sM, sX, sY, anchorPairs, p, alignmentHasRaggedLeftEnd=False,
                                          alignmentHasRaggedRightEnd=False):
    """impl/pairwiseAligner.c:1451: (alignedPairs, gapXPairs, gapYPairs), each int32[n,3] of (score, x, y)."""
    a, ptr, n = _anchor_array(anchorPairs)
    outs = [C.POINTER(C.c_int32)() for _ in range(3)]
    cnts = [C.c_int64() for _ in range(3)]
    _check(lib().cpecan_get_aligned_pairs_with_indels_using_anchors(
        C.byref(sM), _bytes(sX), _bytes(sY), ptr, n, C.byref(p), int(alignmentHasRaggedLeftEnd),
        int(alignmentHasRaggedRightEnd), C.byref(outs[0]), C.byref(cnts[0]), C.byref(outs[1]), C.byref(cnts[1]),
        C.byref(outs[2]), C.byref(cnts[2])), "getAlignedPairsWithIndelsUsingAnchors")
    return tuple(_take_list(o, c.value) for o, c in zip(outs, cnts))


def computeForwardProbability(seqX, seqY, anchorPairs, p, sM, alignmentHasRaggedLeftEnd=False,
                              alignmentHasRaggedRightEnd=False):
    """impl/pairwiseAligner.c:936 (argument order as in the reference)."""
    a, ptr, n = _anchor_array(anchorPairs)
    v = C.c_double()
    _check(lib().cpecan_compute_forward_probability(C.byref(sM), _bytes(seqX), _bytes(seqY), ptr, n, C.byref(p),
                                                    int(alignmentHasRaggedLeftEnd), int(alignmentHasRaggedRightEnd),
                                                    C.byref(v)), "computeForwardProbability")
    return v.value


def getExpectationsUsingAnchors(sM, hmmExpectations, sX, sY, anchorPairs, p, alignmentHasRaggedLeftEnd=False,
                                alignmentHasRaggedRightEnd=False):
    """impl/pairwiseAligner.c:1500: accumulates into hmmExpectations (+=), likelihood included."""
    with Batch(sM, p, emit=EMIT_EXPECT) as b:
        b.add(sX, sY, anchorPairs, alignmentHasRaggedLeftEnd, alignmentHasRaggedRightEnd)
        b.upload()
        b.run()
        b.download()
        b.expectations(hmmExpectations)
    return hmmExpectations


# ---- consumers of the posterior lists; (score, x, y) int32 triples in, GPU in between ----
def _i32_triples(t):
    a = np.ascontiguousarray(np.asarray(t, dtype=np.int32).reshape(-1, 3))
    return a, a.ctypes.data_as(C.POINTER(C.c_int32)), len(a)


def reweightAlignedPairs2(alignedPairs, seqLengthX, seqLengthY, gapGamma):  # impl/pairwiseAligner.c:1550
    a, ptr, n = _i32_triples(alignedPairs)
    a = a.copy()
    _check(lib().cpecan_reweight_aligned_pairs(a.ctypes.data_as(C.POINTER(C.c_int32)), n, seqLengthX, seqLengthY,
                                               float(gapGamma)), "cpecan_reweight_aligned_pairs")
    return a


def _posterior_scores(alignedPairs, lX, lY):
    a, ptr, n = _i32_triples(alignedPairs)
    s0, s1 = C.c_double(), C.c_double()
    _check(lib().cpecan_posterior_scores(ptr, n, lX, lY, C.byref(s0), C.byref(s1)), "cpecan_posterior_scores")
    return s0.value, s1.value


def scoreByPosteriorProbability(lX, lY, alignedPairs):  # impl/pairwiseAligner.c:1587
    return _posterior_scores(alignedPairs, lX, lY)[0]


def scoreByPosteriorProbabilityIgnoringGaps(alignedPairs):  # impl/pairwiseAligner.c:1591
    return _posterior_scores(alignedPairs, 0, 0)[1]


def _identity_scores(alignedPairs, seqX, seqY):
    a, pa, n = _i32_triples(alignedPairs)
    s0, s1 = C.c_double(), C.c_double()
    _check(lib().cpecan_identity_scores(pa, n, _bytes(seqX), _bytes(seqY), C.byref(s0), C.byref(s1)),
           "cpecan_identity_scores")
    return s0.value, s1.value


def scoreByIdentity(subSeqX, subSeqY, lX, lY, alignedPairs):  # impl/pairwiseAligner.c:1572
    return _identity_scores(alignedPairs, subSeqX, subSeqY)[0]


def scoreByIdentityIgnoringGaps(subSeqX, subSeqY, alignedPairs):  # impl/pairwiseAligner.c:1577
    return _identity_scores(alignedPairs, subSeqX, subSeqY)[1]


def filterPairwiseAlignmentToMakePairsOrdered(alignedPairs, seqX, seqY, matchGamma):  # impl/multipleAligner.c:945
    """The heaviest chain of the pairs with weight >= matchGamma, in the reference's output order (reverse input order);
    the reference's random weight jitter (multipleAligner.c:145) is left out."""
    a, pa, n = _i32_triples(alignedPairs)
    out = C.POINTER(C.c_int32)()
    cnt = C.c_int64()
    _check(lib().cpecan_filter_pairs_ordered(pa, n, len(seqX), len(seqY), C.c_float(matchGamma), C.byref(out),
                                             C.byref(cnt)), "cpecan_filter_pairs_ordered")
    return _take_list(out, cnt.value)


def getMaximalExpectedAccuracyPairwiseAlignment(alignedPairs, gapXPairs, gapYPairs, seqXLength, seqYLength, p=None,
                                                gapGamma=None):  # impl/pairwiseAligner.c:1628
    """Returns (alignment int32[n,3], alignmentScore).  gapGamma defaults to the reference's 0.5 (:1345)."""
    a, pa, n = _i32_triples(alignedPairs)
    gx, pgx, ngx = _i32_triples(gapXPairs)
    gy, pgy, ngy = _i32_triples(gapYPairs)
    out = C.POINTER(C.c_int32)()
    cnt = C.c_int64()
    score = C.c_double()
    _check(lib().cpecan_mea_alignment(pa, n, pgx, ngx, pgy, ngy, seqXLength, seqYLength,
                                      C.c_float(0.5 if gapGamma is None else gapGamma), C.byref(out), C.byref(cnt),
                                      C.byref(score)), "cpecan_mea_alignment")
    return _take_list(out, cnt.value), score.value


def leftShiftAlignment(alignedPairs, seqX, seqY):  # impl/pairwiseAligner.c:1726
    a, pa, n = _i32_triples(alignedPairs)
    out = C.POINTER(C.c_int32)()
    cnt = C.c_int64()
    _check(lib().cpecan_left_shift_alignment(pa, n, _bytes(seqX), _bytes(seqY), C.byref(out), C.byref(cnt)),
           "cpecan_left_shift_alignment")
    return _take_list(out, cnt.value)


def getShiftedMEAAlignment(seqX, seqY, anchorAlignment, p, sM, alignmentHasRaggedLeftEnd=False,
                           alignmentHasRaggedRightEnd=False, gapGamma=0.5):  # impl/pairwiseAligner.c:1767
    """Returns (left-shifted MEA alignment int32[n,3], alignmentScore)."""
    arr, ptr, n = _anchor_array(anchorAlignment)
    out = C.POINTER(C.c_int32)()
    cnt = C.c_int64()
    score = C.c_double()
    _check(lib().cpecan_get_shifted_mea_alignment(C.byref(sM), _bytes(seqX), _bytes(seqY), ptr, n, C.byref(p),
                                                  C.c_float(gapGamma), int(alignmentHasRaggedLeftEnd),
                                                  int(alignmentHasRaggedRightEnd), C.byref(out), C.byref(cnt),
                                                  C.byref(score)), "cpecan_get_shifted_mea_alignment")
    return _take_list(out, cnt.value), score.value


def convertPairwiseForwardStrandAlignmentToAnchorPairs(ops, start1, start2, trim, expansion, seqX=None, seqY=None):
    """impl/pairwiseAligner.c:979-1003; ops = [(OP_MATCH | OP_INDEL_X | OP_INDEL_Y, length), ...].  With seqX and seqY the
    exact-match filter of cPecanRealign.c:277-281,529 is applied as well.  Returns int64[n,3] (x, y, expansion)."""
    o = np.ascontiguousarray(np.asarray(ops, dtype=np.int64).reshape(-1, 2))
    cap = int(o[o[:, 0] == OP_MATCH, 1].sum()) if len(o) else 0
    out = np.zeros((max(cap, 1), 3), dtype=np.int64)
    sx = _bytes(seqX) if seqX is not None else None
    sy = _bytes(seqY) if seqY is not None else None
    n = lib().cpecan_anchors_from_alignment(o.ctypes.data_as(C.POINTER(C.c_int64)), len(o), start1, start2, trim, expansion,
                                            sx, len(sx) if sx else 0, sy, len(sy) if sy else 0,
                                            out.ctypes.data_as(C.POINTER(C.c_int64)))
    _check(n, "cpecan_anchors_from_alignment")
    return out[:n].copy()


def anchor_runs_from_alignment(ops, start1, start2, trim, expansion, seqX=None, seqY=None):
    """cpecan_anchor_runs_from_alignment: the anchors of convertPairwiseForwardStrandAlignmentToAnchorPairs (+ exact-match
    filter) as runs, int64[nRuns, 4] (x, y, length, expansion), without the per-column list in between."""
    o = np.ascontiguousarray(np.asarray(ops, dtype=np.int64).reshape(-1, 2))
    sx = _bytes(seqX) if seqX is not None else None
    sy = _bytes(seqY) if seqY is not None else None
    args = (o.ctypes.data_as(C.POINTER(C.c_int64)), len(o), start1, start2, trim, expansion, sx, len(sx) if sx else 0, sy,
            len(sy) if sy else 0)
    n = _check(lib().cpecan_anchor_runs_from_alignment(*args, None, 0), "cpecan_anchor_runs_from_alignment")
    out = np.zeros((max(n, 1), 4), dtype=np.int64)
    _check(lib().cpecan_anchor_runs_from_alignment(*args, out.ctypes.data_as(C.POINTER(C.c_int64)), n),
           "cpecan_anchor_runs_from_alignment")
    return out[:n].copy()


def filterToRemoveOverlap(sortedOverlappingPairs):  # impl/pairwiseAligner.c:1095
    """Pairs (x, y, expansion) sorted by x, then y -> the ones that overlap no other pair, int64[n,3]."""
    a, ptr, n = _anchor_array(sortedOverlappingPairs)
    out = np.zeros((max(n, 1), 3), dtype=np.int64)
    cnt = _check(lib().cpecan_filter_to_remove_overlap(ptr, n, out.ctypes.data_as(C.POINTER(C.c_int64))),
                 "filterToRemoveOverlap")
    return out[:cnt].copy()
