"""ctypes mirror of include/cpecan_realign.h: the cPecanRealign-style batch front end (cPecanRealign.c:354-624).

Cigars are plain Python objects here (`Cigar`); text parsing, formatting and the realign loop itself run in the C library.
"""
import ctypes as C

from . import api

# Every symbol include/cpecan_realign.h declares (checked by tests/test_abi.py).
EXPORTS = [
    "cpecan_cigar_parse", "cpecan_cigar_format", "cpecan_cigar_clear", "cpecan_cigars_free",
    "cpecan_cigar_from_aligned_pairs", "cpecan_cigar_split",
    "cpecan_realign_options_default", "cpecan_realigner_create", "cpecan_realigner_destroy",
    "cpecan_realigner_add_sequence", "cpecan_realigner_read_fasta", "cpecan_realigner_set_posterior_files",
    "cpecan_realigner_realign", "cpecan_realigner_expectations",
    "cpecan_realigner_set_devices", "cpecan_realign_shard_bounds",
]


class _Cigar(C.Structure):
    _fields_ = [("contig1", C.c_void_p), ("contig2", C.c_void_p),
                ("start1", C.c_int64), ("end1", C.c_int64), ("start2", C.c_int64), ("end2", C.c_int64),
                ("strand1", C.c_int32), ("strand2", C.c_int32), ("score", C.c_double),
                ("nOps", C.c_int64), ("ops", C.POINTER(C.c_int64))]


class RealignOptions(C.Structure):
    """cpecan_realign_options: cPecanRealign's command-line options (cPecanRealign.c:354-370)."""
    _fields_ = [("params", api.PairwiseAlignmentParameters), ("constraintDiagonalTrim", C.c_int64),
                ("gapGamma", C.c_float), ("matchGamma", C.c_float),
                ("rescoreOriginalAlignment", C.c_int32), ("rescoreByIdentity", C.c_int32),
                ("rescoreByPosteriorProb", C.c_int32), ("rescoreByIdentityIgnoringGaps", C.c_int32),
                ("rescoreByPosteriorProbIgnoringGaps", C.c_int32), ("splitIndelsLongerThanThis", C.c_int64)]


_bound = False


def _lib():
    global _bound
    L = api.lib()
    if not _bound:
        vp = C.c_void_p
        L.cpecan_cigar_parse.argtypes = [C.c_char_p, C.POINTER(_Cigar)]
        L.cpecan_cigar_format.argtypes = [C.POINTER(_Cigar), C.c_char_p, C.c_int64]
        L.cpecan_cigar_format.restype = C.c_int64
        L.cpecan_cigar_clear.argtypes = [C.POINTER(_Cigar)]
        L.cpecan_cigar_clear.restype = None
        L.cpecan_cigars_free.argtypes = [C.POINTER(_Cigar), C.c_int64]
        L.cpecan_cigars_free.restype = None
        L.cpecan_cigar_from_aligned_pairs.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_int64, C.c_int64,
                                                      C.POINTER(C.c_int64), C.c_int64, C.POINTER(_Cigar)]
        L.cpecan_cigar_split.argtypes = [C.POINTER(_Cigar), C.c_int64, C.POINTER(C.POINTER(_Cigar)), C.POINTER(C.c_int64)]
        L.cpecan_realign_options_default.argtypes = [C.POINTER(RealignOptions)]
        L.cpecan_realign_options_default.restype = None
        L.cpecan_realigner_create.argtypes = [C.POINTER(vp), C.POINTER(api.StateMachine), C.POINTER(RealignOptions), C.c_int]
        L.cpecan_realigner_destroy.argtypes = [vp]
        L.cpecan_realigner_destroy.restype = None
        L.cpecan_realigner_add_sequence.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int64]
        L.cpecan_realigner_read_fasta.argtypes = [vp, C.c_char_p]
        L.cpecan_realigner_read_fasta.restype = C.c_int64
        L.cpecan_realigner_set_posterior_files.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.cpecan_realigner_realign.argtypes = [vp, C.POINTER(_Cigar), C.c_int64, C.POINTER(C.POINTER(_Cigar)),
                                               C.POINTER(C.c_int64)]
        L.cpecan_realigner_expectations.argtypes = [vp, C.POINTER(_Cigar), C.c_int64, C.POINTER(api.Hmm)]
        L.cpecan_realigner_set_devices.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
        L.cpecan_realign_shard_bounds.argtypes = [C.POINTER(_Cigar), C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
        _bound = True
    return L


_OP_CHARS = "MDI"


class Cigar:
    """struct PairwiseAlignment of sonLib: contig1 is sequence X of the aligner, contig2 is Y; ops: [(OP_*, length)]."""

    def __init__(self, contig1, start1, end1, strand1, contig2, start2, end2, strand2, score, ops):
        self.contig1, self.start1, self.end1, self.strand1 = contig1, int(start1), int(end1), bool(strand1)
        self.contig2, self.start2, self.end2, self.strand2 = contig2, int(start2), int(end2), bool(strand2)
        self.score = float(score)
        self.ops = [(int(t), int(n)) for t, n in ops]

    @staticmethod
    def _from_c(c):
        return Cigar(C.string_at(c.contig1).decode(), c.start1, c.end1, c.strand1, C.string_at(c.contig2).decode(), c.start2,
                     c.end2, c.strand2, c.score, [(c.ops[2 * i], c.ops[2 * i + 1]) for i in range(c.nOps)])

    def _to_c(self, keep):
        c = _Cigar()
        n1, n2 = C.create_string_buffer(self.contig1.encode()), C.create_string_buffer(self.contig2.encode())
        ops = (C.c_int64 * max(1, 2 * len(self.ops)))(*[v for op in self.ops for v in op])
        keep.extend([n1, n2, ops])
        c.contig1, c.contig2 = C.cast(n1, C.c_void_p), C.cast(n2, C.c_void_p)
        c.start1, c.end1, c.strand1 = self.start1, self.end1, int(self.strand1)
        c.start2, c.end2, c.strand2 = self.start2, self.end2, int(self.strand2)
        c.score, c.nOps, c.ops = self.score, len(self.ops), C.cast(ops, C.POINTER(C.c_int64))
        return c

    @staticmethod
    def parse(line):  # cigarRead (sonLib), one line
        c = _Cigar()
        api._check(_lib().cpecan_cigar_parse(line.encode(), C.byref(c)), "cpecan_cigar_parse")
        try:
            return Cigar._from_c(c)
        finally:
            _lib().cpecan_cigar_clear(C.byref(c))

    def format(self):  # cigarWrite(fileHandle, pA, 0) (sonLib) without the newline
        keep = []
        c = self._to_c(keep)
        n = api._check(_lib().cpecan_cigar_format(C.byref(c), None, 0), "cpecan_cigar_format")
        buf = C.create_string_buffer(n + 1)
        _lib().cpecan_cigar_format(C.byref(c), buf, n + 1)
        return buf.value.decode()

    @staticmethod
    def from_aligned_pairs(contig1, contig2, score, length1, length2, xy):  # cPecanRealign.c:49
        """convertAlignedPairsToPairwiseAlignment: xy = increasing (x, y) pairs."""
        flat = [int(v) for p in xy for v in p]
        arr = (C.c_int64 * max(1, len(flat)))(*flat)
        c = _Cigar()
        api._check(_lib().cpecan_cigar_from_aligned_pairs(contig1.encode(), contig2.encode(), score, length1, length2, arr,
                                                          len(flat) // 2, C.byref(c)), "cpecan_cigar_from_aligned_pairs")
        try:
            return Cigar._from_c(c)
        finally:
            _lib().cpecan_cigar_clear(C.byref(c))

    def split(self, max_indel_length):  # splitPairwiseAlignment, cPecanRealign.c:117
        keep = []
        c = self._to_c(keep)
        out, n = C.POINTER(_Cigar)(), C.c_int64()
        api._check(_lib().cpecan_cigar_split(C.byref(c), max_indel_length, C.byref(out), C.byref(n)), "cpecan_cigar_split")
        try:
            return [Cigar._from_c(out[i]) for i in range(n.value)]
        finally:
            _lib().cpecan_cigars_free(out, n.value)

    def same_coordinates(self, other):  # sonLib.bioio.PairwiseAlignment.sameCoordinates
        key = lambda c: (c.contig1, c.start1, c.end1, c.strand1, c.contig2, c.start2, c.end2, c.strand2)
        return key(self) == key(other)

    def __eq__(self, other):
        return self.same_coordinates(other) and self.ops == other.ops and self.score == other.score

    def __repr__(self):
        return "Cigar(%s)" % self.format()


def realign_options(**overrides):
    """cPecanRealign's defaults; keyword overrides name fields of the options or of its `params`."""
    o = RealignOptions()
    _lib().cpecan_realign_options_default(C.byref(o))
    for k, v in overrides.items():
        if hasattr(o, k) and k != "params":
            setattr(o, k, v)
        elif hasattr(o.params, k):
            setattr(o.params, k, v)
        else:
            raise AttributeError(k)
    return o


class Realigner:
    """cpecan_realigner: sequences by name plus the options; realign() and expectations() each run ONE GPU batch."""

    def __init__(self, sM=None, options=None, device=0):
        self._h = C.c_void_p()
        self._sm = sM if sM is not None else api.stateMachine5_construct()  # cPecanRealign.c:489
        self._opt = options if options is not None else realign_options()
        api._check(_lib().cpecan_realigner_create(C.byref(self._h), C.byref(self._sm), C.byref(self._opt), device),
                   "cpecan_realigner_create")

    def close(self):
        if self._h:
            _lib().cpecan_realigner_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_sequence(self, header, seq):  # addToSequencesHash, cPecanRealign.c:245
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        api._check(_lib().cpecan_realigner_add_sequence(self._h, header.encode(), s, len(s)), "cpecan_realigner_add_sequence")

    def read_fasta(self, path):
        return api._check(_lib().cpecan_realigner_read_fasta(self._h, path.encode()), "cpecan_realigner_read_fasta")

    def set_posterior_files(self, final_pairs=None, all_pairs=None):
        api._check(_lib().cpecan_realigner_set_posterior_files(
            self._h, final_pairs.encode() if final_pairs else None, all_pairs.encode() if all_pairs else None),
            "cpecan_realigner_set_posterior_files")

    def set_devices(self, devices):
        """The cigars of every later call are cut into one contiguous shard per listed device (cpecan_realigner_set_devices)."""
        arr = (C.c_int * max(1, len(devices)))(*devices)
        api._check(_lib().cpecan_realigner_set_devices(self._h, arr, len(devices)), "cpecan_realigner_set_devices")

    def _pack(self, cigars):
        keep = []
        arr = (_Cigar * max(1, len(cigars)))(*[c._to_c(keep) for c in cigars])
        return arr, keep

    def realign(self, cigars):
        arr, keep = self._pack(cigars)
        out, n = C.POINTER(_Cigar)(), C.c_int64()
        api._check(_lib().cpecan_realigner_realign(self._h, arr, len(cigars), C.byref(out), C.byref(n)),
                   "cpecan_realigner_realign")
        try:
            return [Cigar._from_c(out[i]) for i in range(n.value)]
        finally:
            _lib().cpecan_cigars_free(out, n.value)

    def expectations(self, cigars, hmm):
        arr, keep = self._pack(cigars)
        api._check(_lib().cpecan_realigner_expectations(self._h, arr, len(cigars), C.byref(hmm)),
                   "cpecan_realigner_expectations")
        return hmm


def shard_bounds(cigars, n_shards, expansion=4):
    """cpecan_realign_shard_bounds: the cut points of n_shards contiguous shards of about equal band cells."""
    keep = []
    arr = (_Cigar * max(1, len(cigars)))(*[c._to_c(keep) for c in cigars])
    out = (C.c_int64 * (n_shards + 1))()
    api._check(_lib().cpecan_realign_shard_bounds(arr, len(cigars), expansion, n_shards, out), "cpecan_realign_shard_bounds")
    return list(out)
