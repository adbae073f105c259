/*
 * cpecan_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for cPecan's banded pair-HMM
 * forward / backward / posterior path, plus the device-memory plumbing behind them.
 *
 * What the reference does per alignment (impl/pairwiseAligner.c:756-877, getPosteriorProbsWithBanding):
 * a forward sweep over anti-diagonals with periodic partial tracebacks; each traceback runs the
 * backward recurrence from an end-state prior, refreshes the total probability every 10th diagonal
 * and emits thresholded posteriors.  The arithmetic is log-space fp64 with a piecewise-cubic logAdd
 * (:287-307) whose fold ORDER is part of the result, so the kernel reproduces it term for term.
 *
 * How it is mapped to CDNA4 (one 64-lane wavefront per DP region, persistent, work-queue fed):
 *   - lanes <-> cells of the current anti-diagonal (dense index k = (xmy - xmyL)/2), ceil(W/64) passes;
 *   - the two previous diagonals live in LDS, structure-of-arrays per state with -inf guard cells,
 *     so band edges need no branches (neighbour indices are clamped onto a guard);
 *   - the backward recurrence is evaluated as a GATHER whose term order equals the reference's
 *     scatter order (derivation: DESIGN.md "backward as a gather");
 *   - forward values stream to a per-wave ring in HBM (coalesced, SoA) and are read back once by the
 *     traceback; the ring holds one traceback segment, not the whole matrix;
 *   - the sequential logAdd fold that defines the per-diagonal total probability is transposed:
 *     all refresh points of a segment are folded at once, one lane per refresh point;
 *   - posteriors are thresholded and compacted with wave ballots straight into the output list order.
 * No MFMA: this is an fp64 stencil bounded by HBM traffic and fp64 VALU rate, not a contraction.
 *
 * Built with -ffp-contract=off: the reference's polynomial is separately rounded mul/add.
 */
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpecan_internal.h"
#include "cpecan_band.inl"

#define NEG_INF (-__builtin_huge_val())

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" void cpk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *cpk_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cpk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CPECAN_EHIP;                                                                     \
        }                                                                                           \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------
struct Candidate {  // a cell that may pass the posterior threshold once the total probability is known
    double fb;      // F.match + B.match
    int32_t x, y;   // matrix coordinates
};

// Transition log-probabilities and the threshold travel BY VALUE in the kernel arguments: kernarg loads are scalar
// (s_load), so the hot loops never wait on vector memory for a model constant.
struct KConsts {
    double matchContinue;
    double matchFromShortX, matchFromShortY, matchFromLongX, matchFromLongY;
    double shortOpenX, shortOpenY, shortExtendX, shortExtendY, shortSwitchToX, shortSwitchToY;
    double longOpenX, longOpenY, longExtendX, longExtendY;
    double threshold;
};

struct KArgs {
    KConsts kc;
    const CpkRegion *regions;
    const CpkDiag *diags;
    const CpkSegment *segs;
    const uint8_t *symbols;
    const CpkModel *model;
    CpkGeometry geo;
    double *ring;      // [slots][ringCells*S]   forward values of the live traceback segment
    Candidate *cand;   // [slots][candCells]     posterior candidates of the segment being traced back
    double *cbuf;      // [slots][refreshCells]  per-cell F.B dot products on refresh diagonals, [k][j]
    double *mbuf;      // [slots][refreshCells]  per-cell "match straddling the diagonal" terms, [k][j]
    double *totals;    // [slots][maxRefresh]
    double *groll;     // [slots][rollDoubles]   rolling buffers when they do not fit in LDS
    double *bring;     // [slots][fbCells*S]     expectation emitter: backward values of the emitted cells of the segment
    int32_t *outCounts;  // [nLists][nRegions]
    int32_t *segStarts;  // [nLists][nSegsTotal]
    int32_t *triples;    // [nLists][outTriplesPerList*3]
    int64_t outTriplesPerList;
    int64_t nSegsTotal;
    unsigned int *queue;
    int32_t regionBase, regionCount;  // this launch works on regions [regionBase, regionBase + regionCount)
    double *forwardOut;  // [nRegions] total forward log-probability (forward mode)
    double *expectOut;   // [slots][128] per-wave expectation partial sums (expectation mode)
    double *dbgFb;
    double *dbgTotals;
};

// logAdd, impl/pairwiseAligner.c:287-307.  hi/lo form: with d = hi - lo the reference returns hi when
// lo == -inf or d >= 7.5, else lo + P(d).  d is +inf when only lo is -inf and NaN when both are, and
// both fail (d < 7.5), so one comparison covers the reference's two tests.  The cubic's coefficients are
// float literals in the reference, i.e. float32 values widened to double; Horner with separate mul/add.
// The four cubics live in a 128-byte LDS table [segment][c3,c2,c1,c0] read with two ds_read_b128: selecting
// four 64-bit coefficients with v_cndmask cost 24 VALU instructions per logAdd (47 % of the forward loop).
struct __attribute__((aligned(16))) Cubic {
    double c3, c2, c1, c0;
};

// The segment of d = hi - lo is found without fp64 compares.  For d >= 0 the IEEE bit pattern is monotone; the three
// thresholds (1.0, 2.5, 4.5) have a zero low dword and high dwords that are multiples of 2^17, so the segment is a
// function of the bucket  b = ((bits(d) + 2^49 - 1) >> 49) - (bits(1.0) >> 49), saturated at 0:
// b == 0 -> d <= 1;  1..10 -> (1, 2.5];  11..17 -> (2.5, 4.5];  18.. -> above, i.e. segment = number of set bits of
// {0, 10, 17} below position b.  Exact for every double, thresholds included.  NaN / +inf / d >= 8 may pick any
// segment: the result is `hi` then.  The table keeps 4 rows of 32 bytes (conflict-free for ds_read_b128); a 26-row
// table indexed by bucket measured 100x the LDS bank conflicts.
__device__ __forceinline__ void fill_cubics(double *t) {
    const float c[16] = {-0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                         -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                         -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                         -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f};
    const int l = threadIdx.x;
    if (l < 16) {
        float v = c[0];
#pragma unroll
        for (int i = 1; i < 16; i++) v = l == i ? c[i] : v;
        t[l] = (double)v;
    }
}

__device__ __forceinline__ int cubic_row(double d) {
    // bits(d) > bits(T)  <=>  bits(d) + 2^49 - 1 >= bits(T) + 2^49 for the three thresholds (multiples of 2^49), so the
    // bucket is the high part of one 64-bit add; the saturating subtract sends every d <= 1 (d == 0 included) to
    // bucket 0, and v_bfe_u32 reads only 5 bits of its width, which is harmless: buckets above 24 mean d >= 8.
    const unsigned long long bits = (unsigned long long)__double_as_longlong(d) + ((1ull << 49) - 1ull);
    const unsigned h = (unsigned)(bits >> 32) >> 17;
    const unsigned b = __builtin_elementwise_sub_sat(h, 0x3FF00000u >> 17);
    return __builtin_popcount(__builtin_amdgcn_ubfe((1u << 0) | (1u << 10) | (1u << 17), 0u, b));
}

__device__ __forceinline__ double logadd(const Cubic *tab, double x, double y) {
    const double hi = __builtin_fmax(x, y);
    const double lo = __builtin_fmin(x, y);
    const double d = hi - lo;
    const Cubic q = tab[cubic_row(d)];
    double r = q.c3 * d;
    r = r + q.c2;
    r = r * d;
    r = r + q.c1;
    r = r * d;
    r = r + q.c0;
    r = r + lo;
    return (d < 7.5) ? r : hi;
}

// exp(x) to a relative error of ~1e-7 for x <= ~1 (probabilities): 2^(x log2 e) with the integer part split off in
// double, the fraction through v_exp_f32, and the scaling by v_ldexp_f64 -- 8 instructions instead of the ~35 of the
// double-precision exp.  Only for the expectation sums, whose gate is 1e-5 relative (linear-space
// sums, order-insensitive at 1e-5); the posterior emitters keep the exact exp.
__device__ __forceinline__ double exp_1e7(double x) {
    // branch-free: -inf, NaN (an unreachable transition) and anything below 2^-1100 end as ldexp(.., -1100) == 0
    const double y = __builtin_fmax(x * 1.4426950408889634 /* log2(e) */, -1100.0);
    const double yi = __builtin_rint(y);
    const float yf = (float)(y - yi);         // in [-0.5, 0.5]
    return __builtin_ldexp((double)__builtin_amdgcn_exp2f(yf), (int)yi);
}

// N independent logAdds advanced in lock-step stages (compare/select -> table fetch -> Horner) so that the N LDS
// table fetches are in flight together instead of one fetch + wait per logAdd.  acc[i] = logAdd(acc[i], t[i]).
template <int N>
__device__ __forceinline__ void logadd_n(const Cubic *tab, double (&acc)[N], const double (&t)[N]) {
    double hi[N], lo[N], d[N];
    Cubic q[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        hi[i] = __builtin_fmax(acc[i], t[i]);
        lo[i] = __builtin_fmin(acc[i], t[i]);
        d[i] = hi[i] - lo[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) q[i] = tab[cubic_row(d[i])];
    double r[N];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = q[i].c3 * d[i];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + q[i].c2;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] * d[i];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + q[i].c1;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] * d[i];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + q[i].c0;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + lo[i];
#pragma unroll
    for (int i = 0; i < N; i++) acc[i] = (d[i] < 7.5) ? r[i] : hi[i];
}

// Row position of neighbour cell i of a diagonal that has `w` cells (w = 0: the diagonal does not exist):
// cells sit at positions 1..w, position 0 of every row is a permanent -inf guard.
__device__ __forceinline__ int guard_pos(int i, int w) { return ((unsigned)i < (unsigned)w) ? i + 1 : 0; }

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = v > o ? v : o;
    }
    return v;
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

template <bool GLOBAL_ROLL>
__device__ __forceinline__ void roll_fence() {
    if (GLOBAL_ROLL) {
        __syncthreads();  // workgroup-scope release/acquire on global memory (single-wave workgroup)
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// data this wave wrote earlier in the same launch: always a vector load, never the scalar cache
__device__ __forceinline__ double ld_self(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// 64 consecutive entries of the region's diagonal table, one per lane, handed out with v_readlane: the sweeps
// touch the table once per diagonal and must not wait on a memory load for it.
struct DiagCache {
    const CpkDiag *table;
    int last;  // highest valid diagonal (N)
    int base;  // diagonal held by lane 0
    int lane;
    int eXmyL, eWidth, eRing, eCell;

    __device__ __forceinline__ void load(int b) {
        base = b;
        int i = b + lane;
        i = i < 0 ? 0 : (i > last ? last : i);
        const int4 e = *reinterpret_cast<const int4 *>(table + i);
        eXmyL = e.x;
        eWidth = e.y;
        eRing = e.z;
        eCell = e.w;
        // consume the loaded registers here so the s_waitcnt for this load sits inside the (rare) refill branch;
        // otherwise hipcc puts a vmcnt(0) at the branch merge and every diagonal waits for its ring stores
        asm volatile("" ::"v"(eXmyL), "v"(eWidth), "v"(eRing), "v"(eCell));
    }
    // entry held by lane l of the current chunk (l wave-uniform).  The hot loops walk a chunk with load() outside the
    // loop over its 64 diagonals: a lazy refill inside the loop costs a range check, a branch and a round of VGPR
    // copies at its merge point on every diagonal.
    __device__ __forceinline__ CpkDiag at(int l) const {
        CpkDiag g;
        g.xmyL = __builtin_amdgcn_readlane(eXmyL, l);
        g.width = __builtin_amdgcn_readlane(eWidth, l);
        g.ringOff = __builtin_amdgcn_readlane(eRing, l);
        g.cellOff = __builtin_amdgcn_readlane(eCell, l);
        return g;
    }
    // descending = the caller walks towards lower diagonals (refill so that d is the LAST lane of the chunk)
    __device__ __forceinline__ CpkDiag get(int d, bool descending) {
        if (d < base || d >= base + CPK_WAVE) load(descending ? d - (CPK_WAVE - 1) : d);
        const int l = __builtin_amdgcn_readfirstlane(d - base);
        CpkDiag g;
        g.xmyL = __builtin_amdgcn_readlane(eXmyL, l);
        g.width = __builtin_amdgcn_readlane(eWidth, l);
        g.ringOff = __builtin_amdgcn_readlane(eRing, l);
        g.cellOff = __builtin_amdgcn_readlane(eCell, l);
        return g;
    }
};

constexpr int kLdsCubics = 16;  // 4 rows x 4 coefficients
// doubles of LDS in front of the rolling rows: cubics + emissions (+ expectation sums for that emitter only)
constexpr int kExpectCopies = 4;  // emission-expectation sums are kept in 4 LDS copies (lane & 3): fewer atomic collisions
constexpr int kLdsWeights = 168;  // (emission + transition) sums, see Sweep::wt: 25*5 + 5*4 + 5*4 = 165, padded
__host__ __device__ constexpr int lds_header_doubles(int emit) {
    return kLdsCubics + 40 + kLdsWeights + (emit == CPECAN_EMIT_EXPECT ? kExpectCopies * 80 : 0);
}
// doubles of LDS behind the rolling rows for the candidate staging rings (16-byte Candidates, 128 per list; none for
// the forward-only and expectation emitters)
__host__ __device__ constexpr int lds_stage_doubles(int emit) {
    return emit == CPECAN_EMIT_MATCH ? 2 * 128 : (emit == CPECAN_EMIT_INDEL ? 3 * 2 * 128 : 0);
}
constexpr int kStage = 128;  // LDS staging slots per candidate list (two waves' worth: flushed 64 at a time)
constexpr int kPrefetch = 3;  // passes (of 64 cells) of F.match prefetched one diagonal ahead in the traceback
constexpr float kCandMargin = 3.0f;  // log-space slack of the candidate filter (see DESIGN.md "candidate filter")

// FAST: rolling diagonals and the two symbol strings live in LDS.  !FAST: both stay in global memory (bands wider
// than the LDS budget, or sequences too long for it); same arithmetic, workgroup-scope fences.
template <int S, bool FAST>
struct Sweep {
    const KArgs &a;
    const KConsts &m;  // kernarg-resident constants
    DiagCache dc;
    // padded symbol strings: symbol p of X is the base x-1 (p = 0 and p = lX+1 read as N).  FAST: two symbols per
    // byte in LDS (low nibble = even p); otherwise one byte per symbol in global memory.
    const uint8_t *sxp;
    const uint8_t *syp;
    __device__ __forceinline__ int symX(int p) const { return FAST ? (sxp[p >> 1] >> ((p & 1) * 4)) & 15 : sxp[p]; }
    __device__ __forceinline__ int symY(int p) const { return FAST ? (syp[p >> 1] >> ((p & 1) * 4)) & 15 : syp[p]; }
    double *roll;        // rolling buffers: `stride` positions of R = 2S+1 doubles; position 0 = -inf guard
    const double *em;    // LDS emissions: [0..24] match, [25..29] gapX, [30..34] gapY
    // LDS (emission + transition) sums, the second operand of every DP term `from + (eP + tP)` (pairwiseAligner.c:384):
    //   wt[(cX*5 + cY)*kWM + i]             match emission + {matchContinue, matchFromShortX, matchFromShortY, [matchFromLongX, matchFromLongY]}
    //   wt[25*kWM + cX*kWG + i]             gapX emission  + {open, extend, [longOpen, longExtend] | switchToX}
    //   wt[25*kWM + 5*kWG + cY*kWG + i]     gapY emission  + the same for Y
    // One table fetch replaces an emission fetch plus one fp64 add per term (13 adds per cell and direction).
    const double *wt;
    static constexpr int kWM = S == 5 ? 5 : 3, kWG = S == 5 ? 4 : 3;
    const Cubic *lg;     // LDS logAdd cubics
    double *ring;
    Candidate *cand;
    Candidate *stage;  // LDS: kStage candidates per output list, see traceback()
    double *cbuf, *mbuf, *totals;
    int stride;
    int lane;
    int laneR;  // lane * R
    int N;
    // forward sweep state: the two previous diagonals' table entries
    CpkDiag f1, f2;

    // Rolling buffers, position-major: element (row r, position i) is roll[i * R + r], R = 2S+1 rows, positions
    // 0..stride-1, position 0 of every row is the -inf guard.  With the row a compile-time offset the rows of one
    // position cost one address VGPR and immediate offsets (adjacent rows pair up into ds_read2/ds_write2_b64), and
    // a lane stride of R*8 bytes (R odd) is bank-conflict-free for 8-byte accesses.
    //  forward layout : two diagonals, F[d] = rows [(d&1)*S, (d&1)*S + S); F[d] overwrites F[d-2] in place
    //  backward layout: match row in a ring of three, B[d].match = row (d mod 3); the other states in two alternating
    //                   groups, B[d][s] = row 3 + (d&1)*(S-1) + (s-1) for s >= 1
    //  fbuf1/bM1/bG1 return the row pointer at position 1 (cell 0); bG1(d)[s + kR] is state s >= 1 of cell k
    static constexpr int R = 2 * S + 1;
    // Cell k of a diagonal lives at position k+1, i.e. at element offset k*R from a row pointer that already points at
    // position 1 (fbuf1/bM1/bG1 below).  Cell indices are kept premultiplied by R ("kR"): lane*R is computed once per
    // kernel and everything added to it per diagonal / per group is wave-uniform, so no per-access multiply is left.
    // sel(iR, wR): element offset of neighbour cell i of a diagonal with w cells (wR = w*R; w = 0: no such diagonal),
    // or the offset of the -inf guard (position 0) when the neighbour is outside the band.
    __device__ __forceinline__ static int sel(int iR, int wR) { return ((unsigned)iR < (unsigned)wR) ? iR : -R; }
    __device__ __forceinline__ double *fbuf1(int d) const { return roll + R + (d & 1) * S; }
    __device__ __forceinline__ double *bM1(int d) const { return roll + R + (d + 3) % 3; }
    __device__ __forceinline__ double *bG1(int d) const { return roll + R + 2 + (d & 1) * (S - 1); }
    // Forward ring in HBM, per diagonal of W cells: the match row [W], then the other states cell-major [W][S-1]
    // (the traceback reads the match row on its own; a cell's remaining states go out as one 32-byte run).
    __device__ __forceinline__ static size_t ringIdx(int W, int s, int k) {
        return s == 0 ? (size_t)k : (size_t)W + (size_t)k * (S - 1) + (size_t)(s - 1);
    }
    __device__ __forceinline__ double *ringAt(const CpkDiag &g) const { return ring + (size_t)g.ringOff * S; }

    // ---- forward: impl/pairwiseAligner.c:609-629 with stateMachine{5,3}_cellCalculate as the per-cell body ----
    struct FwdCtx {
        int d, xlo, dlR, w1R, dmR, w2R;  // neighbour shifts and widths premultiplied by R
        const double *p1, *p2;           // F[d-1], F[d-2] rows at position 1
    };

    // NC cells (NC = 1 or 2, 64 lanes apart on the same diagonal) computed together.  Fold order per state is the
    // reference's transition-list order; independent folds advance in lock-step (logadd_n).
    template <int NC>
    __device__ __forceinline__ void fwdCells(const FwdCtx &c, const int (&k)[NC], const int (&kR)[NC],
                                             double (&v)[NC][S]) const {
        int cX[NC], cY[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int x = c.xlo + k[q], y = c.d - x;
            cX[q] = symX(x);
            cY[q] = symY(y);
        }
        fwdCellsSym<NC>(c, cX, cY, kR, v);
    }
    // the same with the cells' symbols given (the packed kernel fetches them itself)
    template <int NC>
    __device__ __forceinline__ void fwdCellsSym(const FwdCtx &c, const int (&cX)[NC], const int (&cY)[NC],
                                                const int (&kR)[NC], double (&v)[NC][S]) const {
        const double *p1 = c.p1, *p2 = c.p2;
        if (S == 5) {
            // states: 0 match, 1 shortGapX, 2 shortGapY, 3 longGapX, 4 longGapY (stateMachine.c:261-263)
            double acc[NC * 5], t[NC * 5], m2[NC], m3[NC], m4[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX[q] * 5 + cY[q]) * kWM, *wX = wt + 25 * kWM + cX[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY[q] * kWG;
                const int iL = sel(kR[q] + c.dlR, c.w1R);
                const int iU = sel(kR[q] + c.dlR + R, c.w1R);
                const int iM = sel(kR[q] + c.dmR, c.w2R);
                const double lM = p1[0 + iL], lSX = p1[1 + iL], lLX = p1[3 + iL];
                const double uM = p1[0 + iU], uSY = p1[2 + iU], uLY = p1[4 + iU];
                const double mM = p2[0 + iM], mSX = p2[1 + iM], mSY = p2[2 + iM],
                             mLX = p2[3 + iM], mLY = p2[4 + iM];
                // first two terms of every state's fold: lower block :454-462, middle :463-470, upper :471-479
                acc[q * 5 + 0] = mM + wM[0];
                t[q * 5 + 0] = mSX + wM[1];
                acc[q * 5 + 1] = lM + wX[0];
                t[q * 5 + 1] = lSX + wX[1];
                acc[q * 5 + 2] = uM + wY[0];
                t[q * 5 + 2] = uSY + wY[1];
                acc[q * 5 + 3] = lM + wX[2];
                t[q * 5 + 3] = lLX + wX[3];
                acc[q * 5 + 4] = uM + wY[2];
                t[q * 5 + 4] = uLY + wY[3];
                m2[q] = mSY + wM[2];
                m3[q] = mLX + wM[3];
                m4[q] = mLY + wM[4];
            }
            logadd_n<NC * 5>(lg, acc, t);
            // the match state folds three more terms, in order
            double am[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) am[q] = acc[q * 5 + 0];
            logadd_n<NC>(lg, am, m2);
            logadd_n<NC>(lg, am, m3);
            logadd_n<NC>(lg, am, m4);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                v[q][0] = am[q];
#pragma unroll
                for (int s2 = 1; s2 < 5; s2++) v[q][s2] = acc[q * 5 + s2];
            }
        } else {
            // states: 0 match, 1 gapX, 2 gapY; stateMachine.c:695-713
            double acc[NC * 3], t[NC * 3], u[NC * 3];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX[q] * 5 + cY[q]) * kWM, *wX = wt + 25 * kWM + cX[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY[q] * kWG;
                const int iL = sel(kR[q] + c.dlR, c.w1R);
                const int iU = sel(kR[q] + c.dlR + R, c.w1R);
                const int iM = sel(kR[q] + c.dmR, c.w2R);
                const double lM = p1[0 + iL], lGX = p1[1 + iL], lGY = p1[2 + iL];
                const double uM = p1[0 + iU], uGX = p1[1 + iU], uGY = p1[2 + iU];
                const double mM = p2[0 + iM], mGX = p2[1 + iM], mGY = p2[2 + iM];
                acc[q * 3 + 0] = mM + wM[0];
                t[q * 3 + 0] = mGX + wM[1];
                u[q * 3 + 0] = mGY + wM[2];
                acc[q * 3 + 1] = lM + wX[0];
                t[q * 3 + 1] = lGX + wX[1];
                u[q * 3 + 1] = lGY + wX[2];
                acc[q * 3 + 2] = uM + wY[0];
                t[q * 3 + 2] = uGY + wY[1];
                u[q * 3 + 2] = uGX + wY[2];
            }
            logadd_n<NC * 3>(lg, acc, t);
            logadd_n<NC * 3>(lg, acc, u);
#pragma unroll
            for (int q = 0; q < NC; q++)
#pragma unroll
                for (int s2 = 0; s2 < 3; s2++) v[q][s2] = acc[q * 3 + s2];
        }
    }

    // ringStates: how many states of F[d] go to the forward ring (0, 1 = match row only, S = all)
    __device__ void forward(int d, const CpkDiag &g, int ringStates) {
        const int W = g.width;
        FwdCtx c;
        c.d = d;
        c.xlo = (d + g.xmyL) >> 1;
        const int dl = (g.xmyL - 1 - f1.xmyL) >> 1;  // lower neighbour (d-1, xmy-1) is cell k+dl, upper is k+dl+1
        const int dm = (g.xmyL - f2.xmyL) >> 1;      // middle neighbour (d-2, xmy) is cell k+dm
        c.dlR = dl * R;
        c.w1R = f1.width * R;
        c.dmR = dm * R;
        c.w2R = d >= 2 ? f2.width * R : 0;
        c.p1 = fbuf1(d - 1);
        c.p2 = fbuf1(d - 2);
        double *cur = fbuf1(d);  // same rows as F[d-2]: updated in place
        double *out = ringAt(g);
        // A group of 64 cells reads F[d-2] at k+dm and writes F[d] at k.  With dm >= 0 ascending groups never read a
        // position an earlier group has overwritten; with dm < 0 descending groups never do (DESIGN.md "LDS layout").
        const int nPass = (W + CPK_WAVE - 1) / CPK_WAVE;
        const bool ascending = dm >= 0;
        for (int i = 0; i < nPass; i++) {
            const int kb = (ascending ? i : nPass - 1 - i) * CPK_WAVE;
            const int k0 = kb + lane;
            if (k0 < W) {
                const int kk[1] = {k0};
                const int kkR[1] = {kb * R + laneR};
                double v[1][S];
                fwdCells<1>(c, kk, kkR, v);
#pragma unroll
                for (int s = 0; s < S; s++) cur[s + kkR[0]] = v[0][s];
                if (ringStates > 0) {
                    out[ringIdx(W, 0, k0)] = v[0][0];
                    if (ringStates > 1) {
#pragma unroll
                        for (int s = 1; s < S; s++) out[ringIdx(W, s, k0)] = v[0][s];
                    }
                }
            }
        }
        roll_fence<!FAST>();
        f2 = f1;
        f1 = g;
    }

    // ---- forward sweep as a STREAM of cells (LDS variant).  Diagonals of 101-155 cells fill groups of 64 lanes to 77 %:
    // the last group of a diagonal is mostly empty.  Here a diagonal's leftover cells (fewer than 64) wait and share
    // a group with the first cells of the next diagonal: lanes [0, r) finish diagonal A, lanes [r, 64) start
    // diagonal B = A+1.  Legal when (1) both diagonals run in the same direction (in-place rule above), and (2) B's
    // cells in the shared group only read cells of A that earlier groups have written:
    //   ascending : B cells [0, b) read F[A] up to index b + dl_B        -> need b + dl_B < first leftover cell of A
    //   descending: B cells [W_B - b, W_B) read F[A] down to W_B - b + dl_B -> need that >= end of A's leftover range
    // Within the shared group every load precedes every store (one instruction stream, LDS in order), so A's
    // reads of F[A-1] and B's in-place writes over F[A-1] do not collide.  Otherwise the leftover is flushed as a
    // partly filled group, as before.  Per-lane parameters of the shared group are selects between A's and B's
    // wave-uniform ones.
    struct FwdTail {
        bool has;
        bool asc;
        int lo, n;  // leftover cells [lo, lo + n)
        int W, ringStates;
        FwdCtx c;
        double *cur, *out;
    };
    FwdTail tail{};
    // expectation emitter: backward values of the emitted cells of the segment being traced back, [cell][S], written by
    // traceback() and read by expectations() (set by the kernel; null for the other emitters)
    double *bring = nullptr;

    // one group of cells of ONE diagonal: cells [kb, kb + 64) clipped to [lo, hi)
    __device__ __forceinline__ void fwdGroupUniform(const FwdCtx &c, double *cur, double *out, int W, int ringStates, int kb,
                                                    int lo, int hi) {
        const int k0 = kb + lane;
        if (k0 >= lo && k0 < hi) {
            const int kk[1] = {k0};
            const int kkR[1] = {kb * R + laneR};
            double v[1][S];
            fwdCells<1>(c, kk, kkR, v);
#pragma unroll
            for (int s = 0; s < S; s++) cur[s + kkR[0]] = v[0][s];
            if (ringStates > 0) {
                out[ringIdx(W, 0, k0)] = v[0][0];
                if (ringStates > 1) {
#pragma unroll
                    for (int s = 1; s < S; s++) out[ringIdx(W, s, k0)] = v[0][s];
                }
            }
        }
    }

    __device__ void flushTail() {
        if (!tail.has) return;
        fwdGroupUniform(tail.c, tail.cur, tail.out, tail.W, tail.ringStates, tail.lo, tail.lo, tail.lo + tail.n);
        tail.has = false;
    }

    __device__ void forwardStream(int d, const CpkDiag &g, int ringStates) {
        const int W = g.width;
        FwdCtx c;
        c.d = d;
        c.xlo = (d + g.xmyL) >> 1;
        const int dl = (g.xmyL - 1 - f1.xmyL) >> 1;
        const int dm = (g.xmyL - f2.xmyL) >> 1;
        c.dlR = dl * R;
        c.w1R = f1.width * R;
        c.dmR = dm * R;
        c.w2R = d >= 2 ? f2.width * R : 0;
        c.p1 = fbuf1(d - 1);
        c.p2 = fbuf1(d - 2);
        double *cur = fbuf1(d);
        double *out = ringAt(g);
        const bool asc = dm >= 0;
        int lo = 0, hi = W;  // cells of this diagonal still to do
        if (tail.has) {
            const int r = tail.n;
            const int b = CPK_WAVE - r < W ? CPK_WAVE - r : W;
            const int kB0 = asc ? 0 : W - b;  // first cell of B's share
            const bool reads_done = asc ? (b + dl < tail.lo) : (kB0 + dl >= tail.lo + tail.n);
            if (tail.asc == asc && reads_done) {
                const bool inA = lane < r;
                const bool inB = !inA && lane - r < b;
                // idle lanes (a narrow B) recompute B's first cell and store nothing
                const int k = inA ? tail.lo + lane : (inB ? kB0 + lane - r : kB0);
                const int kR = inA ? tail.lo * R + laneR : (inB ? (kB0 - r) * R + laneR : kB0 * R);
                FwdCtx m;
                m.d = inA ? tail.c.d : c.d;
                m.xlo = inA ? tail.c.xlo : c.xlo;
                m.dlR = inA ? tail.c.dlR : c.dlR;
                m.w1R = inA ? tail.c.w1R : c.w1R;
                m.dmR = inA ? tail.c.dmR : c.dmR;
                m.w2R = inA ? tail.c.w2R : c.w2R;
                // rows by parity of the diagonal: A writes over F[A-2] in `tail.cur` and reads F[A-1] from the other set,
                // which is the set B = A+1 writes into: two selects cover p1, p2 and cur
                double *curL = inA ? tail.cur : cur;
                m.p1 = inA ? cur : tail.cur;
                m.p2 = curL;
                double *outL = inA ? tail.out : out;
                const int WL = inA ? tail.W : W;
                const int rsL = inA ? tail.ringStates : ringStates;
                const int kk[1] = {k};
                const int kkR[1] = {kR};
                double v[1][S];
                fwdCells<1>(m, kk, kkR, v);
                if (inA || inB) {
#pragma unroll
                    for (int s = 0; s < S; s++) curL[s + kR] = v[0][s];
                    if (rsL > 0) {
                        outL[ringIdx(WL, 0, k)] = v[0][0];
                        if (rsL > 1) {
#pragma unroll
                            for (int s = 1; s < S; s++) outL[ringIdx(WL, s, k)] = v[0][s];
                        }
                    }
                }
                tail.has = false;
                if (asc) lo = b;
                else hi = W - b;
            } else {
                flushTail();
            }
        }
        // whole groups of this diagonal, in its direction; what is left over waits for the next diagonal
        while (hi - lo >= CPK_WAVE) {
            if (asc) {
                fwdGroupUniform(c, cur, out, W, ringStates, lo, lo, hi);
                lo += CPK_WAVE;
            } else {
                fwdGroupUniform(c, cur, out, W, ringStates, hi - CPK_WAVE, lo, hi);
                hi -= CPK_WAVE;
            }
        }
        if (hi > lo && W < CPK_WAVE) {
            // a diagonal of fewer than 64 cells can never share (its first cell would have to be behind the reads of
            // the next diagonal): do it now
            fwdGroupUniform(c, cur, out, W, ringStates, lo, lo, hi);
        } else if (hi > lo) {
            tail.has = true;
            tail.asc = asc;
            tail.lo = lo;
            tail.n = hi - lo;
            tail.W = W;
            tail.ringStates = ringStates;
            tail.c = c;
            tail.cur = cur;
            tail.out = out;
        }
        f2 = f1;
        f1 = g;
    }

    // Puts diagonal d of the forward ring back into its rolling buffer (after a traceback used the buffers).
    __device__ void reloadForward(const CpkDiag &g, int d) {
        const int W = g.width;
        double *cur = fbuf1(d);
        const double *src = ringAt(g);
        for (int kb = 0; kb < W; kb += CPK_WAVE) {
            const int k = kb + lane;
            if (k < W) {
#pragma unroll
                for (int s = 0; s < S; s++) cur[s + kb * R + laneR] = ld_self(src + ringIdx(W, s, k));
            }
        }
        roll_fence<!FAST>();
    }

    struct BwdCtx {
        int d2, xlo, dbR, wBR, daR, wAR;  // source shifts and widths premultiplied by R
        const double *pb, *pa;            // B[d2+1] gap rows, B[d2+2] match row, at position 1
    };
    // B[d2][k] gathered from B[d2+1], B[d2+2] in the reference's scatter order, NC cells at a time
    template <int NC>
    __device__ __forceinline__ void bwdCells(const BwdCtx &c, const int (&k)[NC], const int (&kR)[NC],
                                             double (&v)[NC][S]) const {
        int cX1[NC], cY1[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int x = c.xlo + k[q], y = c.d2 - x;
            cX1[q] = symX(x + 1);  // symbols of the source cells (x+1,.) and (.,y+1)
            cY1[q] = symY(y + 1);
        }
        bwdCellsSym<NC>(c, cX1, cY1, kR, v);
    }
    template <int NC>
    __device__ __forceinline__ void bwdCellsSym(const BwdCtx &c, const int (&cX1)[NC], const int (&cY1)[NC],
                                                const int (&kR)[NC], double (&v)[NC][S]) const {
        const double *pb = c.pb, *pa = c.pa;
        if (S == 5) {
            double acc[NC * 5], t[NC * 5], m2[NC], m3[NC], m4[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX1[q] * 5 + cY1[q]) * kWM, *wX = wt + 25 * kWM + cX1[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY1[q] * kWG;
                const int iU = sel(kR[q] + c.dbR, c.wBR);      // cell (x, y+1): its "upper" neighbour is the target
                const int iL = sel(kR[q] + c.dbR + R, c.wBR);  // cell (x+1, y): its "lower" neighbour is the target
                const int iA = sel(kR[q] + c.daR, c.wAR);      // cell (x+1, y+1): its "middle" neighbour is the target
                const double aM = pa[iA];
                const double uSY = pb[2 + iU], uLY = pb[4 + iU];
                const double lSX = pb[1 + iL], lLX = pb[3 + iL];
                // per target state: (1) middle term from d2+2, (2) upper-block terms, (3) lower-block terms
                acc[q * 5 + 0] = aM + wM[0];
                t[q * 5 + 0] = uSY + wY[0];
                m2[q] = uLY + wY[2];
                m3[q] = lSX + wX[0];
                m4[q] = lLX + wX[2];
                acc[q * 5 + 1] = aM + wM[1];
                t[q * 5 + 1] = lSX + wX[1];
                acc[q * 5 + 2] = aM + wM[2];
                t[q * 5 + 2] = uSY + wY[1];
                acc[q * 5 + 3] = aM + wM[3];
                t[q * 5 + 3] = lLX + wX[3];
                acc[q * 5 + 4] = aM + wM[4];
                t[q * 5 + 4] = uLY + wY[3];
            }
            logadd_n<NC * 5>(lg, acc, t);
            double am[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) am[q] = acc[q * 5 + 0];
            logadd_n<NC>(lg, am, m2);
            logadd_n<NC>(lg, am, m3);
            logadd_n<NC>(lg, am, m4);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                v[q][0] = am[q];
#pragma unroll
                for (int s2 = 1; s2 < 5; s2++) v[q][s2] = acc[q * 5 + s2];
            }
        } else {
            double acc[NC * 3], t[NC * 3], u[NC * 3];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX1[q] * 5 + cY1[q]) * kWM, *wX = wt + 25 * kWM + cX1[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY1[q] * kWG;
                const int iU = sel(kR[q] + c.dbR, c.wBR);
                const int iL = sel(kR[q] + c.dbR + R, c.wBR);
                const int iA = sel(kR[q] + c.daR, c.wAR);
                const double aM = pa[iA];
                const double uGY = pb[2 + iU];
                const double lGX = pb[1 + iL];
                acc[q * 3 + 0] = aM + wM[0];
                t[q * 3 + 0] = uGY + wY[0];
                u[q * 3 + 0] = lGX + wX[0];
                acc[q * 3 + 1] = aM + wM[1];
                t[q * 3 + 1] = uGY + wY[2];
                u[q * 3 + 1] = lGX + wX[1];
                acc[q * 3 + 2] = aM + wM[2];
                t[q * 3 + 2] = uGY + wY[1];
                u[q * 3 + 2] = lGX + wX[2];
            }
            logadd_n<NC * 3>(lg, acc, t);
            logadd_n<NC * 3>(lg, acc, u);
#pragma unroll
            for (int q = 0; q < NC; q++)
#pragma unroll
                for (int s2 = 0; s2 < 3; s2++) v[q][s2] = acc[q * 3 + s2];
        }
    }

    // ---- traceback of one segment (pairwiseAligner.c:796-862).
    // The reference scatters from diagonal d2+1 / d2+2 into d2 (:392-395, :631-634); this gathers the same terms in
    // the same order.  Per emitted diagonal it forms fb = F.s + B.s for the NL emitted
    // states (match; plus gapX, gapY for the indel emitter, :691-733) and keeps the cells that can still reach the
    // threshold once the total is known; on refresh diagonals it also writes the two per-cell series whose sequential
    // logAdd folds give the total probability (:636-653).
    // nCand[l] receives the number of candidates appended to list l (visit order: diagonal descending, x-y ascending).
    template <int NL, bool CANDS>
    __device__ void traceback(const CpkSegment &sg, const double *endPrior, double *dbgFb, int (&nCand)[NL]) {
        const int J = sg.nRefresh;
        const float logThr = (float)log(m.threshold);  // -inf for threshold 0: every cell is a candidate
        // Candidates are staged in LDS (a ring of kStage slots per list) and go to HBM 64 at a time as one coalesced
        // store.  A store per group would sit between the F prefetch below and its use: loads and stores share vmcnt
        // on gfx9, the compiler then waits with vmcnt(0) at every diagonal, i.e. for the write acknowledgement too.
        int pend[NL], head[NL];  // staged entries and ring position of the oldest, per list (wave-uniform)
#pragma unroll
        for (int l = 0; l < NL; l++) nCand[l] = pend[l] = head[l] = 0;
        auto flush = [&](int l, int n) {  // the n <= 64 oldest staged candidates of list l -> cand[l][nCand[l]..]
            if (lane < n) {
                cand[(size_t)l * a.geo.fbCells + nCand[l] + lane] = stage[l * kStage + ((head[l] + lane) & (kStage - 1))];
            }
            head[l] = (head[l] + n) & (kStage - 1);
            pend[l] -= n;
            nCand[l] += n;
        };
        // expectation emitter: cells of the segment are numbered from the first cell of its lowest emitted diagonal
        const int bBase = (!CANDS && bring) ? dc.table[sg.tbPrev + 1].cellOff : 0;
        float lastMax = -__builtin_huge_valf();
        double ep[S];  // end prior: loaded AND waited for here (the empty asm consumes the registers); a value whose
                       // load may still be pending at the loop head costs a vmcnt(0) in front of every group
#pragma unroll
        for (int s = 0; s < S; s++) ep[s] = endPrior[s];
#pragma unroll
        for (int s = 0; s < S; s++) asm volatile("" : "+v"(ep[s]));
        CpkDiag gb{}, ga{};  // table entries of d2+1 and d2+2
        CpkDiag g = dc.get(sg.dTop, true);
        CpkDiag gnext = sg.dTop >= 1 ? dc.get(sg.dTop - 1, true) : CpkDiag{};  // entry of d2-1
        // F rows of the emitted states (list l emits state l), prefetched one diagonal ahead of their use.  wantF: the
        // emitted diagonals plus the one above the first refresh point (its F.m + B.m feeds the straddle term).
        // The loads are unconditional (lanes past the end of the diagonal re-read its last cell, diagonals that are not
        // emitted are read all the same): a predicate per load costs more instructions than the load.
        double fmCur[NL][kPrefetch];
        auto loadRows = [&](const CpkDiag &gd, double (&dst)[NL][kPrefetch]) {
            const double *src = ringAt(gd);
#pragma unroll
            for (int l = 0; l < NL; l++)
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
                    dst[l][q] = ld_self(src + ringIdx(gd.width, l, k < gd.width ? k : gd.width - 1));
                }
        };
        loadRows(g, fmCur);
#pragma unroll
        for (int l = 0; l < NL; l++)
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) asm volatile("" : "+v"(fmCur[l][q]));  // complete before the loop
        // Refresh points (every 10th emitted diagonal, counted from tbFrom) as a countdown: no division per diagonal.
        int untilRefresh = sg.dTop - sg.tbFrom;  // diagonals until the next refresh point
        int jr = 0;                              // ... and its index
        for (int d2 = sg.dTop; d2 > sg.tbPrev;) {
          // Table entries of the 64 diagonals ending at d2-2: each diagonal of the sweep needs one new entry, that of d2-2.
          dc.load(d2 - 2 - (CPK_WAVE - 1));
          for (int ci = CPK_WAVE - 1; ci >= 0 && d2 > sg.tbPrev; ci--, d2--) {
            const bool seeded = d2 == sg.dTop;
            const int W = g.width;
            const bool emit = d2 <= sg.tbFrom;
            const bool refresh = untilRefresh == 0;
            // "Matches straddling diagonal r" (pairwiseAligner.c:643-651) is a middle-block forward step from F[r-1] into the
            // cells of r+1, times B[r+1].  The match state is reached through the middle block only, so that step IS
            // F[r+1].match (same terms, same order: stateMachine.c:463-470 / :703-707), and the series to fold is
            // F[r+1].m + B[r+1].m -- the fb values this loop forms anyway, one diagonal before the refresh point.
            const bool feeds = untilRefresh == 1 && d2 - 1 > sg.tbPrev;
            const int jrNext = jr;
            // issue the loads for diagonal d2-1 now: one diagonal of arithmetic covers the HBM round trip
            double fmNext[NL][kPrefetch];
            loadRows(gnext, fmNext);
            const CpkDiag gnext2 = dc.at(ci);  // entry of d2-2 (of diagonal 0 when d2 < 2: not used then)
            double *curM = bM1(d2), *curG = bG1(d2);
            const double *fsrc = ringAt(g);
            const int xlo = (d2 + g.xmyL) >> 1;
            BwdCtx c;
            c.d2 = d2;
            c.xlo = xlo;
            c.dbR = ((g.xmyL - 1 - gb.xmyL) >> 1) * R;  // source (d2+1, xmy-1) is cell k+db, (d2+1, xmy+1) is k+db+1
            c.wBR = seeded ? 0 : gb.width * R;
            c.daR = ((g.xmyL - ga.xmyL) >> 1) * R;      // source (d2+2, xmy) is cell k+da
            c.wAR = (!seeded && d2 + 2 <= sg.dTop) ? ga.width * R : 0;
            c.pb = bG1(d2 + 1);
            c.pa = bM1(d2 + 2);
            const float keepFrom = lastMax + logThr - kCandMargin;  // wave-uniform
            // Refresh diagonals read the remaining states of F[d2] (cell dot products).  Those loads are issued HERE,
            // before the compute loop of the diagonal, and consumed after it, so their HBM latency hides behind a few
            // thousand cycles of arithmetic.
            double rfC[S][kPrefetch];  // F[d2][s][k], s >= NL   (rows < NL are in fmCur)
            if (refresh) {
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
#pragma unroll
                    for (int s2 = NL; s2 < S; s2++) rfC[s2][q] = k < W ? ld_self(fsrc + ringIdx(W, s2, k)) : 0.0;
                }
            }
            // One group of 64 cells.  Wave-uniform (the candidate counts must stay identical in every lane): lanes past
            // the end of the diagonal recompute its last cell and have their stores masked.  f0 = F[d2][l][k0].
            const int lastR = (W - 1) * R;
            auto group = [&](int kb, const double (&f0)[NL]) {
                const int k0 = kb + lane, kR0 = kb * R + laneR;
                const bool on = k0 < W;
                double v[1][S];
                if (seeded) {
                    // every cell of the top diagonal gets the end-state prior (pairwiseAligner.c:798-799)
#pragma unroll
                    for (int s = 0; s < S; s++) v[0][s] = ep[s];
                } else {
                    const int kk[1] = {on ? k0 : W - 1};
                    const int kkR[1] = {on ? kR0 : lastR};
                    bwdCells<1>(c, kk, kkR, v);
                }
                if (on) {
                    curM[kR0] = v[0][0];
#pragma unroll
                    for (int s = 1; s < S; s++) curG[s + kR0] = v[0][s];
                }
                if (!CANDS && bring && emit && on) {  // kept for the expectation step: it needs B again, not its neighbours
                    double *bo = bring + (size_t)(g.cellOff - bBase + k0) * S;
#pragma unroll
                    for (int s = 0; s < S; s++) bo[s] = v[0][s];
                }
                if (feeds && on) mbuf[(size_t)k0 * J + jrNext] = f0[0] + v[0][0];  // every cell of the diagonal (:647)
                if (emit) {
                    const int x = xlo + k0, y = d2 - x;
                    double fbv[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) fbv[l] = f0[l] + v[0][l];
                    if (on && dbgFb) dbgFb[g.cellOff + k0] = fbv[0];
                    // candidate filter: a cell survives when it is within log(threshold) - margin of the bound on the
                    // total probability (DESIGN.md "candidate filter").  Match cells need x > 0 and y > 0, gapX cells
                    // x > 0, gapY cells y > 0 (pairwiseAligner.c:680, :719, :725).
#pragma unroll
                    for (int l = 0; l < (CANDS ? NL : 0); l++) {
                        const bool cell = l == 0 ? (x > 0 && y > 0) : (l == 1 ? x > 0 : y > 0);
                        const bool keep = on && cell && (float)fbv[l] >= keepFrom;
                        const unsigned long long mask = __ballot(keep);
                        if (keep) {
                            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            Candidate cd;
                            cd.fb = fbv[l];
                            cd.x = x;
                            cd.y = y;
                            stage[l * kStage + ((head[l] + pend[l] + rank) & (kStage - 1))] = cd;
                        }
                        pend[l] += __popcll(mask);
                        if (pend[l] >= CPK_WAVE) flush(l, CPK_WAVE);
                    }
                }
            };
            // the first kPrefetch groups take F from the prefetched registers (compile-time group index) ...
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) {
                if (q * CPK_WAVE < W) {
                    double f0[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) f0[l] = fmCur[l][q];
                    group(q * CPK_WAVE, f0);
                }
            }
            // ... wider diagonals load it on the spot
            for (int kb = kPrefetch * CPK_WAVE; kb < W; kb += CPK_WAVE) {
                double f0[NL];
#pragma unroll
                for (int l = 0; l < NL; l++)
                    f0[l] = ((emit || feeds) && kb + lane < W) ? ld_self(fsrc + ringIdx(W, l, kb + lane)) : 0.0;
                group(kb, f0);
            }
            roll_fence<!FAST>();
            if (refresh) {
                // (a) cell dot products over states (cell_dotProduct, pairwiseAligner.c:402-408) and, for the candidate
                //     bound, this diagonal's largest F.m + B.m: renewed every 10th diagonal as
                //     max(this diagonal's maximum, old bound - 1); the reference itself asserts that consecutive totals
                //     differ by less than 1.0 (:834), so the decayed bound stays below the current total.
                float diagMax = -__builtin_huge_valf();
                auto dotCell = [&](int k, int kR, const double (&fRow)[S]) {
                    double t = fRow[0] + curM[kR];
                    const int x = xlo + k, y = d2 - x;
                    const float fbf = (x > 0 && y > 0) ? (float)t : -__builtin_huge_valf();
#pragma unroll
                    for (int s2 = 1; s2 < S; s2++) t = logadd(lg, t, fRow[s2] + curG[s2 + kR]);
                    cbuf[(size_t)k * J + jr] = t;
                    return fbf;
                };
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
                    float fbf = -__builtin_huge_valf();
                    if (q * CPK_WAVE < W) {  // wave-uniform
                        if (k < W) {
                            double fRow[S];
#pragma unroll
                            for (int s2 = 0; s2 < S; s2++) fRow[s2] = s2 < NL ? fmCur[s2][q] : rfC[s2][q];
                            fbf = dotCell(k, q * CPK_WAVE * R + laneR, fRow);
                        }
                        if (CANDS) diagMax = fmaxf(diagMax, wave_max_f32(fbf));
                    }
                }
                for (int kb = kPrefetch * CPK_WAVE; kb < W; kb += CPK_WAVE) {  // diagonals wider than the prefetch
                    const int k = kb + lane;
                    float fbf = -__builtin_huge_valf();
                    if (k < W) {
                        double fRow[S];
#pragma unroll
                        for (int s2 = 0; s2 < S; s2++) fRow[s2] = ld_self(fsrc + ringIdx(W, s2, k));
                        fbf = dotCell(k, kb * R + laneR, fRow);
                    }
                    if (CANDS) diagMax = fmaxf(diagMax, wave_max_f32(fbf));
                }
                if (CANDS) lastMax = fmaxf(diagMax, lastMax - 1.0f);
            }
            // slide the window of table entries and prefetched F rows down one diagonal
            ga = gb;
            gb = g;
            g = gnext;
            gnext = gnext2;
            // The empty asm consumes the prefetched registers HERE, one whole diagonal after their loads were issued and
            // before the next prefetch goes out: left to itself hipcc waits at the first use inside the next diagonal,
            // behind the next prefetch, with vmcnt(0) -- the full HBM round trip exposed on every diagonal.
#pragma unroll
            for (int l = 0; l < NL; l++)
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    asm volatile("" : "+v"(fmNext[l][q]));
                    fmCur[l][q] = fmNext[l][q];
                }
            if (refresh) {
                untilRefresh = CPK_REFRESH_PERIOD - 1;
                jr++;
            } else {
                untilRefresh--;
            }
          }
        }
#pragma unroll
        for (int l = 0; l < (CANDS ? NL : 0); l++) flush(l, pend[l]);
    }

    // ---- expectation step (diagonalCalculationExpectations, pairwiseAligner.c:735-746; updateExpectations :418-432).
    // Second backward sweep of the segment, run once the totals are known.  For every emitted diagonal d2, every cell
    // of B[d2] and every transition into it: p = exp(F_nbr[from] + B[to] + (eP + tP) - total) with the neighbours
    // taken from F[d2-1] / F[d2-2]; T[from][to] += p, and E[to][cX][cY] += p when neither symbol is N.
    // The reference has already freed F[d2-2] at the lowest diagonal of a segment (:843-845), so the middle block
    // contributes nothing there; reproduced.  Sums are linear-space fp64: order-insensitive at the 1e-5 gate.
    // tAcc: per-lane sums, one per transition in list order; eLds: [state*16 + cX*4 + cY] in LDS (fp64 LDS atomics).
    static constexpr int kNT = S == 5 ? 13 : 9;

    // One group of 64 cells of one emitted diagonal, as the expectation step sees it.
    struct ExpItem {
        int d2, kb, W, xlo, dl, dm, w1, w2, cellOff;
        const double *f1, *f2;
        double total;
        bool valid;
    };
    // What the step reads for its cell: B of the cell, F[d2-1] at the lower / upper neighbour, F[d2-2] at the middle one.
    struct ExpLoads {
        double v[S], fL[S], fU[S], fM[S];
    };

    __device__ void expectations(const CpkSegment &sg, double (&tAcc)[kNT], double *eLds, double &likelihood) {
        const int bBase = dc.table[sg.tbPrev + 1].cellOff;
        // The step has no dependency between cells: it is a stream of (diagonal, group) items.  The loads of the NEXT
        // item are issued before the events of the current one are computed, so the HBM round trip of 16 values per
        // cell runs beside ~250 vector instructions instead of in front of them.
        auto first_of = [&](int d2) {
            ExpItem it{};
            it.valid = d2 > sg.tbPrev;
            if (!it.valid) return it;
            const CpkDiag g = dc.get(d2, true);
            const CpkDiag g1 = dc.get(d2 - 1, true);      // F[d2-1]: always alive (d2-1 >= tbPrev)
            const bool haveM2 = d2 - 2 >= sg.tbPrev;      // F[d2-2] is gone at d2 == tbPrev+1 (:843-845)
            const CpkDiag g2 = haveM2 ? dc.get(d2 - 2, true) : CpkDiag{};
            it.d2 = d2;
            it.kb = 0;
            it.W = g.width;
            it.xlo = (d2 + g.xmyL) >> 1;
            it.dl = (g.xmyL - 1 - g1.xmyL) >> 1;  // lower neighbour (d2-1, xmy-1) is cell k+dl of F[d2-1]
            it.dm = (g.xmyL - g2.xmyL) >> 1;      // middle neighbour (d2-2, xmy) is cell k+dm of F[d2-2]
            it.w1 = g1.width;
            it.w2 = haveM2 ? g2.width : 0;
            it.cellOff = g.cellOff;
            it.f1 = ringAt(g1);
            it.f2 = ringAt(g2);
            it.total = ld_self(totals + (sg.tbFrom - d2) / CPK_REFRESH_PERIOD);
            return it;
        };
        auto next_of = [&](const ExpItem &it) {
            if (it.kb + CPK_WAVE < it.W) {
                ExpItem n = it;
                n.kb += CPK_WAVE;
                return n;
            }
            return first_of(it.d2 - 1);
        };
        auto issue = [&](const ExpItem &it, ExpLoads &L) {
            int k = it.kb + lane;
            k = k < it.W ? k : it.W - 1;  // lanes past the end re-read the last cell
            const int kL = k + it.dl, kU = k + it.dl + 1, kM = k + it.dm;
            const int qL = (unsigned)kL < (unsigned)it.w1 ? kL : 0, qU = (unsigned)kU < (unsigned)it.w1 ? kU : 0;
            const bool okM = (unsigned)kM < (unsigned)it.w2;
            const int qM = okM ? kM : 0;
            const double *bo = bring + (size_t)(it.cellOff - bBase + k) * S;
#pragma unroll
            for (int s = 0; s < S; s++) {
                // 5 states: the lower block reads M, sX, lX, the upper block M, sY, lY; 3 states: all three
                const bool needL = S == 3 || s == 0 || s == 1 || s == 3, needU = S == 3 || s == 0 || s == 2 || s == 4;
                L.v[s] = it.valid ? ld_self(bo + s) : 0.0;
                L.fL[s] = (it.valid && needL) ? ld_self(it.f1 + ringIdx(it.w1, s, qL)) : 0.0;
                L.fU[s] = (it.valid && needU) ? ld_self(it.f1 + ringIdx(it.w1, s, qU)) : 0.0;
                L.fM[s] = (it.valid && it.w2 > 0) ? ld_self(it.f2 + ringIdx(it.w2, s, qM)) : 0.0;
            }
        };
        ExpItem cur = first_of(sg.tbFrom);
        ExpLoads Lc;
        issue(cur, Lc);
        while (cur.valid) {
            const ExpItem nxt = next_of(cur);
            ExpLoads Ln;
            issue(nxt, Ln);
#pragma unroll
            for (int s = 0; s < S; s++) {  // the wait for the current item's loads sits here, one item after their issue
                asm volatile("" : "+v"(Lc.v[s]), "+v"(Lc.fL[s]), "+v"(Lc.fU[s]), "+v"(Lc.fM[s]));
            }
            if (cur.kb == 0) likelihood += cur.total;  // once per diagonal (:743)
            const int k = cur.kb + lane;
            if (k < cur.W) {
                const int kL = k + cur.dl, kU = k + cur.dl + 1, kM = k + cur.dm;
                const bool okL = (unsigned)kL < (unsigned)cur.w1, okU = (unsigned)kU < (unsigned)cur.w1,
                           okM = (unsigned)kM < (unsigned)cur.w2;
                double fL[S], fU[S], fM[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    fL[s] = okL ? Lc.fL[s] : NEG_INF;
                    fU[s] = okU ? Lc.fU[s] : NEG_INF;
                    fM[s] = okM ? Lc.fM[s] : NEG_INF;
                }
                const int x = cur.xlo + k, y = cur.d2 - x;
                const int cX = symX(x), cY = symY(y);
                // (emission + transition) sums of the events, from the same LDS table as the sweeps (Sweep::wt)
                const double *wM = wt + (cX * 5 + cY) * kWM, *wX = wt + 25 * kWM + cX * kWG, *wY = wt + 25 * kWM + 5 * kWG + cY * kWG;
                const bool acgt = cX < CPK_SYM_N && cY < CPK_SYM_N;
                const int eIdx = cX * 4 + cY;
                const double total = cur.total;
                double eAcc[S];  // this cell's events summed per target state: one LDS atomic per state, not per event
#pragma unroll
                for (int s = 0; s < S; s++) eAcc[s] = 0.0;
                // one (transition, emission) event: impl/pairwiseAligner.c:426-431
                auto event = [&](int ti, double from, int to, double w) {
                    const double p = exp_1e7(from + Lc.v[to] + w - total);
                    tAcc[ti] += p;
                    eAcc[to] += p;
                };
                if (S == 5) {
                    event(0, fL[0], 1, wX[0]);   // M -> shortX (open)
                    event(1, fL[1], 1, wX[1]);   // shortX -> shortX
                    event(2, fL[0], 3, wX[2]);   // M -> longX (open)
                    event(3, fL[3], 3, wX[3]);   // longX -> longX
                    event(4, fM[0], 0, wM[0]);   // M -> M
                    event(5, fM[1], 0, wM[1]);   // shortX -> M
                    event(6, fM[2], 0, wM[2]);   // shortY -> M
                    event(7, fM[3], 0, wM[3]);   // longX -> M
                    event(8, fM[4], 0, wM[4]);   // longY -> M
                    event(9, fU[0], 2, wY[0]);   // M -> shortY
                    event(10, fU[2], 2, wY[1]);  // shortY -> shortY
                    event(11, fU[0], 4, wY[2]);  // M -> longY
                    event(12, fU[4], 4, wY[3]);  // longY -> longY
                } else {
                    event(0, fL[0], 1, wX[0]);  // M -> gapX
                    event(1, fL[1], 1, wX[1]);  // gapX -> gapX
                    event(2, fL[2], 1, wX[2]);  // gapY -> gapX (switch)
                    event(3, fM[0], 0, wM[0]);
                    event(4, fM[1], 0, wM[1]);
                    event(5, fM[2], 0, wM[2]);
                    event(6, fU[0], 2, wY[0]);  // M -> gapY
                    event(7, fU[2], 2, wY[1]);  // gapY -> gapY
                    event(8, fU[1], 2, wY[2]);  // gapX -> gapY (switch)
                }
                if (acgt) {  // emissions are counted for ACGT x ACGT cells only (:429)
                    double *copy = eLds + (lane & (kExpectCopies - 1)) * 80;
#pragma unroll
                    for (int s = 0; s < S; s++) atomicAdd(&copy[s * 16 + eIdx], eAcc[s]);
                }
            }
            cur = nxt;
            Lc = Ln;
        }
    }

    // ---- total probability at every refresh point of the segment: one lane per refresh point, each doing the
    // reference's sequential folds (dpDiagonal_dotProduct :513-523, then the straddle term :649).
    __device__ void foldTotals(const CpkSegment &sg, const CpkDiag *table) {
        const int J = sg.nRefresh;
        for (int j0 = 0; j0 < J; j0 += CPK_WAVE) {
            const int j = j0 + lane;
            const bool on = j < J;
            const int r = sg.tbFrom - CPK_REFRESH_PERIOD * (on ? j : 0);
            const int Wc = on ? table[r].width : 0;
            const int Wm = (on && r + 1 <= sg.dTop) ? table[r + 1].width : 0;
            double total = NEG_INF, straddle = NEG_INF;
            const int WcMax = wave_max_i32(Wc), WmMax = wave_max_i32(Wm);
            // loads are issued eight at a time, then folded in order; padding with -inf leaves the fold unchanged
            // because logAdd(x, -inf) returns x exactly
            for (int k = 0; k < WcMax; k += 8) {
                double x[8];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = k + i < Wc ? ld_self(cbuf + (size_t)(k + i) * J + j) : NEG_INF;
#pragma unroll
                for (int i = 0; i < 8; i++) total = logadd(lg, total, x[i]);
            }
            for (int k = 0; k < WmMax; k += 8) {
                double x[8];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = k + i < Wm ? ld_self(mbuf + (size_t)(k + i) * J + j) : NEG_INF;
#pragma unroll
                for (int i = 0; i < 8; i++) straddle = logadd(lg, straddle, x[i]);
            }
            if (on) {
                if (r + 1 <= sg.dTop) total = logadd(lg, total, straddle);
                totals[j] = total;
            }
        }
        roll_fence<true>();
    }

    // ---- thresholded posteriors (pairwiseAligner.c:655-689) from the candidate list, walked backwards so that the
    // output is in the reference's list order (diagonal ascending, x-y descending).
    __device__ int emitMatches(const CpkSegment &sg, const Candidate *cand, int nCand, int32_t *out, int outCap,
                               int count) {
        const double thr = m.threshold;
        for (int top = nCand; top > 0; top -= CPK_WAVE) {
            const int i = top - 1 - lane;
            const bool valid = i >= 0;
            double p = 0.0;
            int x = 0, y = 0;
            if (valid) {
                const double fbv = ld_self(&cand[i].fb);
                const long long xy = __hip_atomic_load(reinterpret_cast<const long long *>(&cand[i].x), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
                x = (int)(xy & 0xffffffffll);
                y = (int)(xy >> 32);
                const double total = ld_self(totals + (sg.tbFrom - (x + y)) / CPK_REFRESH_PERIOD);
                p = exp(fbv - total);
            }
            const bool keep = valid && p >= thr;
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                if (p > 1.0) p = 1.0;
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                const int pos = count + rank;
                if (pos < outCap) {
                    out[3 * (size_t)pos + 0] = (int32_t)floor(p * (double)CPECAN_PROB_1);
                    out[3 * (size_t)pos + 1] = x - 1;
                    out[3 * (size_t)pos + 2] = y - 1;
                }
            }
            count += __popcll(mask);
        }
        return count;
    }
};

// (emission + transition) sums of every DP term, see Sweep::wt for the layout
template <int S>
__device__ __forceinline__ void fill_weights(double *wt, const CpkModel &m, const KConsts &kc, int lane) {
    constexpr int kWM = S == 5 ? 5 : 3, kWG = S == 5 ? 4 : 3;
    for (int i = lane; i < 25 * kWM + 10 * kWG; i += CPK_WAVE) {
        double e, t;
        if (i < 25 * kWM) {
            const int j = i % kWM;
            e = m.matchEm[i / kWM];
            t = j == 0 ? kc.matchContinue : j == 1 ? kc.matchFromShortX : j == 2 ? kc.matchFromShortY
              : j == 3 ? kc.matchFromLongX : kc.matchFromLongY;
        } else {
            const int r = i - 25 * kWM, y = r >= 5 * kWG, c = (r - y * 5 * kWG) / kWG, j = (r - y * 5 * kWG) % kWG;
            e = y ? m.gapYEm[c] : m.gapXEm[c];
            if (S == 5) {
                t = j == 0 ? (y ? kc.shortOpenY : kc.shortOpenX) : j == 1 ? (y ? kc.shortExtendY : kc.shortExtendX)
                  : j == 2 ? (y ? kc.longOpenY : kc.longOpenX) : (y ? kc.longExtendY : kc.longExtendX);
            } else {
                t = j == 0 ? (y ? kc.shortOpenY : kc.shortOpenX) : j == 1 ? (y ? kc.shortExtendY : kc.shortExtendX)
                  : (y ? kc.shortSwitchToY : kc.shortSwitchToX);
            }
        }
        wt[i] = e + t;
    }
}

// EMIT: CPECAN_EMIT_MATCH (0), CPECAN_EMIT_INDEL (1) or kEmitForward (3: forward sweep only, total probability out)
constexpr int kEmitForward = 3;

template <int S, bool FAST, int EMIT>
__global__ void __launch_bounds__(CPK_WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
cpecan_pairhmm_sweep(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const CpkModel &m = *a.model;
    const int stride = a.geo.rollStride;
    constexpr int R = 2 * S + 1;  // rows of the rolling buffers (Sweep::R)

    // LDS (doubles): logAdd cubics | emission tables | expectation sums | rolling buffers (FAST) | symbol strings (FAST)
    fill_cubics(lds);
    const Cubic *lg = reinterpret_cast<const Cubic *>(lds);
    double *em = lds + kLdsCubics;
    if (lane < 25) em[lane] = m.matchEm[lane];
    if (lane < 5) {
        em[25 + lane] = m.gapXEm[lane];
        em[30 + lane] = m.gapYEm[lane];
    }
    double *wt = lds + kLdsCubics + 40;
    fill_weights<S>(wt, m, a.kc, lane);
    double *eLds = lds + kLdsCubics + 40 + kLdsWeights;  // emission-expectation sums of this wave (expectation emitter)
    if (EMIT == CPECAN_EMIT_EXPECT)
        for (int i = lane; i < kExpectCopies * 80; i += CPK_WAVE) eLds[i] = 0.0;
    constexpr int kNT = S == 5 ? 13 : 9;
    double tAcc[kNT];
#pragma unroll
    for (int i = 0; i < kNT; i++) tAcc[i] = 0.0;
    double likelihood = 0.0;
    constexpr int kHeader = lds_header_doubles(EMIT);
    double *roll = FAST ? (lds + kHeader) : (a.groll + (size_t)blockIdx.x * a.geo.rollDoubles);
    Candidate *stageLds = reinterpret_cast<Candidate *>(lds + kHeader + (FAST ? (size_t)(2 * S + 1) * stride : 0));
    constexpr int kStageDoubles = lds_stage_doubles(EMIT);
    uint8_t *seqLds = reinterpret_cast<uint8_t *>(lds + kHeader + (size_t)(2 * S + 1) * stride + kStageDoubles);
    // every rolling cell starts as -inf; position 0 of each row is never written again (the guard)
    for (int i = lane; i < (2 * S + 1) * stride; i += CPK_WAVE) roll[i] = NEG_INF;
    __syncthreads();

    const size_t slot = blockIdx.x;
    for (;;) {
        // Every lane takes part in the ticket fetch (lane 0 adds 1, the others add 0; hipcc folds this into one
        // atomic per wave).  Do NOT write this as `if (lane == 0) ticket = atomicAdd(..)`: hipcc 7.2 jump-threads
        // the lane test across the loop back-edge and re-runs the readfirstlane with 63 lanes -> endless loop.
        const unsigned int ticket = atomicAdd(a.queue, lane == 0 ? 1u : 0u);
        const int tk = __builtin_amdgcn_readfirstlane((int)ticket);
        if (tk >= a.regionCount) break;
        const int r = a.regionBase + tk;

        const CpkRegion &rg = a.regions[r];
        const int lX = rg.lX, lY = rg.lY, N = lX + lY;
        const uint8_t *gx = a.symbols + rg.seqXOff, *gy = a.symbols + rg.seqYOff;
        if (FAST) {
            // stage N + bases + N of both strings into LDS, two symbols per byte (per-cell reads come from here)
            const int bx = (lX + 3) >> 1, by = (lY + 3) >> 1;
            for (int i = lane; i < bx; i += CPK_WAVE) {
                const int lo4 = gx[2 * i], hi4 = 2 * i + 1 < lX + 2 ? gx[2 * i + 1] : CPK_SYM_N;
                seqLds[i] = (uint8_t)(lo4 | (hi4 << 4));
            }
            for (int i = lane; i < by; i += CPK_WAVE) {
                const int lo4 = gy[2 * i], hi4 = 2 * i + 1 < lY + 2 ? gy[2 * i + 1] : CPK_SYM_N;
                seqLds[bx + i] = (uint8_t)(lo4 | (hi4 << 4));
            }
            roll_fence<false>();
        }
        const CpkDiag *table = a.diags + rg.diagOff;
        Sweep<S, FAST> sw{a,
                          a.kc,
                          DiagCache{table, N, 0, lane, 0, 0, 0, 0},
                          FAST ? seqLds : gx,
                          FAST ? seqLds + ((lX + 3) >> 1) : gy,
                          roll,
                          em,
                          wt,
                          lg,
                          a.ring + slot * (size_t)a.geo.ringCells * S,
                          a.cand + slot * (size_t)a.geo.fbCells * (EMIT == CPECAN_EMIT_INDEL ? 3 : 1),
                          stageLds,
                          a.cbuf + slot * (size_t)a.geo.refreshCells,
                          a.mbuf + slot * (size_t)a.geo.refreshCells,
                          a.totals + slot * (size_t)a.geo.maxRefresh,
                          stride,
                          lane,
                          lane * R,
                          N,
                          CpkDiag{},
                          CpkDiag{}};
        if (EMIT == CPECAN_EMIT_EXPECT) sw.bring = a.bring + slot * (size_t)a.geo.fbCells * S;
        constexpr int NL = EMIT == CPECAN_EMIT_INDEL ? 3 : 1;
        int count[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) count[l] = 0;
        if (EMIT == kEmitForward) {
            // getForwardProbWithBanding (pairwiseAligner.c:879-931): forward sweep over the whole matrix, then the
            // total probability of the last diagonal against the end prior; no traceback.
            double total = 0.0;  // LOG_ONE for two empty sequences (:889-891)
            if (N > 0) {
                sw.dc.load(0);
                const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
                const CpkDiag g0 = sw.dc.get(0, false);
                double *cur0 = sw.fbuf1(0);
                if (lane < S) cur0[lane] = startPrior[lane];
                roll_fence<!FAST>();
                sw.f1 = g0;
                sw.f2 = g0;
                for (int d = 1; d <= N;) {
                    sw.dc.load(d);  // table entries of diagonals d .. d+63
                    const int dEnd = d + CPK_WAVE - 1 < N ? d + CPK_WAVE - 1 : N;
                    for (; d <= dEnd; d++) {
                        if (FAST) sw.forwardStream(d, sw.dc.at(d - sw.dc.base), 0);  // nothing reads F back
                        else sw.forward(d, sw.dc.at(d - sw.dc.base), 0);
                    }
                }
                if (FAST) sw.flushTail();
                const double *endPrior = rg.raggedRight ? m.raggedEnd : m.end;
                const double *last = sw.fbuf1(N);
                const int W = sw.f1.width;
                total = NEG_INF;  // dpDiagonal_dotProduct (:513-523) over the cells of diagonal N, every lane alike
                for (int k = 0; k < W; k++) {
                    double t = last[0 + k * R] + endPrior[0];
#pragma unroll
                    for (int s = 1; s < S; s++) t = logadd(lg, t, last[s + k * R] + endPrior[s]);
                    total = logadd(lg, total, t);
                }
            }
            if (lane == 0) a.forwardOut[r] = total;
            continue;
        }
        if (N > 0) {
            sw.dc.load(0);
            // diagonal 0: the single cell (0,0) holds the start prior (pairwiseAligner.c:776-777)
            const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
            {
                const CpkDiag g0 = sw.dc.get(0, false);
                double *cur = sw.fbuf1(0);
                double *o0 = sw.ringAt(g0);
                if (lane < S) {
                    cur[lane] = startPrior[lane];
                    o0[lane] = startPrior[lane];
                }
                roll_fence<!FAST>();
                sw.f1 = g0;
                sw.f2 = g0;
            }
            int d = 1;
            int emitSeg = 0, emitFrom = a.segs[rg.segOff].tbFrom;  // the segment whose traceback emits diagonal d: the first with tbFrom >= d
            for (int si = 0; si < rg.nSeg; si++) {
                const CpkSegment sg = a.segs[rg.segOff + si];
                // Which states of F[d] the traceback will read back: the match row always (posteriors), every state on the
                // refresh points of the segment that emits d (cell dot products, pairwiseAligner.c:636-653; the schedule is
                // known up front) and on the two diagonals the forward sweep is resumed from; the indel and expectation
                // emitters read every state of every diagonal.  For the match emitter this cuts the ring stores from 8*S to
                // ~8 + 0.8*(S-1) bytes per cell.
                while (d <= sg.dTop) {
                    sw.dc.load(d);  // table entries of diagonals d .. d+63
                    const int dEnd = d + CPK_WAVE - 1 < sg.dTop ? d + CPK_WAVE - 1 : sg.dTop;
                    for (; d <= dEnd; d++) {
                        while (d > emitFrom) emitFrom = a.segs[rg.segOff + ++emitSeg].tbFrom;  // the last segment ends at N
                        const bool all = EMIT != CPECAN_EMIT_MATCH || (emitFrom - d) % CPK_REFRESH_PERIOD == 0 || d >= sg.dTop - 1;
                        if (FAST) sw.forwardStream(d, sw.dc.at(d - sw.dc.base), all ? S : 1);
                        else sw.forward(d, sw.dc.at(d - sw.dc.base), all ? S : 1);
                    }
                }
                if (FAST) sw.flushTail();  // the traceback needs every cell of dTop
                if (a.geo.debug & 2) continue;  // diagnostic: time the forward sweep alone (no traceback, no output)
                const double *endPrior = (sg.atEnd && rg.raggedRight) ? m.raggedEnd : m.end;
                int nCand[NL];
                sw.template traceback<NL, EMIT != CPECAN_EMIT_EXPECT>(sg, endPrior, (a.geo.debug & 1) ? a.dbgFb + rg.dbgCellOff : nullptr, nCand);
                roll_fence<true>();  // candidate / cbuf / mbuf stores of all lanes are complete before they are re-read
                sw.foldTotals(sg, table);
                if (a.geo.debug & 1) {
                    for (int d2 = sg.tbPrev + 1 + lane; d2 <= sg.tbFrom; d2 += CPK_WAVE)
                        a.dbgTotals[rg.dbgDiagOff + d2] = ld_self(sw.totals + (sg.tbFrom - d2) / CPK_REFRESH_PERIOD);
                }
                if (EMIT == CPECAN_EMIT_EXPECT) sw.expectations(sg, tAcc, eLds, likelihood);
#pragma unroll
                for (int l = 0; l < (EMIT == CPECAN_EMIT_EXPECT ? 0 : NL); l++) {
                    if (lane == 0) a.segStarts[(size_t)l * a.nSegsTotal + rg.segOff + si] = count[l];
                    count[l] = sw.emitMatches(sg, sw.cand + (size_t)l * a.geo.fbCells, nCand[l],
                                              a.triples + 3 * ((size_t)l * a.outTriplesPerList + rg.outOff), rg.outCap,
                                              count[l]);
                }
                if (!sg.atEnd) {
                    // the traceback reused the rolling buffers: restore F[dTop-1], F[dTop] for the forward sweep
                    const CpkDiag gTopM1 = sw.dc.get(sg.dTop - 1, false);
                    const CpkDiag gTop = sw.dc.get(sg.dTop, false);
                    sw.reloadForward(gTopM1, sg.dTop - 1);
                    sw.reloadForward(gTop, sg.dTop);
                    sw.f2 = gTopM1;
                    sw.f1 = gTop;
                }
            }
        }
#pragma unroll
        for (int l = 0; l < NL; l++)
            if (lane == 0) a.outCounts[(size_t)l * a.geo.nRegions + r] = count[l];
    }
    if (EMIT == CPECAN_EMIT_EXPECT) {
        // one partial result per resident wave: [0,25) transitions [from*S+to], [25,105) emissions, [105] likelihood
        __syncthreads();
        double *dst = a.expectOut + (size_t)blockIdx.x * 128;
        constexpr int kFrom5[13] = {0, 1, 0, 3, 0, 1, 2, 3, 4, 0, 2, 0, 4}, kTo5[13] = {1, 1, 3, 3, 0, 0, 0, 0, 0, 2, 2, 4, 4};
        constexpr int kFrom3[9] = {0, 1, 2, 0, 1, 2, 0, 2, 1}, kTo3[9] = {1, 1, 1, 0, 0, 0, 2, 2, 2};
        for (int i = lane; i < 25; i += CPK_WAVE) dst[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kNT; i++) {
            double v = tAcc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            const int idx = S == 5 ? kFrom5[i] * 5 + kTo5[i] : kFrom3[i] * 3 + kTo3[i];
            if (lane == 0) dst[idx] = v;
        }
        for (int i = lane; i < 80; i += CPK_WAVE) {
            double e = 0.0;
            for (int k = 0; k < kExpectCopies; k++) e += eLds[k * 80 + i];
            dst[25 + i] = e;
        }
        if (lane == 0) dst[105] = likelihood;
    }
}

// The per-diagonal table the sweeps read, built on the device: one thread per region walks its band with the host's
// own iterator (cpecan_band.inl; the host has already validated the anchors with it) and writes
// {x-y of the first cell, width, position in the region's forward ring, cells on earlier diagonals}.  The ring
// position follows the rule the kernels rely on: diagonals are laid end to end and never straddle the ring's end.
__global__ void __launch_bounds__(64) cpecan_build_diag_table(const CpkRegion *regions, int nRegions, const int64_t *anchors,
                                                              CpkDiag *diags, int64_t expansion, int dynamic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRegions) return;
    const CpkRegion rg = regions[i];
    CpkDiag *table = diags + rg.diagOff;
    const int64_t N = (int64_t)rg.lX + rg.lY;
    CpkBandIter it;
    cpk_band_init(&it, anchors + 3 * rg.anchorOff, rg.nAnchors, rg.lX, rg.lY, expansion, dynamic);
    int32_t cells = 0, pos = 0;
    for (int64_t d = 0; d <= N; d++) {
        int64_t lo = 0, hi = 0;
        cpk_band_next(&it, d, &lo, &hi);
        const int32_t w = (int32_t)((hi - lo) / 2 + 1);
        if (pos + w > rg.ringCap) pos = 0;
        CpkDiag e;
        e.xmyL = (int32_t)lo;
        e.width = w;
        e.ringOff = pos;
        e.cellOff = cells;
        table[d] = e;
        pos += w;
        cells += w;
    }
}

// Result compaction: the sweep leaves every region's triples in its own slice, segments in processing order.  One
// workgroup per chunk (a region's segment) copies it to its place in the compact buffer -- problems in order, regions in
// order, segments DEScending (the reference prepends each traceback's pairs, pairwiseAligner.c:1415-1417) -- and adds
// the region offset.  The host then fetches exactly the emitted triples instead of the slices' capacity.
__global__ void __launch_bounds__(256) cpecan_gather_lists(const CpkChunk *chunks, int64_t nChunks,
                                                           const int32_t *triples, int32_t *out) {
    for (int64_t c = blockIdx.x; c < nChunks; c += gridDim.x) {
        const CpkChunk ch = chunks[c];
        const int32_t *src = triples + 3 * ch.src;
        int32_t *dst = out + 3 * ch.dst;
        for (int i = threadIdx.x; i < 3 * ch.len; i += blockDim.x) {
            const int f = i % 3;
            dst[i] = src[i] + (f == 1 ? ch.dx : (f == 2 ? ch.dy : 0));
        }
    }
}



// ------------------------------------------------------------------------------------------------
// Packed kernel for narrow bands (realign-style work: diagonalExpansion 4-10, diagonals of 5-30 cells).
// One wave per region leaves most lanes idle there and pays the per-diagonal bookkeeping for a handful of cells.
// Here a wave runs G = 64 / GW regions at once: lane = (group g, cell c), every diagonal of a narrow region is one
// group of at most GW cells, so there is no loop over groups and no in-place ordering problem.  Everything that is
// wave-uniform in the sweep kernel (diagonal counter, table entries, neighbour shifts, row pointers, traceback
// schedule) is per lane here, identical within a group.  The groups of a wave move in lock-step through the same
// phases -- forward sweep of segment i, traceback of segment i, totals, emission -- each over its own diagonals; a
// group that has nothing to do in a phase idles (regions are handed out sorted by size, so neighbours are alike).
// Arithmetic: the sweep kernel's own cell functions (fwdCellsSym / bwdCellsSym), same order, bit-identical results.
// Match emitter only; symbols are read from global memory (one byte per symbol).
// ------------------------------------------------------------------------------------------------
template <int GW>
__device__ __forceinline__ float group_max_f32(float v) {
#pragma unroll
    for (int off = GW / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// LDS of the packed kernel behind the shared tables, per group: rolling buffers | 64 table entries | two symbol
// windows | candidate staging ring
constexpr int kPackChunk = 64;                  // diagonals per staged chunk
__host__ __device__ constexpr int pack_win_bytes(int gw) { return (kPackChunk + gw + 15) / 8 * 8; }  // a symbol window
__host__ __device__ constexpr int pack_rows_bytes(int S, int gw) { return (8 * (2 * S + 1) * (gw + 1) + 15) / 16 * 16; }
__host__ __device__ constexpr int pack_group_bytes(int S, int gw) {  // a multiple of 16: entries and candidates are 16-byte items
    return pack_rows_bytes(S, gw) + 16 * kPackChunk + 2 * pack_win_bytes(gw) + 16 * 2 * gw;
}

template <int S, int GW>
__global__ void __launch_bounds__(CPK_WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
cpecan_pairhmm_packed(const KArgs a) {
    constexpr int G = CPK_WAVE / GW;
    constexpr int R = 2 * S + 1;
    constexpr int kRowDoubles = R * (GW + 1);
    constexpr int kWin = pack_win_bytes(GW);
    constexpr int kStageP = 2 * GW;  // candidate staging slots per group (flushed GW at a time)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int g = lane / GW, c = lane % GW;
    const CpkModel &m = *a.model;

    fill_cubics(lds);
    const Cubic *lg = reinterpret_cast<const Cubic *>(lds);
    double *wt = lds + kLdsCubics + 40;
    fill_weights<S>(wt, m, a.kc, lane);
    uint8_t *mine = reinterpret_cast<uint8_t *>(lds + kLdsCubics + 40 + kLdsWeights) + (size_t)g * pack_group_bytes(S, GW);
    double *rows = reinterpret_cast<double *>(mine);                                   // rolling buffers
    int4 *ebuf = reinterpret_cast<int4 *>(mine + pack_rows_bytes(S, GW));              // table entries of the chunk
    uint8_t *xwin = mine + pack_rows_bytes(S, GW) + 16 * kPackChunk, *ywin = xwin + kWin;  // symbols of the chunk
    Candidate *stage = reinterpret_cast<Candidate *>(ywin + kWin);                     // candidate ring
    __syncthreads();

    // the cell functions only need the tables; every position-dependent input is passed per call
    Sweep<S, false> sw{a, a.kc, DiagCache{nullptr, 0, 0, lane, 0, 0, 0, 0}, nullptr, nullptr, rows, lds + kLdsCubics, wt, lg,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, GW + 1, lane, lane * R, 0, CpkDiag{}, CpkDiag{}};
    using SW = Sweep<S, false>;
    const unsigned long long groupBits = (GW == 64 ? ~0ull : ((1ull << GW) - 1ull)) << (g * GW);
    const unsigned long long belowMe = groupBits & ((1ull << lane) - 1ull);
    const float logThr = (float)log(a.kc.threshold);
    const double thr = a.kc.threshold;

    for (;;) {
        const unsigned int ticket = atomicAdd(a.queue, lane == 0 ? (unsigned)G : 0u);
        const int tk = __builtin_amdgcn_readfirstlane((int)ticket);
        if (tk >= a.regionCount) break;
        const bool have = tk + g < a.regionCount;
        const int r = a.regionBase + (have ? tk + g : tk);
        const CpkRegion rg = a.regions[r];
        const int N = have ? rg.lX + rg.lY : 0;
        const int nSeg = (have && N > 0) ? rg.nSeg : 0;
        const CpkDiag *table = a.diags + rg.diagOff;
        const CpkSegment *segs = a.segs + rg.segOff;
        const uint8_t *gx = a.symbols + rg.seqXOff, *gy = a.symbols + rg.seqYOff;  // padded: index p = base p-1, N at both ends
        const size_t sub = (size_t)blockIdx.x * G + g;  // scratch sub-slot of this group
        double *ring = a.ring + sub * (size_t)a.geo.ringCells * S;
        Candidate *cand = a.cand + sub * (size_t)a.geo.fbCells;
        double *cbuf = a.cbuf + sub * (size_t)a.geo.refreshCells, *mbuf = a.mbuf + sub * (size_t)a.geo.refreshCells;
        double *totals = a.totals + sub * (size_t)a.geo.maxRefresh;
        int32_t *out = a.triples + 3 * rg.outOff;
        int count = 0;

        for (int i = c; i < kRowDoubles; i += GW) rows[i] = NEG_INF;  // position 0 stays the -inf guard
        auto fbuf1 = [&](int d) { return rows + R + (d & 1) * S; };
        auto bM1 = [&](int d) { return rows + R + (d + 3) % 3; };
        auto bG1 = [&](int d) { return rows + R + 2 + (d & 1) * (S - 1); };
        auto ringAt = [&](const CpkDiag &e) { return ring + (size_t)e.ringOff * S; };
        auto unpack = [](const int4 &t) { return CpkDiag{t.x, t.y, t.z, t.w}; };
        // Stages the table entries of `cnt` (<= 64) diagonals first, first + step, ... into ebuf and the X / Y symbols
        // their cells use (shifted by `shift`: the backward step reads the symbols of (x+1, y+1)) into the two windows.
        // One global round trip per 64 diagonals instead of three per diagonal.
        auto stage_chunk = [&](bool on, int first, int step, int cnt, int shift, int &x0, int &y0) {
            for (int i = c; i < kPackChunk; i += GW) {
                int dd = first + step * (i < cnt ? i : (cnt > 0 ? cnt - 1 : 0));
                dd = dd < 0 ? 0 : (dd > N ? N : dd);
                ebuf[i] = on ? *reinterpret_cast<const int4 *>(table + dd) : int4{0, 1, 0, 0};
            }
            // both ends of the chunk bound the coordinates in between (x and y never decrease with the diagonal)
            const int dA = step > 0 ? first : first - (cnt - 1), dB = step > 0 ? first + (cnt - 1) : first;
            const CpkDiag eA = unpack(ebuf[step > 0 ? 0 : (cnt > 0 ? cnt - 1 : 0)]);
            const CpkDiag eB = unpack(ebuf[step > 0 ? (cnt > 0 ? cnt - 1 : 0) : 0]);
            const int xloA = (dA + eA.xmyL) >> 1, xloB = (dB + eB.xmyL) >> 1;
            x0 = xloA + shift;
            y0 = dA - (xloA + eA.width - 1) + shift;
            const int x1 = xloB + eB.width - 1 + shift, y1 = dB - xloB + shift;
            for (int i = c; i < kWin; i += GW) {
                const int px = x0 + i, py = y0 + i;
                xwin[i] = (on && cnt > 0 && px <= x1 && px <= rg.lX + 1) ? gx[px] : (uint8_t)CPK_SYM_N;
                ywin[i] = (on && cnt > 0 && py <= y1 && py <= rg.lY + 1) ? gy[py] : (uint8_t)CPK_SYM_N;
            }
        };

        CpkDiag e1{}, e2{};  // entries of d-1 and d-2 of the forward sweep
        if (nSeg > 0) {
            const int4 t0 = *reinterpret_cast<const int4 *>(table);
            e1 = e2 = unpack(t0);
            const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
            if (c < S) {  // diagonal 0: the single cell (0,0) holds the start prior (pairwiseAligner.c:776-777)
                fbuf1(0)[c] = startPrior[c];
                ringAt(e1)[c] = startPrior[c];
            }
        }
        int d = 1;
        const int maxSeg = wave_max_i32(nSeg);
        for (int si = 0; si < maxSeg; si++) {
            const bool segOn = si < nSeg;
            CpkSegment sg{};
            if (segOn) sg = segs[si];
            // ---------------- forward sweep up to dTop (pairwiseAligner.c:609-629) ----------------
            while (__ballot(segOn && d <= sg.dTop)) {
                const bool more = segOn && d <= sg.dTop;
                const int cnt = more ? (sg.dTop - d + 1 < kPackChunk ? sg.dTop - d + 1 : kPackChunk) : 0;
                int x0, y0;
                stage_chunk(more, d, 1, cnt, 0, x0, y0);
                for (int i = 0; i < kPackChunk; i++) {
                    if (!__ballot(i < cnt)) break;
                    const bool act = i < cnt;
                    const CpkDiag e = act ? unpack(ebuf[i]) : e1;
                    const int W = e.width;
                    const bool on = act && c < W;
                    typename SW::FwdCtx fc;
                    fc.d = d;
                    fc.xlo = (d + e.xmyL) >> 1;
                    fc.dlR = ((e.xmyL - 1 - e1.xmyL) >> 1) * R;
                    fc.w1R = e1.width * R;
                    fc.dmR = ((e.xmyL - e2.xmyL) >> 1) * R;
                    fc.w2R = d >= 2 ? e2.width * R : 0;
                    fc.p1 = fbuf1(d - 1);
                    fc.p2 = fbuf1(d - 2);
                    const int x = fc.xlo + c, y = d - x;
                    const int cX[1] = {on ? xwin[x - x0] : CPK_SYM_N}, cY[1] = {on ? ywin[y - y0] : CPK_SYM_N};
                    const int kR[1] = {c * R};
                    double v[1][S];
                    sw.template fwdCellsSym<1>(fc, cX, cY, kR, v);
                    if (on) {
                        double *cur = fbuf1(d);
                        double *o = ringAt(e);
#pragma unroll
                        for (int s = 0; s < S; s++) cur[s + c * R] = v[0][s];
#pragma unroll
                        for (int s = 0; s < S; s++) o[SW::ringIdx(W, s, c)] = v[0][s];
                    }
                    if (act) {
                        e2 = e1;
                        e1 = e;
                        d++;
                    }
                }
            }
            // ---------------- traceback of the segment (pairwiseAligner.c:796-862) ----------------
            const double *endPrior = (segOn && sg.atEnd && rg.raggedRight) ? m.raggedEnd : m.end;
            double ep[S];
#pragma unroll
            for (int s = 0; s < S; s++) ep[s] = endPrior[s];
            const int J = sg.nRefresh;
            int nCand = 0, pend = 0, head = 0;  // candidates in HBM; staged in LDS; ring position of the oldest staged
            auto flush = [&](int n) {           // the n <= GW oldest staged candidates of this group -> cand[nCand ..]
                if (c < n) cand[nCand + c] = stage[(head + c) & (kStageP - 1)];
                head = (head + n) & (kStageP - 1);
                pend -= n;
                nCand += n;
            };
            float lastMax = -__builtin_huge_valf();
            int d2 = segOn ? sg.dTop : 0;
            CpkDiag eb{}, ea{};  // entries of d2+1, d2+2
            while (__ballot(segOn && d2 > sg.tbPrev)) {
                const bool more = segOn && d2 > sg.tbPrev;
                const int cnt = more ? (d2 - sg.tbPrev < kPackChunk ? d2 - sg.tbPrev : kPackChunk) : 0;
                int x0, y0;
                stage_chunk(more, d2, -1, cnt, 1, x0, y0);
                // F.match of the chunk's first diagonal; inside the loop the next diagonal's is requested one step ahead
                double fNext = 0.0;
                {
                    const CpkDiag e = unpack(ebuf[0]);
                    if (cnt > 0 && c < e.width) fNext = ld_self(ringAt(e) + SW::ringIdx(e.width, 0, c));
                }
                for (int i = 0; i < kPackChunk; i++) {
                    if (!__ballot(i < cnt)) break;
                    const bool act = i < cnt;
                    const CpkDiag e = act ? unpack(ebuf[i]) : CpkDiag{0, 1, 0, 0};
                    const int W = e.width;
                    const bool on = act && c < W;
                    asm volatile("" : "+v"(fNext));  // the wait for the prefetched value sits here, a whole step after its load
                    const double f0 = fNext;
                    if (i + 1 < kPackChunk) {
                        const CpkDiag en = unpack(ebuf[i + 1]);
                        fNext = (i + 1 < cnt && c < en.width) ? ld_self(ringAt(en) + SW::ringIdx(en.width, 0, c)) : 0.0;
                    }
                    const bool seeded = d2 == sg.dTop;
                    const bool emit = act && d2 <= sg.tbFrom;
                    const int sinceFrom = sg.tbFrom - d2;
                    const bool refresh = emit && sinceFrom % CPK_REFRESH_PERIOD == 0;
                    const int jr = sinceFrom / CPK_REFRESH_PERIOD;
                    // the fb values of the diagonal above a refresh point are its straddle series (see Sweep::traceback)
                    const bool feeds = act && d2 - 1 > sg.tbPrev && d2 - 1 <= sg.tbFrom &&
                                       (sg.tbFrom - (d2 - 1)) % CPK_REFRESH_PERIOD == 0;
                    const int jrNext = (sg.tbFrom - (d2 - 1)) / CPK_REFRESH_PERIOD;
                    const double *fsrc = ringAt(e);
                    typename SW::BwdCtx bc;
                    bc.d2 = d2;
                    bc.xlo = (d2 + e.xmyL) >> 1;
                    bc.dbR = ((e.xmyL - 1 - eb.xmyL) >> 1) * R;
                    bc.wBR = seeded ? 0 : eb.width * R;
                    bc.daR = ((e.xmyL - ea.xmyL) >> 1) * R;
                    bc.wAR = (!seeded && d2 + 2 <= sg.dTop) ? ea.width * R : 0;
                    bc.pb = bG1(d2 + 1);
                    bc.pa = bM1(d2 + 2);
                    const int x = bc.xlo + c, y = d2 - x;
                    // symbols of the source cells (x+1, .) and (., y+1): the windows are staged one to the right
                    const int cX1[1] = {on ? xwin[x + 1 - x0] : CPK_SYM_N}, cY1[1] = {on ? ywin[y + 1 - y0] : CPK_SYM_N};
                    const int kR[1] = {c * R};
                    double v[1][S];
                    sw.template bwdCellsSym<1>(bc, cX1, cY1, kR, v);
                    if (seeded) {  // every cell of the top diagonal gets the end-state prior (:798-799)
#pragma unroll
                        for (int s = 0; s < S; s++) v[0][s] = ep[s];
                    }
                    if (on) {
                        bM1(d2)[c * R] = v[0][0];
                        double *curG = bG1(d2);
#pragma unroll
                        for (int s = 1; s < S; s++) curG[s + c * R] = v[0][s];
                    }
                    const double fbv = f0 + v[0][0];
                    if (feeds && on) mbuf[(size_t)c * J + jrNext] = fbv;
                    {
                        const float keepFrom = lastMax + logThr - kCandMargin;
                        const bool keep = on && emit && x > 0 && y > 0 && (float)fbv >= keepFrom;
                        const unsigned long long mask = __ballot(keep);
                        if (keep) {
                            Candidate cd;
                            cd.fb = fbv;
                            cd.x = x;
                            cd.y = y;
                            stage[(head + pend + __popcll(mask & belowMe)) & (kStageP - 1)] = cd;
                        }
                        pend += __popcll(mask & groupBits);
                        if (__ballot(pend >= GW)) {
                            if (pend >= GW) flush(GW);
                        }
                    }
                    if (__ballot(refresh)) {
                        // cell dot product over the states (cell_dotProduct :402-408): this lane holds its cell's B values
                        double t = fbv;
                        float fbf = -__builtin_huge_valf();
                        if (refresh && on) {
#pragma unroll
                            for (int s2 = 1; s2 < S; s2++)
                                t = logadd(lg, t, ld_self(fsrc + SW::ringIdx(W, s2, c)) + v[0][s2]);
                            cbuf[(size_t)c * J + jr] = t;
                            if (x > 0 && y > 0) fbf = (float)fbv;
                        }
                        const float diagMax = group_max_f32<GW>(fbf);
                        if (refresh) lastMax = fmaxf(diagMax, lastMax - 1.0f);
                    }
                    if (act) {
                        ea = eb;
                        eb = e;
                        d2--;
                    }
                }
            }
            if (__ballot(pend > 0)) flush(pend);
            roll_fence<true>();  // candidate / cbuf / mbuf stores of the group's lanes are visible to each other
            // ---------------- totals at the refresh points (:636-653): lane c takes points c, c + GW, ... ----------------
            for (int j0 = 0; __ballot(segOn && j0 + c < J); j0 += GW) {
                const int j = j0 + c;
                if (segOn && j < J) {
                    const int rr = sg.tbFrom - CPK_REFRESH_PERIOD * j;
                    const int Wc = table[rr].width;
                    const int Wm = rr + 1 <= sg.dTop ? table[rr + 1].width : 0;
                    double total = NEG_INF, straddle = NEG_INF;
                    for (int k = 0; k < Wc; k++) total = logadd(lg, total, ld_self(cbuf + (size_t)k * J + j));
                    for (int k = 0; k < Wm; k++) straddle = logadd(lg, straddle, ld_self(mbuf + (size_t)k * J + j));
                    if (rr + 1 <= sg.dTop) total = logadd(lg, total, straddle);
                    totals[j] = total;
                }
            }
            roll_fence<true>();
            // ---------------- thresholded posteriors from the candidates, walked backwards (:655-689) ----------------
            if (segOn && c == 0) a.segStarts[rg.segOff + si] = count;
            for (int top = nCand; __ballot(segOn && top > 0); top -= GW) {
                const int i = top - 1 - c;
                const bool valid = segOn && top > 0 && i >= 0;
                double p = 0.0;
                int x = 0, y = 0;
                if (valid) {
                    const double fbv = ld_self(&cand[i].fb);
                    const long long xy = __hip_atomic_load(reinterpret_cast<const long long *>(&cand[i].x), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                    x = (int)(xy & 0xffffffffll);
                    y = (int)(xy >> 32);
                    const double total = ld_self(totals + (sg.tbFrom - (x + y)) / CPK_REFRESH_PERIOD);
                    p = exp(fbv - total);
                }
                const bool keep = valid && p >= thr;
                const unsigned long long mask = __ballot(keep);
                if (keep) {
                    if (p > 1.0) p = 1.0;
                    const int pos = count + __popcll(mask & belowMe);
                    if (pos < rg.outCap) {
                        out[3 * (size_t)pos + 0] = (int32_t)floor(p * (double)CPECAN_PROB_1);
                        out[3 * (size_t)pos + 1] = x - 1;
                        out[3 * (size_t)pos + 2] = y - 1;
                    }
                }
                count += __popcll(mask & groupBits);
            }
            // ---------------- the traceback used the rolling buffers: restore F[dTop-1], F[dTop] ----------------
            if (segOn && !sg.atEnd) {
#pragma unroll
                for (int back = 1; back >= 0; back--) {
                    const int dd = sg.dTop - back;
                    const CpkDiag e = back ? e2 : e1;
                    if (c < e.width) {
                        const double *src = ringAt(e);
                        double *cur = fbuf1(dd);
#pragma unroll
                        for (int s = 0; s < S; s++) cur[s + c * R] = ld_self(src + SW::ringIdx(e.width, s, c));
                    }
                }
            }
        }
        if (have && c == 0) a.outCounts[r] = count;
    }
}

// ------------------------------------------------------------------------------------------------
// Consumers of the posterior lists.  Integer / order-defined arithmetic: bit-exact.
// ------------------------------------------------------------------------------------------------
constexpr int kPostReweight = 1, kPostMea = 2, kPostLeftShift = 4;  // == CPECAN_POST_*

// reweightAlignedPairs2 (impl/pairwiseAligner.c:1519-1558) + scoreByPosteriorProbability[IgnoringGaps] (:1578-1597).
// One workgroup per problem.  mass[] = PROB_1 minus the listed mass of every base of X then Y, floored at 0 when read
// (:1529-1533); a pair keeps  score - gapGamma * (massX + massY), evaluated in double and truncated towards zero (:1543).
__global__ void __launch_bounds__(256) cpecan_post_reweight(const CpkPostProblem *problems, int32_t *triples, int32_t *mass,
                                                            double gapGamma, int reweight, double *scores) {
    const CpkPostProblem pb = problems[blockIdx.x];
    int32_t *t = triples + 3 * pb.off[0];
    const int n = pb.n[0];
    __shared__ long long partial[256];
    long long sum = 0;  // exact: |score| <= 1e7 * (1 + 2 gapGamma), n < 2^31
    if (reweight && gapGamma > 0.0) {  // :1551
        int32_t *mx = mass + pb.seqOff, *my = mx + pb.lX;
        for (int i = threadIdx.x; i < pb.lX + pb.lY; i += blockDim.x) mx[i] = CPECAN_PROB_1;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            atomicSub(&mx[t[3 * i + 1]], t[3 * i]);
            atomicSub(&my[t[3 * i + 2]], t[3 * i]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const long long ux = mx[t[3 * i + 1]], uy = my[t[3 * i + 2]];
            const long long unaligned = (ux < 0 ? 0 : ux) + (uy < 0 ? 0 : uy);
            const long long w = (long long)((double)(long long)t[3 * i] - gapGamma * (double)unaligned);
            t[3 * i] = (int32_t)w;
            sum += w;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) sum += t[3 * i];
    }
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) partial[threadIdx.x] += partial[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double total = (double)partial[0];  // the reference adds int64 scores into a double: exact below 2^53
        const long long L = (long long)pb.lX + pb.lY;
        scores[3 * blockIdx.x + 0] = 100.0 * (L == 0 ? 0 : (2.0 * total) / (double)(L * CPECAN_PROB_1));
        scores[3 * blockIdx.x + 1] = 100.0 * total / ((double)n * CPECAN_PROB_1);
    }
}

// getIndelProb (:1621-1625): gap mass of `length` bases starting at `start`
__device__ __forceinline__ long long gap_mass(const long long *cum, long long start, long long length) {
    return length == 0 ? 0 : cum[start + length - 1] - (start > 0 ? cum[start - 1] : 0);
}

// getMaximalExpectedAccuracyPairwiseAlignment (:1628-1724), one LANE per problem: the chain DP walks the pairs in
// list order with a data-dependent walk back.  gapGamma is the float of PairwiseAlignmentParameters, so
// `int64 * gapGamma` and `int64 + that` are float arithmetic, `int64 + double + float` is double truncated to int64.
__global__ void __launch_bounds__(64) cpecan_post_mea(const CpkPostProblem *problems, int64_t nProblems,
                                                      const int32_t *triples, long long *cum, double *best, int32_t *prev,
                                                      uint8_t *record, float gapGamma, int32_t *meaOut, int32_t *counts,
                                                      double *scores) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nProblems) return;
    const CpkPostProblem pb = problems[p];
    const int32_t *pairs = triples + 3 * pb.off[0], *gx = triples + 3 * pb.off[1], *gy = triples + 3 * pb.off[2];
    const int n = pb.n[0];
    const long long lX = pb.lX, lY = pb.lY;
    long long *cx = cum + pb.seqOff, *cy = cx + pb.lX;  // getCumulativeGapProbs (:1603-1619)
    for (long long i = 0; i < lX + lY; i++) cx[i] = 0;
    for (int i = 0; i < pb.n[1]; i++) cx[gx[3 * i + 1]] += gx[3 * i];
    for (int i = 0; i < pb.n[2]; i++) cy[gy[3 * i + 2]] += gy[3 * i];
    for (long long i = 1; i < lX; i++) cx[i] += cx[i - 1];
    for (long long i = 1; i < lY; i++) cy[i] += cy[i - 1];
    double *bs = best + pb.chainOff;
    int32_t *pv = prev + pb.chainOff;
    uint8_t *rec = record + pb.chainOff;
    double top = 0;
    for (int i = 0; i <= n; i++) {
        long long w, x, y;
        if (i == n) {  // sentinel behind both sequences (:1652-1654)
            w = 0;
            x = lX;
            y = lY;
        } else {
            w = pairs[3 * i];
            x = pairs[3 * i + 1];
            y = pairs[3 * i + 2];
        }
        double score = (double)((float)w + (float)(gap_mass(cx, 0, x) + gap_mass(cy, 0, y)) * gapGamma);  // :1660-1661
        int from = -1;
        for (int j = i - 1; j >= 0; j--) {
            const long long x2 = pairs[3 * j + 1], y2 = pairs[3 * j + 2];
            if (x2 < x && y2 < y) {
                const float g = (float)(gap_mass(cx, x2 + 1, x - x2 - 1) + gap_mass(cy, y2 + 1, y - y2 - 1)) * gapGamma;
                const long long sc = (long long)(((double)w + bs[j]) + (double)g);  // :1673-1675
                if ((double)sc > score) {
                    score = (double)sc;
                    from = j;
                }
                if (rec[j]) break;  // :1685
            }
        }
        pv[i] = from;
        bs[i] = score;
        const float tail = (float)((x < lX ? gap_mass(cx, x + 1, lX - x - 1) : 0) + (y < lY ? gap_mass(cy, y + 1, lY - y - 1) : 0)) * gapGamma;
        const double sc = score + (double)tail;  // :1695-1696
        rec[i] = 0;
        if (sc >= top) {
            top = sc;
            rec[i] = 1;
        }
    }
    int count = 0;
    for (int i = pv[n]; i >= 0; i = pv[i]) count++;
    int32_t *out = meaOut + 3 * pb.meaOut;
    int at = count;
    for (int i = pv[n]; i >= 0; i = pv[i]) {  // back to front == built reversed, then flipped (:1714)
        at--;
        out[3 * at] = pairs[3 * i];
        out[3 * at + 1] = pairs[3 * i + 1];
        out[3 * at + 2] = pairs[3 * i + 2];
    }
    counts[2 * p] = count;
    scores[3 * p + 2] = top;
}

// LEFT_SHIFT without MEA: list 0 is the chain to shift; put it where the MEA stage would have put its alignment.
__global__ void __launch_bounds__(256) cpecan_post_copy_chain(const CpkPostProblem *problems, const int32_t *triples,
                                                              int32_t *meaOut, int32_t *counts) {
    const CpkPostProblem pb = problems[blockIdx.x];
    const int32_t *src = triples + 3 * pb.off[0];
    int32_t *dst = meaOut + 3 * pb.meaOut;
    for (int i = threadIdx.x; i < 3 * pb.n[0]; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) counts[2 * blockIdx.x] = pb.n[0];
}

// leftShiftAlignment (:1726-1762), one lane per problem, on the MEA alignment.  chars: raw upper-case sequences.
__global__ void __launch_bounds__(64) cpecan_post_left_shift(const CpkPostProblem *problems, int64_t nProblems,
                                                             const int32_t *mea, const uint8_t *chars, int32_t *shiftOut,
                                                             int32_t *counts) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nProblems) return;
    const CpkPostProblem pb = problems[p];
    const int32_t *pairs = mea + 3 * pb.meaOut;
    const int n = counts[2 * p];
    const uint8_t *sX = chars + pb.charX, *sY = chars + pb.charY;
    int32_t *out = shiftOut + 3 * pb.shiftOut;
    int count = 0;
    int x = pb.lX, y = pb.lY;
    for (int i = n - 1; i >= 0; i--) {
        const int w = pairs[3 * i], x2 = pairs[3 * i + 1], y2 = pairs[3 * i + 2];
        while ((x - x2 > 1 || y - y2 > 1) && sX[x - 1] == sY[y - 1]) {  // :1737-1744
            out[3 * count] = w;
            out[3 * count + 1] = x - 1;
            out[3 * count + 2] = y - 1;
            count++;
            x--;
            y--;
            if (x2 == x || y2 == y) break;
        }
        if (x2 < x && y2 < y) {
            out[3 * count] = w;
            out[3 * count + 1] = x2;
            out[3 * count + 2] = y2;
            count++;
            x = x2;
            y = y2;
        }
    }
    const int first = n > 0 ? pairs[0] : 1;  // :1754
    while (x > 0 && y > 0 && sX[x - 1] == sY[y - 1]) {
        out[3 * count] = first;
        out[3 * count + 1] = x - 1;
        out[3 * count + 2] = y - 1;
        count++;
        x--;
        y--;
    }
    for (int a = 0, b = count - 1; a < b; a++, b--)  // :1759
        for (int f = 0; f < 3; f++) {
            const int32_t t = out[3 * a + f];
            out[3 * a + f] = out[3 * b + f];
            out[3 * b + f] = t;
        }
    counts[2 * p + 1] = count;
}

// ------------------------------------------------------------------------------------------------
// host side of the HIP TU: memory, launch, timing
// ------------------------------------------------------------------------------------------------
struct CpkDevice {
    int device = 0;
    int numCUs = 0;
    CpkGeometry geo{};
    KConsts kc{};
    int nLists = 1;
    int64_t nSegs = 0, nDiags = 0;
    int64_t outTriplesPerList = 0;
    int64_t dbgCells = 0, dbgDiags = 0;
    int slots = 0;        // waves of the sweep kernel (wide regions)
    size_t ldsBytes = 0;
    int pSlots[3] = {0, 0, 0};  // waves of the packed kernel per class (groups of 8 / 16 / 32 lanes)
    size_t pLdsBytes[3] = {0, 0, 0};
    // device buffers
    CpkRegion *dRegions = nullptr;
    CpkDiag *dDiags = nullptr;
    CpkSegment *dSegs = nullptr;
    uint8_t *dSymbols = nullptr;
    CpkModel *dModel = nullptr;
    double *dRing = nullptr; Candidate *dCand = nullptr; double *dForward = nullptr, *dExpect = nullptr; double *dC = nullptr, *dM = nullptr, *dTotals = nullptr, *dGroll = nullptr, *dBring = nullptr;
    int32_t *dCounts = nullptr, *dSegStarts = nullptr, *dTriples = nullptr;
    int32_t *dCompact = nullptr; CpkChunk *dChunks = nullptr; int64_t compactCap = 0, chunkCap = 0;
    unsigned int *dQueue = nullptr;
    double *dDbgFb = nullptr, *dDbgTotals = nullptr;
    int64_t bytes = 0;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    hipStream_t lastStream = nullptr;
    // the packed classes run beside the sweep kernel on streams of their own (fork / join around cpk_device_run)
    hipStream_t sideStream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t sideDone[3] = {nullptr, nullptr, nullptr};
    int64_t subBase[3] = {0, 0, 0};  // first scratch sub-slot of each packed class (behind the sweep kernel's slots)
    bool ran = false;
};

extern "C" int cpk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int cpk_device_create(CpkDevice **out, int device) {
    int n = cpk_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        cpk_set_error("no usable HIP device (count=%d, requested=%d): the HIP path has no CPU fallback", n, device);
        return CPECAN_ENODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    CpkDevice *d = new CpkDevice();
    d->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    d->numCUs = prop.multiProcessorCount;
    HIP_TRY(hipEventCreate(&d->evStart));
    HIP_TRY(hipEventCreate(&d->evStop));
    for (int k = 0; k < 3; k++) {
        HIP_TRY(hipStreamCreateWithFlags(&d->sideStream[k], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&d->sideDone[k], hipEventDisableTiming));
    }
    *out = d;
    return CPECAN_OK;
}

static void free_all(CpkDevice *d) {
    void *ptrs[] = {d->dRegions, d->dDiags, d->dSegs, d->dSymbols, d->dModel, d->dRing, d->dCand, d->dC, d->dM,
                    d->dTotals, d->dGroll, d->dBring, d->dCounts, d->dSegStarts, d->dTriples, d->dQueue, d->dDbgFb, d->dDbgTotals, d->dForward, d->dExpect,
                    d->dCompact, d->dChunks};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    d->dRegions = nullptr; d->dDiags = nullptr; d->dSegs = nullptr; d->dSymbols = nullptr; d->dModel = nullptr;
    d->dRing = d->dC = d->dM = d->dTotals = d->dGroll = d->dBring = nullptr;
    d->dCand = nullptr;
    d->dForward = nullptr;
    d->dExpect = nullptr;
    d->dCounts = d->dSegStarts = d->dTriples = nullptr;
    d->dCompact = nullptr; d->dChunks = nullptr; d->compactCap = d->chunkCap = 0;
    d->dQueue = nullptr;
    d->dDbgFb = d->dDbgTotals = nullptr;
    d->bytes = 0;
}

extern "C" void cpk_device_destroy(CpkDevice *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    free_all(d);
    if (d->evStart) (void)hipEventDestroy(d->evStart);
    if (d->evStop) (void)hipEventDestroy(d->evStop);
    for (int k = 0; k < 3; k++) {
        if (d->sideStream[k]) (void)hipStreamDestroy(d->sideStream[k]);
        if (d->sideDone[k]) (void)hipEventDestroy(d->sideDone[k]);
    }
    delete d;
}

template <typename T>
static int dev_alloc(CpkDevice *d, T **p, size_t count) {
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc((void **)p, bytes));
    d->bytes += (int64_t)bytes;
    return CPECAN_OK;
}

using KernelFn = void (*)(const KArgs);

static KernelFn pick_packed_kernel(const CpkGeometry &g, int cls) {  // class k: groups of 8 << k lanes
    if (g.emit != CPECAN_EMIT_MATCH) return nullptr;
    const bool five = g.nStates == 5;
    switch (cls) {
        case 0: return five ? cpecan_pairhmm_packed<5, 8> : cpecan_pairhmm_packed<3, 8>;
        case 1: return five ? cpecan_pairhmm_packed<5, 16> : cpecan_pairhmm_packed<3, 16>;
        case 2: return five ? cpecan_pairhmm_packed<5, 32> : cpecan_pairhmm_packed<3, 32>;
    }
    return nullptr;
}

static KernelFn pick_kernel(const CpkGeometry &g) {
    const bool fast = !g.useGlobalRoll;  // second template argument = FAST (LDS rolling buffers + LDS symbol strings)
#define CPK_PICK(E)                                                                                          \
    if (g.emit == (E)) {                                                                                     \
        if (g.nStates == 5) return fast ? cpecan_pairhmm_sweep<5, true, (E)> : cpecan_pairhmm_sweep<5, false, (E)>; \
        return fast ? cpecan_pairhmm_sweep<3, true, (E)> : cpecan_pairhmm_sweep<3, false, (E)>;              \
    }
    CPK_PICK(CPECAN_EMIT_MATCH)
    CPK_PICK(CPECAN_EMIT_INDEL)
    CPK_PICK(CPECAN_EMIT_EXPECT)
    CPK_PICK(kEmitForward)
#undef CPK_PICK
    return nullptr;
}

extern "C" int cpk_device_upload(CpkDevice *d, const CpkGeometry *geo, const CpkModel *model, const CpkRegion *regions,
                                 const int64_t *anchors, int64_t nAnchors, int64_t nDiags, int64_t expansion, int dynamic,
                                 const CpkSegment *segs, int64_t nSegs, const uint8_t *symbols, int64_t nSymbolBytes,
                                 int64_t outTriplesPerList, int nLists, int64_t dbgCells, int64_t dbgDiags,
                                 double *h2dMs) {
    HIP_TRY(hipSetDevice(d->device));
    free_all(d);
    d->geo = *geo;
    d->kc = KConsts{model->matchContinue, model->matchFromShortX, model->matchFromShortY, model->matchFromLongX,
                    model->matchFromLongY, model->shortOpenX, model->shortOpenY, model->shortExtendX,
                    model->shortExtendY, model->shortSwitchToX, model->shortSwitchToY, model->longOpenX,
                    model->longOpenY, model->longExtendX, model->longExtendY, model->threshold};
    d->nLists = nLists;
    d->nSegs = nSegs;
    d->nDiags = nDiags;
    d->outTriplesPerList = outTriplesPerList;
    d->dbgCells = dbgCells;
    d->dbgDiags = dbgDiags;
    d->ran = false;
    const int S = geo->nStates;

    // LDS: 40 doubles of emission tables + (fast path) three rolling buffers + both padded symbol strings
    d->ldsBytes = sizeof(double) * (lds_header_doubles(geo->emit) + lds_stage_doubles(geo->emit));
    if (!geo->useGlobalRoll)
        d->ldsBytes += sizeof(double) * (size_t)(2 * S + 1) * geo->rollStride + (size_t)((geo->seqLdsBytes + 15) / 16 * 16);
    KernelFn fn = pick_kernel(*geo);
    if (!fn) {
        cpk_set_error("no kernel for emitter %d", geo->emit);
        return CPECAN_EINVAL;
    }
    if (d->ldsBytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->ldsBytes));
    }
    // Resident single-wave workgroups per CU.  hipOccupancyMaxActiveBlocksPerMultiprocessor answers 3 for this
    // 64-thread kernel (it reports waves per SIMD), so the bound is computed from the register file and LDS
    // directly (MI355X_MICROARCH.md: 512 VGPRs per lane per SIMD in granules of 8, 4 SIMDs, 32 waves, 160 KiB LDS).
    // Over-estimating is harmless: surplus workgroups simply queue, every wave exits when the work queue is empty.
    hipFuncAttributes attr;
    HIP_TRY(hipFuncGetAttributes(&attr, (const void *)fn));
    const int vgprAlloc = ((attr.numRegs > 0 ? attr.numRegs : 128) + 7) / 8 * 8;
    int wavesPerSimd = 512 / vgprAlloc;
    if (wavesPerSimd > 8) wavesPerSimd = 8;
    if (wavesPerSimd < 1) wavesPerSimd = 1;
    int perCU = 4 * wavesPerSimd;
    const size_t ldsTotal = d->ldsBytes + (size_t)attr.sharedSizeBytes;
    const int byLds = (int)((160 * 1024) / (ldsTotal ? ldsTotal : 1));
    if (byLds < perCU) perCU = byLds;
    if (perCU > 32) perCU = 32;
    if (const char *cap = getenv("CPECAN_MAX_WAVES_PER_CU")) {  // tuning/diagnostic knob
        const int c = atoi(cap);
        if (c >= 1 && c < perCU) perCU = c;
    }
    if (perCU < 1) {
        cpk_set_error("kernel does not fit on a CU (LDS %zu bytes)", d->ldsBytes);
        return CPECAN_EHIP;
    }
    const int64_t nWide = geo->nRegions - geo->nPacked[0] - geo->nPacked[1] - geo->nPacked[2];
    int64_t slots = (int64_t)perCU * d->numCUs;
    if (slots > nWide) slots = nWide;
    if (slots < 1) slots = 1;
    if (nWide > 0) {
        // Even out the rounds: with R = ceil(regions / slots) rounds, ceil(regions / R) waves do the same work in the
        // same number of rounds with fewer waves competing per SIMD (10 000 equal pairs: 1667 waves x 6 pairs instead
        // of 1792 waves of which 1044 do 6 and 748 do 5).
        const int64_t rounds = (nWide + slots - 1) / slots;
        const int64_t even = (nWide + rounds - 1) / rounds;
        if (even >= 1 && even < slots) slots = even;
    }
    d->slots = (int)slots;
    // the packed kernel's waves per class (narrow regions, 64 / GW per wave) and their scratch sub-slots
    int64_t subSlots[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        d->pSlots[k] = 0;
        if (geo->nPacked[k] <= 0) continue;
        KernelFn pfn = pick_packed_kernel(*geo, k);
        if (!pfn) {
            cpk_set_error("no packed kernel for emitter %d", geo->emit);
            return CPECAN_EINVAL;
        }
        const int GW = 8 << k, G = CPK_WAVE / GW;
        d->pLdsBytes[k] = sizeof(double) * (size_t)(kLdsCubics + 40 + kLdsWeights) + (size_t)G * pack_group_bytes(S, GW);
        hipFuncAttributes pattr;
        HIP_TRY(hipFuncGetAttributes(&pattr, (const void *)pfn));
        const int pv = ((pattr.numRegs > 0 ? pattr.numRegs : 128) + 7) / 8 * 8;
        int pPerSimd = 512 / pv;
        if (pPerSimd > 8) pPerSimd = 8;
        if (pPerSimd < 1) pPerSimd = 1;
        int pPerCU = 4 * pPerSimd;
        const int pByLds = (int)((160 * 1024) / (d->pLdsBytes[k] + (size_t)pattr.sharedSizeBytes));
        if (pByLds < pPerCU) pPerCU = pByLds;
        if (const char *cap = getenv("CPECAN_MAX_WAVES_PER_CU")) {
            const int c = atoi(cap);
            if (c >= 1 && c < pPerCU) pPerCU = c;
        }
        int64_t pSlots = (int64_t)pPerCU * d->numCUs;
        const int64_t wavesNeeded = (geo->nPacked[k] + G - 1) / G;
        if (pSlots > wavesNeeded) pSlots = wavesNeeded;
        d->pSlots[k] = (int)pSlots;
        subSlots[k] = pSlots * G;
    }
    // scratch per slot: the launches run side by side, each class has its own part of every buffer behind the sweep's
    auto scratch = [&](int64_t wide, const int64_t (&packed)[3]) {
        int64_t all = nWide > 0 ? slots * wide : 0;
        for (int k = 0; k < 3; k++) all += subSlots[k] * packed[k];
        return (size_t)all;
    };
    for (int k = 0; k < 3; k++) d->subBase[k] = subSlots[k];  // sub-slot counts, turned into element offsets at launch
    const int64_t pRing[3] = {geo->pRingCells[0] * S, geo->pRingCells[1] * S, geo->pRingCells[2] * S};
    const int64_t pRefresh[3] = {(int64_t)8 * geo->pMaxRefresh[0], (int64_t)16 * geo->pMaxRefresh[1], (int64_t)32 * geo->pMaxRefresh[2]};
    const int64_t pTotals[3] = {geo->pMaxRefresh[0], geo->pMaxRefresh[1], geo->pMaxRefresh[2]};

    if (int rc = dev_alloc(d, &d->dRegions, (size_t)geo->nRegions)) return rc;
    if (int rc = dev_alloc(d, &d->dDiags, (size_t)nDiags)) return rc;
    if (int rc = dev_alloc(d, &d->dSegs, (size_t)nSegs)) return rc;
    if (int rc = dev_alloc(d, &d->dSymbols, (size_t)nSymbolBytes)) return rc;
    if (int rc = dev_alloc(d, &d->dModel, 1)) return rc;
    if (int rc = dev_alloc(d, &d->dRing, scratch(geo->ringCells * S, pRing))) return rc;
    if (int rc = dev_alloc(d, &d->dCand, scratch(geo->fbCells * (geo->emit == CPECAN_EMIT_INDEL ? 3 : 1), geo->pFbCells))) return rc;
    if (int rc = dev_alloc(d, &d->dForward, (size_t)geo->nRegions)) return rc;
    if (int rc = dev_alloc(d, &d->dExpect, (size_t)slots * 128)) return rc;
    if (int rc = dev_alloc(d, &d->dC, scratch(geo->refreshCells, pRefresh))) return rc;
    if (int rc = dev_alloc(d, &d->dM, scratch(geo->refreshCells, pRefresh))) return rc;
    if (int rc = dev_alloc(d, &d->dTotals, scratch(geo->maxRefresh, pTotals))) return rc;
    if (geo->useGlobalRoll)
        if (int rc = dev_alloc(d, &d->dGroll, (size_t)slots * geo->rollDoubles)) return rc;
    if (int rc = dev_alloc(d, &d->dCounts, (size_t)nLists * geo->nRegions)) return rc;
    if (int rc = dev_alloc(d, &d->dSegStarts, (size_t)nLists * nSegs)) return rc;
    if (int rc = dev_alloc(d, &d->dTriples, (size_t)nLists * outTriplesPerList * 3)) return rc;
    if (geo->emit == CPECAN_EMIT_EXPECT)
        if (int rc = dev_alloc(d, &d->dBring, (size_t)slots * geo->fbCells * S)) return rc;
    if (int rc = dev_alloc(d, &d->dQueue, 4)) return rc;
    HIP_TRY(hipMemset(d->dCounts, 0, sizeof(int32_t) * (size_t)nLists * geo->nRegions));
    HIP_TRY(hipMemset(d->dSegStarts, 0, sizeof(int32_t) * (size_t)nLists * (nSegs ? nSegs : 1)));
    if (geo->debug) {
        if (int rc = dev_alloc(d, &d->dDbgFb, (size_t)dbgCells)) return rc;
        if (int rc = dev_alloc(d, &d->dDbgTotals, (size_t)dbgDiags)) return rc;
        HIP_TRY(hipMemset(d->dDbgFb, 0xff, sizeof(double) * (size_t)dbgCells));      // NaN pattern
        HIP_TRY(hipMemset(d->dDbgTotals, 0xff, sizeof(double) * (size_t)dbgDiags));
    }

    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    HIP_TRY(hipMemcpy(d->dRegions, regions, sizeof(CpkRegion) * (size_t)geo->nRegions, hipMemcpyHostToDevice));
    {
        // anchors -> per-diagonal table, on the device (the anchors are only needed for this)
        int64_t *dAnchors = nullptr;
        HIP_TRY(hipMalloc((void **)&dAnchors, sizeof(int64_t) * 3 * (size_t)(nAnchors > 0 ? nAnchors : 1)));
        if (nAnchors > 0)
            HIP_TRY(hipMemcpy(dAnchors, anchors, sizeof(int64_t) * 3 * (size_t)nAnchors, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cpecan_build_diag_table, dim3((unsigned)((geo->nRegions + 63) / 64)), dim3(64), 0, nullptr,
                           d->dRegions, geo->nRegions, dAnchors, d->dDiags, expansion, dynamic);
        const hipError_t launched = hipGetLastError();
        const hipError_t done = hipDeviceSynchronize();
        (void)hipFree(dAnchors);
        HIP_TRY(launched);
        HIP_TRY(done);
    }
    HIP_TRY(hipMemcpy(d->dSegs, segs, sizeof(CpkSegment) * (size_t)nSegs, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->dSymbols, symbols, (size_t)nSymbolBytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->dModel, model, sizeof(CpkModel), hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (h2dMs) *h2dMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return CPECAN_OK;
}

extern "C" int cpk_device_update_regions(CpkDevice *d, const CpkRegion *regions, int64_t outTriplesPerList) {
    HIP_TRY(hipSetDevice(d->device));
    if (outTriplesPerList != d->outTriplesPerList) {
        if (d->dTriples) {
            (void)hipFree(d->dTriples);
            d->bytes -= (int64_t)sizeof(int32_t) * d->nLists * d->outTriplesPerList * 3;
            d->dTriples = nullptr;
        }
        d->outTriplesPerList = outTriplesPerList;
        if (int rc = dev_alloc(d, &d->dTriples, (size_t)d->nLists * outTriplesPerList * 3)) return rc;
    }
    HIP_TRY(hipMemcpy(d->dRegions, regions, sizeof(CpkRegion) * (size_t)d->geo.nRegions, hipMemcpyHostToDevice));
    return CPECAN_OK;
}

extern "C" int cpk_device_run(CpkDevice *d, void *stream) {
    HIP_TRY(hipSetDevice(d->device));
    hipStream_t st = (hipStream_t)stream;
    KArgs a{};
    a.kc = d->kc;
    a.regions = d->dRegions;
    a.diags = d->dDiags;
    a.segs = d->dSegs;
    a.symbols = d->dSymbols;
    a.model = d->dModel;
    a.geo = d->geo;
    if (const char *skip = getenv("CPECAN_DEBUG_SKIP")) a.geo.debug |= (atoi(skip) & 6);  // diagnostic phase bisection
    a.ring = d->dRing;
    a.cand = d->dCand;
    a.cbuf = d->dC;
    a.mbuf = d->dM;
    a.totals = d->dTotals;
    a.groll = d->dGroll;
    a.bring = d->dBring;
    a.outCounts = d->dCounts;
    a.segStarts = d->dSegStarts;
    a.triples = d->dTriples;
    a.outTriplesPerList = d->outTriplesPerList;
    a.nSegsTotal = d->nSegs;
    a.queue = d->dQueue;
    a.forwardOut = d->dForward;
    a.expectOut = d->dExpect;
    a.dbgFb = d->dDbgFb;
    a.dbgTotals = d->dDbgTotals;
    HIP_TRY(hipMemsetAsync(d->dQueue, 0, 4 * sizeof(unsigned int), st));
    KernelFn fn = pick_kernel(d->geo);
    HIP_TRY(hipEventRecord(d->evStart, st));
    int base = 0;
    const int64_t nWideRun = d->geo.nRegions - d->geo.nPacked[0] - d->geo.nPacked[1] - d->geo.nPacked[2];
    const int SS = d->geo.nStates;
    // element offsets of each class's scratch inside the shared buffers: [sweep slots][class 0][class 1][class 2]
    int64_t oRing = nWideRun > 0 ? (int64_t)d->slots * d->geo.ringCells * SS : 0;
    int64_t oCand = nWideRun > 0 ? (int64_t)d->slots * d->geo.fbCells * (d->geo.emit == CPECAN_EMIT_INDEL ? 3 : 1) : 0;
    int64_t oRef = nWideRun > 0 ? (int64_t)d->slots * d->geo.refreshCells : 0;
    int64_t oTot = nWideRun > 0 ? (int64_t)d->slots * d->geo.maxRefresh : 0;
    for (int k = 0; k < 3; k++) {  // narrow regions, class by class: several to a wave, on a stream of their own
        if (d->geo.nPacked[k] <= 0) continue;
        KArgs p = a;
        p.regionBase = base;
        p.regionCount = d->geo.nPacked[k];
        p.geo.ringCells = d->geo.pRingCells[k];
        p.geo.fbCells = d->geo.pFbCells[k];
        p.geo.refreshCells = (int64_t)(8 << k) * d->geo.pMaxRefresh[k];
        p.geo.maxRefresh = d->geo.pMaxRefresh[k];
        p.ring = d->dRing + oRing;
        p.cand = d->dCand + oCand;
        p.cbuf = d->dC + oRef;
        p.mbuf = d->dM + oRef;
        p.totals = d->dTotals + oTot;
        p.queue = d->dQueue + 1 + k;
        HIP_TRY(hipStreamWaitEvent(d->sideStream[k], d->evStart, 0));
        hipLaunchKernelGGL(pick_packed_kernel(d->geo, k), dim3((unsigned)d->pSlots[k]), dim3(CPK_WAVE), d->pLdsBytes[k],
                           d->sideStream[k], p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(d->sideDone[k], d->sideStream[k]));
        const int64_t subs = d->subBase[k];
        oRing += subs * p.geo.ringCells * SS;
        oCand += subs * p.geo.fbCells;
        oRef += subs * p.geo.refreshCells;
        oTot += subs * p.geo.maxRefresh;
        base += d->geo.nPacked[k];
    }
    a.regionBase = base;
    a.regionCount = d->geo.nRegions - base;
    if (a.regionCount > 0) {
        hipLaunchKernelGGL(fn, dim3((unsigned)d->slots), dim3(CPK_WAVE), d->ldsBytes, st, a);
        HIP_TRY(hipGetLastError());
    }
    for (int k = 0; k < 3; k++)  // join: the caller's stream continues when every class is done
        if (d->geo.nPacked[k] > 0) HIP_TRY(hipStreamWaitEvent(st, d->sideDone[k], 0));
    HIP_TRY(hipEventRecord(d->evStop, st));
    d->lastStream = st;
    d->ran = true;
    return CPECAN_OK;
}

extern "C" int cpk_device_download(CpkDevice *d, int32_t *counts, int32_t *segStarts, double *expect, double *kernelMs,
                                   double *d2hMs) {
    HIP_TRY(hipSetDevice(d->device));
    if (!d->ran) {
        cpk_set_error("download before run");
        return CPECAN_ESTATE;
    }
    HIP_TRY(hipStreamSynchronize(d->lastStream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->evStart, d->evStop));
    if (kernelMs) *kernelMs = ms;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    HIP_TRY(hipMemcpy(counts, d->dCounts, sizeof(int32_t) * (size_t)d->nLists * d->geo.nRegions, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(segStarts, d->dSegStarts, sizeof(int32_t) * (size_t)d->nLists * d->nSegs, hipMemcpyDeviceToHost));
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (d2hMs) *d2hMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (expect && d->geo.emit == kEmitForward)
        HIP_TRY(hipMemcpy(expect, d->dForward, sizeof(double) * (size_t)d->geo.nRegions, hipMemcpyDeviceToHost));
    if (expect && d->geo.emit == CPECAN_EMIT_EXPECT) {
        // sum the per-wave partials (every launched wave wrote its 106 values, zeros included)
        std::vector<double> part((size_t)d->slots * 128);
        HIP_TRY(hipMemcpy(part.data(), d->dExpect, sizeof(double) * part.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < 106; i++) expect[i] = 0.0;
        for (int w = 0; w < d->slots; w++)
            for (int i = 0; i < 106; i++) expect[i] += part[(size_t)w * 128 + i];
    }
    return CPECAN_OK;
}

extern "C" int cpk_device_gather(CpkDevice *d, const CpkChunk *chunks, int64_t nChunks, int64_t total) {
    HIP_TRY(hipSetDevice(d->device));
    if (nChunks <= 0 || total <= 0) return CPECAN_OK;
    if (nChunks > d->chunkCap) {
        if (d->dChunks) (void)hipFree(d->dChunks);
        d->dChunks = nullptr;
        HIP_TRY(hipMalloc((void **)&d->dChunks, sizeof(CpkChunk) * (size_t)nChunks));
        d->chunkCap = nChunks;
    }
    if (total > d->compactCap) {
        if (d->dCompact) (void)hipFree(d->dCompact);
        d->dCompact = nullptr;
        HIP_TRY(hipMalloc((void **)&d->dCompact, sizeof(int32_t) * 3 * (size_t)total));
        d->compactCap = total;
    }
    HIP_TRY(hipMemcpy(d->dChunks, chunks, sizeof(CpkChunk) * (size_t)nChunks, hipMemcpyHostToDevice));
    const int64_t blocks = nChunks < 16384 ? nChunks : 16384;
    hipLaunchKernelGGL(cpecan_gather_lists, dim3((unsigned)blocks), dim3(256), 0, nullptr, d->dChunks, nChunks, d->dTriples,
                       d->dCompact);
    HIP_TRY(hipGetLastError());
    return CPECAN_OK;
}

extern "C" int cpk_device_fetch(CpkDevice *d, int32_t *hostOut, int64_t total, double *d2hMs) {
    HIP_TRY(hipSetDevice(d->device));
    if (total <= 0) return CPECAN_OK;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    HIP_TRY(hipMemcpy(hostOut, d->dCompact, sizeof(int32_t) * 3 * (size_t)total, hipMemcpyDeviceToHost));
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (d2hMs) *d2hMs += ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return CPECAN_OK;
}

// The consumers on a device-resident triple buffer.  Scratch lives for the duration of the call.
namespace {
struct PostScratch {
    std::vector<void *> ptrs;
    ~PostScratch() {
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    }
    template <typename T>
    int alloc(T **out, size_t count) {
        void *p = nullptr;
        if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) {
            cpk_set_error("out of device memory in the list consumers");
            return CPECAN_ENOMEM;
        }
        ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return CPECAN_OK;
    }
};
}  // namespace

static int post_core(int32_t *dTriples, const CpkPostJob *job) {
    const int64_t nP = job->nProblems;
    if (nP <= 0) return CPECAN_OK;
    PostScratch sc;
    CpkPostProblem *dProblems = nullptr;
    double *dScores = nullptr;
    int32_t *dCounts = nullptr;
    if (int rc = sc.alloc(&dProblems, (size_t)nP)) return rc;
    if (int rc = sc.alloc(&dScores, (size_t)nP * 3)) return rc;
    if (int rc = sc.alloc(&dCounts, (size_t)nP * 2)) return rc;
    HIP_TRY(hipMemcpy(dProblems, job->problems, sizeof(CpkPostProblem) * (size_t)nP, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dScores, 0, sizeof(double) * (size_t)nP * 3));
    HIP_TRY(hipMemset(dCounts, 0, sizeof(int32_t) * (size_t)nP * 2));
    {
        int32_t *dMass = nullptr;
        const bool rw = (job->flags & kPostReweight) != 0;
        if (int rc = sc.alloc(&dMass, rw ? (size_t)job->seqSlots : 1)) return rc;
        hipLaunchKernelGGL(cpecan_post_reweight, dim3((unsigned)nP), dim3(256), 0, nullptr, dProblems, dTriples, dMass,
                           job->gapGamma, rw ? 1 : 0, dScores);
        HIP_TRY(hipGetLastError());
    }
    int32_t *dMea = nullptr, *dShift = nullptr;
    const unsigned laneBlocks = (unsigned)((nP + 63) / 64);
    if (job->flags & (kPostMea | kPostLeftShift))
        if (int rc = sc.alloc(&dMea, (size_t)job->meaCap * 3)) return rc;
    if (job->flags & kPostMea) {
        long long *dCum = nullptr;
        double *dBest = nullptr;
        int32_t *dPrev = nullptr;
        uint8_t *dRecord = nullptr;
        if (int rc = sc.alloc(&dCum, (size_t)job->seqSlots)) return rc;
        if (int rc = sc.alloc(&dBest, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dPrev, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dRecord, (size_t)job->chainSlots)) return rc;
        hipLaunchKernelGGL(cpecan_post_mea, dim3(laneBlocks), dim3(64), 0, nullptr, dProblems, nP, dTriples, dCum, dBest,
                           dPrev, dRecord, (float)job->gapGamma, dMea, dCounts, dScores);
        HIP_TRY(hipGetLastError());
    } else if (job->flags & kPostLeftShift) {
        hipLaunchKernelGGL(cpecan_post_copy_chain, dim3((unsigned)nP), dim3(256), 0, nullptr, dProblems, dTriples, dMea,
                           dCounts);
        HIP_TRY(hipGetLastError());
    }
    if (job->flags & kPostLeftShift) {
        uint8_t *dChars = nullptr;
        if (!job->chars) {
            cpk_set_error("left shift needs the raw sequences");
            return CPECAN_EINVAL;
        }
        if (int rc = sc.alloc(&dChars, (size_t)job->nChars)) return rc;
        if (int rc = sc.alloc(&dShift, (size_t)job->shiftCap * 3)) return rc;
        HIP_TRY(hipMemcpy(dChars, job->chars, (size_t)job->nChars, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cpecan_post_left_shift, dim3(laneBlocks), dim3(64), 0, nullptr, dProblems, nP, dMea, dChars,
                           dShift, dCounts);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    if (job->scores) HIP_TRY(hipMemcpy(job->scores, dScores, sizeof(double) * (size_t)nP * 3, hipMemcpyDeviceToHost));
    if (job->counts) HIP_TRY(hipMemcpy(job->counts, dCounts, sizeof(int32_t) * (size_t)nP * 2, hipMemcpyDeviceToHost));
    if (job->mea && dMea) HIP_TRY(hipMemcpy(job->mea, dMea, sizeof(int32_t) * 3 * (size_t)job->meaCap, hipMemcpyDeviceToHost));
    if (job->shift && dShift)
        HIP_TRY(hipMemcpy(job->shift, dShift, sizeof(int32_t) * 3 * (size_t)job->shiftCap, hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

extern "C" int cpk_device_post(CpkDevice *d, const CpkPostJob *job) {
    HIP_TRY(hipSetDevice(d->device));
    if (!d->dCompact && job->nProblems > 0) {
        // every list is empty: the consumers still need a valid base pointer
        HIP_TRY(hipMalloc((void **)&d->dCompact, sizeof(int32_t) * 3));
        d->compactCap = 1;
    }
    return post_core(d->dCompact, job);
}

extern "C" int cpk_post_lists(int device, int32_t *triples, int64_t total, const CpkPostJob *job) {
    const int nDev = cpk_device_count();
    if (nDev <= 0 || device < 0 || device >= nDev) {
        cpk_set_error("no usable HIP device (count=%d, requested=%d): the HIP path has no CPU fallback", nDev, device);
        return CPECAN_ENODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    PostScratch sc;
    int32_t *dTriples = nullptr;
    if (int rc = sc.alloc(&dTriples, (size_t)(total > 0 ? total : 1) * 3)) return rc;
    if (total > 0) HIP_TRY(hipMemcpy(dTriples, triples, sizeof(int32_t) * 3 * (size_t)total, hipMemcpyHostToDevice));
    if (int rc = post_core(dTriples, job)) return rc;
    if (total > 0) HIP_TRY(hipMemcpy(triples, dTriples, sizeof(int32_t) * 3 * (size_t)total, hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

extern "C" int cpk_device_debug_fetch(CpkDevice *d, double *fb, int64_t cells, double *totals, int64_t diags) {
    HIP_TRY(hipSetDevice(d->device));
    if (!d->geo.debug || !d->dDbgFb) {
        cpk_set_error("debug buffers were not enabled before upload");
        return CPECAN_ESTATE;
    }
    if (cells > d->dbgCells || diags > d->dbgDiags) {
        cpk_set_error("debug fetch larger than the debug buffers");
        return CPECAN_EINVAL;
    }
    HIP_TRY(hipMemcpy(fb, d->dDbgFb, sizeof(double) * (size_t)cells, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(totals, d->dDbgTotals, sizeof(double) * (size_t)diags, hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

extern "C" int64_t cpk_device_bytes(const CpkDevice *d) { return d->bytes; }
extern "C" int cpk_device_waves(const CpkDevice *d) { return d->slots + d->pSlots[0] + d->pSlots[1] + d->pSlots[2]; }
