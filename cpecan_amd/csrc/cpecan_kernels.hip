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
#include <cstring>
#include <vector>

#include "cpecan_internal.h"

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
struct KArgs {
    const CpkRegion *regions;
    const CpkDiag *diags;
    const CpkSegment *segs;
    const uint8_t *symbols;
    const CpkModel *model;
    CpkGeometry geo;
    double *ring;    // [slots][ringCells*S]
    double *fb;      // [slots][fbCells]
    double *cbuf;    // [slots][refreshCells]
    double *mbuf;    // [slots][refreshCells]
    double *totals;  // [slots][maxRefresh]
    double *groll;   // [slots][rollDoubles]   (only when geo.useGlobalRoll)
    int32_t *outCounts;  // [nLists][nRegions]
    int32_t *segStarts;  // [nLists][nSegsTotal]
    int32_t *triples;    // [nLists][outTriplesPerList*3]
    int64_t outTriplesPerList;
    int64_t nSegsTotal;
    unsigned int *queue;
    double *dbgFb;
    double *dbgTotals;
};

// logAdd, impl/pairwiseAligner.c:287-307.  hi/lo form: with d = hi - lo the reference returns hi when
// lo == -inf or d >= 7.5, else lo + P(d).  d is +inf when only lo is -inf and NaN when both are, and
// both fail (d < 7.5), so one comparison covers the reference's two tests.  The cubic's coefficients are
// float literals in the reference, i.e. float32 values widened to double; Horner with separate mul/add.
__device__ __forceinline__ double logadd(double x, double y) {
    const double hi = __builtin_fmax(x, y);
    const double lo = __builtin_fmin(x, y);
    const double d = hi - lo;
    const bool s0 = d <= 1.0, s1 = d <= 2.5, s2 = d <= 4.5;
    const double c3 = s0 ? (double)-0.009350833524763f
                         : (s1 ? (double)-0.014532321752540f : (s2 ? (double)-0.004605031767994f : (double)-0.000458661602210f));
    const double c2 = s0 ? (double)0.130659527668286f
                         : (s1 ? (double)0.139942324101744f : (s2 ? (double)0.063427417320019f : (double)0.009695946122598f));
    const double c1 = s0 ? (double)0.498799810682272f
                         : (s1 ? (double)0.495635523139337f : (s2 ? (double)0.695956496475118f : (double)0.930734667215156f));
    const double c0 = s0 ? (double)0.693203116424741f
                         : (s1 ? (double)0.692140569840976f : (s2 ? (double)0.514272634594009f : (double)0.168037164329057f));
    double r = c3 * d;
    r = r + c2;
    r = r * d;
    r = r + c1;
    r = r * d;
    r = r + c0;
    r = r + lo;
    return (d < 7.5) ? r : hi;
}

// v_med3_i32: clamps a neighbour's dense index onto [-1, hi]; -1 and hi address the -inf guard cells of its row
__device__ __forceinline__ int med3(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = v > o ? v : o;
    }
    return v;
}

template <bool GROLL>
__device__ __forceinline__ void roll_fence() {
    if (GROLL) {
        __syncthreads();  // workgroup-scope release/acquire on global memory (single-wave workgroup)
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ double ld_self(const double *p) {
    // data this wave wrote earlier in the same launch: always a vector load, never the scalar cache
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <int S, bool GROLL>
struct Sweep {
    // wave-uniform context
    const KArgs &a;
    const CpkModel &m;
    const CpkDiag *dg;   // region's diagonal table
    const uint8_t *sxp;  // padded symbols: sxp[x] = symbol of base x-1, sxp[0] = N
    const uint8_t *syp;
    double *roll;        // 3 rolling buffers [3][S][stride]
    const double *em;    // LDS emissions: [0..24] match, [25..29] gapX, [30..34] gapY
    double *ring, *fb, *cbuf, *mbuf, *totals;
    int stride;
    int lane;
    int N;

    __device__ __forceinline__ double *rbuf(int d) const { return roll + (size_t)((d + 3) % 3) * S * stride; }
    __device__ __forceinline__ double *ringAt(const CpkDiag &g) const { return ring + (size_t)g.ringOff * S; }

    // ---- forward: impl/pairwiseAligner.c:609-629 with stateMachine{5,3}_cellCalculate as the per-cell body ----
    __device__ void forward(int d) {
        const CpkDiag g = dg[d];
        const CpkDiag g1 = dg[d - 1];
        const int W = g.width;
        const int dl = (g.xmyL - 1 - g1.xmyL) >> 1;  // lower neighbour (d-1, xmy-1) sits at k + dl, upper at k + dl + 1
        const int hi1 = g1.width;
        int dm = 0, hi2 = -1;
        if (d >= 2) {
            const CpkDiag g2 = dg[d - 2];
            dm = (g.xmyL - g2.xmyL) >> 1;  // middle neighbour (d-2, xmy) sits at k + dm
            hi2 = g2.width;
        }
        double *cur = rbuf(d);
        const double *p1 = rbuf(d - 1);
        const double *p2 = rbuf(d - 2);
        double *out = ringAt(g);
        for (int k = lane; k < W; k += CPK_WAVE) {
            const int xmy = g.xmyL + 2 * k;
            const int x = (d + xmy) >> 1, y = d - x;
            const int cX = sxp[x], cY = syp[y];
            const double eX = em[25 + cX], eM = em[cX * 5 + cY], eY = em[30 + cY];
            const int iL = med3(k + dl, -1, hi1) + 1;
            const int iU = med3(k + dl + 1, -1, hi1) + 1;
            const int iM = med3(k + dm, -1, hi2) + 1;
            double v[S];
            if (S == 5) {
                // states: 0 match, 1 shortGapX, 2 shortGapY, 3 longGapX, 4 longGapY (stateMachine.c:261-263)
                const double lM = p1[0 * stride + iL], lSX = p1[1 * stride + iL], lLX = p1[3 * stride + iL];
                const double uM = p1[0 * stride + iU], uSY = p1[2 * stride + iU], uLY = p1[4 * stride + iU];
                const double mM = p2[0 * stride + iM], mSX = p2[1 * stride + iM], mSY = p2[2 * stride + iM],
                             mLX = p2[3 * stride + iM], mLY = p2[4 * stride + iM];
                // lower block, stateMachine.c:454-462
                v[1] = logadd(lM + (eX + m.shortOpenX), lSX + (eX + m.shortExtendX));
                v[3] = logadd(lM + (eX + m.longOpenX), lLX + (eX + m.longExtendX));
                // middle block, :463-470
                double t = mM + (eM + m.matchContinue);
                t = logadd(t, mSX + (eM + m.matchFromShortX));
                t = logadd(t, mSY + (eM + m.matchFromShortY));
                t = logadd(t, mLX + (eM + m.matchFromLongX));
                t = logadd(t, mLY + (eM + m.matchFromLongY));
                v[0] = t;
                // upper block, :471-479
                v[2] = logadd(uM + (eY + m.shortOpenY), uSY + (eY + m.shortExtendY));
                v[4] = logadd(uM + (eY + m.longOpenY), uLY + (eY + m.longExtendY));
            } else {
                // states: 0 match, 1 gapX, 2 gapY; stateMachine.c:695-713
                const double lM = p1[0 * stride + iL], lGX = p1[1 * stride + iL], lGY = p1[2 * stride + iL];
                const double uM = p1[0 * stride + iU], uGX = p1[1 * stride + iU], uGY = p1[2 * stride + iU];
                const double mM = p2[0 * stride + iM], mGX = p2[1 * stride + iM], mGY = p2[2 * stride + iM];
                double t = lM + (eX + m.shortOpenX);
                t = logadd(t, lGX + (eX + m.shortExtendX));
                t = logadd(t, lGY + (eX + m.shortSwitchToX));
                v[1] = t;
                t = mM + (eM + m.matchContinue);
                t = logadd(t, mGX + (eM + m.matchFromShortX));
                t = logadd(t, mGY + (eM + m.matchFromShortY));
                v[0] = t;
                t = uM + (eY + m.shortOpenY);
                t = logadd(t, uGY + (eY + m.shortExtendY));
                t = logadd(t, uGX + (eY + m.shortSwitchToY));
                v[2] = t;
            }
#pragma unroll
            for (int s = 0; s < S; s++) {
                cur[s * stride + k + 1] = v[s];
                out[(size_t)s * W + k] = v[s];
            }
        }
        if (lane < S) cur[lane * stride + W + 1] = NEG_INF;  // end guard of every state row
        roll_fence<GROLL>();
    }

    // Puts diagonal d of the forward ring back into its rolling buffer (after a traceback used the buffers).
    __device__ void reloadForward(int d) {
        const CpkDiag g = dg[d];
        const int W = g.width;
        double *cur = rbuf(d);
        const double *src = ringAt(g);
        for (int k = lane; k < W; k += CPK_WAVE) {
#pragma unroll
            for (int s = 0; s < S; s++) cur[s * stride + k + 1] = ld_self(src + (size_t)s * W + k);
        }
        if (lane < S) cur[lane * stride + W + 1] = NEG_INF;
        roll_fence<GROLL>();
    }

    // Seeds a traceback: every cell of diagonal d gets the end-state prior (pairwiseAligner.c:798-799).
    __device__ void seedBackward(int d, const double *prior) {
        const int W = dg[d].width;
        double *cur = rbuf(d);
        for (int k = lane; k < W; k += CPK_WAVE) {
#pragma unroll
            for (int s = 0; s < S; s++) cur[s * stride + k + 1] = prior[s];
        }
        if (lane < S) cur[lane * stride + W + 1] = NEG_INF;
        roll_fence<GROLL>();
    }

    // ---- backward: the reference scatters from diagonal d2+1 / d2+2 into d2 (pairwiseAligner.c:392-395,
    // 631-634); this gathers the same terms in the same order.
    // Also forms fb = F.match + B.match for emitted diagonals and, on refresh diagonals, the two
    // per-cell series whose sequential logAdd folds give the total probability (:636-653).
    __device__ void backward(int d2, const CpkSegment &sg, bool seeded, int64_t fbBase) {
        const CpkDiag g = dg[d2];
        const int W = g.width;
        const bool emit = d2 <= sg.tbFrom;
        const bool refresh = emit && ((sg.tbFrom - d2) % CPK_REFRESH_PERIOD == 0);
        const int jr = (sg.tbFrom - d2) / CPK_REFRESH_PERIOD;
        const int J = sg.nRefresh;
        double *cur = rbuf(d2);
        int db = 0, hiB = -1, da = 0, hiA = -1;
        const double *pb = rbuf(d2 + 1);
        const double *pa = rbuf(d2 + 2);
        if (!seeded) {
            const CpkDiag gb = dg[d2 + 1];
            db = (g.xmyL - 1 - gb.xmyL) >> 1;  // source (d2+1, xmy-1) at k + db, source (d2+1, xmy+1) at k + db + 1
            hiB = gb.width;
            if (d2 + 2 <= sg.dTop) {
                const CpkDiag ga = dg[d2 + 2];
                da = (g.xmyL - ga.xmyL) >> 1;  // source (d2+2, xmy) at k + da
                hiA = ga.width;
            }
        }
        const double *fsrc = ringAt(g);
        for (int k = lane; k < W; k += CPK_WAVE) {
            double v[S];
            if (seeded) {
#pragma unroll
                for (int s = 0; s < S; s++) v[s] = cur[s * stride + k + 1];
            } else {
                const int xmy = g.xmyL + 2 * k;
                const int x = (d2 + xmy) >> 1, y = d2 - x;
                const int cX1 = sxp[x + 1], cY1 = syp[y + 1];  // symbols of the source cells (x+1,.) and (.,y+1)
                const double eX = em[25 + cX1], eM = em[cX1 * 5 + cY1], eY = em[30 + cY1];
                const int iU = med3(k + db, -1, hiB) + 1;      // cell (x, y+1): its "upper" neighbour is the target
                const int iL = med3(k + db + 1, -1, hiB) + 1;  // cell (x+1, y): its "lower" neighbour is the target
                const int iA = med3(k + da, -1, hiA) + 1;      // cell (x+1, y+1): its "middle" neighbour is the target
                const double aM = pa[0 * stride + iA];
                if (S == 5) {
                    const double uSY = pb[2 * stride + iU], uLY = pb[4 * stride + iU];
                    const double lSX = pb[1 * stride + iL], lLX = pb[3 * stride + iL];
                    double t = aM + (eM + m.matchContinue);
                    t = logadd(t, uSY + (eY + m.shortOpenY));
                    t = logadd(t, uLY + (eY + m.longOpenY));
                    t = logadd(t, lSX + (eX + m.shortOpenX));
                    t = logadd(t, lLX + (eX + m.longOpenX));
                    v[0] = t;
                    v[1] = logadd(aM + (eM + m.matchFromShortX), lSX + (eX + m.shortExtendX));
                    v[2] = logadd(aM + (eM + m.matchFromShortY), uSY + (eY + m.shortExtendY));
                    v[3] = logadd(aM + (eM + m.matchFromLongX), lLX + (eX + m.longExtendX));
                    v[4] = logadd(aM + (eM + m.matchFromLongY), uLY + (eY + m.longExtendY));
                } else {
                    const double uGY = pb[2 * stride + iU];
                    const double lGX = pb[1 * stride + iL];
                    double t = aM + (eM + m.matchContinue);
                    t = logadd(t, uGY + (eY + m.shortOpenY));
                    t = logadd(t, lGX + (eX + m.shortOpenX));
                    v[0] = t;
                    t = aM + (eM + m.matchFromShortX);
                    t = logadd(t, uGY + (eY + m.shortSwitchToY));
                    t = logadd(t, lGX + (eX + m.shortExtendX));
                    v[1] = t;
                    t = aM + (eM + m.matchFromShortY);
                    t = logadd(t, uGY + (eY + m.shortExtendY));
                    t = logadd(t, lGX + (eX + m.shortSwitchToX));
                    v[2] = t;
                }
#pragma unroll
                for (int s = 0; s < S; s++) cur[s * stride + k + 1] = v[s];
            }
            if (emit) {
                const double f0 = ld_self(fsrc + k);
                const double fbv = f0 + v[0];
                fb[(size_t)(g.cellOff - fbBase) + k] = fbv;
                if (refresh) {
                    // cell_dotProduct over states, pairwiseAligner.c:402-408
                    double t = fbv;
#pragma unroll
                    for (int s = 1; s < S; s++) t = logadd(t, ld_self(fsrc + (size_t)s * W + k) + v[s]);
                    cbuf[(size_t)k * J + jr] = t;
                }
            }
        }
        if (!seeded && lane < S) cur[lane * stride + W + 1] = NEG_INF;
        roll_fence<GROLL>();
        if (refresh && d2 + 1 <= sg.dTop) {
            // matches straddling d2: middle-block forward step from F[d2-1] into the cells of d2+1, times B[d2+1]
            // (pairwiseAligner.c:643-651).  Non-match states of the temporary stay -inf and drop out exactly.
            const CpkDiag gn = dg[d2 + 1];
            const CpkDiag gp = dg[d2 - 1];
            const int Wn = gn.width, Wp = gp.width;
            const int dmm = (gn.xmyL - gp.xmyL) >> 1;
            const double *fprev = ringAt(gp);
            const double *bn = rbuf(d2 + 1);
            for (int k = lane; k < Wn; k += CPK_WAVE) {
                const int xmy = gn.xmyL + 2 * k;
                const int x = (d2 + 1 + xmy) >> 1, y = d2 + 1 - x;
                const double eM = em[sxp[x] * 5 + syp[y]];
                const int kp = k + dmm;
                const bool ok = kp >= 0 && kp < Wp;
                const int kq = ok ? kp : 0;
                double f[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const double val = ld_self(fprev + (size_t)s * Wp + kq);
                    f[s] = ok ? val : NEG_INF;
                }
                double t = f[0] + (eM + m.matchContinue);
                t = logadd(t, f[1] + (eM + m.matchFromShortX));
                t = logadd(t, f[2] + (eM + m.matchFromShortY));
                if (S == 5) {
                    t = logadd(t, f[3] + (eM + m.matchFromLongX));
                    t = logadd(t, f[4] + (eM + m.matchFromLongY));
                }
                mbuf[(size_t)k * J + jr] = t + bn[0 * stride + k + 1];
            }
        }
    }

    // ---- total probability at every refresh point of the segment: one lane per refresh point, each doing the
    // reference's sequential folds (dpDiagonal_dotProduct :513-523, then the straddle term :649).
    __device__ void foldTotals(const CpkSegment &sg) {
        const int J = sg.nRefresh;
        for (int j0 = 0; j0 < J; j0 += CPK_WAVE) {
            const int j = j0 + lane;
            const bool on = j < J;
            const int r = sg.tbFrom - CPK_REFRESH_PERIOD * (on ? j : 0);
            const int Wc = on ? dg[r].width : 0;
            const int Wm = (on && r + 1 <= sg.dTop) ? dg[r + 1].width : 0;
            double total = NEG_INF, straddle = NEG_INF;
            const int Wc_max = wave_max(Wc), Wm_max = wave_max(Wm);
            for (int k = 0; k < Wc_max; k++) {
                if (k < Wc) total = logadd(total, ld_self(cbuf + (size_t)k * J + j));
            }
            for (int k = 0; k < Wm_max; k++) {
                if (k < Wm) straddle = logadd(straddle, ld_self(mbuf + (size_t)k * J + j));
            }
            if (on) {
                if (r + 1 <= sg.dTop) total = logadd(total, straddle);
                totals[j] = total;
            }
        }
        roll_fence<true>();
    }

    // ---- thresholded posteriors in list order (diagonal ascending, x-y descending), pairwiseAligner.c:655-689
    __device__ int emitMatches(const CpkSegment &sg, int32_t *out, int outCap, int count, int64_t fbBase,
                               double *dbgTot, double *dbgFb) {
        const double thr = m.threshold;
        const double logThrLo = log(thr) - 1e-6;
        for (int d2 = sg.tbPrev + 1; d2 <= sg.tbFrom; d2++) {
            const CpkDiag g = dg[d2];
            const int W = g.width;
            const double total = ld_self(totals + (sg.tbFrom - d2) / CPK_REFRESH_PERIOD);
            if (dbgTot) dbgTot[d2] = total;  // every lane stores the same value: no lane test next to the ballots below
            for (int base = ((W - 1) / CPK_WAVE) * CPK_WAVE; base >= 0; base -= CPK_WAVE) {
                const int k = base + (CPK_WAVE - 1 - lane);
                const bool in = k < W;
                double z = NEG_INF;
                int x = 0, y = 0;
                bool valid = false;  // only cells with x > 0 and y > 0 are match cells (pairwiseAligner.c:680)
                if (in) {
                    const int xmy = g.xmyL + 2 * k;
                    x = (d2 + xmy) >> 1;
                    y = d2 - x;
                    const double fbv = ld_self(fb + (size_t)(g.cellOff - fbBase) + k);
                    if (dbgFb) dbgFb[g.cellOff + k] = fbv;
                    valid = x > 0 && y > 0;
                    if (valid) z = fbv - total;
                }
                const bool cand = valid && z >= logThrLo;
                if (__ballot(cand) == 0ull) continue;
                double p = exp(z);
                const bool keep = cand && p >= thr;
                const unsigned long long mask = __ballot(keep);
                if (mask == 0ull) continue;
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
        }
        return count;
    }
};

template <int S, bool GROLL>
__global__ void __launch_bounds__(CPK_WAVE) cpecan_pairhmm_sweep(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const CpkModel &m = *a.model;
    const int stride = a.geo.rollStride;

    // emission tables -> LDS (per-lane indexed reads)
    double *em = lds;
    if (lane < 25) em[lane] = m.matchEm[lane];
    if (lane < 5) {
        em[25 + lane] = m.gapXEm[lane];
        em[30 + lane] = m.gapYEm[lane];
    }
    double *roll = GROLL ? (a.groll + (size_t)blockIdx.x * a.geo.rollDoubles) : (lds + 40);
    // every rolling cell starts as -inf so that guards (position 0 of each row) are valid forever
    for (int i = lane; i < 3 * S * stride; i += CPK_WAVE) roll[i] = NEG_INF;
    __syncthreads();

    const size_t slot = blockIdx.x;
    for (;;) {
        // Every lane takes part in the ticket fetch (lane 0 adds 1, the others add 0; hipcc folds this into one
        // atomic per wave).  Do NOT write this as `if (lane == 0) ticket = atomicAdd(..)`: hipcc 7.2 jump-threads
        // the lane test across the loop back-edge and re-runs the readfirstlane with 63 lanes -> endless loop.
        const unsigned int ticket = atomicAdd(a.queue, lane == 0 ? 1u : 0u);
        const int r = __builtin_amdgcn_readfirstlane((int)ticket);
        if (r >= a.geo.nRegions) break;

        const CpkRegion &rg = a.regions[r];
        Sweep<S, GROLL> sw{a,
                           m,
                           a.diags + rg.diagOff,
                           a.symbols + rg.seqXOff,
                           a.symbols + rg.seqYOff,
                           roll,
                           em,
                           a.ring + slot * (size_t)a.geo.ringCells * S,
                           a.fb + slot * (size_t)a.geo.fbCells,
                           a.cbuf + slot * (size_t)a.geo.refreshCells,
                           a.mbuf + slot * (size_t)a.geo.refreshCells,
                           a.totals + slot * (size_t)a.geo.maxRefresh,
                           stride,
                           lane,
                           rg.lX + rg.lY};
        const int N = sw.N;
        int32_t *out = a.triples + 3 * rg.outOff;
        int count = 0;
        if (N > 0) {
            // diagonal 0: the single cell (0,0) holds the start prior (pairwiseAligner.c:776-777)
            const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
            {
                double *cur = sw.rbuf(0);
                double *o0 = sw.ringAt(sw.dg[0]);
                if (lane < S) {
                    cur[lane * stride + 1] = startPrior[lane];
                    cur[lane * stride + 2] = NEG_INF;
                    o0[lane] = startPrior[lane];
                }
                roll_fence<GROLL>();
            }
            int d = 1;
            for (int si = 0; si < rg.nSeg; si++) {
                const CpkSegment sg = a.segs[rg.segOff + si];
                for (; d <= sg.dTop; d++) sw.forward(d);
                // traceback (pairwiseAligner.c:796-862)
                const double *endPrior = (sg.atEnd && rg.raggedRight) ? m.raggedEnd : m.end;
                const int64_t fbBase = sw.dg[sg.tbPrev + 1].cellOff;
                sw.seedBackward(sg.dTop, endPrior);
                sw.backward(sg.dTop, sg, true, fbBase);
                for (int d2 = sg.dTop - 1; d2 > sg.tbPrev; d2--) sw.backward(d2, sg, false, fbBase);
                roll_fence<true>();  // fb / cbuf / mbuf stores of all lanes are complete before they are re-read
                sw.foldTotals(sg);
                if (lane == 0) a.segStarts[rg.segOff + si] = count;
                count = sw.emitMatches(sg, out, rg.outCap, count, fbBase,
                                       a.geo.debug ? a.dbgTotals + rg.dbgDiagOff : nullptr,
                                       a.geo.debug ? a.dbgFb + rg.dbgCellOff : nullptr);
                if (!sg.atEnd) {
                    sw.reloadForward(sg.dTop - 1);
                    sw.reloadForward(sg.dTop);
                }
            }
        }
        if (lane == 0) a.outCounts[r] = count;
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the HIP TU: memory, launch, timing
// ------------------------------------------------------------------------------------------------
struct CpkDevice {
    int device = 0;
    int numCUs = 0;
    CpkGeometry geo{};
    int nLists = 1;
    int64_t nSegs = 0, nDiags = 0;
    int64_t outTriplesPerList = 0;
    int64_t dbgCells = 0, dbgDiags = 0;
    int slots = 0;
    size_t ldsBytes = 0;
    // device buffers
    CpkRegion *dRegions = nullptr;
    CpkDiag *dDiags = nullptr;
    CpkSegment *dSegs = nullptr;
    uint8_t *dSymbols = nullptr;
    CpkModel *dModel = nullptr;
    double *dRing = nullptr, *dFb = nullptr, *dC = nullptr, *dM = nullptr, *dTotals = nullptr, *dGroll = nullptr;
    int32_t *dCounts = nullptr, *dSegStarts = nullptr, *dTriples = nullptr;
    unsigned int *dQueue = nullptr;
    double *dDbgFb = nullptr, *dDbgTotals = nullptr;
    int64_t bytes = 0;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    hipStream_t lastStream = nullptr;
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
    *out = d;
    return CPECAN_OK;
}

static void free_all(CpkDevice *d) {
    void *ptrs[] = {d->dRegions, d->dDiags, d->dSegs, d->dSymbols, d->dModel, d->dRing, d->dFb, d->dC, d->dM,
                    d->dTotals, d->dGroll, d->dCounts, d->dSegStarts, d->dTriples, d->dQueue, d->dDbgFb, d->dDbgTotals};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    d->dRegions = nullptr; d->dDiags = nullptr; d->dSegs = nullptr; d->dSymbols = nullptr; d->dModel = nullptr;
    d->dRing = d->dFb = d->dC = d->dM = d->dTotals = d->dGroll = nullptr;
    d->dCounts = d->dSegStarts = d->dTriples = nullptr;
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

static KernelFn pick_kernel(const CpkGeometry &g) {
    if (g.nStates == 5) return g.useGlobalRoll ? cpecan_pairhmm_sweep<5, true> : cpecan_pairhmm_sweep<5, false>;
    return g.useGlobalRoll ? cpecan_pairhmm_sweep<3, true> : cpecan_pairhmm_sweep<3, false>;
}

extern "C" int cpk_device_upload(CpkDevice *d, const CpkGeometry *geo, const CpkModel *model, const CpkRegion *regions,
                                 const CpkDiag *diags, int64_t nDiags, const CpkSegment *segs, int64_t nSegs,
                                 const uint8_t *symbols, int64_t nSymbolBytes, int64_t outTriplesPerList, int nLists,
                                 int64_t dbgCells, int64_t dbgDiags, double *h2dMs) {
    HIP_TRY(hipSetDevice(d->device));
    free_all(d);
    d->geo = *geo;
    d->nLists = nLists;
    d->nSegs = nSegs;
    d->nDiags = nDiags;
    d->outTriplesPerList = outTriplesPerList;
    d->dbgCells = dbgCells;
    d->dbgDiags = dbgDiags;
    d->ran = false;
    const int S = geo->nStates;

    // LDS: 40 doubles of emission tables + three rolling buffers (unless they live in global memory)
    d->ldsBytes = sizeof(double) * (40 + (geo->useGlobalRoll ? 0 : (size_t)3 * S * geo->rollStride));
    KernelFn fn = pick_kernel(*geo);
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
    if (perCU < 1) {
        cpk_set_error("kernel does not fit on a CU (LDS %zu bytes)", d->ldsBytes);
        return CPECAN_EHIP;
    }
    int64_t slots = (int64_t)perCU * d->numCUs;
    if (slots > geo->nRegions) slots = geo->nRegions;
    if (slots < 1) slots = 1;
    d->slots = (int)slots;

    if (int rc = dev_alloc(d, &d->dRegions, (size_t)geo->nRegions)) return rc;
    if (int rc = dev_alloc(d, &d->dDiags, (size_t)nDiags)) return rc;
    if (int rc = dev_alloc(d, &d->dSegs, (size_t)nSegs)) return rc;
    if (int rc = dev_alloc(d, &d->dSymbols, (size_t)nSymbolBytes)) return rc;
    if (int rc = dev_alloc(d, &d->dModel, 1)) return rc;
    if (int rc = dev_alloc(d, &d->dRing, (size_t)slots * geo->ringCells * S)) return rc;
    if (int rc = dev_alloc(d, &d->dFb, (size_t)slots * geo->fbCells)) return rc;
    if (int rc = dev_alloc(d, &d->dC, (size_t)slots * geo->refreshCells)) return rc;
    if (int rc = dev_alloc(d, &d->dM, (size_t)slots * geo->refreshCells)) return rc;
    if (int rc = dev_alloc(d, &d->dTotals, (size_t)slots * geo->maxRefresh)) return rc;
    if (geo->useGlobalRoll)
        if (int rc = dev_alloc(d, &d->dGroll, (size_t)slots * geo->rollDoubles)) return rc;
    if (int rc = dev_alloc(d, &d->dCounts, (size_t)nLists * geo->nRegions)) return rc;
    if (int rc = dev_alloc(d, &d->dSegStarts, (size_t)nLists * nSegs)) return rc;
    if (int rc = dev_alloc(d, &d->dTriples, (size_t)nLists * outTriplesPerList * 3)) return rc;
    if (int rc = dev_alloc(d, &d->dQueue, 1)) return rc;
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
    HIP_TRY(hipMemcpy(d->dDiags, diags, sizeof(CpkDiag) * (size_t)nDiags, hipMemcpyHostToDevice));
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
    a.regions = d->dRegions;
    a.diags = d->dDiags;
    a.segs = d->dSegs;
    a.symbols = d->dSymbols;
    a.model = d->dModel;
    a.geo = d->geo;
    a.ring = d->dRing;
    a.fb = d->dFb;
    a.cbuf = d->dC;
    a.mbuf = d->dM;
    a.totals = d->dTotals;
    a.groll = d->dGroll;
    a.outCounts = d->dCounts;
    a.segStarts = d->dSegStarts;
    a.triples = d->dTriples;
    a.outTriplesPerList = d->outTriplesPerList;
    a.nSegsTotal = d->nSegs;
    a.queue = d->dQueue;
    a.dbgFb = d->dDbgFb;
    a.dbgTotals = d->dDbgTotals;
    HIP_TRY(hipMemsetAsync(d->dQueue, 0, sizeof(unsigned int), st));
    KernelFn fn = pick_kernel(d->geo);
    HIP_TRY(hipEventRecord(d->evStart, st));
    hipLaunchKernelGGL(fn, dim3((unsigned)d->slots), dim3(CPK_WAVE), d->ldsBytes, st, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(d->evStop, st));
    d->lastStream = st;
    d->ran = true;
    return CPECAN_OK;
}

extern "C" int cpk_device_download(CpkDevice *d, int32_t *counts, int32_t *segStarts, int32_t *triples, double *expect,
                                   double *kernelMs, double *d2hMs) {
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
    HIP_TRY(hipMemcpy(triples, d->dTriples, sizeof(int32_t) * (size_t)d->nLists * d->outTriplesPerList * 3,
                      hipMemcpyDeviceToHost));
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (d2hMs) *d2hMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)expect;
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
extern "C" int cpk_device_waves(const CpkDevice *d) { return d->slots; }
