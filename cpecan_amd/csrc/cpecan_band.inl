/*
 * cpecan_band.inl -- the banded region of one DP problem as a stream of anti-diagonals, shared by the C host code
 * (planning: cell counts, traceback schedule, scratch sizes) and the HIP TU (the kernel that materialises the
 * per-diagonal table on the device).  Semantics: band_construct / band_constructDynamic,
 * impl/pairwiseAligner.c:128-234: between consecutive anchors (matrix coordinates = sequence coordinates + 1, a
 * virtual first anchor (0,0) and last (lX,lY)) the band is the rectangle x in [x_prev - E/2, x_next + E/2],
 * y in [y_prev - E/2, y_next + E/2], clamped to the matrix and cut by each anti-diagonal.
 */
#ifndef CPECAN_BAND_INL_
#define CPECAN_BAND_INL_

#include <stdint.h>

#ifdef __HIPCC__
#define CPK_HD __host__ __device__ static inline
#else
#define CPK_HD static inline
#endif

/* The batch's own copy of the anchors: 32-bit values (coordinates are below 2^30), `stride` of them per anchor -- (x, y)
 * for a fixed expansion, (x, y, expansion) for per-anchor expansions.  At 8 bytes an anchor instead of the API's 24 the
 * anchors of a realignment batch (one per aligned column) stop being the largest upload. */
typedef int32_t cpk_anchor_t;

typedef struct {
    const cpk_anchor_t *anchors; /* (x, y[, expansion]), coordinates relative to the region */
    int stride;
    int64_t n, lX, lY;
    int64_t used;
    int64_t pX, pY;         /* previous anchor (matrix coordinates) */
    int64_t qX, qY, qSum;   /* next anchor and its anti-diagonal */
    int64_t xLo, xHi, yLo, yHi;
    int64_t e;              /* expansion in force */
    int dynamic;
    /* The anchors as RUNS of diagonal neighbours (round 4; the host's planning of realign-style batches, whose anchors
     * arrive as runs and are expanded to one anchor per column on the device only): `runs` holds (x, y, length, -) per run,
     * anchor number `used` is anchor `ro` of run `ri`; qRi / qRo: where the anchor now in (qX, qY) sits.  NULL: `anchors`. */
    const int32_t *runs;
    int64_t nRuns, ri, qRi;
    int32_t ro, qRo;
} CpkBandIter;

CPK_HD int64_t cpk_clamp(int64_t v, int64_t hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* Returns 0, or -1 for parameters the reference asserts on (pairwiseAligner.c:131, :186). */
CPK_HD int cpk_band_init(CpkBandIter *it, const cpk_anchor_t *anchors, int stride, int64_t n, int64_t lX, int64_t lY,
                         int64_t expansion, int dynamic) {
    if (lX < 0 || lY < 0) return -1;
    if (!dynamic && (expansion < 0 || expansion % 2 != 0)) return -1;
    it->anchors = anchors;
    it->stride = stride;
    it->n = n;
    it->lX = lX;
    it->lY = lY;
    it->used = 0;
    it->pX = it->pY = 0;
    it->qX = it->qY = it->qSum = 0;
    it->xLo = it->xHi = it->yLo = it->yHi = 0;
    it->e = dynamic ? 0 : expansion;
    it->dynamic = dynamic;
    it->runs = 0;
    it->nRuns = it->ri = it->qRi = 0;
    it->ro = it->qRo = 0;
    return 0;
}

/* The same over runs (fixed expansion only): n = the number of ANCHORS the runs stand for. */
CPK_HD int cpk_band_init_runs(CpkBandIter *it, const int32_t *runs, int64_t nRuns, int64_t n, int64_t lX, int64_t lY,
                              int64_t expansion) {
    const int rc = cpk_band_init(it, 0, 2, n, lX, lY, expansion, 0);
    it->runs = runs;
    it->nRuns = nRuns;
    return rc;
}

/* The next anchor's own diagonal has been emitted: the iterator moves on to the interval behind it.  Returns 0, or -1
 * for anchors that do not describe a valid band. */
CPK_HD int cpk_band_advance(CpkBandIter *it) {
    it->pX = it->qX;
    it->pY = it->qY;
    it->qX = it->lX;
    it->qY = it->lY;
    if (it->used < it->n) {
        if (it->runs) {
            const int32_t *q = it->runs + 4 * it->ri;
            it->qX = (int64_t)q[0] + it->ro + 1;
            it->qY = (int64_t)q[1] + it->ro + 1;
            it->qRi = it->ri;
            it->qRo = it->ro;
            if (++it->ro >= q[2]) {
                it->ro = 0;
                it->ri++;
            }
        } else {
            const cpk_anchor_t *q = it->anchors + (int64_t)it->stride * it->used;
            it->qX = (int64_t)q[0] + 1;
            it->qY = (int64_t)q[1] + 1;
            if (it->dynamic) it->e = q[2]; /* stride 3 whenever dynamic */
        }
        it->used++;
        if (it->qX <= it->pX || it->qY <= it->pY || it->qX > it->lX || it->qY > it->lY || it->e < 0 || it->e % 2 != 0)
            return -1;
    }
    it->qSum = it->qX + it->qY;
    it->xLo = cpk_clamp(it->pX - it->e / 2, it->lX);
    it->yHi = cpk_clamp(it->qY + it->e / 2, it->lY);
    it->xHi = cpk_clamp(it->qX + it->e / 2, it->lX);
    it->yLo = cpk_clamp(it->pY - it->e / 2, it->lY);
    return 0;
}

/* Inside a run of diagonal-neighbour anchors -- the interval (X, Y) -> (X + 1, Y + 1), its rectangle clear of the matrix
 * edges -- the band holds exactly two diagonals: x-y in [X-Y-E-1, X-Y+E+1] (E + 2 cells), then [X-Y-E, X-Y+E] (E + 1
 * cells); cpk_band_next gives the same (planning and the device's table builder skip its arithmetic there).  True when
 * the iterator stands in front of the first diagonal d of such an interval. */
CPK_HD int cpk_band_in_run(const CpkBandIter *it, int64_t d) {
    const int64_t h = it->e / 2;
    return !it->dynamic && it->used >= 2 && d == it->pX + it->pY + 1 && it->qX == it->pX + 1 && it->qY == it->pY + 1 &&
           it->pX - h >= 0 && it->pY - h >= 0 && it->qX + h <= it->lX && it->qY + h <= it->lY;
}

/* Diagonal d = 0, 1, ..., lX+lY in order: *xmyL / *xmyR receive its inclusive x-y range.  Returns 0, or -1 when the
 * anchors do not describe a valid band (diagonal_construct would throw, pairwiseAligner.c:31; asserts :159-166). */
CPK_HD int cpk_band_next(CpkBandIter *it, int64_t d, int64_t *xmyL, int64_t *xmyR) {
    const int64_t a = it->xLo > d - it->yHi ? it->xLo : d - it->yHi;
    const int64_t b = it->xHi < d - it->yLo ? it->xHi : d - it->yLo;
    if (a > b) return -1;
    *xmyL = 2 * a - d;
    *xmyR = 2 * b - d;
    if (it->qSum != d) return 0;
    return cpk_band_advance(it); /* the anchor's own diagonal has been emitted: move on to the next interval */
}

#endif
