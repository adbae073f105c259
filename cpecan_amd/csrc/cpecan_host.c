/*
 * cpecan_host.c -- host side (plain C99) of libcpecan_hip.so.
 *
 * Everything that is integer bookkeeping in the reference stays on the host, in C, as the north star asks:
 * the band (impl/pairwiseAligner.c:94-234), the split into regions (:1206-1326), the traceback schedule
 * (:791-810), packing of problems into flat arrays for the GPU and re-assembly of the returned triples into
 * the reference's list order (:1411-1418).  All floating-point DP work happens in cpecan_kernels.hip; there is
 * no CPU implementation of it in this library.
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <omp.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "cpecan_internal.h"
#include "cpecan_band.inl"

/* ------------------------------------------------------------------------------------------------
 * models: impl/stateMachine.c
 * ---------------------------------------------------------------------------------------------- */
enum { kMatch = 0, kShortX = 1, kShortY = 2, kLongX = 3, kLongY = 4 }; /* stateMachine.c:261-263 */

static int is_five(int32_t type) { return type == CPECAN_FIVE_STATE || type == CPECAN_FIVE_STATE_ASYM; }
static int is_three(int32_t type) { return type == CPECAN_THREE_STATE || type == CPECAN_THREE_STATE_ASYM; }

int cpecan_model_default(cpecan_model *m, int32_t type) {
    if (!m || !(is_five(type) || is_three(type))) return CPECAN_EINVAL;
    memset(m, 0, sizeof *m);
    m->type = type;
    /* emissions: stateMachine.c:269-292 */
    const double eMatch = -2.1149196655034745, eTransversion = -4.5691014376830479, eTransition = -3.9833860032220842;
    for (int x = 0; x < 4; x++) {
        for (int y = 0; y < 4; y++) {
            m->emissionMatch[x * 4 + y] = x == y ? eMatch : (((x ^ y) == 2) ? eTransition : eTransversion);
        }
        m->emissionGapX[x] = m->emissionGapY[x] = -1.6094379124341003;
    }
    m->matchContinue = -0.030064059121770816;
    m->matchFromShortGapX = m->matchFromShortGapY = -1.272871422049609;
    m->gapShortExtendX = m->gapShortExtendY = -0.3388262689231553;
    m->gapShortSwitchToX = m->gapShortSwitchToY = -4.910694825551255;
    if (is_five(type)) { /* stateMachine.c:484-501 */
        m->matchFromLongGapX = m->matchFromLongGapY = -5.673280173170473;
        m->gapShortOpenX = m->gapShortOpenY = -4.34381910900448;
        m->gapLongOpenX = m->gapLongOpenY = -6.30810595366929;
        m->gapLongExtendX = m->gapLongExtendY = -0.003442492794189331;
        m->gapLongSwitchToX = m->gapLongSwitchToY = -6.30810595366929;
    } else { /* stateMachine.c:718-726 */
        m->gapShortOpenX = m->gapShortOpenY = -4.21256642;
    }
    return CPECAN_OK;
}

int cpecan_hmm_init(cpecan_hmm *h, int32_t type, double pseudo) { /* stateMachine.c:23-48 */
    if (!h || !(is_five(type) || is_three(type))) return CPECAN_EINVAL;
    h->type = type;
    h->stateNumber = is_five(type) ? 5 : 3;
    for (int i = 0; i < 25; i++) h->transitions[i] = pseudo;
    for (int i = 0; i < 80; i++) h->emissions[i] = pseudo;
    h->likelihood = 0.0;
    return CPECAN_OK;
}

int cpecan_hmm_normalise(cpecan_hmm *h) { /* stateMachine.c:88-112 */
    if (!h) return CPECAN_EINVAL;
    const int S = h->stateNumber;
    for (int from = 0; from < S; from++) {
        double sum = 0.0;
        for (int to = 0; to < S; to++) sum += h->transitions[from * S + to];
        for (int to = 0; to < S; to++) h->transitions[from * S + to] /= sum;
    }
    for (int s = 0; s < S; s++) {
        double sum = 0.0;
        for (int i = 0; i < 16; i++) sum += h->emissions[s * 16 + i];
        for (int i = 0; i < 16; i++) h->emissions[s * 16 + i] /= sum;
    }
    return CPECAN_OK;
}

int cpecan_hmm_write(const cpecan_hmm *h, const char *path) { /* stateMachine.c:133-143 */
    if (!h || !path) return CPECAN_EINVAL;
    FILE *f = fopen(path, "w");
    if (!f) return CPECAN_EINVAL;
    const int S = h->stateNumber;
    fprintf(f, "%i\t", h->type);
    for (int i = 0; i < S * S; i++) fprintf(f, "%f\t", h->transitions[i]);
    fprintf(f, "%f\n", h->likelihood);
    for (int i = 0; i < S * 16; i++) fprintf(f, "%f\t", h->emissions[i]);
    fprintf(f, "\n");
    fclose(f);
    return CPECAN_OK;
}

int cpecan_hmm_load(cpecan_hmm *h, const char *path) { /* stateMachine.c:145-202 */
    if (!h || !path) return CPECAN_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) return CPECAN_EINVAL;
    int type = -1, rc = CPECAN_EINVAL;
    if (fscanf(f, "%i", &type) == 1 && cpecan_hmm_init(h, type, 0.0) == CPECAN_OK) {
        const int S = h->stateNumber;
        int ok = 1;
        for (int i = 0; i < S * S && ok; i++) ok = fscanf(f, "%lf", &h->transitions[i]) == 1;
        ok = ok && fscanf(f, "%lf", &h->likelihood) == 1;
        for (int i = 0; i < S * 16 && ok; i++) ok = fscanf(f, "%lf", &h->emissions[i]) == 1;
        if (ok) rc = CPECAN_OK;
    }
    fclose(f);
    return rc;
}

static double tr(const cpecan_hmm *h, int from, int to) { return h->transitions[from * h->stateNumber + to]; }
static double emi(const cpecan_hmm *h, int s, int x, int y) { return h->emissions[s * 16 + x * 4 + y]; }

/* emissions_loadGapProbs, stateMachine.c:327-349: collapse gap-state emission matrices onto one sequence */
static void gap_emissions(const cpecan_hmm *h, double *out, const int *xs, int nx, const int *ys, int ny) {
    double acc[4] = {0, 0, 0, 0}, sum = 0.0;
    for (int i = 0; i < nx; i++)
        for (int x = 0; x < 4; x++)
            for (int y = 0; y < 4; y++) acc[x] += emi(h, xs[i], x, y);
    for (int i = 0; i < ny; i++)
        for (int x = 0; x < 4; x++)
            for (int y = 0; y < 4; y++) acc[y] += emi(h, ys[i], x, y);
    for (int i = 0; i < 4; i++) sum += acc[i];
    for (int i = 0; i < 4; i++) out[i] = log(acc[i] / sum);
}

static void exchange(double *a, double *b) {
    double t = *a;
    *a = *b;
    *b = t;
}

/* keep "long" the state with the larger extend probability, stateMachine.c:544-550 */
static void order_short_long(double *extS, double *extL, double *fromS, double *fromL, double *openS, double *openL,
                             double *swS, double *swL) {
    if (*extS > *extL) {
        exchange(extS, extL);
        exchange(fromS, fromL);
        exchange(openS, openL);
        exchange(swS, swL);
    }
}

int cpecan_model_from_hmm(cpecan_model *m, const cpecan_hmm *h) { /* stateMachine.c:529-620, 747-819 */
    if (!m || !h) return CPECAN_EINVAL;
    if (cpecan_model_default(m, h->type) != CPECAN_OK) return CPECAN_EINVAL;
    const int sym = (h->type == CPECAN_FIVE_STATE || h->type == CPECAN_THREE_STATE);
    const int xs5[2] = {kShortX, kLongX}, ys5[2] = {kShortY, kLongY}, xs3[1] = {kShortX}, ys3[1] = {kShortY};
    m->matchContinue = log(tr(h, kMatch, kMatch));
    /* match emissions: stateMachine.c:298-317 */
    for (int x = 0; x < 4; x++) {
        for (int y = 0; y < 4; y++) {
            m->emissionMatch[x * 4 + y] =
                (!sym || x == y) ? log(emi(h, kMatch, x, y)) : log((emi(h, kMatch, x < y ? x : y, x < y ? y : x) +
                                                                    emi(h, kMatch, x < y ? y : x, x < y ? x : y)) / 2.0);
        }
    }
    if (h->type == CPECAN_FIVE_STATE) {
        m->matchFromShortGapX = log((tr(h, kShortX, kMatch) + tr(h, kShortY, kMatch)) / 2);
        m->matchFromLongGapX = log((tr(h, kLongX, kMatch) + tr(h, kLongY, kMatch)) / 2);
        m->gapShortOpenX = log((tr(h, kMatch, kShortX) + tr(h, kMatch, kShortY)) / 2);
        m->gapShortExtendX = log((tr(h, kShortX, kShortX) + tr(h, kShortY, kShortY)) / 2);
        m->gapShortSwitchToX = log((tr(h, kShortX, kShortY) + tr(h, kShortY, kShortX)) / 2);
        m->gapLongOpenX = log((tr(h, kMatch, kLongX) + tr(h, kMatch, kLongY)) / 2);
        m->gapLongExtendX = log((tr(h, kLongX, kLongX) + tr(h, kLongY, kLongY)) / 2);
        m->gapLongSwitchToX = log((tr(h, kLongX, kLongY) + tr(h, kLongY, kLongX)) / 2);
        order_short_long(&m->gapShortExtendX, &m->gapLongExtendX, &m->matchFromShortGapX, &m->matchFromLongGapX,
                         &m->gapShortOpenX, &m->gapLongOpenX, &m->gapShortSwitchToX, &m->gapLongSwitchToX);
        m->matchFromShortGapY = m->matchFromShortGapX;
        m->matchFromLongGapY = m->matchFromLongGapX;
        m->gapShortOpenY = m->gapShortOpenX;
        m->gapShortExtendY = m->gapShortExtendX;
        m->gapShortSwitchToY = m->gapShortSwitchToX;
        m->gapLongOpenY = m->gapLongOpenX;
        m->gapLongExtendY = m->gapLongExtendX;
        m->gapLongSwitchToY = m->gapLongSwitchToX;
        gap_emissions(h, m->emissionGapX, xs5, 2, ys5, 2);
        gap_emissions(h, m->emissionGapY, xs5, 2, ys5, 2);
    } else if (h->type == CPECAN_FIVE_STATE_ASYM) {
        m->matchFromShortGapX = log(tr(h, kShortX, kMatch));
        m->matchFromLongGapX = log(tr(h, kLongX, kMatch));
        m->gapShortOpenX = log(tr(h, kMatch, kShortX));
        m->gapShortExtendX = log(tr(h, kShortX, kShortX));
        m->gapShortSwitchToX = log(tr(h, kShortY, kShortX));
        m->gapLongOpenX = log(tr(h, kMatch, kLongX));
        m->gapLongExtendX = log(tr(h, kLongX, kLongX));
        m->gapLongSwitchToX = log(tr(h, kLongY, kLongX));
        order_short_long(&m->gapShortExtendX, &m->gapLongExtendX, &m->matchFromShortGapX, &m->matchFromLongGapX,
                         &m->gapShortOpenX, &m->gapLongOpenX, &m->gapShortSwitchToX, &m->gapLongSwitchToX);
        m->matchFromShortGapY = log(tr(h, kShortY, kMatch));
        m->matchFromLongGapY = log(tr(h, kLongY, kMatch));
        m->gapShortOpenY = log(tr(h, kMatch, kShortY));
        m->gapShortExtendY = log(tr(h, kShortY, kShortY));
        m->gapShortSwitchToY = log(tr(h, kShortX, kShortY));
        m->gapLongOpenY = log(tr(h, kMatch, kLongY));
        m->gapLongExtendY = log(tr(h, kLongY, kLongY));
        m->gapLongSwitchToY = log(tr(h, kLongX, kLongY));
        order_short_long(&m->gapShortExtendY, &m->gapLongExtendY, &m->matchFromShortGapY, &m->matchFromLongGapY,
                         &m->gapShortOpenY, &m->gapLongOpenY, &m->gapShortSwitchToY, &m->gapLongSwitchToY);
        gap_emissions(h, m->emissionGapX, xs5, 2, NULL, 0);
        gap_emissions(h, m->emissionGapY, NULL, 0, ys5, 2);
    } else if (h->type == CPECAN_THREE_STATE) {
        m->matchFromShortGapX = m->matchFromShortGapY = log((tr(h, kShortX, kMatch) + tr(h, kShortY, kMatch)) / 2.0);
        m->gapShortOpenX = m->gapShortOpenY = log((tr(h, kMatch, kShortX) + tr(h, kMatch, kShortY)) / 2.0);
        m->gapShortExtendX = m->gapShortExtendY = log((tr(h, kShortX, kShortX) + tr(h, kShortY, kShortY)) / 2.0);
        m->gapShortSwitchToX = m->gapShortSwitchToY = log((tr(h, kShortY, kShortX) + tr(h, kShortX, kShortY)) / 2.0);
        gap_emissions(h, m->emissionGapX, xs3, 1, ys3, 1);
        gap_emissions(h, m->emissionGapY, xs3, 1, ys3, 1);
    } else {
        m->matchFromShortGapX = log(tr(h, kShortX, kMatch));
        m->matchFromShortGapY = log(tr(h, kShortY, kMatch));
        m->gapShortOpenX = log(tr(h, kMatch, kShortX));
        m->gapShortOpenY = log(tr(h, kMatch, kShortY));
        m->gapShortExtendX = log(tr(h, kShortX, kShortX));
        m->gapShortExtendY = log(tr(h, kShortY, kShortY));
        m->gapShortSwitchToX = log(tr(h, kShortY, kShortX));
        m->gapShortSwitchToY = log(tr(h, kShortX, kShortY));
        gap_emissions(h, m->emissionGapX, xs3, 1, NULL, 0);
        gap_emissions(h, m->emissionGapY, NULL, 0, ys3, 1);
    }
    return CPECAN_OK;
}

int cpecan_params_default(cpecan_params *p) { /* pairwiseAligner.c:1334-1348 */
    if (!p) return CPECAN_EINVAL;
    memset(p, 0, sizeof *p);
    p->threshold = 0.01;
    p->minDiagsBetweenTraceBack = 1000;
    p->traceBackDiagonals = 40;
    p->diagonalExpansion = 20;
    p->splitMatrixBiggerThanThis = (int64_t)3000 * 3000;
    p->dynamicAnchorExpansion = 0;
    return CPECAN_OK;
}

/* Kernel-side view of the model: priors (stateMachine.c:401-448, 648-687) and N-padded emissions (:351-366). */
static void kernel_model(const cpecan_model *m, double threshold, CpkModel *k) {
    const double ninf = -INFINITY;
    memset(k, 0, sizeof *k);
    k->type = m->type;
    k->nStates = is_five(m->type) ? 5 : 3;
    k->threshold = threshold;
    for (int s = 0; s < CPK_MAX_STATES; s++) k->start[s] = k->raggedStart[s] = k->end[s] = k->raggedEnd[s] = ninf;
    k->start[kMatch] = 0.0;
    k->end[kMatch] = m->matchContinue;
    k->end[kShortX] = m->matchFromShortGapX;
    k->end[kShortY] = m->matchFromShortGapY;
    if (k->nStates == 5) {
        k->raggedStart[kLongX] = k->raggedStart[kLongY] = 0.0;
        k->end[kLongX] = m->matchFromLongGapX;
        k->end[kLongY] = m->matchFromLongGapY;
        k->raggedEnd[kMatch] = m->gapLongOpenX;
        k->raggedEnd[kShortX] = m->gapLongOpenX;
        k->raggedEnd[kShortY] = m->gapLongOpenY;
        k->raggedEnd[kLongX] = m->gapLongExtendX;
        k->raggedEnd[kLongY] = m->gapLongExtendY;
    } else {
        k->raggedStart[kShortX] = k->raggedStart[kShortY] = 0.0;
        k->raggedEnd[kMatch] = (m->gapShortOpenX + m->gapShortOpenY) / 2.0;
        k->raggedEnd[kShortX] = m->gapShortExtendX;
        k->raggedEnd[kShortY] = m->gapShortExtendY;
    }
    k->matchContinue = m->matchContinue;
    k->matchFromShortX = m->matchFromShortGapX;
    k->matchFromShortY = m->matchFromShortGapY;
    k->matchFromLongX = m->matchFromLongGapX;
    k->matchFromLongY = m->matchFromLongGapY;
    k->shortOpenX = m->gapShortOpenX;
    k->shortOpenY = m->gapShortOpenY;
    k->shortExtendX = m->gapShortExtendX;
    k->shortExtendY = m->gapShortExtendY;
    k->shortSwitchToX = m->gapShortSwitchToX;
    k->shortSwitchToY = m->gapShortSwitchToY;
    k->longOpenX = m->gapLongOpenX;
    k->longOpenY = m->gapLongOpenY;
    k->longExtendX = m->gapLongExtendX;
    k->longExtendY = m->gapLongExtendY;
    for (int x = 0; x < 5; x++) {
        k->gapXEm[x] = x == CPK_SYM_N ? -1.386294361 : m->emissionGapX[x];
        k->gapYEm[x] = x == CPK_SYM_N ? -1.386294361 : m->emissionGapY[x];
        for (int y = 0; y < 5; y++)
            k->matchEm[x * 5 + y] = (x == CPK_SYM_N || y == CPK_SYM_N) ? -2.772588722 : m->emissionMatch[x * 4 + y];
    }
}

/* ------------------------------------------------------------------------------------------------
 * band and split geometry (integers only)
 * ---------------------------------------------------------------------------------------------- */
/* CPECAN_TRACE_HOST=1: wall time of the host stages of upload and download on stderr (diagnostic) */
static double now_ms(void) { return 1e3 * omp_get_wtime(); }
static int trace_host(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("CPECAN_TRACE_HOST");
        v = e && atoi(e) != 0;
    }
    return v;
}
static int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }
static int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }

/* band_construct / band_constructDynamic (pairwiseAligner.c:128-234) through the shared iterator (cpecan_band.inl). */
int cpecan_band(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY, int64_t expansion, int dynamic,
                int64_t *out) {
    const int64_t n = lX + lY + 1;
    if (n <= 0 || !out || nAnchors < 0) return CPECAN_EINVAL;
    if (nAnchors > 0 && !anchors) return CPECAN_EINVAL;
    int32_t *a32 = malloc(sizeof(int32_t) * 3 * (size_t)(nAnchors ? nAnchors : 1)); /* the iterator reads 32-bit values */
    if (!a32) return CPECAN_ENOMEM;
    int rc = CPECAN_OK;
    for (int64_t i = 0; i < 3 * nAnchors; i++) {
        if (anchors[i] < INT32_MIN || anchors[i] > INT32_MAX) rc = CPECAN_EINVAL;
        a32[i] = (int32_t)anchors[i];
    }
    CpkBandIter it;
    if (rc == CPECAN_OK && cpk_band_init(&it, a32, 3, nAnchors, lX, lY, expansion, dynamic)) rc = CPECAN_EINVAL;
    for (int64_t d = 0; rc == CPECAN_OK && d < n; d++) {
        int64_t lo, hi;
        if (cpk_band_next(&it, d, &lo, &hi)) {
            rc = CPECAN_EINVAL;
            break;
        }
        out[3 * d] = d;
        out[3 * d + 1] = lo;
        out[3 * d + 2] = hi;
    }
    free(a32);
    return rc;
}

/* getSplitPoints, pairwiseAligner.c:1206-1257.  A gap between consecutive anchors whose matrix exceeds
 * maxMatrixSize closes the current rectangle half-way into the gap (at most sqrt(max) deep) and opens the
 * next one the same distance before the following anchor. */
typedef struct {
    int64_t x1, y1; /* start of the rectangle being grown */
    int64_t *out;
    int64_t count;
    int64_t limit, depth;
} SplitState;

static int cut_if_large(SplitState *s, int64_t fromX, int64_t fromY, int64_t toX, int64_t toY, int suppress) {
    const int64_t gx = toX - fromX, gy = toY - fromY;
    if (gx * gy <= s->limit) return 0;
    const int64_t hx = imin(gx / 2, s->depth), hy = imin(gy / 2, s->depth);
    if (!suppress) {
        int64_t *r = s->out + 4 * s->count++;
        r[0] = s->x1;
        r[1] = s->y1;
        r[2] = fromX + hx;
        r[3] = fromY + hy;
    }
    s->x1 = toX - hx;
    s->y1 = toY - hy;
    return 1;
}

int64_t cpecan_split_points(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                            int raggedLeft, int raggedRight, int64_t *out) {
    if (lX < 0 || lY < 0 || nAnchors < 0 || !out) return CPECAN_EINVAL;
    SplitState s = {0, 0, out, 0, maxMatrixSize, (int64_t)sqrt((double)maxMatrixSize)};
    int64_t fromX = 0, fromY = 0;
    for (int64_t i = 0; i < nAnchors; i++) {
        const int64_t ax = anchors[3 * i], ay = anchors[3 * i + 1];
        if (ax < fromX || ay < fromY || ax >= lX || ay >= lY) return CPECAN_EINVAL; /* :1240-1243 */
        cut_if_large(&s, fromX, fromY, ax, ay, raggedLeft && i == 0);
        fromX = ax + 1;
        fromY = ay + 1;
    }
    const int cutAtEnd = cut_if_large(&s, fromX, fromY, lX, lY, raggedLeft && nAnchors == 0);
    if (!cutAtEnd || !raggedRight) {
        int64_t *r = out + 4 * s.count++;
        r[0] = s.x1;
        r[1] = s.y1;
        r[2] = lX;
        r[3] = lY;
    }
    return s.count;
}

/* ------------------------------------------------------------------------------------------------
 * the batch object
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t problem;   /* owning problem */
    int64_t x1, y1;    /* offset of the region inside the problem (coordinate correction, :1411-1418) */
    int64_t lX, lY;
    int64_t seqXOff, seqYOff; /* into the symbol blob */
    int64_t anchorOff, nAnchors;
    int64_t runOff, nRuns; /* its runs in cpecan_batch.runs (a batch that keeps its anchors as runs) */
    int raggedLeft, raggedRight;
    /* filled by upload */
    int64_t cells;
    int64_t devIndex; /* position in the cost-sorted device array */
} HostRegion;

typedef struct {
    int64_t firstRegion, nRegions;
    int32_t *triples[4]; /* lists 0..2 into cpecan_batch.results, list 3 (MEA alignment) into postMea/postShift; not owned */
    int64_t nTriples[4];
    int64_t lX, lY;          /* of the whole problem */
    int64_t charX, charY;    /* raw upper-case sequences in cpecan_batch.chars */
    double scores[CPK_POST_SCORES]; /* byPosterior, byPosteriorIgnoringGaps, MEA alignment score, byIdentity, ...IgnoringGaps */
} HostProblem;

struct cpecan_batch {
    cpecan_model model;
    cpecan_params params;
    int emit, device, debug;
    int frozen, ran, downloaded;
    HostProblem *problems;
    int64_t nProblems, capProblems;
    HostRegion *regions;
    int64_t nRegions, capRegions;
    uint8_t *symbols;
    int64_t nSymbols, capSymbols;
    int32_t *anchors; /* cpk_anchor_t: anchorStride values per anchor, coordinates relative to the region */
    int anchorStride; /* 2: (x, y); 3: (x, y, expansion) for per-anchor expansions */
    int64_t nAnchorVals, capAnchorVals;
    /* Round 4: a batch of fixed expansion whose first problems arrive as runs (cpecan_batch_add_many_runs) KEEPS them as runs
     * -- (x, y, length, index of the run's first anchor among the batch's anchors) -- and never holds one anchor per column
     * on the host: planning walks the runs (cpk_band_init_runs) and the device expands them in front of its table builder.
     * BASELINE config 4: 6e7 anchors, 0.5 GB written by every add and read again by every band walk.  runForm: -1 not
     * decided yet, 0 anchors, 1 runs.  nAnchorVals still counts the expanded anchors. */
    int32_t *runs;
    int64_t nRunVals, capRunVals;
    int runForm;
    /* frozen state */
    CpkRegion *devRegions; /* cost-sorted */
    int64_t *devToHost;    /* devRegions[i] describes regions[devToHost[i]] */
    CpkSegment *segs;
    int64_t nSegs;
    int64_t nDiags;
    CpkGeometry geo;
    int64_t outTriples; /* per list */
    int nLists;
    int64_t dbgCells, dbgDiags;
    CpkDevice *dev;
    double *forward; /* [nRegions] in device order, FORWARD emitter */
    int32_t *results; /* every emitted triple of the batch, list-ordered: [list][problem][triple] */
    uint8_t *chars;   /* raw upper-case sequences (leftShiftAlignment compares letters, not symbols) */
    int64_t nChars, capChars;
    int postFlags;    /* CPECAN_POST_* applied by download */
    double postGapGamma;
    float postMatchGamma; /* ORDERED: matchGamma of cPecanRealign.c:355 */
    int32_t *postMea, *postShift;
    cpecan_stats stats;
    /* cpecan_batch_download_begin / _end */
    pthread_t dlThread;
    int dlActive, dlResult;
    char dlError[256]; /* the helper's error text: cpk_last_error is per thread */
};

/* The batch's arrays live in blocks of the host pool (cpk_host_alloc: pinned and recycled when there is a GPU). */
static int grow(void **p, int64_t *cap, int64_t need, size_t elem) {
    if (need <= *cap) return 0;
    int64_t c = *cap ? *cap : 16;
    while (c < need) c *= 2;
    void *q = cpk_host_grow(*p, (size_t)*cap * elem, (size_t)c * elem);
    if (!q) return -1;
    *p = q;
    *cap = c;
    return 0;
}
static void *host_zalloc(size_t count, size_t elem) {
    void *q = cpk_host_alloc((count ? count : 1) * elem);
    if (q) memset(q, 0, (count ? count : 1) * elem);
    return q;
}

int64_t cpecan_anchors_from_alignment(const int64_t *ops, int64_t nOps, int64_t start1, int64_t start2, int64_t trim,
                                      int64_t expansion, const char *sX, int64_t lX, const char *sY, int64_t lY,
                                      int64_t *anchors) {
    if (nOps < 0 || (nOps > 0 && !ops) || !anchors || trim < 0) return CPECAN_EINVAL;
    const int filter = sX && sY;
    int64_t x = start1, y = start2, n = 0;
    for (int64_t i = 0; i < nOps; i++) {
        const int64_t type = ops[2 * i], len = ops[2 * i + 1];
        if (len < 0 || type < CPECAN_OP_MATCH || type > CPECAN_OP_INDEL_Y) return CPECAN_EINVAL;
        if (type == CPECAN_OP_MATCH) {
            for (int64_t l = trim; l < len - trim; l++) { /* :987-989 */
                const int64_t ax = x + l, ay = y + l;
                if (filter) {
                    if (ax < 0 || ay < 0 || ax >= lX || ay >= lY) return CPECAN_EINVAL;
                    int cx = (unsigned char)sX[ax], cy = (unsigned char)sY[ay]; /* toupper, "C" locale */
                    cx -= (cx >= 'a' && cx <= 'z') ? 'a' - 'A' : 0;
                    cy -= (cy >= 'a' && cy <= 'z') ? 'a' - 'A' : 0;
                    if (cx != cy || cx == 'N') continue; /* cPecanRealign.c:277-281 */
                }
                anchors[3 * n] = ax;
                anchors[3 * n + 1] = ay;
                anchors[3 * n + 2] = expansion;
                n++;
            }
        }
        if (type != CPECAN_OP_INDEL_Y) x += len; /* :991-996 */
        if (type != CPECAN_OP_INDEL_X) y += len;
    }
    return n;
}

/* The same anchors as runs (cpecan_batch_add_many_runs): the kept columns of a match operation that follow each other
 * on a matrix diagonal -- a mismatched column or the end of the operation ends a run -- as (x, y, length, expansion). */
int64_t cpecan_anchor_runs_from_alignment(const int64_t *ops, int64_t nOps, int64_t start1, int64_t start2, int64_t trim,
                                          int64_t expansion, const char *sX, int64_t lX, const char *sY, int64_t lY,
                                          int64_t *runs, int64_t cap) {
    if (nOps < 0 || (nOps > 0 && !ops) || trim < 0 || cap < 0 || (cap > 0 && !runs)) return CPECAN_EINVAL;
    const int filter = sX && sY;
    int64_t x = start1, y = start2, n = 0, rx = 0, ry = 0, len = 0; /* the open run: (rx, ry) .. len columns, len == 0: none */
    for (int64_t i = 0; i < nOps; i++) {
        const int64_t type = ops[2 * i], opLen = ops[2 * i + 1];
        if (opLen < 0 || type < CPECAN_OP_MATCH || type > CPECAN_OP_INDEL_Y) return CPECAN_EINVAL;
        if (type == CPECAN_OP_MATCH) {
            for (int64_t l = trim; l < opLen - trim; l++) {
                const int64_t ax = x + l, ay = y + l;
                if (filter) {
                    if (ax < 0 || ay < 0 || ax >= lX || ay >= lY) return CPECAN_EINVAL;
                    int cx = (unsigned char)sX[ax], cy = (unsigned char)sY[ay];
                    cx -= (cx >= 'a' && cx <= 'z') ? 'a' - 'A' : 0;
                    cy -= (cy >= 'a' && cy <= 'z') ? 'a' - 'A' : 0;
                    if (cx != cy || cx == 'N') continue;
                }
                if (len > 0 && ax == rx + len && ay == ry + len) {
                    len++;
                    continue;
                }
                if (len > 0) {
                    if (n < cap) {
                        runs[4 * n] = rx;
                        runs[4 * n + 1] = ry;
                        runs[4 * n + 2] = len;
                        runs[4 * n + 3] = expansion;
                    }
                    n++;
                }
                rx = ax;
                ry = ay;
                len = 1;
            }
        }
        if (type != CPECAN_OP_INDEL_Y) x += opLen;
        if (type != CPECAN_OP_INDEL_X) y += opLen;
    }
    if (len > 0) {
        if (n < cap) {
            runs[4 * n] = rx;
            runs[4 * n + 1] = ry;
            runs[4 * n + 2] = len;
            runs[4 * n + 3] = expansion;
        }
        n++;
    }
    return n;
}

/* filterToRemoveOverlap (impl/pairwiseAligner.c:1095-1135): one backward and one forward pass over the sorted pairs */
int64_t cpecan_filter_to_remove_overlap(const int64_t *pairs, int64_t n, int64_t *out) {
    if (n < 0 || (n > 0 && (!pairs || !out))) return CPECAN_EINVAL;
    uint8_t *below = malloc((size_t)(n ? n : 1)); /* strictly below every later pair in both coordinates */
    if (!below) return CPECAN_ENOMEM;
    int64_t mX = INT64_MAX, mY = INT64_MAX;
    for (int64_t i = n - 1; i >= 0; i--) {
        const int64_t x = pairs[3 * i], y = pairs[3 * i + 1];
        below[i] = x < mX && y < mY;
        mX = x < mX ? x : mX;
        mY = y < mY ? y : mY;
        if (i > 0 && (pairs[3 * i - 3] > x || (pairs[3 * i - 3] == x && pairs[3 * i - 2] > y))) { /* asserted sorted, :1124-1127 */
            free(below);
            cpk_set_error("filterToRemoveOverlap: the pairs are not sorted by x, then y");
            return CPECAN_EINVAL;
        }
    }
    int64_t count = 0;
    mX = INT64_MIN;
    mY = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        const int64_t x = pairs[3 * i], y = pairs[3 * i + 1];
        if (below[i] && x > mX && y > mY) {
            out[3 * count] = x;
            out[3 * count + 1] = y;
            out[3 * count + 2] = pairs[3 * i + 2];
            count++;
        }
        mX = x > mX ? x : mX;
        mY = y > mY ? y : mY;
    }
    free(below);
    return count;
}

int cpecan_device_count(void) { return cpk_device_count(); }
int cpecan_current_device(void) { return cpk_current_device(); }
int64_t cpecan_cache_trim(int device) { return cpk_cache_trim(device); }

int cpecan_ref_cells(const cpecan_model *model, int mode, const cpecan_cell_op *ops, int64_t n, double *cells,
                     int64_t nDoubles, double total) {
    /* the kernel's operation count is an int */
    if (!model || !ops || !cells || n < 0 || n > INT32_MAX || nDoubles < 0 || mode < CPECAN_CELLS_FORWARD || mode > CPECAN_CELLS_POSTERIOR ||
        model->type < CPECAN_FIVE_STATE || model->type > CPECAN_THREE_STATE_ASYM) {
        cpk_set_error("cpecan_ref_cells: invalid arguments");
        return CPECAN_EINVAL;
    }
    const int S = is_five(model->type) ? 5 : 3;
    for (int64_t i = 0; i < n; i++) { /* every offset names a whole cell inside the buffer: the kernel trusts them */
        const int32_t o[4] = {ops[i].cur, ops[i].lower, ops[i].middle, ops[i].upper};
        for (int f = 0; f < 4; f++) {
            const int need = mode == CPECAN_CELLS_POSTERIOR ? (f == 2 ? 0 : 1) : S;
            const int may_be_null = mode != CPECAN_CELLS_POSTERIOR && f > 0;
            if (need == 0 || (may_be_null && o[f] < 0)) continue;
            if (o[f] < 0 || (int64_t)o[f] + need > nDoubles) {
                cpk_set_error("cpecan_ref_cells: operation %lld names a cell outside the buffer", (long long)i);
                return CPECAN_EINVAL;
            }
        }
        if (mode != CPECAN_CELLS_POSTERIOR && (ops[i].cX < 0 || ops[i].cX > 4 || ops[i].cY < 0 || ops[i].cY > 4)) {
            cpk_set_error("cpecan_ref_cells: operation %lld has a symbol outside 0..4", (long long)i);
            return CPECAN_EINVAL;
        }
    }
    CpkModel km;
    kernel_model(model, 0.0, &km);
    return cpk_ref_cells(cpk_current_device(), &km, mode, (const CpkCellOp *)ops, n, cells, nDoubles, total);
}
const char *cpecan_last_error(void) { return cpk_last_error(); }

int cpecan_batch_create(cpecan_batch **out, const cpecan_model *model, const cpecan_params *params, int emit,
                        int device) {
    if (!out || !model || !params) return CPECAN_EINVAL;
    if (!(is_five(model->type) || is_three(model->type))) return CPECAN_EINVAL;
    if (emit != CPECAN_EMIT_MATCH && emit != CPECAN_EMIT_INDEL && emit != CPECAN_EMIT_FORWARD && emit != CPECAN_EMIT_EXPECT) {
        cpk_set_error("unknown emitter %d", emit);
        return CPECAN_EINVAL;
    }
    /* preconditions of getPosteriorProbsWithBanding, pairwiseAligner.c:761-765 */
    if (params->traceBackDiagonals < 1 || params->diagonalExpansion < 0 || params->diagonalExpansion % 2 != 0 ||
        params->minDiagsBetweenTraceBack < 2 || params->traceBackDiagonals + 1 >= params->minDiagsBetweenTraceBack ||
        !(params->threshold >= 0.0 && params->threshold <= 1.0)) {
        cpk_set_error("invalid banding parameters");
        return CPECAN_EINVAL;
    }
    /* The GPU is bound at upload time: adding problems and planning (bands, schedules) are host-only integer work. */
    cpecan_batch *b = calloc(1, sizeof *b);
    if (!b) return CPECAN_ENOMEM;
    b->model = *model;
    b->params = *params;
    b->emit = emit;
    b->device = device;
    b->dev = NULL;
    b->nLists = emit == CPECAN_EMIT_INDEL ? 3 : 1;
    b->anchorStride = params->dynamicAnchorExpansion ? 3 : 2;
    b->runForm = -1;
    b->postMatchGamma = 0.85f; /* cPecanRealign.c:355 */
    *out = b;
    return CPECAN_OK;
}

static void free_results(cpecan_batch *b) {
    for (int64_t i = 0; i < b->nProblems; i++)
        for (int l = 0; l < 4; l++) {
            b->problems[i].triples[l] = NULL;
            b->problems[i].nTriples[l] = 0;
        }
    cpk_host_free(b->results);
    cpk_host_free(b->postMea);
    cpk_host_free(b->postShift);
    b->results = b->postMea = b->postShift = NULL;
}

void cpecan_batch_destroy(cpecan_batch *b) {
    if (!b) return;
    if (b->dlActive) (void)cpecan_batch_download_end(b);
    free_results(b);
    cpk_device_destroy(b->dev);
    cpk_host_free(b->problems);
    cpk_host_free(b->regions);
    cpk_host_free(b->symbols);
    cpk_host_free(b->anchors);
    cpk_host_free(b->runs);
    cpk_host_free(b->devRegions);
    cpk_host_free(b->devToHost);
    cpk_host_free(b->segs);
    free(b->forward);
    cpk_host_free(b->chars);
    free(b);
}

/* A download running on the helper thread (cpecan_batch_download_begin) rewrites the batch's result arrays, segment
 * table and device order: until cpecan_batch_download_end every other entry point refuses the batch. */
static _Thread_local const cpecan_batch *tl_helperOf = NULL; /* set by the helper thread itself: the batch it downloads */
static int dl_busy(const cpecan_batch *b) {
    /* dlActive is written before pthread_create and after pthread_join only, both of which order it against the helper;
     * the helper recognises itself by its own thread-local mark, not by comparing ids with a field pthread_create is
     * still writing */
    if (!b || !b->dlActive || tl_helperOf == b) return 0;
    cpk_set_error("the batch is being downloaded on its helper thread: call cpecan_batch_download_end first");
    return 1;
}

int cpecan_batch_set_debug(cpecan_batch *b, int on) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || b->frozen) return CPECAN_ESTATE;
    b->debug = on != 0;
    return CPECAN_OK;
}

/* symbol_convertCharToSymbol (pairwiseAligner.c:317-334) and toupper as byte tables: both run over every base added */
static uint8_t g_symbolOf[256], g_upperOf[256];
/* Filled ONCE, when the library is loaded.  (Round 1 refilled them in every cpecan_batch_create, resetting every entry to
 * N first: a batch created on one host thread while another thread converted bases saw A/C/G/T as N for an instant --
 * the intermittent mismatch of the two-host-thread test, found in round 2.) */
__attribute__((constructor)) static void init_byte_tables(void) {
    for (int c = 0; c < 256; c++) {
        g_symbolOf[c] = CPK_SYM_N;
        g_upperOf[c] = (uint8_t)((c >= 'a' && c <= 'z') ? c - 'a' + 'A' : c);
    }
    g_symbolOf['A'] = g_symbolOf['a'] = 0;
    g_symbolOf['C'] = g_symbolOf['c'] = 1;
    g_symbolOf['G'] = g_symbolOf['g'] = 2;
    g_symbolOf['T'] = g_symbolOf['t'] = 3;
}

/* Writes N + symbols + N at dst[at..] (so that index x addresses base x-1 and x = 0 / x = l+1 read as N); returns the
 * offset behind them. */
static int64_t put_symbols(uint8_t *dst, int64_t at, const char *s, int64_t l) {
    dst[at] = CPK_SYM_N;
    for (int64_t i = 0; i < l; i++) dst[at + 1 + i] = g_symbolOf[(unsigned char)s[i]];
    dst[at + l + 1] = CPK_SYM_N;
    return at + l + 2;
}

/* What one problem adds to the batch's arrays; filled by the counting pass of cpecan_batch_add_many. */
typedef struct {
    int64_t nRects, symbolBytes, nAnchorsKept, nRunsKept;
} AddCount;

/* One problem of an add call: its anchors as triples (x, y, expansion), one per anchor, or -- `runs` -- as quadruples
 * (x, y, length, expansion), one per run of diagonal neighbours (cpecan_batch_add_many_runs). */
typedef struct {
    const char *sX;
    int64_t lX;
    const char *sY;
    int64_t lY;
    const int64_t *anchors;
    int64_t nAnchors; /* entries: anchors or runs */
    int32_t raggedLeft, raggedRight;
    int runs;
} ProblemView;

/* getSplitPoints over runs: between two anchors of a run the gap is empty (0 x 0 cells), so only the gaps in front of a
 * run and behind the last one can be cut -- the same rectangles as cpecan_split_points on the expanded anchors. */
static int64_t split_points_runs(const int64_t *runs, int64_t nRuns, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                                 int raggedLeft, int raggedRight, int64_t *out) {
    SplitState s = {0, 0, out, 0, maxMatrixSize, (int64_t)sqrt((double)maxMatrixSize)};
    int64_t fromX = 0, fromY = 0;
    for (int64_t i = 0; i < nRuns; i++) {
        const int64_t ax = runs[4 * i], ay = runs[4 * i + 1], len = runs[4 * i + 2];
        if (ax < fromX || ay < fromY || len < 1 || ax + len > lX || ay + len > lY) return CPECAN_EINVAL;
        cut_if_large(&s, fromX, fromY, ax, ay, raggedLeft && i == 0);
        fromX = ax + len;
        fromY = ay + len;
    }
    const int cutAtEnd = cut_if_large(&s, fromX, fromY, lX, lY, raggedLeft && nRuns == 0);
    if (!cutAtEnd || !raggedRight) {
        int64_t *r = out + 4 * s.count++;
        r[0] = s.x1;
        r[1] = s.y1;
        r[2] = lX;
        r[3] = lY;
    }
    return s.count;
}

/* The split rectangles of one problem (getSplitPoints semantics, pairwiseAligner.c:1230-1271) into *rects, grown as
 * needed.  Returns their number or < 0. */
static int64_t problem_rects(const cpecan_batch *b, const ProblemView *it, int64_t **rects, int64_t *cap) {
    if (4 * (it->nAnchors + 2) > *cap) { /* the caller's scratch: plain heap memory, freed with free() */
        const int64_t c = 8 * (it->nAnchors + 2);
        int64_t *q = realloc(*rects, sizeof(int64_t) * (size_t)c);
        if (!q) return CPECAN_ENOMEM;
        *rects = q;
        *cap = c;
    }
    if (b->emit == CPECAN_EMIT_FORWARD) { /* computeForwardProbability never splits (pairwiseAligner.c:936-949) */
        (*rects)[0] = 0;
        (*rects)[1] = 0;
        (*rects)[2] = it->lX;
        (*rects)[3] = it->lY;
        return 1;
    }
    if (it->runs)
        return split_points_runs(it->anchors, it->nAnchors, it->lX, it->lY, b->params.splitMatrixBiggerThanThis, it->raggedLeft,
                                 it->raggedRight, *rects);
    return cpecan_split_points(it->anchors, it->nAnchors, it->lX, it->lY, b->params.splitMatrixBiggerThanThis, it->raggedLeft,
                               it->raggedRight, *rects);
}

static int problem_valid(const ProblemView *it) {
    if (it->lX < 0 || it->lY < 0 || it->nAnchors < 0 || (it->lX > 0 && !it->sX) || (it->lY > 0 && !it->sY) ||
        (it->nAnchors > 0 && !it->anchors))
        return 0;
    if (it->lX + it->lY >= (int64_t)1 << 30) return 0;
    if (it->runs) { /* strictly increasing from the last anchor of one run to the first of the next */
        int64_t px = -1, py = -1;
        for (int64_t i = 0; i < it->nAnchors; i++) {
            const int64_t x = it->anchors[4 * i], y = it->anchors[4 * i + 1], len = it->anchors[4 * i + 2];
            if (len < 1 || x <= px || y <= py || x + len > it->lX || y + len > it->lY) return 0;
            if (it->anchors[4 * i + 3] < INT32_MIN || it->anchors[4 * i + 3] > INT32_MAX) return 0;
            px = x + len - 1;
            py = y + len - 1;
        }
        return 1;
    }
    /* anchors must be strictly increasing in both coordinates (pairwiseAligner.c:159-164) */
    for (int64_t i = 0; i < it->nAnchors; i++) {
        const int64_t x = it->anchors[3 * i], y = it->anchors[3 * i + 1];
        if (x < 0 || y < 0 || x >= it->lX || y >= it->lY) return 0;
        if (it->anchors[3 * i + 2] < INT32_MIN || it->anchors[3 * i + 2] > INT32_MAX) return 0; /* kept as 32-bit values */
        if (i > 0 && (x <= it->anchors[3 * (i - 1)] || y <= it->anchors[3 * (i - 1) + 1])) return 0;
    }
    return 1;
}

/* problem i of an add call in either form (the two public structs differ in what their anchor arrays hold) */
static ProblemView view_of(const void *items, int runs, int64_t i) {
    ProblemView v;
    if (runs) {
        const cpecan_problem_runs *p = (const cpecan_problem_runs *)items + i;
        v = (ProblemView){p->sX, p->lX, p->sY, p->lY, p->runs, p->nRuns, p->raggedLeft, p->raggedRight, 1};
    } else {
        const cpecan_problem *p = (const cpecan_problem *)items + i;
        v = (ProblemView){p->sX, p->lX, p->sY, p->lY, p->anchors, p->nAnchors, p->raggedLeft, p->raggedRight, 0};
    }
    return v;
}

static int64_t add_many(cpecan_batch *b, const void *items, int runs, int64_t n) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || b->frozen) return CPECAN_ESTATE;
    if (n < 0 || (n > 0 && !items)) return CPECAN_EINVAL;
    if (n == 0) return b->nProblems;
    if (b->runForm < 0) { /* decided by the batch's first problems.  CPECAN_KEEP_RUNS=0: one anchor per column on the host, as rounds 1-3 */
        const char *env = getenv("CPECAN_KEEP_RUNS");
        b->runForm = (runs && b->anchorStride == 2 && !(env && atoi(env) == 0)) ? 1 : 0;
    }
    const int keepRuns = b->runForm == 1;
    AddCount *cnt = malloc(sizeof(AddCount) * (size_t)n);
    int64_t *offs = malloc(sizeof(int64_t) * 4 * (size_t)n); /* first region, symbol byte, anchor triple, char of each */
    if (!cnt || !offs) {
        free(cnt);
        free(offs);
        return CPECAN_ENOMEM;
    }
    /* pass 1: what every problem needs (the problems are independent: OpenMP when there are enough of them) */
    int64_t firstBad = n;
    int bad = CPECAN_OK;
#pragma omp parallel num_threads(cpk_host_threads()) if (n >= 64)
    {
        int64_t *rects = NULL, cap = 0;
#pragma omp for schedule(dynamic, 32)
        for (int64_t i = 0; i < n; i++) {
            const ProblemView view = view_of(items, runs, i), *it = &view;
            int64_t nRects = problem_valid(it) ? problem_rects(b, it, &rects, &cap) : CPECAN_EINVAL;
            if (nRects < 0) {
#pragma omp critical(cpk_add)
                if (i < firstBad) {
                    firstBad = i;
                    bad = nRects == CPECAN_ENOMEM ? CPECAN_ENOMEM : CPECAN_EINVAL;
                }
                cnt[i].nRects = cnt[i].symbolBytes = cnt[i].nAnchorsKept = cnt[i].nRunsKept = 0;
                continue;
            }
            int64_t sym = 0, kept = 0;
            int64_t at = 0; /* entries (anchors or runs) handed to rectangles so far; a rectangle ends inside a gap, never inside a run */
            for (int64_t k = 0; k < nRects; k++) {
                sym += (rects[4 * k + 2] - rects[4 * k]) + (rects[4 * k + 3] - rects[4 * k + 1]) + 4; /* N + bases + N, twice */
                if (it->runs) {
                    for (; at < it->nAnchors && it->anchors[4 * at] + it->anchors[4 * at + 1] < rects[4 * k + 2] + rects[4 * k + 3]; at++)
                        kept += it->anchors[4 * at + 2];
                } else {
                    while (kept < it->nAnchors && it->anchors[3 * kept] + it->anchors[3 * kept + 1] < rects[4 * k + 2] + rects[4 * k + 3]) kept++;
                }
            }
            cnt[i].nRects = nRects;
            cnt[i].symbolBytes = sym;
            cnt[i].nAnchorsKept = kept;
            /* a batch that keeps runs: the problem's runs as they are, or -- anchors given one by one -- a run of one each */
            cnt[i].nRunsKept = it->runs ? at : kept;
        }
        free(rects);
    }
    if (firstBad < n) {
        free(cnt);
        free(offs);
        if (bad == CPECAN_EINVAL) cpk_set_error("problem %lld of the call: bad lengths or anchors", (long long)firstBad);
        return bad;
    }
    int64_t nRegions = b->nRegions, nSymbols = b->nSymbols, nAnchorVals = b->nAnchorVals, nChars = b->nChars, nRunVals = b->nRunVals;
    int64_t *runAt = keepRuns ? malloc(sizeof(int64_t) * (size_t)n) : NULL; /* first run value of each problem */
    if (keepRuns && !runAt) {
        free(cnt);
        free(offs);
        return CPECAN_ENOMEM;
    }
    for (int64_t i = 0; i < n; i++) {
        offs[4 * i] = nRegions;
        offs[4 * i + 1] = nSymbols;
        offs[4 * i + 2] = nAnchorVals;
        offs[4 * i + 3] = nChars;
        if (keepRuns) {
            runAt[i] = nRunVals;
            nRunVals += 4 * cnt[i].nRunsKept;
        }
        nRegions += cnt[i].nRects;
        nSymbols += cnt[i].symbolBytes;
        nAnchorVals += b->anchorStride * cnt[i].nAnchorsKept;
        const ProblemView view = view_of(items, runs, i);
        nChars += view.lX + view.lY;
    }
    if (grow((void **)&b->problems, &b->capProblems, b->nProblems + n, sizeof(HostProblem)) ||
        grow((void **)&b->regions, &b->capRegions, nRegions, sizeof(HostRegion)) ||
        (keepRuns ? grow((void **)&b->runs, &b->capRunVals, nRunVals, sizeof(int32_t))
                  : grow((void **)&b->anchors, &b->capAnchorVals, nAnchorVals, sizeof(int32_t))) ||
        grow((void **)&b->symbols, &b->capSymbols, nSymbols, 1) || grow((void **)&b->chars, &b->capChars, nChars + 1, 1) ||
        (keepRuns && nAnchorVals / 2 >= ((int64_t)1 << 31))) { /* (a run's first anchor is a 32-bit index) */
        free(cnt);
        free(offs);
        free(runAt);
        return CPECAN_ENOMEM;
    }
    /* pass 2: every problem writes its own slices */
    const int64_t firstProblem = b->nProblems;
    int oom = 0;
#pragma omp parallel num_threads(cpk_host_threads()) if (n >= 64)
    {
        int64_t *rects = NULL, cap = 0;
#pragma omp for schedule(dynamic, 32)
        for (int64_t i = 0; i < n; i++) {
            const ProblemView view = view_of(items, runs, i), *it = &view;
            const int64_t nRects = problem_rects(b, it, &rects, &cap);
            if (nRects != cnt[i].nRects) { /* only an allocation failure can change the answer */
#pragma omp atomic write
                oom = 1;
                continue;
            }
            HostProblem *pr = &b->problems[firstProblem + i];
            memset(pr, 0, sizeof *pr);
            pr->firstRegion = offs[4 * i];
            pr->nRegions = nRects;
            pr->lX = it->lX;
            pr->lY = it->lY;
            uint8_t *ch = b->chars + offs[4 * i + 3];
            pr->charX = offs[4 * i + 3];
            for (int64_t k = 0; k < it->lX; k++) ch[k] = g_upperOf[(unsigned char)it->sX[k]];
            pr->charY = pr->charX + it->lX;
            for (int64_t k = 0; k < it->lY; k++) ch[it->lX + k] = g_upperOf[(unsigned char)it->sY[k]];
            int64_t next = 0, symAt = offs[4 * i + 1], anchorAt = offs[4 * i + 2]; /* anchors go to regions in order, :1296-1308 */
            int64_t runValAt = keepRuns ? runAt[i] : 0;
            for (int64_t k = 0; k < nRects; k++) {
                const int64_t x1 = rects[4 * k], y1 = rects[4 * k + 1], x2 = rects[4 * k + 2], y2 = rects[4 * k + 3];
                HostRegion *r = &b->regions[offs[4 * i] + k];
                memset(r, 0, sizeof *r);
                r->problem = firstProblem + i;
                r->x1 = x1;
                r->y1 = y1;
                r->lX = x2 - x1;
                r->lY = y2 - y1;
                r->raggedLeft = (it->raggedLeft || k > 0) ? 1 : 0;
                r->raggedRight = (it->raggedRight || k < nRects - 1) ? 1 : 0;
                r->seqXOff = symAt;
                symAt = put_symbols(b->symbols, symAt, it->sX + x1, r->lX);
                r->seqYOff = symAt;
                symAt = put_symbols(b->symbols, symAt, it->sY + y1, r->lY);
                r->anchorOff = anchorAt / b->anchorStride;
                if (keepRuns) { /* the runs stay runs: 16 bytes each, whatever their length */
                    r->runOff = runValAt / 4;
                    for (; next < it->nAnchors; next++) {
                        const int64_t ax = it->runs ? it->anchors[4 * next] : it->anchors[3 * next];
                        const int64_t ay = it->runs ? it->anchors[4 * next + 1] : it->anchors[3 * next + 1];
                        if (ax + ay >= x2 + y2) break;
                        const int64_t len = it->runs ? it->anchors[4 * next + 2] : 1;
                        int32_t *q = b->runs + runValAt;
                        q[0] = (int32_t)(ax - x1);
                        q[1] = (int32_t)(ay - y1);
                        q[2] = (int32_t)len;
                        q[3] = (int32_t)(anchorAt / 2); /* its first anchor among the batch's (expanded on the device) */
                        runValAt += 4;
                        anchorAt += 2 * len;
                        r->nAnchors += len;
                        r->nRuns++;
                    }
                    continue;
                }
                if (it->runs) { /* a run becomes its anchors: the batch's own 8 (12) bytes each, never the API's 24 */
                    for (; next < it->nAnchors && it->anchors[4 * next] + it->anchors[4 * next + 1] < x2 + y2; next++) {
                        const int32_t rx = (int32_t)(it->anchors[4 * next] - x1), ry = (int32_t)(it->anchors[4 * next + 1] - y1);
                        const int64_t len = it->anchors[4 * next + 2];
                        int32_t *a = b->anchors + anchorAt;
                        if (b->anchorStride == 3) {
                            const int32_t e = (int32_t)it->anchors[4 * next + 3];
                            for (int64_t q = 0; q < len; q++, a += 3) {
                                a[0] = rx + (int32_t)q;
                                a[1] = ry + (int32_t)q;
                                a[2] = e;
                            }
                        } else {
                            for (int64_t q = 0; q < len; q++, a += 2) {
                                a[0] = rx + (int32_t)q;
                                a[1] = ry + (int32_t)q;
                            }
                        }
                        anchorAt += b->anchorStride * len;
                        r->nAnchors += len;
                    }
                    continue;
                }
                while (next < it->nAnchors && it->anchors[3 * next] + it->anchors[3 * next + 1] < x2 + y2) {
                    int32_t *a = b->anchors + anchorAt;
                    a[0] = (int32_t)(it->anchors[3 * next] - x1);
                    a[1] = (int32_t)(it->anchors[3 * next + 1] - y1);
                    if (b->anchorStride == 3) a[2] = (int32_t)it->anchors[3 * next + 2];
                    anchorAt += b->anchorStride;
                    r->nAnchors++;
                    next++;
                }
            }
        }
        free(rects);
    }
    free(cnt);
    free(offs);
    free(runAt);
    if (oom) return CPECAN_ENOMEM;
    b->nProblems += n;
    b->nRegions = nRegions;
    b->nSymbols = nSymbols;
    b->nAnchorVals = nAnchorVals;
    b->nRunVals = nRunVals;
    b->nChars = nChars;
    return firstProblem;
}

int64_t cpecan_batch_add_many(cpecan_batch *b, const cpecan_problem *items, int64_t n) { return add_many(b, items, 0, n); }
int64_t cpecan_batch_add_many_runs(cpecan_batch *b, const cpecan_problem_runs *items, int64_t n) { return add_many(b, items, 1, n); }

int64_t cpecan_anchor_runs(const int64_t *anchors, int64_t nAnchors, int64_t *out, int64_t cap) {
    if (nAnchors < 0 || cap < 0 || (nAnchors > 0 && !anchors) || (cap > 0 && !out)) return CPECAN_EINVAL;
    int64_t n = 0;
    for (int64_t i = 0; i < nAnchors;) {
        int64_t len = 1;
        while (i + len < nAnchors && anchors[3 * (i + len)] == anchors[3 * i] + len && anchors[3 * (i + len) + 1] == anchors[3 * i + 1] + len &&
               anchors[3 * (i + len) + 2] == anchors[3 * i + 2])
            len++;
        if (n < cap) {
            out[4 * n] = anchors[3 * i];
            out[4 * n + 1] = anchors[3 * i + 1];
            out[4 * n + 2] = len;
            out[4 * n + 3] = anchors[3 * i + 2];
        }
        n++;
        i += len;
    }
    return n;
}

int64_t cpecan_batch_add(cpecan_batch *b, const char *sX, int64_t lX, const char *sY, int64_t lY,
                         const int64_t *anchors, int64_t nAnchors, int raggedLeft, int raggedRight) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    const cpecan_problem it = {sX, lX, sY, lY, anchors, nAnchors, raggedLeft ? 1 : 0, raggedRight ? 1 : 0};
    return cpecan_batch_add_many(b, &it, 1);
}

typedef struct {
    int64_t cells;
    int64_t index;
    int cls; /* 0, 1, 2: narrow, at most 8 / 16 / 32 cells per diagonal (packed kernel); 3 ..: the wide classes */
} CostKey;

/* what planning keeps of a region's band walk */
typedef struct {
    int64_t maxW, liveMax, fbMax, nSeg, refreshMax;
    int smooth; /* both edges of the band move by exactly one x-y step per diagonal (fixed expansions do): the region may run
                 * under the absolute-position sweeps (cpk_table_gather.inl, "positions") */
    int64_t winBytes; /* LDS bytes of the largest symbol window of a traceback segment (both strings, two symbols per byte):
                       * what the absolute-position sweeps stage per segment instead of the whole strings (cpk_sweep.inl) */
} RegionPlan;

static int by_cost_desc(const void *a, const void *b) {
    const CostKey *p = a, *q = b;
    if (p->cls != q->cls) return p->cls < q->cls ? -1 : 1;
    if (p->cells != q->cells) return p->cells > q->cells ? -1 : 1;
    return p->index < q->index ? -1 : (p->index > q->index);
}

static int64_t default_out_cap(const cpecan_batch *b, const HostRegion *r) {
    int64_t cap = 6 * (r->lX + r->lY) + 64;
    if (b->params.threshold <= 0.0 || cap > r->cells) cap = r->cells;
    return cap < 1 ? 1 : cap;
}

/* Threads for the host's parallel loops: OpenMP's default is every hardware thread of the machine, which on a shared
 * multi-GPU host is far more than this process's share; 16 unless CPECAN_THREADS or a smaller OMP_NUM_THREADS says otherwise. */
int cpk_host_threads(void) {
    const char *env = getenv("CPECAN_THREADS");
    int n = env ? atoi(env) : 0;
    if (n < 1) {
        n = omp_get_max_threads();
        if (n > 16) n = 16;
    }
    return n < 1 ? 1 : n;
}

int cpecan_batch_upload(cpecan_batch *b) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || b->frozen) return CPECAN_ESTATE;
    if (b->nRegions == 0) {
        if (cpk_device_count() < 1) {
            cpk_set_error("no usable HIP device: the HIP path has no CPU fallback");
            return CPECAN_ENODEVICE;
        }
        b->frozen = 1;
        return CPECAN_OK;
    }
    const cpecan_params *p = &b->params;
    const int S = is_five(b->model.type) ? 5 : 3;
    int rc = CPECAN_OK;

    /* Planning walks every region's band ONCE as a stream of diagonals (cpecan_band.inl) and keeps per-region sums only:
     * cell count, widest diagonal, the traceback schedule (pairwiseAligner.c:791-810) and the scratch sizes it implies.
     * The per-diagonal table the kernels read (16 bytes per diagonal, 640 MB at 10 000 pairs x 2 kb) is built on the
     * device from the anchors by the same iterator (cpk_device_upload).  Regions are independent: OpenMP. */
    const double tU0 = now_ms();
    int64_t totalDiags = 0;
    int64_t *diagStart = malloc(sizeof(int64_t) * (size_t)b->nRegions);
    int64_t *segStart = malloc(sizeof(int64_t) * (size_t)b->nRegions);
    CostKey *keys = malloc(sizeof(CostKey) * (size_t)b->nRegions);
    RegionPlan *plan = calloc((size_t)b->nRegions, sizeof(RegionPlan));
    CpkSegment *segs = NULL;
    int64_t nSegs = 0; /* slots: every region gets room for an upper bound on its segment count */
    if (!diagStart || !segStart || !keys || !plan) {
        rc = CPECAN_ENOMEM;
        goto fail1;
    }
    for (int64_t i = 0; i < b->nRegions; i++) {
        const int64_t N = b->regions[i].lX + b->regions[i].lY;
        diagStart[i] = totalDiags;
        totalDiags += N + 1;
        segStart[i] = nSegs;
        /* consecutive traceback points are at least minDiagsBetweenTraceBack - (traceBackDiagonals + 1) >= 1 apart */
        nSegs += N / (p->minDiagsBetweenTraceBack - p->traceBackDiagonals - 1) + 2;
    }
    segs = host_zalloc((size_t)nSegs, sizeof(CpkSegment));
    if (!segs) {
        rc = CPECAN_ENOMEM;
        goto fail1;
    }
    const int dynamic = b->emit == CPECAN_EMIT_FORWARD ? 0 : p->dynamicAnchorExpansion; /* :894: forward uses the static band */
    const char *fwEnv = getenv("CPECAN_FAST_WALK");
    const int fastWalk = !(fwEnv && atoi(fwEnv) == 0);
    int64_t badRegion = -1;
#pragma omp parallel num_threads(cpk_host_threads())
    {
        /* cell offsets of the last traceBackDiagonals + 3 diagonals: the schedule looks that far back */
        const int64_t K = p->traceBackDiagonals + 3;
        int64_t *histOff = malloc(sizeof(int64_t) * (size_t)K * 4), *histW = histOff ? histOff + K : NULL;
        int64_t *histXlo = histOff ? histOff + 2 * K : NULL, *histYlo = histOff ? histOff + 3 * K : NULL;
        if (!histOff) {
#pragma omp critical(cpk_plan)
            rc = CPECAN_ENOMEM;
        }
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < b->nRegions; i++) {
            HostRegion *r = &b->regions[i];
            RegionPlan *pl = &plan[i];
            const int64_t N = r->lX + r->lY;
            keys[i].cells = 0;
            keys[i].index = i;
            if (!histOff) continue;
            CpkBandIter it;
            memset(&it, 0, sizeof it);
            const int keepRuns = b->runForm == 1;
            int bad = keepRuns ? cpk_band_init_runs(&it, b->runs + 4 * r->runOff, r->nRuns, r->nAnchors, r->lX, r->lY, p->diagonalExpansion)
                               : cpk_band_init(&it, b->anchors + (int64_t)b->anchorStride * r->anchorOff, b->anchorStride, r->nAnchors,
                                               r->lX, r->lY, p->diagonalExpansion, dynamic);
            CpkSegment *sg = segs + segStart[i];
            int64_t cells = 0, tracedBackTo = 0;
            int64_t offTracedBackTo = 0; /* cells before diagonal tracedBackTo */
            int64_t offAfter = 0;        /* cells before diagonal tracedBackTo + 1 */
            int64_t slot = 0; /* d % K without the division: this loop runs once per diagonal of the batch */
            int64_t maxW = 0;
            int64_t prevLo = 0, prevHi = 0;
            int64_t winXlo = 0, winYlo = 0; /* smallest x and y of diagonal tracedBackTo + 1 (symbol windows, RegionPlan::winBytes) */
            int smooth = !dynamic;
            const int64_t minBetween = p->minDiagsBetweenTraceBack, narrow = p->diagonalExpansion * 2 + 1;
            /* Runs of diagonal neighbours among the anchors (realign-style input: one anchor per matching column) are walked
             * in closed form.  Between two anchors (X, Y) and (X + 1, Y + 1) of a run, away from the matrix edges, the band
             * holds exactly two diagonals: x-y in [X-Y-E-1, X-Y+E+1] (E + 2 cells) and [X-Y-E, X-Y+E] (E + 1 cells) -- both
             * "narrow" for E >= 2 (<= 2E + 1) and both edges one x-y step apart from their neighbours'.  So a stretch of s such
             * intervals adds s (2E + 3) cells and nothing else, unless a traceback point may fall into it or behind it within the
             * K diagonals whose offsets the schedule looks back at: those diagonals are walked one by one.
             * (BASELINE config 4: 1.25e8 diagonals at ~14 cycles each were 31 ms of every batch on 16 threads.)  CPECAN_FAST_WALK=0
             * walks every diagonal (tests compare the two). */
            const cpk_anchor_t *ra = keepRuns ? NULL : b->anchors + (int64_t)b->anchorStride * r->anchorOff;
            const int64_t E = p->diagonalExpansion, hE = E / 2;
            const int fastOk = fastWalk && !dynamic && E >= 2 && b->anchorStride == 2;  /* (runs are kept for stride 2 only) */
            for (int64_t d = 0; d <= N && !bad; d++, slot = slot + 1 == K ? 0 : slot + 1) {
                if (fastOk && cpk_band_in_run(&it, d)) {
                    /* in the interval (A_j -> A_j+1) of a run, j = used - 2, about to emit its first diagonal */
                    const int64_t j = it.used - 2;
                    int64_t sMax = (tracedBackTo + minBetween - (K + 1) - d) / 2; /* every skipped diagonal stays K + 1 below the next traceback point */
                    sMax = imin(sMax, imin(r->lX - hE - it.pX, r->lY - hE - it.pY)); /* the last interval's rectangle inside the matrix */
                    if (it.pX - hE < 0 || it.pY - hE < 0) sMax = 0;                  /* ... and the first one's */
                    sMax = imin(sMax, (N - 2 - d) / 2);
                    int64_t sRun = 0; /* intervals of the run from here: anchors j .. j + sRun are diagonal neighbours */
                    if (keepRuns) {
                        /* anchor j + 1 is the one in (qX, qY): anchor qRo of run qRi, with length - 1 - qRo neighbours behind it
                         * in its run (a run that happens to continue in the next one starts a stretch of its own there) */
                        sRun = 1 + (int64_t)b->runs[4 * (r->runOff + it.qRi) + 2] - 1 - it.qRo;
                        sRun = sRun < sMax ? sRun : sMax;
                        sRun = sRun < 0 ? 0 : sRun;
                    } else {
                        while (sRun < sMax && j + sRun + 1 < r->nAnchors && ra[2 * (j + sRun + 1)] == ra[2 * (j + sRun)] + 1 &&
                               ra[2 * (j + sRun + 1) + 1] == ra[2 * (j + sRun) + 1] + 1)
                            sRun++;
                    }
                    if (sRun >= 2) {
                        const int64_t add = sRun * (2 * E + 3);
                        if (cells + add >= (int64_t)1 << 31) {
                            bad = 1;
                            break;
                        }
                        const int64_t xmy0 = it.pX - it.pY;
                        if (d > 0) { /* the step into the stretch: from the diagonal before to its first one */
                            const int64_t dl = (xmy0 - E - 1) - prevLo, dh = (xmy0 + E + 1) - prevHi;
                            smooth &= (dl == 1 || dl == -1) && (dh == 1 || dh == -1);
                        }
                        cells += add;
                        maxW = E + 2 > maxW ? E + 2 : maxW;
                        prevLo = xmy0 - E; /* the stretch's last diagonal: the second one of an interval */
                        prevHi = xmy0 + E;
                        /* the iterator as it stands behind anchor j + sRun's own diagonal */
                        const int64_t jn = j + sRun;
                        if (keepRuns) {
                            /* anchor jn = anchor j + 1 moved sRun - 1 steps along its run; the cursor goes to the anchor behind it */
                            it.pX = it.qX + (sRun - 1);
                            it.pY = it.qY + (sRun - 1);
                            it.ri = it.qRi;
                            it.ro = it.qRo + (int32_t)sRun;
                            if (it.ro >= b->runs[4 * (r->runOff + it.ri) + 2]) {
                                it.ro = 0;
                                it.ri++;
                            }
                        } else {
                            it.pX = (int64_t)ra[2 * jn] + 1;
                            it.pY = (int64_t)ra[2 * jn + 1] + 1;
                        }
                        it.used = jn + 1;
                        it.qX = r->lX;
                        it.qY = r->lY;
                        if (it.used < it.n) {
                            if (keepRuns) {
                                const int32_t *q = b->runs + 4 * (r->runOff + it.ri);
                                it.qX = (int64_t)q[0] + it.ro + 1;
                                it.qY = (int64_t)q[1] + it.ro + 1;
                                it.qRi = it.ri;
                                it.qRo = it.ro;
                                if (++it.ro >= q[2]) {
                                    it.ro = 0;
                                    it.ri++;
                                }
                            } else {
                                it.qX = (int64_t)ra[2 * it.used] + 1;
                                it.qY = (int64_t)ra[2 * it.used + 1] + 1;
                            }
                            it.used++;
                            if (it.qX <= it.pX || it.qY <= it.pY || it.qX > it.lX || it.qY > it.lY) {
                                bad = 1;
                                break;
                            }
                        }
                        it.qSum = it.qX + it.qY;
                        it.xLo = cpk_clamp(it.pX - hE, it.lX);
                        it.yHi = cpk_clamp(it.qY + hE, it.lY);
                        it.xHi = cpk_clamp(it.qX + hE, it.lX);
                        it.yLo = cpk_clamp(it.pY - hE, it.lY);
                        d += 2 * sRun - 1; /* the loop's own step makes it 2 sRun */
                        slot = (slot + 2 * sRun - 1) % K;
                        continue;
                    }
                }
                int64_t lo, hi;
                if (cpk_band_next(&it, d, &lo, &hi)) {
                    bad = 1;
                    break;
                }
                const int64_t w = (hi - lo) / 2 + 1;
                if (cells + w >= (int64_t)1 << 31) {
                    bad = 1;
                    break;
                }
                if (d > 0) {
                    const int64_t dl = lo - prevLo, dh = hi - prevHi;
                    smooth &= (dl == 1 || dl == -1) && (dh == 1 || dh == -1);
                }
                prevLo = lo;
                prevHi = hi;
                histOff[slot] = cells;
                histW[slot] = w;
                histXlo[slot] = (d + lo) >> 1;      /* x of the diagonal's first cell */
                histYlo[slot] = d - ((d + hi) >> 1); /* y of its last cell: the smallest */
                if (d == 1) {
                    offAfter = cells; /* tracedBackTo == 0 for the first segment */
                    winXlo = histXlo[slot];
                    winYlo = histYlo[slot];
                }
                maxW = w > maxW ? w : maxW;
                cells += w;
                if (d == 0) continue;
                const int atEnd = d == N;
                const int tracebackPoint = d >= tracedBackTo + minBetween && w <= narrow;
                if (!atEnd && !tracebackPoint) continue;
                memset(sg, 0, sizeof *sg);
                sg->tbPrev = (int32_t)tracedBackTo;
                sg->dTop = (int32_t)d;
                sg->tbFrom = (int32_t)(d - (atEnd ? 0 : p->traceBackDiagonals + 1));
                sg->atEnd = atEnd;
                sg->nRefresh = (int32_t)((sg->tbFrom - (sg->tbPrev + 1)) / CPK_REFRESH_PERIOD + 1);
                pl->refreshMax = sg->nRefresh > pl->refreshMax ? sg->nRefresh : pl->refreshMax;
                /* forward diagonals tbPrev..dTop are live during this traceback */
                pl->liveMax = imax(pl->liveMax, cells - offTracedBackTo);
                const int64_t tf = sg->tbFrom;
                const int64_t tfSlot = tf % K; /* once per segment */
                const int64_t fbCells = histOff[tfSlot] + histW[tfSlot] - offAfter;
                pl->fbMax = imax(pl->fbMax, fbCells);
                sg->emitCells = (int32_t)fbCells;
                {
                    /* the symbols the segment's diagonals tbPrev + 1 .. d touch, the kernel's arithmetic (cpk_sweep.inl,
                     * "Symbol windows"): from the even index at or below the smallest x (y) to one past the largest */
                    const int64_t x0 = winXlo & ~(int64_t)1, y0 = winYlo & ~(int64_t)1;
                    const int64_t x1 = imin(((d + hi) >> 1) + 1, r->lX + 1), y1 = imin(d - ((d + lo) >> 1) + 1, r->lY + 1);
                    pl->winBytes = imax(pl->winBytes, (x1 - x0 + 2) / 2 + (y1 - y0 + 2) / 2);
                }
                pl->nSeg++;
                sg++;

                /* the next segment starts from tbFrom: remember the cell offsets of tbFrom and tbFrom + 1 */
                tracedBackTo = tf;
                offTracedBackTo = histOff[tfSlot];
                offAfter = histOff[tfSlot] + histW[tfSlot];
                if (tf < d) { /* diagonal tf + 1 has been walked already: within the last K */
                    const int64_t s1 = (tf + 1) % K;
                    winXlo = histXlo[s1];
                    winYlo = histYlo[s1];
                }
            }
            pl->maxW = maxW;
            pl->smooth = smooth;
            if (bad) {
#pragma omp critical(cpk_plan)
                {
                    rc = CPECAN_EINVAL;
                    if (badRegion < 0 || i < badRegion) badRegion = i;
                }
                continue;
            }
            r->cells = cells;
            keys[i].cells = N > 0 ? cells : 0;
        }
        free(histOff);
    }
    const double tU1 = now_ms();
    if (rc != CPECAN_OK) {
        if (badRegion >= 0)
            cpk_set_error("region %lld of problem %lld: anchors do not define a valid band (or the band exceeds 2^31 cells)",
                          (long long)badRegion, (long long)b->regions[badRegion].problem);
        goto fail1;
    }
    /* Narrow regions (no diagonal wider than 32 cells: realign-style bands) are packed several to a wave by their own
     * kernel; they come first in the device order.  Within each class: longest first (the work queues are LPT). */
    {
        const char *env = getenv("CPECAN_PACKED"); /* diagnostic: 0 = one wave per region for every region */
        /* (bands with per-anchor expansions run the packed kernel's DYN variant: their edges may move backwards) */
        const int enabled = (b->emit == CPECAN_EMIT_MATCH || b->emit == CPECAN_EMIT_INDEL || b->emit == CPECAN_EMIT_EXPECT) && !b->debug &&
                            !(env && atoi(env) == 0);
        /* a launch is not worth fewer regions (CPECAN_PACKED=2: always, for tests).  The indel emitter's packed form pays a
         * pass of its own over every emitted cell: measured on realign-style batches it ties with one wave per region at
         * 10 000 alignments (32.1 against 30.5 ms) and wins 2.4x at 50 000 (61 against 148 ms) -- from ~12 000 regions of
         * a class, i.e. once a wave has half a dozen rounds to go (profiles/r04_packed_indel_emitter.txt) */
        const int64_t minCount = (env && atoi(env) >= 2) ? 1 : (b->emit == CPECAN_EMIT_INDEL ? 12000 : 64);
        int64_t perClass[4] = {0, 0, 0, 0};
        for (int64_t i = 0; i < b->nRegions; i++) {
            const int64_t w = plan[i].maxW;
            const int empty = b->regions[i].lX + b->regions[i].lY == 0;
            keys[i].cls = (!enabled || empty || w > 32) ? 3 : (w <= 8 ? 0 : (w <= 16 ? 1 : 2));
            perClass[keys[i].cls]++;
        }
        for (int k = 0; k < 3; k++) /* a class too small for its own launch joins the next wider one */
            if (perClass[k] > 0 && perClass[k] < minCount) {
                for (int64_t i = 0; i < b->nRegions; i++)
                    if (keys[i].cls == k) keys[i].cls = k + 1;
                perClass[k + 1] += perClass[k];
                perClass[k] = 0;
            }
        /* the wide regions by the LDS their rolling buffers and symbol strings need (classes 3..9; the last one keeps
         * both in global memory): a launch per class, each with its own occupancy and per-wave scratch */
        for (int64_t i = 0; i < b->nRegions; i++) {
            if (keys[i].cls != 3) continue;
            const int64_t w = plan[i].maxW;
            const size_t lds = sizeof(double) * (size_t)(544 + 768 + (2 * S + 1) * (w + 1)) +
                               (size_t)((b->regions[i].lX + 3) / 2 + (b->regions[i].lY + 3) / 2) + 16;
            static const int64_t edge[CPK_WIDE_CLASSES - 2] = {64, 128, 192, 256, 384, 512};
            /* up to 64 cells: a class of its own for the expectation emitter only -- its in-sweep kernel has a build unrolled
             * for one 64-lane group per diagonal (BASELINE config 5: 99 % of the regions); the other emitters' kernels loop
             * over groups and gain nothing from another launch */
            int k = b->emit == CPECAN_EMIT_EXPECT ? 0 : 1;
            while (k < CPK_WIDE_CLASSES - 2 && w > edge[k]) k++; /* CPK_WIDE_CLASSES - 2: wider, but its LDS still fits */
            keys[i].cls = 3 + (lds > 64 * 1024 ? CPK_WIDE_CLASSES - 1 : k);
        }
    }
    const double tU2 = now_ms();
    /* (class, cells descending, index): the keys stand in index order, so a STABLE sort by (class, -cells) is the order
     * by_cost_desc defines -- three counting passes of 12 bits instead of qsort's n log n comparator calls (4.3 ms of every
     * config-4 batch on one thread). */
    {
        const int64_t n = b->nRegions;
        CostKey *tmp = malloc(sizeof(CostKey) * (size_t)(n ? n : 1));
        if (!tmp) {
            qsort(keys, (size_t)n, sizeof(CostKey), by_cost_desc);
        } else {
            CostKey *src = keys, *dst = tmp;
            for (int pass = 0; pass < 3; pass++) {
                int64_t count[4097];
                memset(count, 0, sizeof count);
                /* sort value: class in the top 4 of 36 bits, then 2^31 - 1 - cells (cells < 2^31): ascending = by_cost_desc */
#define CPK_SORT_DIGIT(k) ((int)(((((uint64_t)(k).cls) << 32 | (uint64_t)(0x7fffffffll - (k).cells)) >> (12 * pass)) & 0xfff))
                for (int64_t i = 0; i < n; i++) count[CPK_SORT_DIGIT(src[i]) + 1]++;
                for (int q = 0; q < 4096; q++) count[q + 1] += count[q];
                for (int64_t i = 0; i < n; i++) dst[count[CPK_SORT_DIGIT(src[i])]++] = src[i];
#undef CPK_SORT_DIGIT
                CostKey *t = src;
                src = dst;
                dst = t;
            }
            if (src != keys) memcpy(keys, src, sizeof(CostKey) * (size_t)n); /* three passes: the result is in tmp */
            free(tmp);
        }
    }
    const double tU3 = now_ms();

    b->devRegions = host_zalloc((size_t)b->nRegions, sizeof(CpkRegion));
    b->devToHost = cpk_host_alloc(sizeof(int64_t) * (size_t)b->nRegions);
    if (!b->devRegions || !b->devToHost) {
        rc = CPECAN_ENOMEM;
        goto fail2;
    }
    CpkGeometry geo;
    memset(&geo, 0, sizeof geo);
    geo.nRegions = (int32_t)b->nRegions;
    geo.nStates = S;
    geo.emit = b->emit;
    geo.debug = b->debug;
    int64_t outAt = 0, dbgCells = 0, dbgDiags = 0, totalCells = 0;
    /* Two passes: what a region's entry holds of its own (the regions are visited in sorted order, i.e. at random in
     * memory: on one thread this loop was 6-9 ms of every config-4 batch), then the running offsets and the class maxima. */
    int64_t tooLarge = -1;
#pragma omp parallel num_threads(cpk_host_threads()) if (b->nRegions >= 4096)
    {
    CpkGeometry lg; /* this thread's class counts and maxima */
    memset(&lg, 0, sizeof lg);
#pragma omp for schedule(static)
    for (int64_t di = 0; di < b->nRegions; di++) {
        const int64_t hiRegion = keys[di].index;
        HostRegion *r = &b->regions[hiRegion];
        const RegionPlan *pl = &plan[hiRegion];
        CpkRegion *g = &b->devRegions[di];
        r->devIndex = di;
        b->devToHost[di] = hiRegion;
        g->seqXOff = r->seqXOff;
        g->seqYOff = r->seqYOff;
        g->diagOff = diagStart[hiRegion];
        g->segOff = segStart[hiRegion];
        g->anchorOff = r->anchorOff;
        g->nAnchors = (int32_t)r->nAnchors;
        g->nSeg = (int32_t)pl->nSeg;
        g->lX = (int32_t)r->lX;
        g->lY = (int32_t)r->lY;
        g->raggedLeft = r->raggedLeft;
        g->raggedRight = r->raggedRight;
        g->maxWidth = (int32_t)pl->maxW;
        g->absOk = pl->smooth;
        /* ring: diagonals are laid down one after another and never straddle the end of the ring */
        g->ringCap = (int32_t)imin(pl->liveMax + pl->maxW, ((int64_t)1 << 31) - 1);
        g->cells = r->cells;
        g->outCap = (int32_t)default_out_cap(b, r);
        /* every traceback segment gets a part of the region's output slice of its own (used when the segments run as
         * separate queue items, see CpkItem): in proportion to its emitted diagonals, the parts add up to outCap */
        {
            CpkSegment *sg = segs + segStart[hiRegion];
            const int64_t N = r->lX + r->lY;
            int64_t at = 0;
            for (int64_t si = 0; si < pl->nSeg; si++) {
                int64_t cap = N > 0 ? (int64_t)g->outCap * (sg[si].tbFrom - sg[si].tbPrev) / N + 16 : 1;
                /* every cell may be emitted: the cells of the segment's own emitted diagonals (round 3 gave every segment
                 * the whole region's cells, nSeg times what the slice can ever hold) */
                if (b->params.threshold <= 0.0) cap = sg[si].emitCells;
                sg[si].outOff = (int32_t)at;
                sg[si].outCap = (int32_t)cap;
                at += cap;
            }
            if (at > ((int64_t)1 << 31) - 1) {
#pragma omp critical(cpk_order)
                if (at > tooLarge) tooLarge = at;
                continue;
            }
            if (at > g->outCap) g->outCap = (int32_t)at;
        }
        if (keys[di].cls < 3) { /* narrow: scratch of the packed kernel's sub-slots */
            const int k = keys[di].cls;
            lg.nPacked[k]++;
            lg.pMaxRefresh[k] = pl->refreshMax > lg.pMaxRefresh[k] ? (int32_t)pl->refreshMax : lg.pMaxRefresh[k];
            lg.pRingCells[k] = imax(lg.pRingCells[k], pl->liveMax + pl->maxW);
            lg.pFbCells[k] = imax(lg.pFbCells[k], pl->fbMax);
        } else {
            const int k = keys[di].cls - 3;
            lg.nWide[k]++;
            lg.wMaxWidth[k] = g->maxWidth > lg.wMaxWidth[k] ? g->maxWidth : lg.wMaxWidth[k];
            lg.wMaxRefresh[k] = pl->refreshMax > lg.wMaxRefresh[k] ? (int32_t)pl->refreshMax : lg.wMaxRefresh[k];
            lg.wRingCells[k] = imax(lg.wRingCells[k], pl->liveMax + pl->maxW);
            lg.wFbCells[k] = imax(lg.wFbCells[k], pl->fbMax);
            lg.wSeqLdsBytes[k] = (int32_t)imax(lg.wSeqLdsBytes[k], imin((r->lX + 3) / 2 + (r->lY + 3) / 2, (int64_t)1 << 30));
            lg.wWinLdsBytes[k] = (int32_t)imax(lg.wWinLdsBytes[k], imin(pl->winBytes, (int64_t)1 << 30));
        }
    }
#pragma omp critical(cpk_order)
    {
        for (int k = 0; k < 3; k++) {
            geo.nPacked[k] += lg.nPacked[k];
            geo.pMaxRefresh[k] = lg.pMaxRefresh[k] > geo.pMaxRefresh[k] ? lg.pMaxRefresh[k] : geo.pMaxRefresh[k];
            geo.pRingCells[k] = imax(geo.pRingCells[k], lg.pRingCells[k]);
            geo.pFbCells[k] = imax(geo.pFbCells[k], lg.pFbCells[k]);
        }
        for (int k = 0; k < CPK_WIDE_CLASSES; k++) {
            geo.nWide[k] += lg.nWide[k];
            geo.wMaxWidth[k] = lg.wMaxWidth[k] > geo.wMaxWidth[k] ? lg.wMaxWidth[k] : geo.wMaxWidth[k];
            geo.wMaxRefresh[k] = lg.wMaxRefresh[k] > geo.wMaxRefresh[k] ? lg.wMaxRefresh[k] : geo.wMaxRefresh[k];
            geo.wRingCells[k] = imax(geo.wRingCells[k], lg.wRingCells[k]);
            geo.wFbCells[k] = imax(geo.wFbCells[k], lg.wFbCells[k]);
            geo.wSeqLdsBytes[k] = lg.wSeqLdsBytes[k] > geo.wSeqLdsBytes[k] ? lg.wSeqLdsBytes[k] : geo.wSeqLdsBytes[k];
            geo.wWinLdsBytes[k] = lg.wWinLdsBytes[k] > geo.wWinLdsBytes[k] ? lg.wWinLdsBytes[k] : geo.wWinLdsBytes[k];
        }
    }
    }
    if (tooLarge >= 0) {
        /* segment offsets and the region's slice are 32-bit: threshold <= 0 on a region of more than 2^31 cells x
         * segments cannot be laid out (the reference would return a list of that many tuples) */
        cpk_set_error("a region's output slice exceeds 2^31 triples (%lld): raise the threshold or split the region", (long long)tooLarge);
        rc = CPECAN_EINVAL;
        goto fail2;
    }
    for (int64_t di = 0; di < b->nRegions; di++) { /* the running offsets, in device order */
        CpkRegion *g = &b->devRegions[di];
        g->dbgCellOff = dbgCells;
        g->dbgDiagOff = dbgDiags;
        if (b->debug) {
            dbgCells += g->cells;
            dbgDiags += (int64_t)g->lX + g->lY + 1;
        }
        totalCells += g->cells;
        g->outOff = outAt;
        outAt += g->outCap;
    }
    for (int k = 0; k < 3; k++)
        if (geo.nPacked[k]) {
            if (geo.pMaxRefresh[k] < 1) geo.pMaxRefresh[k] = 1;
            if (geo.pRingCells[k] < 1) geo.pRingCells[k] = 1;
            if (geo.pFbCells[k] < 1) geo.pFbCells[k] = 1;
        }
    for (int k = 0; k < CPK_WIDE_CLASSES; k++)
        if (geo.nWide[k]) {
            if (b->emit == CPECAN_EMIT_FORWARD) { /* no traceback: nothing is kept of the forward matrix */
                geo.wRingCells[k] = 1;
                geo.wFbCells[k] = 1;
            }
            if (geo.wMaxRefresh[k] < 1) geo.wMaxRefresh[k] = 1;
            if (geo.wRingCells[k] < 1) geo.wRingCells[k] = 1;
            if (geo.wFbCells[k] < 1) geo.wFbCells[k] = 1;
        }
    /* the scalar geometry (maxWidth, rollStride, ringCells, ..., useGlobalRoll) is filled per launch from the class
     * arrays by the device side (cpk_class_geometry) */
    b->geo = geo;
    b->segs = segs;
    b->nSegs = nSegs;
    b->nDiags = totalDiags;
    b->outTriples = outAt < 1 ? 1 : outAt;
    b->dbgCells = dbgCells < 1 ? 1 : dbgCells;
    b->dbgDiags = dbgDiags < 1 ? 1 : dbgDiags;

    b->stats.problems = b->nProblems;
    b->stats.regions = b->nRegions;
    b->stats.cells = totalCells;
    b->stats.diagonals = totalDiags;
    CpkModel km;
    kernel_model(&b->model, p->threshold, &km);
    const double tU4 = now_ms();
    if (trace_host())
        fprintf(stderr, "cpecan upload: %lld regions, %lld segment slots: band walk %.1f ms, classes %.1f, sort %.1f, device order %.1f\n",
                (long long)b->nRegions, (long long)nSegs, tU1 - tU0, tU2 - tU1, tU3 - tU2, tU4 - tU3);
    if (!b->dev) {
        rc = cpk_device_create(&b->dev, b->device); /* fails with CPECAN_ENODEVICE when there is no GPU */
        if (rc != CPECAN_OK) goto fail2;
    }
    rc = cpk_device_upload(b->dev, &geo, &km, b->devRegions, b->runForm == 1 ? NULL : b->anchors, b->anchorStride,
                           b->nAnchorVals / b->anchorStride, b->runForm == 1 ? b->runs : NULL, b->runForm == 1 ? b->nRunVals / 4 : 0, totalDiags,
                           p->diagonalExpansion, dynamic, segs, nSegs, b->symbols, b->nSymbols, b->outTriples, b->nLists,
                           b->dbgCells, b->dbgDiags, &b->stats.h2dMs);
    if (rc != CPECAN_OK) goto fail2;
    b->stats.deviceBytes = cpk_device_bytes(b->dev);
    b->stats.wavesPerLaunch = cpk_device_waves(b->dev);
    b->stats.launchForm = cpk_device_form(b->dev);
    b->frozen = 1;
    free(diagStart);
    free(segStart);
    free(keys);
    free(plan);
    return CPECAN_OK;

fail2:
    cpk_host_free(b->devRegions);
    cpk_host_free(b->devToHost);
    b->devRegions = NULL;
    b->devToHost = NULL;
    b->segs = NULL;
    b->nSegs = 0;
fail1:
    if (!b->segs) cpk_host_free(segs);
    free(diagStart);
    free(segStart);
    free(keys);
    free(plan);
    return rc;
}

int cpecan_batch_run(cpecan_batch *b, void *stream) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->frozen) return CPECAN_ESTATE;
    b->downloaded = 0;
    if (b->nRegions == 0) {
        b->ran = 1;
        return CPECAN_OK;
    }
    int rc = cpk_device_run(b->dev, stream);
    if (rc == CPECAN_OK) {
        b->ran = 1;
        b->stats.launches = 1;
    }
    return rc;
}

/* The copy plan of cpk_device_gather for the whole batch: per list, problems in order, regions in order, traceback
 * segments in DEScending order (each traceback's pairs are prepended, pairwiseAligner.c:1415-1417), every triple shifted
 * by its region's offset (:1411-1418).  Sets the per-problem result pointers into b->results. */
static int plan_results(cpecan_batch *b, const int32_t *counts, const int32_t *segStarts, const int32_t *segCounts,
                        CpkChunk **chunksOut, int64_t *nChunksOut, int64_t *totalOut) {
    int64_t total = 0;
    for (int l = 0; l < b->nLists; l++)
        for (int64_t di = 0; di < b->nRegions; di++) total += counts[(size_t)l * b->nRegions + di];
    CpkChunk *chunks = cpk_host_alloc(sizeof(CpkChunk) * (size_t)(b->nLists * (b->nSegs ? b->nSegs : 1)));
    b->results = cpk_host_alloc(sizeof(int32_t) * 3 * (size_t)(total ? total : 1));
    if (!chunks || !b->results) {
        cpk_host_free(chunks);
        return CPECAN_ENOMEM;
    }
    int64_t nChunks = 0, at = 0;
    for (int l = 0; l < b->nLists; l++) {
        const int32_t *cnt = counts + (size_t)l * b->nRegions;
        const int32_t *ss = segStarts + (size_t)l * b->nSegs;
        const int32_t *sc = segCounts + (size_t)l * b->nSegs;
        for (int64_t pi = 0; pi < b->nProblems; pi++) {
            HostProblem *pr = &b->problems[pi];
            pr->triples[l] = b->results + 3 * at;
            const int64_t first = at;
            for (int64_t i = 0; i < pr->nRegions; i++) {
                const HostRegion *r = &b->regions[pr->firstRegion + i];
                const CpkRegion *g = &b->devRegions[r->devIndex];
                const int32_t n = cnt[r->devIndex];
                for (int sgi = g->nSeg - 1; sgi >= 0; sgi--) {
                    const int32_t from = ss[g->segOff + sgi];
                    /* a split region's segments were written apart, each with its own count */
                    const int32_t to = g->split ? from + sc[g->segOff + sgi] : (sgi + 1 < g->nSeg ? ss[g->segOff + sgi + 1] : n);
                    if (to <= from) continue;
                    CpkChunk *c = &chunks[nChunks++];
                    c->src = (int64_t)l * b->outTriples + g->outOff + from;
                    c->dst = at;
                    c->len = to - from;
                    c->dx = (int32_t)r->x1;
                    c->dy = (int32_t)r->y1;
                    c->pad = 0;
                    at += to - from;
                }
            }
            pr->nTriples[l] = at - first;
        }
    }
    *chunksOut = chunks;
    *nChunksOut = nChunks;
    *totalOut = at;
    return CPECAN_OK;
}

/* Fills one consumer descriptor per problem and the scratch totals of the job. */
static void post_layout(CpkPostJob *job, CpkPostProblem *pp, int64_t i, const int64_t off[3], const int64_t n[3],
                        int64_t lX, int64_t lY, int64_t charX, int64_t charY) {
    memset(pp, 0, sizeof *pp);
    for (int l = 0; l < 3; l++) {
        pp->off[l] = off[l];
        pp->n[l] = (int32_t)n[l];
    }
    pp->lX = (int32_t)lX;
    pp->lY = (int32_t)lY;
    pp->seqOff = job->seqSlots;
    pp->chainOff = job->chainSlots;
    pp->charX = charX;
    pp->charY = charY;
    pp->meaOut = job->meaCap;
    pp->shiftOut = job->shiftCap;
    job->seqSlots += lX + 2 * lY; /* ORDERED: column heads + the staircase's two arrays; the other consumers use lX + lY */
    job->chainSlots += n[0] + 1;
    job->meaCap += n[0];
    job->shiftCap += n[0] + imin(lX, lY) + 1;
    (void)i;
}

static int run_post(cpecan_batch *b) {
    CpkPostJob job;
    memset(&job, 0, sizeof job);
    job.flags = b->postFlags;
    job.gapGamma = b->postGapGamma;
    job.matchGamma = (double)b->postMatchGamma; /* float widened, as the reference passes it (multipleAligner.c:945) */
    job.nProblems = b->nProblems;
    CpkPostProblem *pp = malloc(sizeof(CpkPostProblem) * (size_t)(b->nProblems ? b->nProblems : 1));
    double *scores = malloc(sizeof(double) * CPK_POST_SCORES * (size_t)(b->nProblems ? b->nProblems : 1));
    int32_t *counts = malloc(sizeof(int32_t) * 2 * (size_t)(b->nProblems ? b->nProblems : 1));
    int rc = (pp && scores && counts) ? CPECAN_OK : CPECAN_ENOMEM;
    for (int64_t i = 0; rc == CPECAN_OK && i < b->nProblems; i++) {
        const HostProblem *pr = &b->problems[i];
        int64_t off[3] = {0, 0, 0}, n[3] = {0, 0, 0};
        for (int l = 0; l < b->nLists; l++) {
            off[l] = (pr->triples[l] - b->results) / 3;
            n[l] = pr->nTriples[l];
        }
        post_layout(&job, &pp[i], i, off, n, pr->lX, pr->lY, pr->charX, pr->charY);
    }
    if (rc == CPECAN_OK) {
        job.problems = pp;
        job.scores = scores;
        job.counts = counts;
        /* identity scores and the left shift compare letters; neither exists without a consumer (cpecan_batch_identity_scores) */
        job.chars = job.flags ? b->chars : NULL;
        job.nChars = job.flags ? b->nChars : 0;
        if (job.flags & (CPECAN_POST_MEA | CPECAN_POST_ORDERED)) {
            b->postMea = cpk_host_alloc(sizeof(int32_t) * 3 * (size_t)(job.meaCap ? job.meaCap : 1));
            job.mea = b->postMea;
            if (!b->postMea) rc = CPECAN_ENOMEM;
        }
        if (rc == CPECAN_OK && (job.flags & CPECAN_POST_LEFT_SHIFT)) {
            b->postShift = cpk_host_alloc(sizeof(int32_t) * 3 * (size_t)(job.shiftCap ? job.shiftCap : 1));
            job.shift = b->postShift;
            if (!b->postShift) rc = CPECAN_ENOMEM;
        }
    }
    if (rc == CPECAN_OK) rc = cpk_device_post(b->dev, &job);
    for (int64_t i = 0; rc == CPECAN_OK && i < b->nProblems; i++) {
        HostProblem *pr = &b->problems[i];
        for (int k = 0; k < CPK_POST_SCORES; k++) pr->scores[k] = scores[CPK_POST_SCORES * i + k];
        if (job.flags & CPECAN_POST_LEFT_SHIFT) {
            pr->triples[3] = b->postShift + 3 * pp[i].shiftOut;
            pr->nTriples[3] = counts[2 * i + 1];
        } else if (job.flags & (CPECAN_POST_MEA | CPECAN_POST_ORDERED)) {
            pr->triples[3] = b->postMea + 3 * pp[i].meaOut;
            pr->nTriples[3] = counts[2 * i];
        }
    }
    free(pp);
    free(scores);
    free(counts);
    return rc;
}

int cpecan_batch_set_post(cpecan_batch *b, int flags, double gapGamma) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b) return CPECAN_EINVAL;
    if (flags & ~(CPECAN_POST_REWEIGHT | CPECAN_POST_MEA | CPECAN_POST_LEFT_SHIFT | CPECAN_POST_ORDERED)) return CPECAN_EINVAL;
    if ((flags & CPECAN_POST_LEFT_SHIFT) && !(flags & CPECAN_POST_MEA)) return CPECAN_EINVAL;
    if ((flags & CPECAN_POST_ORDERED) && (flags & (CPECAN_POST_MEA | CPECAN_POST_LEFT_SHIFT))) return CPECAN_EINVAL;
    if ((flags & CPECAN_POST_MEA) && (flags & CPECAN_POST_REWEIGHT)) return CPECAN_EINVAL; /* alternatives in the reference */
    if ((flags & CPECAN_POST_MEA) && b->emit != CPECAN_EMIT_INDEL) {
        cpk_set_error("the MEA alignment needs the gap lists: create the batch with CPECAN_EMIT_INDEL");
        return CPECAN_EINVAL;
    }
    if (flags && (b->emit == CPECAN_EMIT_FORWARD || b->emit == CPECAN_EMIT_EXPECT)) return CPECAN_EINVAL;
    b->postFlags = flags;
    b->postGapGamma = gapGamma;
    return CPECAN_OK;
}

int cpecan_batch_set_match_gamma(cpecan_batch *b, float matchGamma) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !(matchGamma >= 0.0f)) return CPECAN_EINVAL;
    b->postMatchGamma = matchGamma;
    return CPECAN_OK;
}

int cpecan_batch_identity_scores(const cpecan_batch *b, int64_t problem, double *byIdentity, double *byIdentityIgnoringGaps) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->downloaded) return CPECAN_ESTATE;
    if (problem < 0 || problem >= b->nProblems || b->emit == CPECAN_EMIT_FORWARD || b->emit == CPECAN_EMIT_EXPECT)
        return CPECAN_EINVAL;
    if (!b->postFlags) {
        cpk_set_error("identity scores are computed by the consumer stage: select one with cpecan_batch_set_post");
        return CPECAN_ESTATE;
    }
    if (byIdentity) *byIdentity = b->problems[problem].scores[3];
    if (byIdentityIgnoringGaps) *byIdentityIgnoringGaps = b->problems[problem].scores[4];
    return CPECAN_OK;
}

int cpecan_batch_scores(const cpecan_batch *b, int64_t problem, double *byPosterior, double *byPosteriorIgnoringGaps,
                        double *meaScore) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->downloaded) return CPECAN_ESTATE;
    if (problem < 0 || problem >= b->nProblems || b->emit == CPECAN_EMIT_FORWARD || b->emit == CPECAN_EMIT_EXPECT)
        return CPECAN_EINVAL;
    if (byPosterior) *byPosterior = b->problems[problem].scores[0];
    if (byPosteriorIgnoringGaps) *byPosteriorIgnoringGaps = b->problems[problem].scores[1];
    if (meaScore) *meaScore = b->problems[problem].scores[2];
    return CPECAN_OK;
}

int cpecan_batch_download(cpecan_batch *b) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->ran) return CPECAN_ESTATE;
    free_results(b);
    b->stats.pairs = 0;
    if (b->nRegions == 0) {
        b->downloaded = 1;
        return CPECAN_OK;
    }
    int rc = CPECAN_OK;
    int32_t *counts = NULL, *segStarts = NULL, *segCounts = NULL;
    const double tD0 = now_ms();
    double tD1 = tD0;
    for (int attempt = 0; attempt < 3; attempt++) {
        cpk_host_free(counts);
        cpk_host_free(segStarts);
        cpk_host_free(segCounts);
        counts = cpk_host_alloc(sizeof(int32_t) * (size_t)b->nLists * b->nRegions);
        segStarts = cpk_host_alloc(sizeof(int32_t) * (size_t)b->nLists * (b->nSegs ? b->nSegs : 1));
        segCounts = cpk_host_alloc(sizeof(int32_t) * (size_t)b->nLists * (b->nSegs ? b->nSegs : 1));
        if (!counts || !segStarts || !segCounts) {
            rc = CPECAN_ENOMEM;
            break;
        }
        if ((b->emit == CPECAN_EMIT_FORWARD || b->emit == CPECAN_EMIT_EXPECT) && !b->forward) {
            b->forward = malloc(sizeof(double) * (size_t)(b->nRegions > 106 ? b->nRegions : 106));
            if (!b->forward) {
                rc = CPECAN_ENOMEM;
                break;
            }
        }
        rc = cpk_device_download(b->dev, counts, segStarts, segCounts, b->forward, &b->stats.kernelMs, &b->stats.d2hMs);
        if (rc != CPECAN_OK) break;
        b->stats.h2dMs = cpk_device_h2d_ms(b->dev);
        tD1 = now_ms();
        if (b->emit == CPECAN_EMIT_FORWARD || b->emit == CPECAN_EMIT_EXPECT) break; /* these emitters produce no lists */
        /* did any region overflow its output slice?  If so enlarge exactly and run once more. */
        int overflow = 0;
        int64_t outAt = 0;
        for (int64_t di = 0; di < b->nRegions; di++) {
            CpkRegion *g = &b->devRegions[di];
            if (g->split) {
                /* the segments were written apart: a region's count is the sum, and each segment has its own capacity */
                int64_t at = 0;
                for (int l = 0; l < b->nLists; l++) counts[(size_t)l * b->nRegions + di] = 0;
                for (int32_t si = 0; si < g->nSeg; si++) {
                    CpkSegment *sg = &b->segs[g->segOff + si];
                    int32_t need = 0;
                    for (int l = 0; l < b->nLists; l++) {
                        const int32_t c = segCounts[(size_t)l * b->nSegs + g->segOff + si];
                        need = c > need ? c : need;
                        counts[(size_t)l * b->nRegions + di] += c < sg->outCap ? c : sg->outCap;
                    }
                    if (need > sg->outCap) {
                        overflow = 1;
                        sg->outCap = need;
                    }
                    sg->outOff = (int32_t)at;
                    at += sg->outCap;
                }
                if (at > g->outCap) g->outCap = (int32_t)at;
            } else {
                int32_t need = 0;
                for (int l = 0; l < b->nLists; l++) {
                    const int32_t c = counts[(size_t)l * b->nRegions + di];
                    need = c > need ? c : need;
                }
                if (need > g->outCap) {
                    overflow = 1;
                    g->outCap = need;
                }
            }
            g->outOff = outAt;
            outAt += g->outCap;
        }
        if (!overflow) break;
        b->outTriples = outAt;
        rc = cpk_device_update_regions(b->dev, b->devRegions, b->segs, b->outTriples);
        if (rc != CPECAN_OK) break;
        rc = cpk_device_rerun(b->dev); /* same stream as the run that overflowed */
        if (rc != CPECAN_OK) break;
        b->stats.launches++;
        if (attempt == 2) {
            cpk_set_error("output overflow persisted after re-running");
            rc = CPECAN_ESTATE;
        }
    }
    if (rc == CPECAN_OK && b->emit != CPECAN_EMIT_FORWARD && b->emit != CPECAN_EMIT_EXPECT) {
        /* the lists are put in order on the device; only the emitted triples cross PCIe */
        CpkChunk *chunks = NULL;
        int64_t nChunks = 0, total = 0;
        const double tD2 = now_ms();
        rc = plan_results(b, counts, segStarts, segCounts, &chunks, &nChunks, &total);
        const double tD3 = now_ms();
        if (rc == CPECAN_OK) rc = cpk_device_gather(b->dev, chunks, nChunks, total);
        cpk_host_free(chunks);
        const double tD4 = now_ms();
        if (rc == CPECAN_OK) rc = run_post(b); /* consumers of the lists, on the device, before they leave it */
        const double tD5 = now_ms();
        if (rc == CPECAN_OK) rc = cpk_device_fetch(b->dev, b->results, total, &b->stats.d2hMs);
        b->stats.pairs = total;
        if (trace_host())
            fprintf(stderr, "cpecan download: wait + counts %.1f ms, overflow scan %.1f, list plan %.1f (%lld chunks), gather %.1f, consumers %.1f, fetch %.1f (%lld triples)\n",
                    tD1 - tD0, tD2 - tD1, tD3 - tD2, (long long)nChunks, tD4 - tD3, tD5 - tD4, now_ms() - tD5, (long long)total);
    }
    if (rc == CPECAN_OK) b->downloaded = 1;
    cpk_host_free(counts);
    cpk_host_free(segStarts);
    cpk_host_free(segCounts);
    return rc;
}

static void *download_helper(void *arg) {
    cpecan_batch *b = arg;
    tl_helperOf = b;
    b->dlResult = cpecan_batch_download(b);
    if (b->dlResult != CPECAN_OK) {
        strncpy(b->dlError, cpk_last_error(), sizeof b->dlError - 1);
        b->dlError[sizeof b->dlError - 1] = 0;
    }
    return NULL;
}

int cpecan_batch_download_begin(cpecan_batch *b) {
    if (!b || !b->ran || b->dlActive) return CPECAN_ESTATE;
    b->dlResult = CPECAN_OK;
    b->dlError[0] = 0;
    b->dlActive = 1;
    if (pthread_create(&b->dlThread, NULL, download_helper, b) != 0) {
        b->dlActive = 0;
        cpk_set_error("cannot start the download helper thread");
        return CPECAN_ENOMEM;
    }
    return CPECAN_OK;
}

int cpecan_batch_download_end(cpecan_batch *b) {
    if (!b || !b->dlActive) return CPECAN_ESTATE;
    pthread_join(b->dlThread, NULL);
    b->dlActive = 0;
    if (b->dlResult != CPECAN_OK) cpk_set_error("%s", b->dlError);
    return b->dlResult;
}

int cpecan_batch_result(const cpecan_batch *b, int64_t problem, int which, const int32_t **triples, int64_t *n) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->downloaded) return CPECAN_ESTATE;
    if (problem < 0 || problem >= b->nProblems || which < 0) return CPECAN_EINVAL;
    if (which >= b->nLists && !(which == 3 && (b->postFlags & (CPECAN_POST_MEA | CPECAN_POST_ORDERED)))) return CPECAN_EINVAL;
    *triples = b->problems[problem].triples[which];
    *n = b->problems[problem].nTriples[which];
    return CPECAN_OK;
}

int cpecan_batch_forward_prob(const cpecan_batch *b, int64_t problem, double *logProb) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->downloaded || b->emit != CPECAN_EMIT_FORWARD) return CPECAN_ESTATE;
    if (problem < 0 || problem >= b->nProblems || !logProb) return CPECAN_EINVAL;
    *logProb = b->forward[b->regions[b->problems[problem].firstRegion].devIndex];
    return CPECAN_OK;
}

int cpecan_batch_expectations(const cpecan_batch *b, cpecan_hmm *acc) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->downloaded || b->emit != CPECAN_EMIT_EXPECT) return CPECAN_ESTATE;
    const int S = is_five(b->model.type) ? 5 : 3;
    if (!acc || acc->stateNumber != S) return CPECAN_EINVAL;
    if (b->nRegions == 0) return CPECAN_OK;
    /* b->forward holds the batch sums: [0,25) transitions [from*S+to], [25,105) emissions, [105] likelihood */
    for (int i = 0; i < S * S; i++) acc->transitions[i] += b->forward[i];
    for (int i = 0; i < S * 16; i++) acc->emissions[i] += b->forward[25 + i];
    acc->likelihood += b->forward[105];
    return CPECAN_OK;
}

int cpecan_batch_stats(const cpecan_batch *b, cpecan_stats *s) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !s) return CPECAN_EINVAL;
    *s = b->stats;
    return CPECAN_OK;
}

int cpecan_batch_debug_fetch(const cpecan_batch *b, int64_t problem, double *fbMatch, int64_t cells, double *totalUsed,
                             int64_t diagonals) {
    if (dl_busy(b)) return CPECAN_ESTATE;
    if (!b || !b->downloaded || !b->debug) return CPECAN_ESTATE;
    if (problem < 0 || problem >= b->nProblems || b->problems[problem].nRegions != 1) return CPECAN_EINVAL;
    const HostRegion *r = &b->regions[b->problems[problem].firstRegion];
    const CpkRegion *g = &b->devRegions[r->devIndex];
    if (cells != r->cells || diagonals != r->lX + r->lY + 1) return CPECAN_EINVAL;
    double *fbAll = malloc(sizeof(double) * (size_t)b->dbgCells), *totAll = malloc(sizeof(double) * (size_t)b->dbgDiags);
    if (!fbAll || !totAll) {
        free(fbAll);
        free(totAll);
        return CPECAN_ENOMEM;
    }
    int rc = cpk_device_debug_fetch(b->dev, fbAll, b->dbgCells, totAll, b->dbgDiags);
    if (rc == CPECAN_OK) {
        memcpy(fbMatch, fbAll + g->dbgCellOff, sizeof(double) * (size_t)cells);
        memcpy(totalUsed, totAll + g->dbgDiagOff, sizeof(double) * (size_t)diagonals);
    }
    free(fbAll);
    free(totAll);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * single-problem convenience
 * ---------------------------------------------------------------------------------------------- */
/* ---- the list consumers on host-held lists: one problem through cpk_post_lists ---- */
typedef struct {
    int32_t *buf;    /* concatenated lists */
    int64_t total;
    CpkPostProblem pp;
    CpkPostJob job;
    uint8_t *chars;
    double scores[CPK_POST_SCORES];
    int32_t counts[2];
} PostSingle;

static void post_single_free(PostSingle *ps) {
    free(ps->buf);
    free(ps->chars);
    free(ps->job.mea);
    free(ps->job.shift);
}

static int post_single_run(PostSingle *ps, int flags, double gapGamma, double matchGamma, const int32_t *lists[3],
                           const int64_t n[3], int64_t lX, int64_t lY, const char *sX, const char *sY) {
    memset(ps, 0, sizeof *ps);
    ps->job.matchGamma = matchGamma;
    if (lX < 0 || lY < 0 || lX + lY >= (int64_t)1 << 30) return CPECAN_EINVAL;
    int64_t off[3], at = 0;
    for (int l = 0; l < 3; l++) {
        if (n[l] < 0 || n[l] >= (int64_t)1 << 30 || (n[l] > 0 && !lists[l])) return CPECAN_EINVAL;
        off[l] = at;
        at += n[l];
    }
    ps->total = at;
    ps->buf = malloc(sizeof(int32_t) * 3 * (size_t)(at ? at : 1));
    if (!ps->buf) return CPECAN_ENOMEM;
    for (int l = 0; l < 3; l++) {
        for (int64_t i = 0; (flags || sX) && i < n[l]; i++) { /* coordinates index the mass arrays and the sequences: check them */
            const int32_t x = lists[l][3 * i + 1], y = lists[l][3 * i + 2];
            const int okX = x >= 0 && x < lX, okY = y >= 0 && y < lY;
            if ((l == 0 && !(okX && okY)) || (l == 1 && !okX) || (l == 2 && !okY)) return CPECAN_EINVAL;
        }
        if (n[l]) memcpy(ps->buf + 3 * off[l], lists[l], sizeof(int32_t) * 3 * (size_t)n[l]);
    }
    ps->job.flags = flags;
    ps->job.gapGamma = gapGamma;
    ps->job.nProblems = 1;
    post_layout(&ps->job, &ps->pp, 0, off, n, lX, lY, 0, lX);
    ps->job.problems = &ps->pp;
    ps->job.scores = ps->scores;
    ps->job.counts = ps->counts;
    if (flags & (CPECAN_POST_MEA | CPECAN_POST_ORDERED)) {
        ps->job.mea = malloc(sizeof(int32_t) * 3 * (size_t)(ps->job.meaCap ? ps->job.meaCap : 1));
        if (!ps->job.mea) return CPECAN_ENOMEM;
    }
    if ((flags & CPECAN_POST_LEFT_SHIFT) && (!sX || !sY)) return CPECAN_EINVAL;
    if (sX && sY) {
        ps->chars = malloc((size_t)(lX + lY + 1));
        if (!ps->chars) return CPECAN_ENOMEM;
        for (int64_t i = 0; i < lX; i++) ps->chars[i] = (uint8_t)toupper((unsigned char)sX[i]);
        for (int64_t i = 0; i < lY; i++) ps->chars[lX + i] = (uint8_t)toupper((unsigned char)sY[i]);
        ps->job.chars = ps->chars;
        ps->job.nChars = lX + lY;
    }
    if (flags & CPECAN_POST_LEFT_SHIFT) {
        ps->job.shift = malloc(sizeof(int32_t) * 3 * (size_t)(ps->job.shiftCap ? ps->job.shiftCap : 1));
        if (!ps->job.shift) return CPECAN_ENOMEM;
    }
    return cpk_post_lists(cpk_current_device(), ps->buf, ps->total, &ps->job); /* the caller's current device */
}

static int take_list(const int32_t *src, int64_t n, int32_t **out, int64_t *nOut) {
    int32_t *t = malloc(sizeof(int32_t) * 3 * (size_t)(n ? n : 1));
    if (!t) return CPECAN_ENOMEM;
    if (n) memcpy(t, src, sizeof(int32_t) * 3 * (size_t)n);
    *out = t;
    *nOut = n;
    return CPECAN_OK;
}

int cpecan_reweight_aligned_pairs(int32_t *triples, int64_t n, int64_t lX, int64_t lY, double gapGamma) {
    const int32_t *lists[3] = {triples, NULL, NULL};
    const int64_t ns[3] = {n, 0, 0};
    PostSingle ps;
    int rc = post_single_run(&ps, CPECAN_POST_REWEIGHT, gapGamma, 0.0, lists, ns, lX, lY, NULL, NULL);
    if (rc == CPECAN_OK && n) memcpy(triples, ps.buf, sizeof(int32_t) * 3 * (size_t)n);
    post_single_free(&ps);
    return rc;
}

int cpecan_posterior_scores(const int32_t *triples, int64_t n, int64_t lX, int64_t lY, double *byPosterior,
                            double *byPosteriorIgnoringGaps) {
    const int32_t *lists[3] = {triples, NULL, NULL};
    const int64_t ns[3] = {n, 0, 0};
    PostSingle ps;
    int rc = post_single_run(&ps, 0, 0.0, 0.0, lists, ns, lX, lY, NULL, NULL);
    if (rc == CPECAN_OK) {
        if (byPosterior) *byPosterior = ps.scores[0];
        if (byPosteriorIgnoringGaps) *byPosteriorIgnoringGaps = ps.scores[1];
    }
    post_single_free(&ps);
    return rc;
}

int cpecan_identity_scores(const int32_t *triples, int64_t n, const char *sX, const char *sY, double *byIdentity,
                           double *byIdentityIgnoringGaps) {
    if (!sX || !sY) return CPECAN_EINVAL;
    const int32_t *lists[3] = {triples, NULL, NULL};
    const int64_t ns[3] = {n, 0, 0};
    PostSingle ps;
    int rc = post_single_run(&ps, 0, 0.0, 0.0, lists, ns, (int64_t)strlen(sX), (int64_t)strlen(sY), sX, sY);
    if (rc == CPECAN_OK) {
        if (byIdentity) *byIdentity = ps.scores[3];
        if (byIdentityIgnoringGaps) *byIdentityIgnoringGaps = ps.scores[4];
    }
    post_single_free(&ps);
    return rc;
}

int cpecan_filter_pairs_ordered(const int32_t *pairs, int64_t n, int64_t lX, int64_t lY, float matchGamma, int32_t **out,
                                int64_t *nOut) {
    if (!out || !nOut || !(matchGamma >= 0.0f)) return CPECAN_EINVAL;
    const int32_t *lists[3] = {pairs, NULL, NULL};
    const int64_t ns[3] = {n, 0, 0};
    PostSingle ps;
    int rc = post_single_run(&ps, CPECAN_POST_ORDERED, 0.0, (double)matchGamma, lists, ns, lX, lY, NULL, NULL);
    if (rc == CPECAN_OK) rc = take_list(ps.job.mea, ps.counts[0], out, nOut);
    post_single_free(&ps);
    return rc;
}

int cpecan_mea_alignment(const int32_t *pairs, int64_t n, const int32_t *gapX, int64_t nGapX, const int32_t *gapY,
                         int64_t nGapY, int64_t lX, int64_t lY, float gapGamma, int32_t **out, int64_t *nOut,
                         double *alignmentScore) {
    if (!out || !nOut) return CPECAN_EINVAL;
    const int32_t *lists[3] = {pairs, gapX, gapY};
    const int64_t ns[3] = {n, nGapX, nGapY};
    PostSingle ps;
    int rc = post_single_run(&ps, CPECAN_POST_MEA, (double)gapGamma, 0.0, lists, ns, lX, lY, NULL, NULL);
    if (rc == CPECAN_OK) rc = take_list(ps.job.mea, ps.counts[0], out, nOut);
    if (rc == CPECAN_OK && alignmentScore) *alignmentScore = ps.scores[2];
    post_single_free(&ps);
    return rc;
}

int cpecan_left_shift_alignment(const int32_t *pairs, int64_t n, const char *sX, const char *sY, int32_t **out,
                                int64_t *nOut) {
    if (!out || !nOut || !sX || !sY) return CPECAN_EINVAL;
    /* LEFT_SHIFT without MEA: the consumer stage shifts list 0 itself (it must be a chain, as in the reference) */
    const int32_t *lists[3] = {pairs, NULL, NULL};
    const int64_t ns[3] = {n, 0, 0};
    PostSingle ps;
    int rc = post_single_run(&ps, CPECAN_POST_LEFT_SHIFT, 0.0, 0.0, lists, ns, (int64_t)strlen(sX), (int64_t)strlen(sY), sX, sY);
    if (rc == CPECAN_OK) rc = take_list(ps.job.shift, ps.counts[1], out, nOut);
    post_single_free(&ps);
    return rc;
}

int cpecan_get_shifted_mea_alignment(const cpecan_model *m, const char *sX, const char *sY, const int64_t *anchors,
                                     int64_t nAnchors, const cpecan_params *p, float gapGamma, int raggedLeft,
                                     int raggedRight, int32_t **out, int64_t *nOut, double *alignmentScore) {
    if (!m || !sX || !sY || !p || !out || !nOut) return CPECAN_EINVAL;
    cpecan_batch *b = NULL;
    int rc = cpecan_batch_create(&b, m, p, CPECAN_EMIT_INDEL, cpk_current_device());
    if (rc != CPECAN_OK) return rc;
    rc = cpecan_batch_set_post(b, CPECAN_POST_MEA | CPECAN_POST_LEFT_SHIFT, (double)gapGamma);
    if (rc == CPECAN_OK) {
        const int64_t idx = cpecan_batch_add(b, sX, (int64_t)strlen(sX), sY, (int64_t)strlen(sY), anchors, nAnchors,
                                             raggedLeft, raggedRight);
        rc = idx < 0 ? (int)idx : CPECAN_OK;
    }
    if (rc == CPECAN_OK) rc = cpecan_batch_upload(b);
    if (rc == CPECAN_OK) rc = cpecan_batch_run(b, NULL);
    if (rc == CPECAN_OK) rc = cpecan_batch_download(b);
    if (rc == CPECAN_OK) rc = take_list(b->problems[0].triples[3], b->problems[0].nTriples[3], out, nOut);
    if (rc == CPECAN_OK && alignmentScore) *alignmentScore = b->problems[0].scores[2];
    cpecan_batch_destroy(b);
    return rc;
}

void cpecan_free(void *p) { free(p); }

/* one problem through a batch of one: create, add, upload, run, download */
static int run_single(cpecan_batch **out, const cpecan_model *m, const char *sX, const char *sY, const int64_t *anchors,
                      int64_t nAnchors, const cpecan_params *p, int raggedLeft, int raggedRight, int emit) {
    if (!m || !sX || !sY || !p) return CPECAN_EINVAL;
    cpecan_batch *b = NULL;
    int rc = cpecan_batch_create(&b, m, p, emit, cpk_current_device()); /* single-problem calls: the caller's current device */
    if (rc != CPECAN_OK) return rc;
    int64_t idx = cpecan_batch_add(b, sX, (int64_t)strlen(sX), sY, (int64_t)strlen(sY), anchors, nAnchors, raggedLeft,
                                   raggedRight);
    rc = idx < 0 ? (int)idx : CPECAN_OK;
    if (rc == CPECAN_OK) rc = cpecan_batch_upload(b);
    if (rc == CPECAN_OK) rc = cpecan_batch_run(b, NULL);
    if (rc == CPECAN_OK) rc = cpecan_batch_download(b);
    if (rc != CPECAN_OK) {
        cpecan_batch_destroy(b);
        return rc;
    }
    *out = b;
    return CPECAN_OK;
}

static int copy_list(const cpecan_batch *b, int which, int32_t **triples, int64_t *n) {
    const int32_t *src;
    int64_t cnt;
    int rc = cpecan_batch_result(b, 0, which, &src, &cnt);
    if (rc != CPECAN_OK) return rc;
    *triples = malloc(sizeof(int32_t) * 3 * (size_t)(cnt ? cnt : 1));
    if (!*triples) return CPECAN_ENOMEM;
    memcpy(*triples, src, sizeof(int32_t) * 3 * (size_t)cnt);
    *n = cnt;
    return CPECAN_OK;
}

int cpecan_get_aligned_pairs_using_anchors(const cpecan_model *m, const char *sX, const char *sY,
                                           const int64_t *anchors, int64_t nAnchors, const cpecan_params *p,
                                           int raggedLeft, int raggedRight, int32_t **triples, int64_t *n) {
    if (!triples || !n) return CPECAN_EINVAL;
    cpecan_batch *b = NULL;
    int rc = run_single(&b, m, sX, sY, anchors, nAnchors, p, raggedLeft, raggedRight, CPECAN_EMIT_MATCH);
    if (rc != CPECAN_OK) return rc;
    rc = copy_list(b, 0, triples, n);
    cpecan_batch_destroy(b);
    return rc;
}

int cpecan_get_aligned_pairs_with_indels_using_anchors(const cpecan_model *m, const char *sX, const char *sY,
                                                       const int64_t *anchors, int64_t nAnchors, const cpecan_params *p,
                                                       int raggedLeft, int raggedRight, int32_t **match, int64_t *nMatch,
                                                       int32_t **gapX, int64_t *nGapX, int32_t **gapY, int64_t *nGapY) {
    if (!match || !nMatch || !gapX || !nGapX || !gapY || !nGapY) return CPECAN_EINVAL;
    cpecan_batch *b = NULL;
    int rc = run_single(&b, m, sX, sY, anchors, nAnchors, p, raggedLeft, raggedRight, CPECAN_EMIT_INDEL);
    if (rc != CPECAN_OK) return rc;
    *match = *gapX = *gapY = NULL;
    rc = copy_list(b, 0, match, nMatch);
    if (rc == CPECAN_OK) rc = copy_list(b, 1, gapX, nGapX);
    if (rc == CPECAN_OK) rc = copy_list(b, 2, gapY, nGapY);
    if (rc != CPECAN_OK) {
        free(*match);
        free(*gapX);
        free(*gapY);
    }
    cpecan_batch_destroy(b);
    return rc;
}

int cpecan_compute_forward_probability(const cpecan_model *m, const char *sX, const char *sY, const int64_t *anchors,
                                       int64_t nAnchors, const cpecan_params *p, int raggedLeft, int raggedRight,
                                       double *logProb) {
    if (!logProb) return CPECAN_EINVAL;
    cpecan_batch *b = NULL;
    int rc = run_single(&b, m, sX, sY, anchors, nAnchors, p, raggedLeft, raggedRight, CPECAN_EMIT_FORWARD);
    if (rc != CPECAN_OK) return rc;
    rc = cpecan_batch_forward_prob(b, 0, logProb);
    cpecan_batch_destroy(b);
    return rc;
}
