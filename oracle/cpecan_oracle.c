/*
 * cpecan_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See cpecan_oracle.h.
 *
 * Plain-C restatement of cPecan's banded pair-HMM posterior path.  Every function cites the
 * reference lines whose behaviour it restates (paths relative to the reference root).
 * Layout differences from the reference (deliberate): flat arrays instead of stList /
 * DpMatrix / malloc-per-diagonal, a data-driven transition table instead of callback
 * code, explicit "alive" flags instead of create/delete of diagonals.
 */
#include "cpecan_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)
#define SYM_N 4

void orc_free(void *p) { free(p); }

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}

/* ------------------------------------------------------------------------------------------
 * logAdd: impl/pairwiseAligner.c:287-307.  Four cubics in Horner form whose coefficients are
 * float literals (so each is the float32-rounded value widened to double), cutoff 7.5.
 * ---------------------------------------------------------------------------------------- */
static const float kLogAddCoef[4][4] = {
    /* c3, c2, c1, c0 for d <= 1.0, <= 2.5, <= 4.5, else */
    {-0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f},
    {-0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f},
    {-0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f},
    {-0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f}};

static inline double softplus_poly(double d) {
    const float *c = kLogAddCoef[d <= 1.00f ? 0 : (d <= 2.50f ? 1 : (d <= 4.50f ? 2 : 3))];
    double r = (double)c[0] * d;
    r = r + (double)c[1];
    r = r * d;
    r = r + (double)c[2];
    r = r * d;
    r = r + (double)c[3];
    return r;
}

double orc_logAdd(double x, double y) {
    if (x < y) {
        return (x == NEG_INF || y - x >= 7.5) ? y : softplus_poly(y - x) + x;
    }
    return (y == NEG_INF || x - y >= 7.5) ? x : softplus_poly(x - y) + y;
}

/* impl/pairwiseAligner.c:317-334 */
int32_t orc_symbol(char ch) {
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return SYM_N;
    }
}

static int32_t *symbols_of(const char *s, int64_t n) {
    int32_t *out = xmalloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) out[i] = orc_symbol(s[i]);
    return out;
}

/* impl/pairwiseAligner.c:1334-1348 (only the fields the DP reads) */
void orc_params_default(OrcParams *p) {
    p->threshold = 0.01;
    p->minDiagsBetweenTraceBack = 1000;
    p->traceBackDiagonals = 40;
    p->diagonalExpansion = 20;
    p->splitMatrixBiggerThanThis = (int64_t)3000 * 3000;
    p->dynamicAnchorExpansion = 0;
}

/* ------------------------------------------------------------------------------------------
 * Diagonals and the band: impl/pairwiseAligner.c:30-43, 94-234.
 * ---------------------------------------------------------------------------------------- */
int orc_diagonal_valid(int64_t xay, int64_t xmyL, int64_t xmyR) {
    return !((xay + xmyL) % 2 != 0 || (xay + xmyR) % 2 != 0 || xmyL > xmyR);
}

static inline int64_t diag_width(OrcDiagonal d) { return (d.xmyR - d.xmyL) / 2 + 1; }

static int64_t match_parity(int64_t xay, int64_t xmy) { return (xay + xmy) % 2 == 0 ? xmy : xmy + 1; }

static int64_t clampi(int64_t z, int64_t hi) { return z < 0 ? 0 : (z > hi ? hi : z); }

/* Intersect anti-diagonal xay with the rectangle whose min-xmy corner is (xL,yL) and max-xmy
 * corner is (xU,yU): impl/pairwiseAligner.c:104-122. */
static int band_cut(int64_t xay, int64_t xL, int64_t yL, int64_t xU, int64_t yU, OrcDiagonal *out) {
    int64_t lo = match_parity(xay, xL - yL);
    int64_t hi = match_parity(xay, xU - yU);
    int64_t x = (xay + lo) / 2;
    if (x < xL) lo += 2 * (xL - x);
    int64_t y = (xay - lo) / 2;
    if (yL < y) lo += 2 * (y - yL);
    x = (xay + hi) / 2;
    if (xU < x) hi -= 2 * (x - xU);
    y = (xay - hi) / 2;
    if (y < yU) hi -= 2 * (yU - y);
    if (!orc_diagonal_valid(xay, lo, hi)) return -1;
    out->xay = xay;
    out->xmyL = lo;
    out->xmyR = hi;
    return 0;
}

/* band_construct (:183-234) and band_constructDynamic (:128-181) in one routine. */
int orc_band(const int64_t *anchors, int64_t n, int64_t lX, int64_t lY, int64_t expansion, int dynamic,
             OrcDiagonal *out) {
    int64_t next = 0;           /* index of the next unused anchor */
    int64_t pSum = 0, pDiff = 0; /* previous anchor in matrix coords, as x+y / x-y */
    int64_t nSum = 0, nDiff = 0; /* next anchor */
    int64_t xL = 0, yL = 0, xU = 0, yU = 0;
    int64_t e = dynamic ? 0 : expansion;
    for (int64_t xay = 0; xay <= lX + lY; xay++) {
        if (band_cut(xay, xL, yL, xU, yU, &out[xay]) != 0) return -1;
        if (nSum != xay) continue;
        /* reached the "next" anchor: it becomes "previous", fetch a new "next" */
        pSum = nSum;
        pDiff = nDiff;
        int64_t ax = lX, ay = lY;
        if (next < n) {
            ax = anchors[3 * next] + 1; /* matrix coordinates are sequence coordinates + 1 */
            ay = anchors[3 * next + 1] + 1;
            if (dynamic) e = anchors[3 * next + 2];
            next++;
        }
        nSum = ax + ay;
        nDiff = ax - ay;
        xL = clampi((pSum + (pDiff - e)) / 2, lX);
        yL = clampi((nSum - (nDiff - e)) / 2, lY);
        xU = clampi((nSum + (nDiff + e)) / 2, lX);
        yU = clampi((pSum - (pDiff + e)) / 2, lY);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Split rectangles: impl/pairwiseAligner.c:1206-1257.
 * ---------------------------------------------------------------------------------------- */
static int split_gap(int64_t *x1, int64_t *y1, int64_t x2, int64_t y2, int64_t x3, int64_t y3, int64_t *out,
                     int64_t *nOut, int64_t maxMatrix, int skip) {
    int64_t gx = x3 - x2, gy = y3 - y2;
    if (gx * gy <= maxMatrix) return 0;
    int64_t side = (int64_t)sqrt((double)maxMatrix);
    int64_t hX = gx / 2 > side ? side : gx / 2;
    int64_t hY = gy / 2 > side ? side : gy / 2;
    if (!skip) {
        int64_t *r = out + 4 * (*nOut)++;
        r[0] = *x1; r[1] = *y1; r[2] = x2 + hX; r[3] = y2 + hY;
    }
    *x1 = x3 - hX;
    *y1 = y3 - hY;
    return 1;
}

int64_t orc_split_points(const int64_t *anchors, int64_t n, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                         int raggedLeft, int raggedRight, int64_t *out) {
    int64_t x1 = 0, y1 = 0, x2 = 0, y2 = 0, cnt = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t x3 = anchors[3 * i], y3 = anchors[3 * i + 1];
        split_gap(&x1, &y1, x2, y2, x3, y3, out, &cnt, maxMatrixSize, raggedLeft && i == 0);
        x2 = x3 + 1;
        y2 = y3 + 1;
    }
    int tailSplit = split_gap(&x1, &y1, x2, y2, lX, lY, out, &cnt, maxMatrixSize, raggedLeft && n == 0);
    if (!tailSplit || !raggedRight) {
        int64_t *r = out + 4 * cnt++;
        r[0] = x1; r[1] = y1; r[2] = lX; r[3] = lY;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * Model: impl/stateMachine.c:255-823.
 * ---------------------------------------------------------------------------------------- */
enum { ST_M = 0, ST_SX = 1, ST_SY = 2, ST_LX = 3, ST_LY = 4 };

/* Named transition log-probabilities; union of StateMachine5 (:377-399) and StateMachine3 (:631-646) */
typedef struct {
    double matchContinue;
    double matchFromShortX, matchFromShortY, matchFromLongX, matchFromLongY;
    double shortOpenX, shortOpenY, shortExtendX, shortExtendY, shortSwitchToX, shortSwitchToY;
    double longOpenX, longOpenY, longExtendX, longExtendY, longSwitchToX, longSwitchToY;
    double matchEm[16], gapXEm[4], gapYEm[4];
} NamedParams;

static void default_emissions(NamedParams *q) {
    /* impl/stateMachine.c:269-292 */
    const double M = -2.1149196655034745, TV = -4.5691014376830479, TS = -3.9833860032220842;
    const double tab[16] = {M, TV, TS, TV, TV, M, TV, TS, TS, TV, M, TV, TV, TS, TV, M};
    memcpy(q->matchEm, tab, sizeof tab);
    for (int i = 0; i < 4; i++) q->gapXEm[i] = q->gapYEm[i] = -1.6094379124341003;
}

static void add_tr(OrcModel *m, int block, int from, int to, double tP) {
    OrcTransition *t = &m->tr[m->nTransitions++];
    t->block = block; t->from = from; t->to = to; t->tP = tP;
}

static void finish_model(OrcModel *m, const NamedParams *q, int32_t type) {
    memset(m, 0, sizeof *m);
    m->type = type;
    m->matchState = ST_M; m->gapXState = ST_SX; m->gapYState = ST_SY;
    for (int x = 0; x < 5; x++) {
        m->gapXEm[x] = x == SYM_N ? -1.386294361 : q->gapXEm[x]; /* :351-357 */
        m->gapYEm[x] = x == SYM_N ? -1.386294361 : q->gapYEm[x];
        for (int y = 0; y < 5; y++) /* :359-366 */
            m->matchEm[x * 5 + y] = (x == SYM_N || y == SYM_N) ? -2.772588722 : q->matchEm[x * 4 + y];
    }
    for (int s = 0; s < ORC_MAX_STATES; s++) m->start[s] = m->raggedStart[s] = m->end[s] = m->raggedEnd[s] = NEG_INF;
    if (type == ORC_FIVE_STATE || type == ORC_FIVE_STATE_ASYM) {
        m->S = 5;
        /* per-cell ordered list, impl/stateMachine.c:450-480 (switch transitions are commented out there) */
        add_tr(m, 0, ST_M, ST_SX, q->shortOpenX);
        add_tr(m, 0, ST_SX, ST_SX, q->shortExtendX);
        add_tr(m, 0, ST_M, ST_LX, q->longOpenX);
        add_tr(m, 0, ST_LX, ST_LX, q->longExtendX);
        add_tr(m, 1, ST_M, ST_M, q->matchContinue);
        add_tr(m, 1, ST_SX, ST_M, q->matchFromShortX);
        add_tr(m, 1, ST_SY, ST_M, q->matchFromShortY);
        add_tr(m, 1, ST_LX, ST_M, q->matchFromLongX);
        add_tr(m, 1, ST_LY, ST_M, q->matchFromLongY);
        add_tr(m, 2, ST_M, ST_SY, q->shortOpenY);
        add_tr(m, 2, ST_SY, ST_SY, q->shortExtendY);
        add_tr(m, 2, ST_M, ST_LY, q->longOpenY);
        add_tr(m, 2, ST_LY, ST_LY, q->longExtendY);
        m->start[ST_M] = 0.0;                          /* :401-405 */
        m->raggedStart[ST_LX] = m->raggedStart[ST_LY] = 0.0; /* :407-410 */
        m->end[ST_M] = q->matchContinue;               /* :412-429 */
        m->end[ST_SX] = q->matchFromShortX;
        m->end[ST_SY] = q->matchFromShortY;
        m->end[ST_LX] = q->matchFromLongX;
        m->end[ST_LY] = q->matchFromLongY;
        m->raggedEnd[ST_M] = q->longOpenX;             /* :431-448 */
        m->raggedEnd[ST_SX] = q->longOpenX;
        m->raggedEnd[ST_SY] = q->longOpenY;
        m->raggedEnd[ST_LX] = q->longExtendX;
        m->raggedEnd[ST_LY] = q->longExtendY;
    } else {
        m->S = 3;
        /* impl/stateMachine.c:689-714 (switch transitions ARE used here) */
        add_tr(m, 0, ST_M, ST_SX, q->shortOpenX);
        add_tr(m, 0, ST_SX, ST_SX, q->shortExtendX);
        add_tr(m, 0, ST_SY, ST_SX, q->shortSwitchToX);
        add_tr(m, 1, ST_M, ST_M, q->matchContinue);
        add_tr(m, 1, ST_SX, ST_M, q->matchFromShortX);
        add_tr(m, 1, ST_SY, ST_M, q->matchFromShortY);
        add_tr(m, 2, ST_M, ST_SY, q->shortOpenY);
        add_tr(m, 2, ST_SY, ST_SY, q->shortExtendY);
        add_tr(m, 2, ST_SX, ST_SY, q->shortSwitchToY);
        m->start[ST_M] = 0.0;                           /* :648-652 */
        m->raggedStart[ST_SX] = m->raggedStart[ST_SY] = 0.0; /* :654-657 */
        m->end[ST_M] = q->matchContinue;                /* :659-672 */
        m->end[ST_SX] = q->matchFromShortX;
        m->end[ST_SY] = q->matchFromShortY;
        m->raggedEnd[ST_M] = (q->shortOpenX + q->shortOpenY) / 2.0; /* :674-687 */
        m->raggedEnd[ST_SX] = q->shortExtendX;
        m->raggedEnd[ST_SY] = q->shortExtendY;
    }
}

void orc_model_default(OrcModel *m, int32_t type) {
    NamedParams q;
    memset(&q, 0, sizeof q);
    default_emissions(&q);
    q.matchContinue = -0.030064059121770816;
    q.matchFromShortX = q.matchFromShortY = -1.272871422049609;
    q.shortExtendX = q.shortExtendY = -0.3388262689231553;
    q.shortSwitchToX = q.shortSwitchToY = -4.910694825551255;
    if (type == ORC_FIVE_STATE || type == ORC_FIVE_STATE_ASYM) { /* impl/stateMachine.c:482-501 */
        q.matchFromLongX = q.matchFromLongY = -5.673280173170473;
        q.shortOpenX = q.shortOpenY = -4.34381910900448;
        q.longOpenX = q.longOpenY = -6.30810595366929;
        q.longExtendX = q.longExtendY = -0.003442492794189331;
        q.longSwitchToX = q.longSwitchToY = -6.30810595366929;
    } else { /* impl/stateMachine.c:716-726 */
        q.shortOpenX = q.shortOpenY = -4.21256642;
    }
    finish_model(m, &q, type);
}

void orc_hmm_init(OrcHmm *h, int32_t type, double pseudo) { /* impl/stateMachine.c:23-48 */
    h->type = type;
    h->S = (type == ORC_FIVE_STATE || type == ORC_FIVE_STATE_ASYM) ? 5 : 3;
    for (int i = 0; i < ORC_MAX_STATES * ORC_MAX_STATES; i++) h->T[i] = pseudo;
    for (int i = 0; i < ORC_MAX_STATES * 16; i++) h->E[i] = pseudo;
    h->likelihood = 0.0;
}

void orc_hmm_normalise(OrcHmm *h) { /* impl/stateMachine.c:88-112 */
    int S = h->S;
    for (int from = 0; from < S; from++) {
        double tot = 0.0;
        for (int to = 0; to < S; to++) tot += h->T[from * S + to];
        for (int to = 0; to < S; to++) h->T[from * S + to] = h->T[from * S + to] / tot;
    }
    for (int s = 0; s < S; s++) {
        double tot = 0.0;
        for (int i = 0; i < 16; i++) tot += h->E[s * 16 + i];
        for (int i = 0; i < 16; i++) h->E[s * 16 + i] = h->E[s * 16 + i] / tot;
    }
}

static double hT(const OrcHmm *h, int from, int to) { return h->T[from * h->S + to]; }
static double hE(const OrcHmm *h, int s, int x, int y) { return h->E[s * 16 + x * 4 + y]; }

/* impl/stateMachine.c:319-349 */
static void gap_emissions_from(const OrcHmm *h, double *out, const int *xStates, int nX, const int *yStates, int nY) {
    for (int i = 0; i < 4; i++) out[i] = 0.0;
    for (int k = 0; k < nX; k++)
        for (int x = 0; x < 4; x++)
            for (int y = 0; y < 4; y++) out[x] += hE(h, xStates[k], x, y);
    for (int k = 0; k < nY; k++)
        for (int x = 0; x < 4; x++)
            for (int y = 0; y < 4; y++) out[y] += hE(h, yStates[k], x, y);
    double tot = 0.0;
    for (int i = 0; i < 4; i++) tot += out[i];
    for (int i = 0; i < 4; i++) out[i] = log(out[i] / tot);
}

static void match_emissions_from(const OrcHmm *h, double *out, int symmetric) { /* :298-317 */
    for (int x = 0; x < 4; x++) {
        if (!symmetric) {
            for (int y = 0; y < 4; y++) out[x * 4 + y] = log(hE(h, ST_M, x, y));
            continue;
        }
        out[x * 4 + x] = log(hE(h, ST_M, x, x));
        for (int y = x + 1; y < 4; y++) {
            double v = log((hE(h, ST_M, x, y) + hE(h, ST_M, y, x)) / 2.0);
            out[x * 4 + y] = v;
            out[y * 4 + x] = v;
        }
    }
}

static void swapd(double *a, double *b) { double c = *a; *a = *b; *b = c; }

/* hmm_getStateMachine and the four load routines: impl/stateMachine.c:529-620, 747-819 */
int orc_model_from_hmm(OrcModel *m, const OrcHmm *h) {
    NamedParams q;
    memset(&q, 0, sizeof q);
    const int xs5[2] = {ST_SX, ST_LX}, ys5[2] = {ST_SY, ST_LY}, xs3[1] = {ST_SX}, ys3[1] = {ST_SY};
    switch (h->type) {
    case ORC_FIVE_STATE: { /* :576-620 */
        q.matchContinue = log(hT(h, ST_M, ST_M));
        q.matchFromShortX = log((hT(h, ST_SX, ST_M) + hT(h, ST_SY, ST_M)) / 2);
        q.matchFromLongX = log((hT(h, ST_LX, ST_M) + hT(h, ST_LY, ST_M)) / 2);
        q.shortOpenX = log((hT(h, ST_M, ST_SX) + hT(h, ST_M, ST_SY)) / 2);
        q.shortExtendX = log((hT(h, ST_SX, ST_SX) + hT(h, ST_SY, ST_SY)) / 2);
        q.shortSwitchToX = log((hT(h, ST_SX, ST_SY) + hT(h, ST_SY, ST_SX)) / 2);
        q.longOpenX = log((hT(h, ST_M, ST_LX) + hT(h, ST_M, ST_LY)) / 2);
        q.longExtendX = log((hT(h, ST_LX, ST_LX) + hT(h, ST_LY, ST_LY)) / 2);
        q.longSwitchToX = log((hT(h, ST_LX, ST_LY) + hT(h, ST_LY, ST_LX)) / 2);
        if (q.shortExtendX > q.longExtendX) {
            swapd(&q.shortExtendX, &q.longExtendX);
            swapd(&q.matchFromShortX, &q.matchFromLongX);
            swapd(&q.shortOpenX, &q.longOpenX);
            swapd(&q.shortSwitchToX, &q.longSwitchToX);
        }
        q.matchFromShortY = q.matchFromShortX; q.matchFromLongY = q.matchFromLongX;
        q.shortOpenY = q.shortOpenX; q.shortExtendY = q.shortExtendX; q.shortSwitchToY = q.shortSwitchToX;
        q.longOpenY = q.longOpenX; q.longExtendY = q.longExtendX; q.longSwitchToY = q.longSwitchToX;
        match_emissions_from(h, q.matchEm, 1);
        gap_emissions_from(h, q.gapXEm, xs5, 2, ys5, 2);
        gap_emissions_from(h, q.gapYEm, xs5, 2, ys5, 2);
        break;
    }
    case ORC_FIVE_STATE_ASYM: { /* :529-574 */
        q.matchContinue = log(hT(h, ST_M, ST_M));
        q.matchFromShortX = log(hT(h, ST_SX, ST_M));
        q.matchFromLongX = log(hT(h, ST_LX, ST_M));
        q.shortOpenX = log(hT(h, ST_M, ST_SX));
        q.shortExtendX = log(hT(h, ST_SX, ST_SX));
        q.shortSwitchToX = log(hT(h, ST_SY, ST_SX));
        q.longOpenX = log(hT(h, ST_M, ST_LX));
        q.longExtendX = log(hT(h, ST_LX, ST_LX));
        q.longSwitchToX = log(hT(h, ST_LY, ST_LX));
        if (q.shortExtendX > q.longExtendX) {
            swapd(&q.shortExtendX, &q.longExtendX);
            swapd(&q.matchFromShortX, &q.matchFromLongX);
            swapd(&q.shortOpenX, &q.longOpenX);
            swapd(&q.shortSwitchToX, &q.longSwitchToX);
        }
        q.matchFromShortY = log(hT(h, ST_SY, ST_M));
        q.matchFromLongY = log(hT(h, ST_LY, ST_M));
        q.shortOpenY = log(hT(h, ST_M, ST_SY));
        q.shortExtendY = log(hT(h, ST_SY, ST_SY));
        q.shortSwitchToY = log(hT(h, ST_SX, ST_SY));
        q.longOpenY = log(hT(h, ST_M, ST_LY));
        q.longExtendY = log(hT(h, ST_LY, ST_LY));
        q.longSwitchToY = log(hT(h, ST_LX, ST_LY));
        if (q.shortExtendY > q.longExtendY) {
            swapd(&q.shortExtendY, &q.longExtendY);
            swapd(&q.matchFromShortY, &q.matchFromLongY);
            swapd(&q.shortOpenY, &q.longOpenY);
            swapd(&q.shortSwitchToY, &q.longSwitchToY);
        }
        match_emissions_from(h, q.matchEm, 0);
        gap_emissions_from(h, q.gapXEm, xs5, 2, NULL, 0);
        gap_emissions_from(h, q.gapYEm, NULL, 0, ys5, 2);
        break;
    }
    case ORC_THREE_STATE: { /* :767-789 */
        q.matchContinue = log(hT(h, ST_M, ST_M));
        q.matchFromShortX = q.matchFromShortY = log((hT(h, ST_SX, ST_M) + hT(h, ST_SY, ST_M)) / 2.0);
        q.shortOpenX = q.shortOpenY = log((hT(h, ST_M, ST_SX) + hT(h, ST_M, ST_SY)) / 2.0);
        q.shortExtendX = q.shortExtendY = log((hT(h, ST_SX, ST_SX) + hT(h, ST_SY, ST_SY)) / 2.0);
        q.shortSwitchToX = q.shortSwitchToY = log((hT(h, ST_SY, ST_SX) + hT(h, ST_SX, ST_SY)) / 2.0);
        match_emissions_from(h, q.matchEm, 1);
        gap_emissions_from(h, q.gapXEm, xs3, 1, ys3, 1);
        gap_emissions_from(h, q.gapYEm, xs3, 1, ys3, 1);
        break;
    }
    case ORC_THREE_STATE_ASYM: { /* :747-765 */
        q.matchContinue = log(hT(h, ST_M, ST_M));
        q.matchFromShortX = log(hT(h, ST_SX, ST_M));
        q.matchFromShortY = log(hT(h, ST_SY, ST_M));
        q.shortOpenX = log(hT(h, ST_M, ST_SX));
        q.shortOpenY = log(hT(h, ST_M, ST_SY));
        q.shortExtendX = log(hT(h, ST_SX, ST_SX));
        q.shortExtendY = log(hT(h, ST_SY, ST_SY));
        q.shortSwitchToX = log(hT(h, ST_SY, ST_SX));
        q.shortSwitchToY = log(hT(h, ST_SX, ST_SY));
        match_emissions_from(h, q.matchEm, 0);
        gap_emissions_from(h, q.gapXEm, xs3, 1, NULL, 0);
        gap_emissions_from(h, q.gapYEm, NULL, 0, ys3, 1);
        break;
    }
    default:
        return -1;
    }
    finish_model(m, &q, h->type);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * One DP cell.  Walks the ordered transition table; each term is from + (eP + tP)
 * (impl/pairwiseAligner.c:382-395).  A NULL neighbour skips its block
 * (impl/stateMachine.c:454,463,471).
 * ---------------------------------------------------------------------------------------- */
static inline double emission_of(const OrcModel *m, int block, int32_t cX, int32_t cY) {
    return block == 0 ? m->gapXEm[cX] : (block == 1 ? m->matchEm[cX * 5 + cY] : m->gapYEm[cY]);
}

void orc_cell_forward(const OrcModel *m, double *current, const double *lower, const double *middle,
                      const double *upper, int32_t cX, int32_t cY) {
    const double *nbr[3] = {lower, middle, upper};
    for (int i = 0; i < m->nTransitions; i++) {
        const OrcTransition *t = &m->tr[i];
        const double *src = nbr[t->block];
        if (!src) continue;
        double w = emission_of(m, t->block, cX, cY) + t->tP;
        current[t->to] = orc_logAdd(current[t->to], src[t->from] + w);
    }
}

/* The reference's backward step is a scatter from the current cell into its three
 * earlier neighbours (impl/pairwiseAligner.c:392-395); kept as a scatter here so the
 * oracle is an independent check of the HIP kernel's gather formulation. */
void orc_cell_backward(const OrcModel *m, const double *current, double *lower, double *middle, double *upper,
                       int32_t cX, int32_t cY) {
    double *nbr[3] = {lower, middle, upper};
    for (int i = 0; i < m->nTransitions; i++) {
        const OrcTransition *t = &m->tr[i];
        double *dst = nbr[t->block];
        if (!dst) continue;
        double w = emission_of(m, t->block, cX, cY) + t->tP;
        dst[t->from] = orc_logAdd(dst[t->from], current[t->to] + w);
    }
}

/* impl/pairwiseAligner.c:418-432: posterior of one (transition, emission) event */
static void cell_expectation(const OrcModel *m, const double *current, const double *lower, const double *middle,
                             const double *upper, int32_t cX, int32_t cY, double total, OrcHmm *acc) {
    const double *nbr[3] = {lower, middle, upper};
    for (int i = 0; i < m->nTransitions; i++) {
        const OrcTransition *t = &m->tr[i];
        const double *src = nbr[t->block];
        if (!src) continue;
        double w = emission_of(m, t->block, cX, cY) + t->tP;
        double p = exp(src[t->from] + current[t->to] + w - total);
        acc->T[t->from * acc->S + t->to] += p;
        if (cX < SYM_N && cY < SYM_N) acc->E[t->to * 16 + cX * 4 + cY] += p;
    }
}

/* ------------------------------------------------------------------------------------------
 * DP workspace for one (sub-)alignment: flat F and B arrays over all band cells plus
 * per-diagonal "alive" flags that stand in for dpMatrix_createDiagonal/deleteDiagonal
 * (impl/pairwiseAligner.c:567-586); a dead diagonal reads as NULL (:556-561).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const OrcModel *m;
    const OrcParams *p;
    const int32_t *sx, *sy;
    int64_t lX, lY, N;
    OrcDiagonal *band;
    int64_t *off; /* off[d] = first cell of diagonal d, off[N+1] = total */
    double *F, *B;
    char *fAlive, *bAlive;
} Dp;

static double *cell_at(const Dp *dp, double *arr, const char *alive, int64_t d, int64_t xmy) {
    if (d < 0 || d > dp->N || !alive[d]) return NULL;
    if (xmy < dp->band[d].xmyL || xmy > dp->band[d].xmyR) return NULL; /* :489-495 */
    return arr + (dp->off[d] + (xmy - dp->band[d].xmyL) / 2) * dp->m->S;
}

static void fill_diag(const Dp *dp, double *arr, char *alive, int64_t d, const double *perState) {
    alive[d] = 1;
    int S = dp->m->S;
    double *c = arr + dp->off[d] * S;
    int64_t w = diag_width(dp->band[d]);
    for (int64_t k = 0; k < w; k++)
        for (int s = 0; s < S; s++) c[k * S + s] = perState ? perState[s] : NEG_INF;
}

static inline int32_t sym_x(const Dp *dp, int64_t xay, int64_t xmy) { /* :597-601 */
    int64_t x = (xay + xmy) / 2;
    return x > 0 ? dp->sx[x - 1] : SYM_N;
}
static inline int32_t sym_y(const Dp *dp, int64_t xay, int64_t xmy) { /* :603-607 */
    int64_t y = (xay - xmy) / 2;
    return y > 0 ? dp->sy[y - 1] : SYM_N;
}

static int dp_open(Dp *dp, const OrcModel *m, const OrcParams *p, const char *sX, int64_t lX, const char *sY,
                   int64_t lY, const int64_t *anchors, int64_t n, int forceStaticBand) {
    memset(dp, 0, sizeof *dp);
    dp->m = m; dp->p = p; dp->lX = lX; dp->lY = lY; dp->N = lX + lY;
    dp->sx = symbols_of(sX, lX);
    dp->sy = symbols_of(sY, lY);
    dp->band = xmalloc(sizeof(OrcDiagonal) * (size_t)(dp->N + 1));
    int dynamic = forceStaticBand ? 0 : p->dynamicAnchorExpansion; /* :773 vs :894 */
    if (orc_band(anchors, n, lX, lY, p->diagonalExpansion, dynamic, dp->band) != 0) return -1;
    dp->off = xmalloc(sizeof(int64_t) * (size_t)(dp->N + 2));
    dp->off[0] = 0;
    for (int64_t d = 0; d <= dp->N; d++) dp->off[d + 1] = dp->off[d] + diag_width(dp->band[d]);
    size_t cells = (size_t)dp->off[dp->N + 1];
    dp->F = xmalloc(sizeof(double) * cells * m->S);
    dp->B = xmalloc(sizeof(double) * cells * m->S);
    dp->fAlive = calloc((size_t)dp->N + 2, 1);
    dp->bAlive = calloc((size_t)dp->N + 2, 1);
    return 0;
}

static void dp_close(Dp *dp) {
    free((void *)dp->sx); free((void *)dp->sy); free(dp->band); free(dp->off);
    free(dp->F); free(dp->B); free(dp->fAlive); free(dp->bAlive);
}

/* diagonalCalculationForward: impl/pairwiseAligner.c:609-629 */
static void sweep_forward(Dp *dp, int64_t d) {
    OrcDiagonal g = dp->band[d];
    for (int64_t xmy = g.xmyL; xmy <= g.xmyR; xmy += 2) {
        orc_cell_forward(dp->m, cell_at(dp, dp->F, dp->fAlive, d, xmy), cell_at(dp, dp->F, dp->fAlive, d - 1, xmy - 1),
                         cell_at(dp, dp->F, dp->fAlive, d - 2, xmy), cell_at(dp, dp->F, dp->fAlive, d - 1, xmy + 1),
                         sym_x(dp, d, xmy), sym_y(dp, d, xmy));
    }
}

/* diagonalCalculationBackward: impl/pairwiseAligner.c:631-634 */
static void sweep_backward(Dp *dp, int64_t d) {
    OrcDiagonal g = dp->band[d];
    for (int64_t xmy = g.xmyL; xmy <= g.xmyR; xmy += 2) {
        orc_cell_backward(dp->m, cell_at(dp, dp->B, dp->bAlive, d, xmy), cell_at(dp, dp->B, dp->bAlive, d - 1, xmy - 1),
                          cell_at(dp, dp->B, dp->bAlive, d - 2, xmy), cell_at(dp, dp->B, dp->bAlive, d - 1, xmy + 1),
                          sym_x(dp, d, xmy), sym_y(dp, d, xmy));
    }
}

/* cell_dotProduct :402-408 */
static double dot_states(const double *a, const double *b, int S) {
    double t = a[0] + b[0];
    for (int s = 1; s < S; s++) t = orc_logAdd(t, a[s] + b[s]);
    return t;
}

/* diagonalCalculationTotalProbability: impl/pairwiseAligner.c:636-653 (+ dpDiagonal_dotProduct :513-523) */
static double total_probability(Dp *dp, int64_t d) {
    int S = dp->m->S;
    OrcDiagonal g = dp->band[d];
    double total = NEG_INF;
    for (int64_t xmy = g.xmyL; xmy <= g.xmyR; xmy += 2)
        total = orc_logAdd(total, dot_states(cell_at(dp, dp->F, dp->fAlive, d, xmy),
                                             cell_at(dp, dp->B, dp->bAlive, d, xmy), S));
    /* matches that straddle diagonal d: from F[d-1] into cells of d+1, dotted with B[d+1] */
    if (d + 1 <= dp->N && dp->bAlive[d + 1] && d - 1 >= 0 && dp->fAlive[d - 1]) {
        OrcDiagonal h = dp->band[d + 1];
        double straddle = NEG_INF;
        for (int64_t xmy = h.xmyL; xmy <= h.xmyR; xmy += 2) {
            double tmp[ORC_MAX_STATES];
            for (int s = 0; s < S; s++) tmp[s] = NEG_INF;
            orc_cell_forward(dp->m, tmp, NULL, cell_at(dp, dp->F, dp->fAlive, d - 1, xmy), NULL, sym_x(dp, d + 1, xmy),
                             sym_y(dp, d + 1, xmy));
            straddle = orc_logAdd(straddle, dot_states(tmp, cell_at(dp, dp->B, dp->bAlive, d + 1, xmy), S));
        }
        total = orc_logAdd(total, straddle);
    }
    return total;
}

/* ---- emitters ---- */
typedef struct {
    int64_t *v;
    int64_t n, cap;
} Triples;

static void push3(Triples *t, int64_t a, int64_t b, int64_t c) {
    if (t->n == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 256;
        t->v = realloc(t->v, sizeof(int64_t) * 3 * (size_t)t->cap);
        if (!t->v) abort();
    }
    int64_t *r = t->v + 3 * t->n++;
    r[0] = a; r[1] = b; r[2] = c;
}

/* addPosteriorProb: impl/pairwiseAligner.c:655-664 */
static void keep_if_probable(Triples *out, int64_t x, int64_t y, double prob, double threshold) {
    if (prob >= threshold) {
        if (prob > 1.0) prob = 1.0;
        push3(out, (int64_t)floor(prob * ORC_PROB_1), x - 1, y - 1);
    }
}

typedef struct {
    int mode; /* 0 match posteriors, 1 match+indel posteriors, 2 expectations */
    Triples *match, *gapX, *gapY;
    OrcHmm *acc;
    OrcTrace *trace;
} Emitter;

static void emit_diagonal(Dp *dp, int64_t d, double total, Emitter *e) {
    const OrcModel *m = dp->m;
    OrcDiagonal g = dp->band[d];
    if (e->trace) e->trace->totalUsed[d] = total;
    if (e->mode == 2) { /* diagonalCalculationExpectations :735-746 */
        e->acc->likelihood += total;
        for (int64_t xmy = g.xmyL; xmy <= g.xmyR; xmy += 2)
            cell_expectation(m, cell_at(dp, dp->B, dp->bAlive, d, xmy), cell_at(dp, dp->F, dp->fAlive, d - 1, xmy - 1),
                             cell_at(dp, dp->F, dp->fAlive, d - 2, xmy), cell_at(dp, dp->F, dp->fAlive, d - 1, xmy + 1),
                             sym_x(dp, d, xmy), sym_y(dp, d, xmy), total, e->acc);
        return;
    }
    /* diagonalCalculationPosteriorMatchProbs :666-689 / diagonalCalculationPosteriorProbs :691-733 */
    for (int64_t xmy = g.xmyL; xmy <= g.xmyR; xmy += 2) {
        int64_t x = (d + xmy) / 2, y = (d - xmy) / 2;
        const double *f = cell_at(dp, dp->F, dp->fAlive, d, xmy);
        const double *b = cell_at(dp, dp->B, dp->bAlive, d, xmy);
        if (x > 0 && y > 0) {
            double fb = f[m->matchState] + b[m->matchState];
            if (e->trace) e->trace->fbMatch[dp->off[d] + (xmy - g.xmyL) / 2] = fb;
            keep_if_probable(e->match, x, y, exp(fb - total), dp->p->threshold);
        }
        if (e->mode == 1) {
            if (x > 0) keep_if_probable(e->gapX, x, y, exp((f[m->gapXState] + b[m->gapXState]) - total), dp->p->threshold);
            if (y > 0) keep_if_probable(e->gapY, x, y, exp((f[m->gapYState] + b[m->gapYState]) - total), dp->p->threshold);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * getPosteriorProbsWithBanding: impl/pairwiseAligner.c:756-877.  Forward sweep with periodic
 * partial tracebacks; every diagonal 1..N is emitted exactly once.
 * ---------------------------------------------------------------------------------------- */
static int banded_posteriors(const OrcModel *m, const char *sX, int64_t lX, const char *sY, int64_t lY,
                             const int64_t *anchors, int64_t n, const OrcParams *p, int raggedLeft, int raggedRight,
                             Emitter *e, int64_t *cellsOut) {
    if (lX + lY == 0) { /* :767-770 */
        if (cellsOut) *cellsOut = 0;
        return 0;
    }
    Dp dp;
    if (dp_open(&dp, m, p, sX, lX, sY, lY, anchors, n, 0) != 0) { dp_close(&dp); return -1; }
    const int64_t N = dp.N;
    if (cellsOut) *cellsOut = dp.off[N + 1];
    if (e->trace) {
        OrcTrace *t = e->trace;
        t->nDiagonals = N + 1;
        t->nCells = dp.off[N + 1];
        t->nTracebacks = 0;
        t->cellOffset = xmalloc(sizeof(int64_t) * (size_t)(N + 2));
        memcpy(t->cellOffset, dp.off, sizeof(int64_t) * (size_t)(N + 2));
        t->totalUsed = xmalloc(sizeof(double) * (size_t)(N + 1));
        t->fbMatch = xmalloc(sizeof(double) * (size_t)t->nCells);
        t->forward = xmalloc(sizeof(double) * (size_t)t->nCells * m->S);
        for (int64_t i = 0; i <= N; i++) t->totalUsed[i] = NAN;
        for (int64_t i = 0; i < t->nCells; i++) t->fbMatch[i] = NAN;
    }
    fill_diag(&dp, dp.F, dp.fAlive, 0, raggedLeft ? m->raggedStart : m->start); /* :776-777 */
    if (e->trace) memcpy(e->trace->forward, dp.F, sizeof(double) * m->S);

    int64_t tracedBackTo = 0;
    for (int64_t d = 1; d <= N; d++) {
        fill_diag(&dp, dp.F, dp.fAlive, d, NULL); /* :788 */
        sweep_forward(&dp, d);
        if (e->trace)
            memcpy(e->trace->forward + dp.off[d] * m->S, dp.F + dp.off[d] * m->S,
                   sizeof(double) * (size_t)(diag_width(dp.band[d]) * m->S));
        int atEnd = d == N;
        int tracebackPoint = d >= tracedBackTo + p->minDiagsBetweenTraceBack &&
                             diag_width(dp.band[d]) <= p->diagonalExpansion * 2 + 1; /* :792-793 */
        if (!atEnd && !tracebackPoint) continue;

        if (e->trace) e->trace->nTracebacks++;
        fill_diag(&dp, dp.B, dp.bAlive, d, (atEnd && raggedRight) ? m->raggedEnd : m->end); /* :798-799 */
        if (d > tracedBackTo + 1) fill_diag(&dp, dp.B, dp.bAlive, d - 1, NULL);               /* :800-804 */
        int64_t tracedBackFrom = d - (atEnd ? 0 : p->traceBackDiagonals + 1);                 /* :810 */
        double total = NEG_INF;
        int64_t emitted = 0;
        for (int64_t d2 = d; d2 > tracedBackTo; d2--) {
            if (d2 > tracedBackTo + 2) fill_diag(&dp, dp.B, dp.bAlive, d2 - 2, NULL); /* :815-819 */
            if (d2 > tracedBackTo + 1) sweep_backward(&dp, d2);                       /* :820-822 */
            if (d2 <= tracedBackFrom) {
                if (emitted++ % 10 == 0) total = total_probability(&dp, d2); /* :830-838 */
                emit_diagonal(&dp, d2, total, e);                           /* :840 */
                if (d2 < tracedBackFrom || atEnd) dp.fAlive[d2] = 0;        /* :843-845 */
            }
            if (d2 + 1 <= N) dp.bAlive[d2 + 1] = 0; /* :847-849 */
        }
        dp.bAlive[tracedBackTo + 1] = 0; /* :854 */
        dp.fAlive[tracedBackTo] = 0;     /* :855 */
        tracedBackTo = tracedBackFrom;   /* :852 */
    }
    dp_close(&dp);
    return 0;
}

void orc_trace_free(OrcTrace *t) {
    free(t->cellOffset); free(t->totalUsed); free(t->fbMatch); free(t->forward);
    memset(t, 0, sizeof *t);
}

/* ------------------------------------------------------------------------------------------
 * getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps: impl/pairwiseAligner.c:1273-1326,
 * with the coordinate correction + list reversal of :1259-1271, :1411-1429.
 * ---------------------------------------------------------------------------------------- */
static void drain_reversed(Triples *sub, Triples *dst, int64_t offX, int64_t offY) {
    while (sub->n > 0) {
        int64_t *r = sub->v + 3 * --sub->n;
        push3(dst, r[0], r[1] + offX, r[2] + offY);
    }
}

static int64_t run_regions(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors, int64_t n,
                           const OrcParams *p, int raggedLeft, int raggedRight, int mode, Triples out[3],
                           OrcHmm *acc, OrcTrace *trace) {
    int64_t lX = (int64_t)strlen(sX), lY = (int64_t)strlen(sY);
    int64_t *regions = xmalloc(sizeof(int64_t) * 4 * (size_t)(n + 2));
    int64_t nRegions = orc_split_points(anchors, n, lX, lY, p->splitMatrixBiggerThanThis, raggedLeft, raggedRight, regions);
    int64_t j = 0, cells = 0;
    Triples sub[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int64_t i = 0; i < nRegions; i++) {
        int64_t x1 = regions[4 * i], y1 = regions[4 * i + 1], x2 = regions[4 * i + 2], y2 = regions[4 * i + 3];
        char *subX = xmalloc((size_t)(x2 - x1) + 1), *subY = xmalloc((size_t)(y2 - y1) + 1);
        memcpy(subX, sX + x1, (size_t)(x2 - x1)); subX[x2 - x1] = 0;
        memcpy(subY, sY + y1, (size_t)(y2 - y1)); subY[y2 - y1] = 0;
        int64_t first = j;
        while (j < n && anchors[3 * j] + anchors[3 * j + 1] < x2 + y2) j++; /* :1296-1308 */
        int64_t nSub = j - first;
        int64_t *subAnchors = xmalloc(sizeof(int64_t) * 3 * (size_t)(nSub ? nSub : 1));
        for (int64_t k = 0; k < nSub; k++) {
            subAnchors[3 * k] = anchors[3 * (first + k)] - x1;
            subAnchors[3 * k + 1] = anchors[3 * (first + k) + 1] - y1;
            subAnchors[3 * k + 2] = anchors[3 * (first + k) + 2];
        }
        Emitter e = {mode, &sub[0], &sub[1], &sub[2], acc, trace};
        int64_t c = 0;
        banded_posteriors(m, subX, x2 - x1, subY, y2 - y1, subAnchors, nSub, p, raggedLeft || i > 0,
                          raggedRight || i < nRegions - 1, &e, &c);
        cells += c;
        if (mode != 2) {
            for (int q = 0; q < 3; q++) drain_reversed(&sub[q], &out[q], x1, y1);
        }
        free(subAnchors); free(subX); free(subY);
    }
    for (int q = 0; q < 3; q++) free(sub[q].v);
    free(regions);
    return cells;
}

/* getAlignedPairsUsingAnchors: impl/pairwiseAligner.c:1431-1449 */
int64_t orc_aligned_pairs(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors, int64_t n,
                          const OrcParams *p, int raggedLeft, int raggedRight, int64_t **outTriples) {
    Triples out[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    run_regions(m, sX, sY, anchors, n, p, raggedLeft, raggedRight, 0, out, NULL, NULL);
    *outTriples = out[0].v;
    return out[0].n;
}

int64_t orc_aligned_pairs_traced(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors,
                                 int64_t n, const OrcParams *p, int raggedLeft, int raggedRight,
                                 int64_t **outTriples, OrcTrace *trace) {
    Triples out = {0, 0, 0}, sub = {0, 0, 0}, unused = {0, 0, 0};
    memset(trace, 0, sizeof *trace);
    Emitter e = {0, &sub, &unused, &unused, NULL, trace};
    banded_posteriors(m, sX, (int64_t)strlen(sX), sY, (int64_t)strlen(sY), anchors, n, p, raggedLeft, raggedRight, &e, NULL);
    drain_reversed(&sub, &out, 0, 0);
    free(sub.v);
    *outTriples = out.v;
    return out.n;
}

/* getAlignedPairsWithIndelsUsingAnchors: impl/pairwiseAligner.c:1451-1479 */
void orc_aligned_pairs_with_indels(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors,
                                   int64_t n, const OrcParams *p, int raggedLeft, int raggedRight,
                                   int64_t **match, int64_t *nMatch, int64_t **gapX, int64_t *nGapX,
                                   int64_t **gapY, int64_t *nGapY) {
    Triples out[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    run_regions(m, sX, sY, anchors, n, p, raggedLeft, raggedRight, 1, out, NULL, NULL);
    *match = out[0].v; *nMatch = out[0].n;
    *gapX = out[1].v; *nGapX = out[1].n;
    *gapY = out[2].v; *nGapY = out[2].n;
}

/* getExpectationsUsingAnchors: impl/pairwiseAligner.c:1500-1505 */
void orc_expectations(const OrcModel *m, OrcHmm *acc, const char *sX, const char *sY, const int64_t *anchors,
                      int64_t n, const OrcParams *p, int raggedLeft, int raggedRight) {
    Triples out[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    run_regions(m, sX, sY, anchors, n, p, raggedLeft, raggedRight, 2, out, acc, NULL);
}

int64_t orc_band_cells(const char *sX, const char *sY, const int64_t *anchors, int64_t n, const OrcParams *p,
                       int raggedLeft, int raggedRight) {
    int64_t lX = (int64_t)strlen(sX), lY = (int64_t)strlen(sY);
    int64_t *regions = xmalloc(sizeof(int64_t) * 4 * (size_t)(n + 2));
    int64_t nRegions = orc_split_points(anchors, n, lX, lY, p->splitMatrixBiggerThanThis, raggedLeft, raggedRight, regions);
    int64_t j = 0, cells = 0;
    for (int64_t i = 0; i < nRegions; i++) {
        int64_t x1 = regions[4 * i], y1 = regions[4 * i + 1], x2 = regions[4 * i + 2], y2 = regions[4 * i + 3];
        int64_t first = j;
        while (j < n && anchors[3 * j] + anchors[3 * j + 1] < x2 + y2) j++;
        int64_t nSub = j - first, N = (x2 - x1) + (y2 - y1);
        if (N == 0) continue;
        int64_t *subAnchors = xmalloc(sizeof(int64_t) * 3 * (size_t)(nSub ? nSub : 1));
        for (int64_t k = 0; k < nSub; k++) {
            subAnchors[3 * k] = anchors[3 * (first + k)] - x1;
            subAnchors[3 * k + 1] = anchors[3 * (first + k) + 1] - y1;
            subAnchors[3 * k + 2] = anchors[3 * (first + k) + 2];
        }
        OrcDiagonal *band = xmalloc(sizeof(OrcDiagonal) * (size_t)(N + 1));
        orc_band(subAnchors, nSub, x2 - x1, y2 - y1, p->diagonalExpansion, p->dynamicAnchorExpansion, band);
        for (int64_t d = 0; d <= N; d++) cells += diag_width(band[d]);
        free(band); free(subAnchors);
    }
    free(regions);
    return cells;
}

/* getForwardProbWithBanding / computeForwardProbability: impl/pairwiseAligner.c:879-949.
 * Keeps every forward diagonal; always the static band (:894). */
double orc_forward_prob(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors, int64_t n,
                        const OrcParams *p, int raggedLeft, int raggedRight) {
    int64_t lX = (int64_t)strlen(sX), lY = (int64_t)strlen(sY);
    if (lX + lY == 0) return 0.0; /* LOG_ONE */
    Dp dp;
    if (dp_open(&dp, m, p, sX, lX, sY, lY, anchors, n, 1) != 0) { dp_close(&dp); return NAN; }
    fill_diag(&dp, dp.F, dp.fAlive, 0, raggedLeft ? m->raggedStart : m->start);
    for (int64_t d = 1; d <= dp.N; d++) {
        fill_diag(&dp, dp.F, dp.fAlive, d, NULL);
        sweep_forward(&dp, d);
    }
    fill_diag(&dp, dp.B, dp.bAlive, dp.N, raggedRight ? m->raggedEnd : m->end);
    double total = total_probability(&dp, dp.N);
    dp_close(&dp);
    return total;
}

int64_t orc_batch_aligned_pairs(const OrcModel *m, const char *seqBlob, const int64_t *seqOff,
                                const int64_t *anchors, const int64_t *anchorOff, int64_t nPairs,
                                const OrcParams *p, int raggedLeft, int raggedRight, int nThreads,
                                int64_t *cells) {
    int64_t totalPairs = 0, totalCells = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(nThreads > 0 ? nThreads : 1) reduction(+ : totalPairs, totalCells)
#endif
    for (int64_t i = 0; i < nPairs; i++) {
        const char *sX = seqBlob + seqOff[2 * i], *sY = seqBlob + seqOff[2 * i + 1];
        const int64_t *a = anchors + 3 * anchorOff[i];
        int64_t na = anchorOff[i + 1] - anchorOff[i];
        Triples out[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        totalCells += run_regions(m, sX, sY, a, na, p, raggedLeft, raggedRight, 0, out, NULL, NULL);
        totalPairs += out[0].n;
        free(out[0].v);
    }
    (void)nThreads;
    if (cells) *cells = totalCells;
    return totalPairs;
}

/* The expectation step over a batch (cPecanRealign.c:509-534 with --outputExpectations, one problem after the other;
 * here: OpenMP over problems, every thread summing into an Hmm of its own, the thread sums added at the end -- linear-space
 * sums, order-insensitive at the 1e-5 gate).  Returns the band cells processed. */
int64_t orc_batch_expectations(const OrcModel *m, OrcHmm *acc, const char *seqBlob, const int64_t *seqOff,
                               const int64_t *anchors, const int64_t *anchorOff, int64_t nPairs, const OrcParams *p,
                               int raggedLeft, int raggedRight, int nThreads) {
    int64_t totalCells = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nThreads > 0 ? nThreads : 1) reduction(+ : totalCells)
#endif
    {
        OrcHmm mine = *acc;
        for (int i = 0; i < ORC_MAX_STATES * ORC_MAX_STATES; i++) mine.T[i] = 0.0;
        for (int i = 0; i < ORC_MAX_STATES * 16; i++) mine.E[i] = 0.0;
        mine.likelihood = 0.0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int64_t i = 0; i < nPairs; i++) {
            const char *sX = seqBlob + seqOff[2 * i], *sY = seqBlob + seqOff[2 * i + 1];
            Triples out[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            totalCells += run_regions(m, sX, sY, anchors + 3 * anchorOff[i], anchorOff[i + 1] - anchorOff[i], p, raggedLeft,
                                      raggedRight, 2, out, &mine, NULL);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            for (int i = 0; i < ORC_MAX_STATES * ORC_MAX_STATES; i++) acc->T[i] += mine.T[i];
            for (int i = 0; i < ORC_MAX_STATES * 16; i++) acc->E[i] += mine.E[i];
            acc->likelihood += mine.likelihood;
        }
    }
    (void)nThreads;
    return totalCells;
}

/* ------------------------------------------------------------------------------------------------
 * Consumers of the posterior lists
 * ---------------------------------------------------------------------------------------------- */

/* getIndelProbabilities, impl/pairwiseAligner.c:1519-1534: probability mass of each base that is NOT in a listed pair */
static int64_t *unaligned_mass(const int64_t *triples, int64_t n, int64_t len, int coord) {
    int64_t *mass = malloc(sizeof(int64_t) * (size_t)(len > 0 ? len : 1));
    for (int64_t i = 0; i < len; i++) mass[i] = ORC_PROB_1;
    for (int64_t i = 0; i < n; i++) mass[triples[3 * i + coord]] -= triples[3 * i];
    for (int64_t i = 0; i < len; i++)
        if (mass[i] < 0) mass[i] = 0;
    return mass;
}

void orc_reweight_aligned_pairs(int64_t *triples, int64_t n, int64_t lX, int64_t lY, double gapGamma) {
    if (gapGamma <= 0.0) return; /* :1551 */
    int64_t *mx = unaligned_mass(triples, n, lX, 1), *my = unaligned_mass(triples, n, lY, 2);
    for (int64_t i = 0; i < n; i++) {
        /* :1543: int64 - double * int64, evaluated in double and truncated towards zero on assignment */
        const int64_t w = triples[3 * i] - gapGamma * (mx[triples[3 * i + 1]] + my[triples[3 * i + 2]]);
        triples[3 * i] = w;
    }
    free(mx);
    free(my);
}

static double sum_scores(const int64_t *triples, int64_t n) { /* totalScore, :1578-1585 */
    double t = 0.0;
    for (int64_t i = 0; i < n; i++) t += triples[3 * i];
    return t;
}

double orc_score_by_posterior(int64_t lX, int64_t lY, const int64_t *triples, int64_t n) { /* :1587-1589 */
    return 100.0 * ((lX + lY) == 0 ? 0 : (2.0 * sum_scores(triples, n)) / ((lX + lY) * ORC_PROB_1));
}

double orc_score_by_posterior_ignoring_gaps(const int64_t *triples, int64_t n) { /* :1591-1593 */
    return 100.0 * sum_scores(triples, n) / ((double)n * ORC_PROB_1);
}

/* getCumulativeGapProbs, :1603-1619 */
static int64_t *cumulative_gap_mass(const int64_t *gaps, int64_t n, int64_t len, int coord) {
    int64_t *cum = calloc((size_t)(len > 0 ? len : 1), sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) cum[gaps[3 * i + coord]] += gaps[3 * i];
    for (int64_t i = 1; i < len; i++) cum[i] += cum[i - 1];
    return cum;
}

/* getIndelProb, :1621-1625: gap mass of `length` bases starting at `start` */
static int64_t gap_mass(const int64_t *cum, int64_t start, int64_t length) {
    return length == 0 ? 0 : cum[start + length - 1] - (start > 0 ? cum[start - 1] : 0);
}

int64_t orc_mea_alignment(const int64_t *pairs, int64_t n, const int64_t *gapX, int64_t nGapX, const int64_t *gapY,
                          int64_t nGapY, int64_t lX, int64_t lY, float gapGamma, int64_t *out, double *alignmentScore) {
    double *best = calloc((size_t)n + 1, sizeof(double));   /* score of the best chain ending at pair i */
    int64_t *prev = calloc((size_t)n + 1, sizeof(int64_t)); /* its predecessor */
    char *record = calloc((size_t)n + 1, 1);                /* chain score incl. trailing gaps is a running maximum */
    int64_t *cy = cumulative_gap_mass(gapY, nGapY, lY, 2), *cx = cumulative_gap_mass(gapX, nGapX, lX, 1);
    double top = 0;
    for (int64_t i = 0; i <= n; i++) {
        int64_t w, x, y;
        if (i == n) { /* sentinel pair behind both sequences, :1652-1654 */
            w = 0;
            x = lX;
            y = lY;
        } else {
            w = pairs[3 * i];
            x = pairs[3 * i + 1];
            y = pairs[3 * i + 2];
        }
        /* :1660-1661: int64 + (int64 * float): the product and the sum are float arithmetic */
        double score = w + (gap_mass(cx, 0, x) + gap_mass(cy, 0, y)) * gapGamma;
        int64_t from = -1;
        for (int64_t j = i - 1; j >= 0; j--) {
            const int64_t x2 = pairs[3 * j + 1], y2 = pairs[3 * j + 2];
            if (x2 < x && y2 < y) {
                /* :1673-1675: (int64 + double) + float product, truncated to int64 */
                const int64_t s = w + best[j] + (gap_mass(cx, x2 + 1, x - x2 - 1) + gap_mass(cy, y2 + 1, y - y2 - 1)) * gapGamma;
                if (s > score) {
                    score = s;
                    from = j;
                }
                if (record[j]) break; /* :1685: nothing further back can do better */
            }
        }
        prev[i] = from;
        best[i] = score;
        /* :1695-1696 */
        const double s = score + ((x < lX ? gap_mass(cx, x + 1, lX - x - 1) : 0) + (y < lY ? gap_mass(cy, y + 1, lY - y - 1) : 0)) * gapGamma;
        if (s >= top) {
            top = s;
            record[i] = 1;
        }
    }
    int64_t count = 0;
    for (int64_t i = prev[n]; i >= 0; i = prev[i]) count++;
    int64_t at = count;
    for (int64_t i = prev[n]; i >= 0; i = prev[i]) { /* written back to front == built reversed, then flipped (:1714) */
        at--;
        out[3 * at] = pairs[3 * i];
        out[3 * at + 1] = pairs[3 * i + 1];
        out[3 * at + 2] = pairs[3 * i + 2];
    }
    free(best);
    free(prev);
    free(record);
    free(cx);
    free(cy);
    if (alignmentScore) *alignmentScore = top;
    return count;
}

static int up(char c) { return toupper((unsigned char)c); }

int64_t orc_left_shift_alignment(const int64_t *pairs, int64_t n, const char *sX, const char *sY, int64_t *out) {
    const int64_t lX = (int64_t)strlen(sX), lY = (int64_t)strlen(sY);
    int64_t count = 0; /* pairs are produced from the right end, reversed at the end (:1759) */
    int64_t x = lX, y = lY;
    for (int64_t i = n - 1; i >= 0; i--) {
        const int64_t w = pairs[3 * i], x2 = pairs[3 * i + 1], y2 = pairs[3 * i + 2];
        /* a gap lies between this pair and the last placed one, and the bases left of the gap's right edge match:
         * slide the edge left, borrowing this pair's score (:1737-1744) */
        while ((x - x2 > 1 || y - y2 > 1) && up(sX[x - 1]) == up(sY[y - 1])) {
            out[3 * count] = w;
            out[3 * count + 1] = x - 1;
            out[3 * count + 2] = y - 1;
            count++;
            x--;
            y--;
            if (x2 == x || y2 == y) break; /* slid over an existing pair */
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
    /* left end (:1753-1757); the score is borrowed from the first input pair, 1 for an empty input */
    while (((x > 0) & (y > 0)) && up(sX[x - 1]) == up(sY[y - 1])) {
        out[3 * count] = n > 0 ? pairs[0] : 1;
        out[3 * count + 1] = x - 1;
        out[3 * count + 2] = y - 1;
        count++;
        x--;
        y--;
    }
    for (int64_t a = 0, b = count - 1; a < b; a++, b--)
        for (int f = 0; f < 3; f++) {
            const int64_t t = out[3 * a + f];
            out[3 * a + f] = out[3 * b + f];
            out[3 * b + f] = t;
        }
    return count;
}

/* ---- filterPairwiseAlignmentToMakePairsOrdered, impl/multipleAligner.c:945-972, and the identity scores ---- */

/* getNumberOfMatchingAlignedPairs, impl/pairwiseAligner.c:1562-1570 */
static int64_t matching_pairs(const char *sX, const char *sY, const int64_t *triples, int64_t n) {
    int64_t matches = 0;
    for (int64_t i = 0; i < n; i++) {
        const int cx = up(sX[triples[3 * i + 1]]), cy = up(sY[triples[3 * i + 2]]);
        matches += cx == cy && cx != 'N';
    }
    return matches;
}

double orc_score_by_identity(const char *sX, const char *sY, int64_t lX, int64_t lY, const int64_t *triples, int64_t n) {
    const int64_t matches = matching_pairs(sX, sY, triples, n); /* :1572-1575 */
    return 100.0 * ((lX + lY) == 0 ? 0 : (2.0 * matches) / (lX + lY));
}

double orc_score_by_identity_ignoring_gaps(const char *sX, const char *sY, const int64_t *triples, int64_t n) {
    return 100.0 * matching_pairs(sX, sY, triples, n) / (double)n; /* :1577-1580 */
}

typedef struct {
    int64_t y, pair; /* pair: index of the aligned pair, -1 for the buffering first pair (multipleAligner.c:378) */
    double score;
} FrontierEntry;

static const int64_t *g_sortPairs;
static int cmp_pair_by_xy(const void *a, const void *b) {
    const int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
    const int64_t *p = g_sortPairs + 3 * i, *q = g_sortPairs + 3 * j;
    if (p[1] != q[1]) return p[1] < q[1] ? -1 : 1;
    if (p[2] != q[2]) return p[2] < q[2] ? -1 : 1;
    return i < j ? -1 : (i > j ? 1 : 0);
}

int64_t orc_filter_pairs_ordered(const int64_t *pairs, int64_t n, int64_t lX, int64_t lY, double matchGamma, int64_t *out) {
    /* With two sequences every column holds one base, every weight links one X base to one Y base with
     * numberOfWeights == 1 (multipleAligner.c:140-147), both sequences carry the same number of weights so X stays X
     * (:361-367), and getMultipleSequenceAlignmentProgressive makes the single call of pairwiseAlignColumns (:358-492)
     * restated here.  The st_random() * 0.00001 added to every weight (:145) is left out. */
    (void)lX;
    int64_t *order = malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *pred = malloc(sizeof(int64_t) * (size_t)(n + 1));
    double *score = malloc(sizeof(double) * (size_t)(n + 1));
    char *chosen = calloc((size_t)n + 1, 1);
    FrontierEntry *f = malloc(sizeof(FrontierEntry) * (size_t)(n + 2)); /* bestScoringAlignments, sorted by y (:375) */
    for (int64_t i = 0; i < n; i++) order[i] = i;
    g_sortPairs = pairs;
    qsort(order, (size_t)n, sizeof(int64_t), cmp_pair_by_xy); /* the adjacency lists: per X column, by Y position */
    int64_t nf = 0;
    f[nf++] = (FrontierEntry){-1, -1, 0.0};
    f[nf++] = (FrontierEntry){lY, -2, (double)INT64_MAX}; /* :379 */
    for (int64_t a = 0; a < n;) {
        int64_t b = a;
        while (b < n && pairs[3 * order[b] + 1] == pairs[3 * order[a] + 1]) b++;
        /* all pairs of this X column are scored against the pairs of earlier columns (:389-409) ... */
        for (int64_t k = a; k < b; k++) {
            const int64_t i = order[k];
            const double w = (double)pairs[3 * i] / ORC_PROB_1;
            pred[i] = -3; /* not a candidate */
            if (w >= matchGamma && w > 0.0) {
                int64_t at = 0; /* searchLessThan: the entry with the largest y below this one */
                while (f[at + 1].y < pairs[3 * i + 2]) at++;
                pred[i] = f[at].pair;
                score[i] = f[at].score + w * 1.0;
            }
        }
        /* ... and then put into the frontier from the right (:412-433) */
        for (int64_t k = b - 1; k >= a; k--) {
            const int64_t i = order[k];
            if (pred[i] == -3) continue;
            const int64_t y = pairs[3 * i + 2];
            int64_t at = 0; /* searchGreaterThanOrEqual */
            while (f[at].y < y) at++;
            if (score[i] >= f[at].score || f[at].y > y) {
                int64_t end = at;
                while (score[i] >= f[end].score) end++; /* entries at or right of y that score no better */
                if (end == at) { /* make room */
                    memmove(&f[at + 1], &f[at], sizeof(FrontierEntry) * (size_t)(nf - at));
                    nf++;
                } else if (end > at + 1) {
                    memmove(&f[at + 1], &f[end], sizeof(FrontierEntry) * (size_t)(nf - end));
                    nf -= end - at - 1;
                }
                f[at] = (FrontierEntry){y, i, score[i]};
            }
        }
        a = b;
    }
    /* trace back from the right-most, i.e. best, entry (:437-475) */
    for (int64_t i = f[nf - 2].pair; i >= 0; i = pred[i]) chosen[i] = 1;
    /* filterMultipleAlignedPairs keeps the pairs whose bases share a column (:569-601); the three list conversions
     * (:621-651, :582) pop from the back, which leaves the survivors in reverse input order */
    int64_t count = 0;
    for (int64_t i = n - 1; i >= 0; i--)
        if (chosen[i]) {
            out[3 * count] = pairs[3 * i];
            out[3 * count + 1] = pairs[3 * i + 1];
            out[3 * count + 2] = pairs[3 * i + 2];
            count++;
        }
    free(order);
    free(pred);
    free(score);
    free(chosen);
    free(f);
    return count;
}

/* filterToRemoveOverlap, impl/pairwiseAligner.c:1095-1135: pairs sorted by (x, y, expansion); a pair survives when it is
 * strictly below every later pair in both coordinates (the backward pass, :1101-1110) and strictly above every earlier
 * one (the forward pass, :1116-1131).  out holds n triples; returns the number kept. */
int64_t orc_filter_to_remove_overlap(const int64_t *pairs, int64_t n, int64_t *out) {
    char *inSet = calloc((size_t)n + 1, 1);
    int64_t pX = INT64_MAX, pY = INT64_MAX;
    for (int64_t i = n - 1; i >= 0; i--) {
        const int64_t x = pairs[3 * i], y = pairs[3 * i + 1];
        if (x < pX && y < pY) inSet[i] = 1;
        pX = x < pX ? x : pX;
        pY = y < pY ? y : pY;
    }
    int64_t count = 0;
    pX = INT64_MIN;
    pY = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        const int64_t x = pairs[3 * i], y = pairs[3 * i + 1];
        if (x > pX && y > pY && inSet[i]) {
            out[3 * count] = x;
            out[3 * count + 1] = y;
            out[3 * count + 2] = pairs[3 * i + 2];
            count++;
        }
        pX = x > pX ? x : pX;
        pY = y > pY ? y : pY;
    }
    free(inSet);
    return count;
}
