/*
 * cpecan_dropin.c -- reference-named entry points (include/cpecan_dropin.h) on top of the C ABI.
 * Host-side glue only: list <-> array conversion, struct flattening, error mapping (st_errAbort semantics:
 * print and abort, as impl/pairwiseAligner.c:1044 and impl/stateMachine.c:36 do).
 */
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "cpecan_dropin.h"

static void die(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    abort();
}

/* ---------------- minimal sonLib containers (weak) ---------------- */
struct _stList {
    void **items;
    int64_t length, capacity;
    void (*destructElement)(void *);
};
struct _stIntTuple {
    int64_t length;
    int64_t v[4];
};
#define WEAK __attribute__((weak))

WEAK stList *stList_construct3(int64_t size, void (*destructElement)(void *)) {
    stList *l = calloc(1, sizeof *l);
    if (!l) die("out of memory");
    l->capacity = size > 8 ? size : 8;
    l->items = calloc((size_t)l->capacity, sizeof(void *));
    l->length = size;
    l->destructElement = destructElement;
    return l;
}
WEAK stList *stList_construct(void) { return stList_construct3(0, NULL); }
WEAK void stList_destruct(stList *l) {
    if (!l) return;
    if (l->destructElement)
        for (int64_t i = 0; i < l->length; i++)
            if (l->items[i]) l->destructElement(l->items[i]);
    free(l->items);
    free(l);
}
WEAK int64_t stList_length(stList *l) { return l ? l->length : 0; }
WEAK void *stList_get(stList *l, int64_t i) { return l->items[i]; }
WEAK void stList_append(stList *l, void *item) {
    if (l->length == l->capacity) {
        l->capacity *= 2;
        l->items = realloc(l->items, sizeof(void *) * (size_t)l->capacity);
        if (!l->items) die("out of memory");
    }
    l->items[l->length++] = item;
}
WEAK stIntTuple *stIntTuple_construct3(int64_t a, int64_t b, int64_t c) {
    stIntTuple *t = malloc(sizeof *t);
    if (!t) die("out of memory");
    t->length = 3;
    t->v[0] = a; t->v[1] = b; t->v[2] = c; t->v[3] = 0;
    return t;
}
WEAK stIntTuple *stIntTuple_construct2(int64_t a, int64_t b) {
    stIntTuple *t = stIntTuple_construct3(a, b, 0);
    t->length = 2;
    return t;
}
static stIntTuple *tuple4(int64_t a, int64_t b, int64_t c, int64_t d) {
    stIntTuple *t = stIntTuple_construct3(a, b, c);
    t->length = 4;
    t->v[3] = d;
    return t;
}
WEAK void stIntTuple_destruct(stIntTuple *t) { free(t); }
WEAK int64_t stIntTuple_get(stIntTuple *t, int64_t i) { return t->v[i]; }
WEAK int64_t stIntTuple_length(stIntTuple *t) { return t->length; }

/* ---------------- state machines ---------------- */
typedef struct {
    StateMachine model; /* must be first: callers hold StateMachine* */
    uint64_t magic;
    cpecan_model flat;
} OwnStateMachine;
#define OWN_MAGIC 0x6350656341644d44ull

const cpecan_model *stateMachine_flat(StateMachine *sM) {
    OwnStateMachine *o = (OwnStateMachine *)sM;
    return (sM && o->magic == OWN_MAGIC) ? &o->flat : NULL;
}
static const cpecan_model *flat_or_die(StateMachine *sM) {
    const cpecan_model *m = stateMachine_flat(sM);
    if (!m) die("cpecan_hip: StateMachine with a foreign vtable cannot run on the GPU path");
    return m;
}
/* priors: impl/stateMachine.c:401-448 (five state), :648-687 (three state) */
static double prior_start(StateMachine *sM, int64_t s) { (void)sM; return s == 0 ? 0.0 : -INFINITY; }
static double prior_ragged_start(StateMachine *sM, int64_t s) {
    if (sM->stateNumber == 5) return (s == 3 || s == 4) ? 0.0 : -INFINITY;
    return (s == 1 || s == 2) ? 0.0 : -INFINITY;
}
static double prior_end(StateMachine *sM, int64_t s) {
    const cpecan_model *m = flat_or_die(sM);
    switch (s) {
    case 0: return m->matchContinue;
    case 1: return m->matchFromShortGapX;
    case 2: return m->matchFromShortGapY;
    case 3: return m->matchFromLongGapX;
    case 4: return m->matchFromLongGapY;
    }
    return 0.0;
}
static double prior_ragged_end(StateMachine *sM, int64_t s) {
    const cpecan_model *m = flat_or_die(sM);
    if (sM->stateNumber == 5) {
        switch (s) {
        case 0: case 1: return m->gapLongOpenX;
        case 2: return m->gapLongOpenY;
        case 3: return m->gapLongExtendX;
        case 4: return m->gapLongExtendY;
        }
        return 0.0;
    }
    switch (s) {
    case 0: return (m->gapShortOpenX + m->gapShortOpenY) / 2.0;
    case 1: return m->gapShortExtendX;
    case 2: return m->gapShortExtendY;
    }
    return 0.0;
}
/* sM->cellCalculate (inc/stateMachine.h:47-50; stateMachine5_cellCalculate impl/stateMachine.c:450-480,
 * stateMachine3_cellCalculate :689-714): the ordered list of transitions into `current` handed to the CALLER's
 * doTransition, one call per transition, blocks of absent neighbours skipped.  This is the vtable's contract and holds no
 * arithmetic of the recurrence -- what a transition does (a logAdd fold forwards or backwards, an expectation update) is
 * the callback's business; the library's own DP never comes through here (it runs on the GPU, and cell_calculateForward /
 * Backward below evaluate the reference's two callbacks there).  A symbol outside a, c, g, t emits as n
 * (emissions_getGapProb / getMatchProb, :351-366). */
static void cell_calculate_dispatch(StateMachine *sM, double *current, double *lower, double *middle, double *upper, Symbol cX,
                                    Symbol cY, void (*doTransition)(double *, double *, int64_t, int64_t, double, double, void *),
                                    void *extraArgs) {
    const cpecan_model *m = flat_or_die(sM);
    const int x = (int)cX, y = (int)cY;
    const int nX = x < 0 || x > 3, nY = y < 0 || y > 3;
    const double eX = nX ? -1.386294361 : m->emissionGapX[x], eY = nY ? -1.386294361 : m->emissionGapY[y];
    const double eM = (nX || nY) ? -2.772588722 : m->emissionMatch[x * 4 + y];
    if (sM->stateNumber == 5) { /* 0 match, 1 shortGapX, 2 shortGapY, 3 longGapX, 4 longGapY */
        if (lower) {
            doTransition(lower, current, 0, 1, eX, m->gapShortOpenX, extraArgs);
            doTransition(lower, current, 1, 1, eX, m->gapShortExtendX, extraArgs);
            doTransition(lower, current, 0, 3, eX, m->gapLongOpenX, extraArgs);
            doTransition(lower, current, 3, 3, eX, m->gapLongExtendX, extraArgs);
        }
        if (middle) {
            doTransition(middle, current, 0, 0, eM, m->matchContinue, extraArgs);
            doTransition(middle, current, 1, 0, eM, m->matchFromShortGapX, extraArgs);
            doTransition(middle, current, 2, 0, eM, m->matchFromShortGapY, extraArgs);
            doTransition(middle, current, 3, 0, eM, m->matchFromLongGapX, extraArgs);
            doTransition(middle, current, 4, 0, eM, m->matchFromLongGapY, extraArgs);
        }
        if (upper) {
            doTransition(upper, current, 0, 2, eY, m->gapShortOpenY, extraArgs);
            doTransition(upper, current, 2, 2, eY, m->gapShortExtendY, extraArgs);
            doTransition(upper, current, 0, 4, eY, m->gapLongOpenY, extraArgs);
            doTransition(upper, current, 4, 4, eY, m->gapLongExtendY, extraArgs);
        }
        return;
    }
    if (lower) { /* 0 match, 1 gapX, 2 gapY */
        doTransition(lower, current, 0, 1, eX, m->gapShortOpenX, extraArgs);
        doTransition(lower, current, 1, 1, eX, m->gapShortExtendX, extraArgs);
        doTransition(lower, current, 2, 1, eX, m->gapShortSwitchToX, extraArgs);
    }
    if (middle) {
        doTransition(middle, current, 0, 0, eM, m->matchContinue, extraArgs);
        doTransition(middle, current, 1, 0, eM, m->matchFromShortGapX, extraArgs);
        doTransition(middle, current, 2, 0, eM, m->matchFromShortGapY, extraArgs);
    }
    if (upper) {
        doTransition(upper, current, 0, 2, eY, m->gapShortOpenY, extraArgs);
        doTransition(upper, current, 2, 2, eY, m->gapShortExtendY, extraArgs);
        doTransition(upper, current, 1, 2, eY, m->gapShortSwitchToY, extraArgs);
    }
}
static StateMachine *wrap_model(const cpecan_model *m) {
    OwnStateMachine *o = calloc(1, sizeof *o);
    if (!o) die("out of memory");
    o->magic = OWN_MAGIC;
    o->flat = *m;
    o->model.type = (StateMachineType)m->type;
    o->model.stateNumber = (m->type == CPECAN_FIVE_STATE || m->type == CPECAN_FIVE_STATE_ASYM) ? 5 : 3;
    o->model.matchState = 0;
    o->model.gapXState = 1;
    o->model.gapYState = 2;
    o->model.startStateProb = prior_start;
    o->model.endStateProb = prior_end;
    o->model.raggedStartStateProb = prior_ragged_start;
    o->model.raggedEndStateProb = prior_ragged_end;
    o->model.cellCalculate = cell_calculate_dispatch;
    return &o->model;
}
StateMachine *stateMachine5_construct(StateMachineType type) {
    if (type != fiveState && type != fiveStateAsymmetric) die("Wrong type for five state %i", (int)type);
    cpecan_model m;
    cpecan_model_default(&m, (int32_t)type);
    return wrap_model(&m);
}
StateMachine *stateMachine3_construct(StateMachineType type) {
    if (type != threeState && type != threeStateAsymmetric)
        die("Tried to create a three state state-machine with the wrong type");
    cpecan_model m;
    cpecan_model_default(&m, (int32_t)type);
    return wrap_model(&m);
}
void stateMachine_destruct(StateMachine *sM) { free(sM); }

/* ---------------- Hmm ---------------- */
Hmm *hmm_constructEmpty(double pseudo, StateMachineType type) {
    cpecan_hmm f;
    if (cpecan_hmm_init(&f, (int32_t)type, pseudo) != CPECAN_OK) die("Unrecognised state type: %i", (int)type);
    Hmm *h = malloc(sizeof *h);
    if (!h) die("out of memory");
    h->type = type;
    h->stateNumber = f.stateNumber;
    h->transitions = malloc(sizeof(double) * (size_t)(h->stateNumber * h->stateNumber));
    h->emissions = malloc(sizeof(double) * (size_t)(h->stateNumber * 16));
    for (int64_t i = 0; i < h->stateNumber * h->stateNumber; i++) h->transitions[i] = pseudo;
    for (int64_t i = 0; i < h->stateNumber * 16; i++) h->emissions[i] = pseudo;
    h->likelihood = 0.0;
    return h;
}
void hmm_destruct(Hmm *h) {
    free(h->transitions);
    free(h->emissions);
    free(h);
}
static void to_flat(const Hmm *h, cpecan_hmm *f) {
    cpecan_hmm_init(f, (int32_t)h->type, 0.0);
    memcpy(f->transitions, h->transitions, sizeof(double) * (size_t)(h->stateNumber * h->stateNumber));
    memcpy(f->emissions, h->emissions, sizeof(double) * (size_t)(h->stateNumber * 16));
    f->likelihood = h->likelihood;
}
static void from_flat(const cpecan_hmm *f, Hmm *h) {
    memcpy(h->transitions, f->transitions, sizeof(double) * (size_t)(h->stateNumber * h->stateNumber));
    memcpy(h->emissions, f->emissions, sizeof(double) * (size_t)(h->stateNumber * 16));
    h->likelihood = f->likelihood;
}
double hmm_getTransition(Hmm *h, int64_t from, int64_t to) { return h->transitions[from * h->stateNumber + to]; }
void hmm_setTransition(Hmm *h, int64_t from, int64_t to, double p) { h->transitions[from * h->stateNumber + to] = p; }
void hmm_addToTransitionExpectation(Hmm *h, int64_t from, int64_t to, double p) { h->transitions[from * h->stateNumber + to] += p; }
double hmm_getEmissionsExpectation(Hmm *h, int64_t s, Symbol x, Symbol y) { return h->emissions[s * 16 + x * 4 + y]; }
void hmm_setEmissionsExpectation(Hmm *h, int64_t s, Symbol x, Symbol y, double p) { h->emissions[s * 16 + x * 4 + y] = p; }
void hmm_addToEmissionsExpectation(Hmm *h, int64_t s, Symbol x, Symbol y, double p) { h->emissions[s * 16 + x * 4 + y] += p; }
void hmm_normalise(Hmm *h) {
    cpecan_hmm f;
    to_flat(h, &f);
    cpecan_hmm_normalise(&f);
    from_flat(&f, h);
}
void hmm_write(Hmm *h, FILE *fh) { /* impl/stateMachine.c:133-143 */
    fprintf(fh, "%i\t", (int)h->type);
    for (int64_t i = 0; i < h->stateNumber * h->stateNumber; i++) fprintf(fh, "%f\t", h->transitions[i]);
    fprintf(fh, "%f\n", h->likelihood);
    for (int64_t i = 0; i < h->stateNumber * 16; i++) fprintf(fh, "%f\t", h->emissions[i]);
    fprintf(fh, "\n");
}
Hmm *hmm_loadFromFile(const char *fileName) {
    cpecan_hmm f;
    if (cpecan_hmm_load(&f, fileName) != CPECAN_OK) die("Failed to parse the input state machine file %s", fileName);
    Hmm *h = hmm_constructEmpty(0.0, (StateMachineType)f.type);
    from_flat(&f, h);
    return h;
}
StateMachine *hmm_getStateMachine(Hmm *h) {
    cpecan_hmm f;
    to_flat(h, &f);
    cpecan_model m;
    if (cpecan_model_from_hmm(&m, &f) != CPECAN_OK) return NULL;
    return wrap_model(&m);
}

/* ---------------- parameters ---------------- */
PairwiseAlignmentParameters *pairwiseAlignmentBandingParameters_construct(void) { /* impl/pairwiseAligner.c:1334-1348 */
    PairwiseAlignmentParameters *p = malloc(sizeof *p);
    if (!p) die("out of memory");
    p->threshold = 0.01;
    p->minDiagsBetweenTraceBack = 1000;
    p->traceBackDiagonals = 40;
    p->diagonalExpansion = 20;
    p->constraintDiagonalTrim = 14;
    p->anchorMatrixBiggerThanThis = 500 * 500;
    p->repeatMaskMatrixBiggerThanThis = 500 * 500;
    p->splitMatrixBiggerThanThis = (int64_t)3000 * 3000;
    p->alignAmbiguityCharacters = 0;
    p->gapGamma = 0.5;
    p->dynamicAnchorExpansion = 0;
    return p;
}
void pairwiseAlignmentBandingParameters_destruct(PairwiseAlignmentParameters *p) { free(p); }

static void flatten_params(const PairwiseAlignmentParameters *p, cpecan_params *q) {
    cpecan_params_default(q);
    q->threshold = p->threshold;
    q->minDiagsBetweenTraceBack = p->minDiagsBetweenTraceBack;
    q->traceBackDiagonals = p->traceBackDiagonals;
    q->diagonalExpansion = p->diagonalExpansion;
    q->splitMatrixBiggerThanThis = p->splitMatrixBiggerThanThis;
    q->dynamicAnchorExpansion = p->dynamicAnchorExpansion ? 1 : 0;
}
static int64_t *flatten_anchors(stList *anchorPairs, int64_t *n) {
    *n = anchorPairs ? stList_length(anchorPairs) : 0;
    int64_t *a = malloc(sizeof(int64_t) * 3 * (size_t)(*n ? *n : 1));
    if (!a) die("out of memory");
    for (int64_t i = 0; i < *n; i++) {
        stIntTuple *tp = stList_get(anchorPairs, i);
        a[3 * i] = stIntTuple_get(tp, 0);
        a[3 * i + 1] = stIntTuple_get(tp, 1);
        a[3 * i + 2] = stIntTuple_length(tp) > 2 ? stIntTuple_get(tp, 2) : 0;
    }
    return a;
}
static stList *list_of(const int32_t *tr, int64_t n) {
    stList *l = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int64_t i = 0; i < n; i++) stList_append(l, stIntTuple_construct3(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2]));
    return l;
}
static void check(int rc, const char *what) {
    if (rc != CPECAN_OK) die("cpecan_hip: %s failed (%d): %s", what, rc, cpecan_last_error());
}

/* pairwiseAlignmentParameters_jsonParse, impl/pairwiseAligner.c:1354-1408.  The reference tokenises with jsmn (through
 * sonLib's stJson) and walks key/value token pairs of one flat object; the same here without the tokeniser. */
static const char *json_skip(const char *s, const char *end) {
    while (s < end && (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r' || *s == ',' || *s == ':')) s++;
    return s;
}
PairwiseAlignmentParameters *pairwiseAlignmentParameters_jsonParse(char *buf, size_t r) {
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    const char *s = buf, *end = buf + r;
    s = json_skip(s, end);
    if (s >= end || *s != '{') die("cpecan_hip: pairwise alignment parameters json: expected an object");
    s++;
    for (;;) {
        s = json_skip(s, end);
        if (s >= end) die("cpecan_hip: pairwise alignment parameters json: unterminated object");
        if (*s == '}') break;
        if (*s != '"') die("cpecan_hip: pairwise alignment parameters json: expected a key");
        const char *key = ++s;
        while (s < end && *s != '"') s++;
        if (s >= end) die("cpecan_hip: pairwise alignment parameters json: unterminated key");
        const size_t keyLen = (size_t)(s - key);
        s = json_skip(s + 1, end);
        char val[64];
        size_t n = 0;
        while (s < end && *s != ',' && *s != '}' && *s != ' ' && *s != '\n' && *s != '\r' && *s != '\t' && n + 1 < sizeof val)
            val[n++] = *s++;
        val[n] = 0;
        if (n == 0) die("cpecan_hip: pairwise alignment parameters json: key without a value");
#define KEY(name) (keyLen == strlen(name) && strncmp(key, name, keyLen) == 0)
        const int truth = strcmp(val, "true") == 0 ? 1 : (strcmp(val, "false") == 0 ? 0 : (int)strtoll(val, NULL, 10) != 0);
        if (KEY("threshold")) p->threshold = strtod(val, NULL);
        else if (KEY("minDiagsBetweenTraceBack")) p->minDiagsBetweenTraceBack = strtoll(val, NULL, 10);
        else if (KEY("traceBackDiagonals")) p->traceBackDiagonals = strtoll(val, NULL, 10);
        else if (KEY("diagonalExpansion")) p->diagonalExpansion = strtoll(val, NULL, 10);
        else if (KEY("constraintDiagonalTrim")) p->constraintDiagonalTrim = strtoll(val, NULL, 10);
        else if (KEY("anchorMatrixBiggerThanThis")) p->anchorMatrixBiggerThanThis = strtoll(val, NULL, 10);
        else if (KEY("repeatMaskMatrixBiggerThanThis")) p->repeatMaskMatrixBiggerThanThis = strtoll(val, NULL, 10);
        else if (KEY("splitMatrixBiggerThanThis")) p->splitMatrixBiggerThanThis = strtoll(val, NULL, 10);
        else if (KEY("alignAmbiguityCharacters")) p->alignAmbiguityCharacters = truth;
        else if (KEY("gapGamma")) p->gapGamma = (float)strtod(val, NULL);
        else if (KEY("dynamicAnchorExpansion")) p->dynamicAnchorExpansion = truth;
        else die("cpecan_hip: ERROR: Unrecognised key in pairwise alignment parameters json: %.*s", (int)keyLen, key);
#undef KEY
    }
    return p;
}

/* convertPairwiseForwardStrandAlignmentToAnchorPairs, impl/pairwiseAligner.c:979-1003 */
stList *convertPairwiseForwardStrandAlignmentToAnchorPairs(struct PairwiseAlignment *pA, int64_t trim, int64_t diagonalExpansion) {
    if (!pA || !pA->strand1 || !pA->strand2) die("cpecan_hip: the alignment must be on the forward strands (:981-982)");
    const int64_t nOps = pA->operationList ? pA->operationList->length : 0;
    int64_t *ops = malloc(sizeof(int64_t) * 2 * (size_t)(nOps ? nOps : 1)), matches = 0;
    if (!ops) die("cpecan_hip: out of memory");
    for (int64_t i = 0; i < nOps; i++) {
        const struct AlignmentOperation *op = pA->operationList->list[i];
        ops[2 * i] = op->opType == PAIRWISE_MATCH ? CPECAN_OP_MATCH : (op->opType == PAIRWISE_INDEL_X ? CPECAN_OP_INDEL_X : CPECAN_OP_INDEL_Y);
        ops[2 * i + 1] = op->length;
        if (op->opType == PAIRWISE_MATCH) matches += op->length;
    }
    int64_t *anchors = malloc(sizeof(int64_t) * 3 * (size_t)(matches ? matches : 1));
    if (!anchors) die("cpecan_hip: out of memory");
    const int64_t n = cpecan_anchors_from_alignment(ops, nOps, pA->start1, pA->start2, trim, diagonalExpansion, NULL, 0, NULL, 0,
                                                    anchors);
    if (n < 0) die("cpecan_hip: convertPairwiseForwardStrandAlignmentToAnchorPairs failed: %s", cpecan_last_error());
    stList *l = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int64_t i = 0; i < n; i++) stList_append(l, stIntTuple_construct3(anchors[3 * i], anchors[3 * i + 1], anchors[3 * i + 2]));
    free(ops);
    free(anchors);
    return l;
}

stList *filterToRemoveOverlap(stList *sortedOverlappingPairs) {
    int64_t n;
    int64_t *in = flatten_anchors(sortedOverlappingPairs, &n);
    int64_t *out = malloc(sizeof(int64_t) * 3 * (size_t)(n ? n : 1));
    if (!out) die("cpecan_hip: out of memory");
    const int64_t kept = cpecan_filter_to_remove_overlap(in, n, out);
    if (kept < 0) die("cpecan_hip: filterToRemoveOverlap failed: %s", cpecan_last_error());
    stList *l = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int64_t i = 0; i < kept; i++) stList_append(l, stIntTuple_construct3(out[3 * i], out[3 * i + 1], out[3 * i + 2]));
    free(in);
    free(out);
    return l;
}

/* ---------------- the path: impl/pairwiseAligner.c:1431-1513, :936 ---------------- */
stList *getAlignedPairsUsingAnchors(StateMachine *sM, const char *sX, const char *sY, stList *anchorPairs,
                                    PairwiseAlignmentParameters *p, bool raggedLeft, bool raggedRight) {
    cpecan_params q;
    flatten_params(p, &q);
    int64_t n, cnt = 0;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    int32_t *tr = NULL;
    check(cpecan_get_aligned_pairs_using_anchors(flat_or_die(sM), sX, sY, anchors, n, &q, raggedLeft, raggedRight, &tr, &cnt),
          "getAlignedPairsUsingAnchors");
    free(anchors);
    stList *l = list_of(tr, cnt);
    cpecan_free(tr);
    return l;
}
void getAlignedPairsWithIndelsUsingAnchors(StateMachine *sM, const char *sX, const char *sY, stList *anchorPairs,
                                           PairwiseAlignmentParameters *p, stList **alignedPairs, stList **gapXPairs,
                                           stList **gapYPairs, bool raggedLeft, bool raggedRight) {
    cpecan_params q;
    flatten_params(p, &q);
    int64_t n, c0 = 0, c1 = 0, c2 = 0;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    int32_t *t0 = NULL, *t1 = NULL, *t2 = NULL;
    check(cpecan_get_aligned_pairs_with_indels_using_anchors(flat_or_die(sM), sX, sY, anchors, n, &q, raggedLeft, raggedRight,
                                                             &t0, &c0, &t1, &c1, &t2, &c2),
          "getAlignedPairsWithIndelsUsingAnchors");
    free(anchors);
    *alignedPairs = list_of(t0, c0);
    *gapXPairs = list_of(t1, c1);
    *gapYPairs = list_of(t2, c2);
    cpecan_free(t0);
    cpecan_free(t1);
    cpecan_free(t2);
}
void getExpectationsUsingAnchors(StateMachine *sM, Hmm *hmmExpectations, const char *sX, const char *sY, stList *anchorPairs,
                                 PairwiseAlignmentParameters *p, bool raggedLeft, bool raggedRight) {
    cpecan_params q;
    flatten_params(p, &q);
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    cpecan_batch *b = NULL;
    check(cpecan_batch_create(&b, flat_or_die(sM), &q, CPECAN_EMIT_EXPECT, cpecan_current_device()), "cpecan_batch_create");
    if (cpecan_batch_add(b, sX, (int64_t)strlen(sX), sY, (int64_t)strlen(sY), anchors, n, raggedLeft, raggedRight) < 0)
        die("cpecan_hip: invalid anchors");
    check(cpecan_batch_upload(b), "cpecan_batch_upload");
    check(cpecan_batch_run(b, NULL), "cpecan_batch_run");
    check(cpecan_batch_download(b), "cpecan_batch_download");
    cpecan_hmm acc;
    to_flat(hmmExpectations, &acc);
    check(cpecan_batch_expectations(b, &acc), "cpecan_batch_expectations");
    from_flat(&acc, hmmExpectations);
    cpecan_batch_destroy(b);
    free(anchors);
}
/* ---------------- consumers of the posterior lists: impl/pairwiseAligner.c:1519-1790 ---------------- */
static int32_t *flatten_triples(stList *l, int64_t *n) { /* (score, x, y) tuples -> int32 triples */
    *n = l ? stList_length(l) : 0;
    int32_t *t = malloc(sizeof(int32_t) * 3 * (size_t)(*n ? *n : 1));
    if (!t) die("out of memory");
    for (int64_t i = 0; i < *n; i++) {
        stIntTuple *tp = stList_get(l, i);
        for (int f = 0; f < 3; f++) t[3 * i + f] = (int32_t)stIntTuple_get(tp, f);
    }
    return t;
}
stList *reweightAlignedPairs2(stList *alignedPairs, int64_t seqLengthX, int64_t seqLengthY, double gapGamma) {
    if (gapGamma <= 0.0) return alignedPairs; /* :1551 */
    int64_t n;
    int32_t *t = flatten_triples(alignedPairs, &n);
    check(cpecan_reweight_aligned_pairs(t, n, seqLengthX, seqLengthY, gapGamma), "reweightAlignedPairs2");
    stList *l = list_of(t, n);
    free(t);
    stList_destruct(alignedPairs); /* the reference consumes its input (:1546) */
    return l;
}
double scoreByPosteriorProbability(int64_t lX, int64_t lY, stList *alignedPairs) {
    int64_t n;
    int32_t *t = flatten_triples(alignedPairs, &n);
    double s = 0.0;
    check(cpecan_posterior_scores(t, n, lX, lY, &s, NULL), "scoreByPosteriorProbability");
    free(t);
    return s;
}
double scoreByPosteriorProbabilityIgnoringGaps(stList *alignedPairs) {
    int64_t n;
    int32_t *t = flatten_triples(alignedPairs, &n);
    double s = 0.0;
    check(cpecan_posterior_scores(t, n, 0, 0, NULL, &s), "scoreByPosteriorProbabilityIgnoringGaps");
    free(t);
    return s;
}
double scoreByIdentity(char *subSeqX, char *subSeqY, int64_t lX, int64_t lY, stList *alignedPairs) {
    (void)lX; /* the reference's callers pass strlen of the two strings (cPecanRealign.c:560) */
    (void)lY;
    int64_t n;
    int32_t *t = flatten_triples(alignedPairs, &n);
    double s = 0.0;
    check(cpecan_identity_scores(t, n, subSeqX, subSeqY, &s, NULL), "scoreByIdentity");
    free(t);
    return s;
}
double scoreByIdentityIgnoringGaps(char *subSeqX, char *subSeqY, stList *alignedPairs) {
    int64_t n;
    int32_t *t = flatten_triples(alignedPairs, &n);
    double s = 0.0;
    check(cpecan_identity_scores(t, n, subSeqX, subSeqY, NULL, &s), "scoreByIdentityIgnoringGaps");
    free(t);
    return s;
}
stList *filterPairwiseAlignmentToMakePairsOrdered(stList *alignedPairs, const char *seqX, const char *seqY, float matchGamma) {
    int64_t n, cnt = 0;
    int32_t *t = flatten_triples(alignedPairs, &n), *out = NULL;
    check(cpecan_filter_pairs_ordered(t, n, (int64_t)strlen(seqX), (int64_t)strlen(seqY), matchGamma, &out, &cnt),
          "filterPairwiseAlignmentToMakePairsOrdered");
    stList *l = list_of(out, cnt);
    cpecan_free(out);
    free(t);
    stList_destruct(alignedPairs); /* "Destroys input list of aligned pairs in process" (multipleAligner.c:943) */
    return l;
}
/* getBlastPairsForPairwiseAlignmentParameters (:1162-1166) up to the size at which the reference turns to lastz */
static stList *no_anchors_or_die(const char *sX, const char *sY, const PairwiseAlignmentParameters *p, const char *what) {
    const int64_t lX = (int64_t)strlen(sX), lY = (int64_t)strlen(sY);
    if (lX * lY > p->anchorMatrixBiggerThanThis)
        die("cpecan_hip: %s on a %lld x %lld matrix needs lastz anchors (anchorMatrixBiggerThanThis = %lld), which this "
            "library does not compute: use the *UsingAnchors entry point",
            what, (long long)lX, (long long)lY, (long long)p->anchorMatrixBiggerThanThis);
    return stList_construct();
}
stList *getAlignedPairs(StateMachine *sM, const char *sX, const char *sY, PairwiseAlignmentParameters *p, bool raggedLeft,
                        bool raggedRight) {
    stList *anchors = no_anchors_or_die(sX, sY, p, "getAlignedPairs");
    stList *l = getAlignedPairsUsingAnchors(sM, sX, sY, anchors, p, raggedLeft, raggedRight);
    stList_destruct(anchors);
    return l;
}
void getAlignedPairsWithIndels(StateMachine *sM, const char *sX, const char *sY, PairwiseAlignmentParameters *p,
                               stList **alignedPairs, stList **gapXPairs, stList **gapYPairs, bool raggedLeft, bool raggedRight) {
    stList *anchors = no_anchors_or_die(sX, sY, p, "getAlignedPairsWithIndels");
    getAlignedPairsWithIndelsUsingAnchors(sM, sX, sY, anchors, p, alignedPairs, gapXPairs, gapYPairs, raggedLeft, raggedRight);
    stList_destruct(anchors);
}
void getExpectations(StateMachine *sM, Hmm *hmmExpectations, const char *sX, const char *sY, PairwiseAlignmentParameters *p,
                     bool raggedLeft, bool raggedRight) {
    stList *anchors = no_anchors_or_die(sX, sY, p, "getExpectations");
    getExpectationsUsingAnchors(sM, hmmExpectations, sX, sY, anchors, p, raggedLeft, raggedRight);
    stList_destruct(anchors);
}
stList *getMaximalExpectedAccuracyPairwiseAlignment(stList *alignedPairs, stList *gapXPairs, stList *gapYPairs,
                                                    int64_t seqXLength, int64_t seqYLength, double *alignmentScore,
                                                    PairwiseAlignmentParameters *p) {
    int64_t n, nx, ny, cnt = 0;
    int32_t *t = flatten_triples(alignedPairs, &n), *gx = flatten_triples(gapXPairs, &nx), *gy = flatten_triples(gapYPairs, &ny);
    int32_t *out = NULL;
    double score = 0.0;
    check(cpecan_mea_alignment(t, n, gx, nx, gy, ny, seqXLength, seqYLength, p->gapGamma, &out, &cnt, &score),
          "getMaximalExpectedAccuracyPairwiseAlignment");
    stList *l = list_of(out, cnt);
    cpecan_free(out);
    free(t);
    free(gx);
    free(gy);
    if (alignmentScore) *alignmentScore = score;
    return l;
}
stList *leftShiftAlignment(stList *alignedPairs, char *seqX, char *seqY) {
    int64_t n, cnt = 0;
    int32_t *t = flatten_triples(alignedPairs, &n), *out = NULL;
    check(cpecan_left_shift_alignment(t, n, seqX, seqY, &out, &cnt), "leftShiftAlignment");
    stList *l = list_of(out, cnt);
    cpecan_free(out);
    free(t);
    return l;
}
stList *getShiftedMEAAlignment(char *seqX, char *seqY, stList *anchorAlignment, PairwiseAlignmentParameters *p,
                               StateMachine *sM, bool raggedLeft, bool raggedRight, double *alignmentScore) {
    cpecan_params q;
    flatten_params(p, &q);
    int64_t n, cnt = 0;
    int64_t *anchors = flatten_anchors(anchorAlignment, &n);
    int32_t *out = NULL;
    double score = 0.0;
    check(cpecan_get_shifted_mea_alignment(flat_or_die(sM), seqX, seqY, anchors, n, &q, p->gapGamma, raggedLeft, raggedRight,
                                           &out, &cnt, &score),
          "getShiftedMEAAlignment");
    free(anchors);
    stList *l = list_of(out, cnt);
    cpecan_free(out);
    if (alignmentScore) *alignmentScore = score;
    return l;
}
double computeForwardProbability(char *seqX, char *seqY, stList *anchorPairs, PairwiseAlignmentParameters *p, StateMachine *sM,
                                 bool raggedLeft, bool raggedRight) {
    cpecan_params q;
    flatten_params(p, &q);
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    double lp = 0.0;
    check(cpecan_compute_forward_probability(flat_or_die(sM), seqX, seqY, anchors, n, &q, raggedLeft, raggedRight, &lp),
          "computeForwardProbability");
    free(anchors);
    return lp;
}

/* ---------------- geometry: impl/pairwiseAligner.c:30-78, 183-277, 1230 ---------------- */
int64_t diagonal_getXay(Diagonal d) { return d.xay; }
int64_t diagonal_getMinXmy(Diagonal d) { return d.xmyL; }
int64_t diagonal_getMaxXmy(Diagonal d) { return d.xmyR; }
int64_t diagonal_getWidth(Diagonal d) { return (d.xmyR - d.xmyL) / 2 + 1; }
int64_t diagonal_getXCoordinate(int64_t xay, int64_t xmy) { return (xay + xmy) / 2; }
int64_t diagonal_getYCoordinate(int64_t xay, int64_t xmy) { return (xay - xmy) / 2; }
int64_t diagonal_equals(Diagonal p, Diagonal q) { return p.xay == q.xay && p.xmyL == q.xmyL && p.xmyR == q.xmyR; }

struct _band {
    Diagonal *diagonals;
    int64_t lXalY;
};
Band *band_construct(stList *anchorPairs, int64_t lX, int64_t lY, int64_t expansion) {
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    int64_t *flat = malloc(sizeof(int64_t) * 3 * (size_t)(lX + lY + 1));
    check(cpecan_band(anchors, n, lX, lY, expansion, 0, flat), "band_construct");
    Band *b = malloc(sizeof *b);
    b->lXalY = lX + lY;
    b->diagonals = malloc(sizeof(Diagonal) * (size_t)(lX + lY + 1));
    for (int64_t d = 0; d <= lX + lY; d++) {
        b->diagonals[d].xay = flat[3 * d];
        b->diagonals[d].xmyL = flat[3 * d + 1];
        b->diagonals[d].xmyR = flat[3 * d + 2];
    }
    free(flat);
    free(anchors);
    return b;
}
void band_destruct(Band *b) {
    free(b->diagonals);
    free(b);
}
struct _bandIterator {
    Band *band;
    int64_t index;
};
BandIterator *bandIterator_construct(Band *band) {
    BandIterator *it = malloc(sizeof *it);
    it->band = band;
    it->index = 0;
    return it;
}
BandIterator *bandIterator_clone(BandIterator *it) {
    BandIterator *c2 = malloc(sizeof *c2);
    *c2 = *it;
    return c2;
}
void bandIterator_destruct(BandIterator *it) { free(it); }
Diagonal bandIterator_getNext(BandIterator *it) { /* saturates at the last diagonal, :263-270 */
    Diagonal d = it->band->diagonals[it->index > it->band->lXalY ? it->band->lXalY : it->index];
    if (it->index <= it->band->lXalY) it->index++;
    return d;
}
Diagonal bandIterator_getPrevious(BandIterator *it) { /* saturates at the first, :272-277 */
    if (it->index > 0) it->index--;
    return it->band->diagonals[it->index];
}
Symbol symbol_convertCharToSymbol(char i) {
    switch (i) {
    case 'A': case 'a': return a;
    case 'C': case 'c': return c;
    case 'G': case 'g': return g;
    case 'T': case 't': return t;
    default: return n;
    }
}
char symbol_convertSymbolToChar(Symbol i) {
    static const char k[] = "ACGTN";
    return (i >= a && i <= t) ? k[i] : 'N';
}
stList *getSplitPoints(stList *anchorPairs, int64_t lX, int64_t lY, int64_t maxMatrixSize, bool raggedLeft, bool raggedRight) {
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    int64_t *out = malloc(sizeof(int64_t) * 4 * (size_t)(n + 2));
    int64_t cnt = cpecan_split_points(anchors, n, lX, lY, maxMatrixSize, raggedLeft, raggedRight, out);
    if (cnt < 0) die("cpecan_hip: invalid anchors for getSplitPoints");
    stList *l = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int64_t i = 0; i < cnt; i++) stList_append(l, tuple4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]));
    free(out);
    free(anchors);
    return l;
}


/* ---------------- the rest of the reference header's hot-path surface ---------------- */

/* inc/pairwiseAligner.h:23, impl/pairwiseAligner.c:29 */
const char *PAIRWISE_ALIGNMENT_EXCEPTION_ID = "PAIRWISE_ALIGNMENT_EXCEPTION";

/* The reference raises sonLib exceptions (stThrowNew, setjmp-based) where a diagonal is malformed; sonLib is not
 * vendored.  The raise goes through this WEAK hook: a program that links sonLib overrides it with one line
 * (`void cpecan_dropin_throw(const char *id, const char *msg) { stThrowNew(id, "%s", msg); }`); the default does what
 * an uncaught stExcept does -- prints and aborts. */
__attribute__((weak)) void cpecan_dropin_throw(const char *exceptionId, const char *message) {
    die("cpecan_hip: uncaught %s: %s", exceptionId, message);
}

/* diagonal_construct, impl/pairwiseAligner.c:30-42 (inc/pairwiseAligner.h:122) */
Diagonal diagonal_construct(int64_t xay, int64_t xmyL, int64_t xmyR) {
    if ((xay + xmyL) % 2 != 0 || (xay + xmyR) % 2 != 0 || xmyL > xmyR) {
        char msg[192];
        snprintf(msg, sizeof msg, "Attempt to create diagonal with invalid coordinates: xay %lld xmyL %lld xmyR %lld",
                 (long long)xay, (long long)xmyL, (long long)xmyR);
        cpecan_dropin_throw(PAIRWISE_ALIGNMENT_EXCEPTION_ID, msg);
    }
    Diagonal d = {xay, xmyL, xmyR};
    return d;
}

/* logAdd, impl/pairwiseAligner.c:290-307 (inc/pairwiseAligner.h:167), operation for operation: this is the host-side
 * function a caller links, not the device code (cpk_device_common.inl). */
static inline double logadd_lookup(double x) {
    if (x <= 1.00f) return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x + 0.693203116424741f;
    if (x <= 2.50f) return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x + 0.692140569840976f;
    if (x <= 4.50f) return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x + 0.514272634594009f;
    return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x + 0.168037164329057f;
}
double logAdd(double x, double y) {
    if (x < y) return (x == LOG_ZERO || y - x >= 7.5) ? y : logadd_lookup(y - x) + x;
    return (y == LOG_ZERO || x - y >= 7.5) ? x : logadd_lookup(x - y) + y;
}

/* band_constructDynamic, impl/pairwiseAligner.c:128-181: every anchor tuple carries its own expansion (third element) */
Band *band_constructDynamic(stList *anchorPairs, int64_t lX, int64_t lY) {
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    int64_t *flat = malloc(sizeof(int64_t) * 3 * (size_t)(lX + lY + 1));
    if (!flat) die("cpecan_hip: out of memory");
    check(cpecan_band(anchors, n, lX, lY, 0, 1, flat), "band_constructDynamic");
    Band *b = malloc(sizeof *b);
    b->lXalY = lX + lY;
    b->diagonals = malloc(sizeof(Diagonal) * (size_t)(lX + lY + 1));
    for (int64_t d = 0; d <= lX + lY; d++) {
        b->diagonals[d].xay = flat[3 * d];
        b->diagonals[d].xmyL = flat[3 * d + 1];
        b->diagonals[d].xmyR = flat[3 * d + 2];
    }
    free(flat);
    free(anchors);
    return b;
}

/* symbols: impl/pairwiseAligner.c:336-366 */
Symbol *symbol_convertStringToSymbols(const char *s, int64_t sL) {
    Symbol *cS = malloc(sizeof(Symbol) * (size_t)(sL > 0 ? sL : 1));
    if (!cS) die("cpecan_hip: out of memory");
    for (int64_t i = 0; i < sL; i++) cS[i] = symbol_convertCharToSymbol(s[i]);
    return cS;
}
SymbolString symbolString_construct(const char *sequence, int64_t length) {
    SymbolString sS;
    sS.sequence = symbol_convertStringToSymbols(sequence, length);
    sS.length = length;
    return sS;
}

/* sonLib's st_random (a uniform double in [0, 1)); weak, so that a linked sonLib's takes precedence */
__attribute__((weak)) double st_random(void) {
    static uint64_t state = 0x9E3779B97F4A7C15ull;
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* hmm_randomise, impl/stateMachine.c:114-131 */
void hmm_randomise(Hmm *hmm) {
    for (int64_t from = 0; from < hmm->stateNumber; from++)
        for (int64_t to = 0; to < hmm->stateNumber; to++) hmm_setTransition(hmm, from, to, st_random());
    for (int64_t state = 0; state < hmm->stateNumber; state++)
        for (int64_t x = 0; x < SYMBOL_NUMBER_NO_N; x++)
            for (int64_t y = 0; y < SYMBOL_NUMBER_NO_N; y++) hmm_setEmissionsExpectation(hmm, state, (Symbol)x, (Symbol)y, st_random());
    hmm_normalise(hmm);
}

/* hmm_jsonParse, impl/stateMachine.c:204-253: {"type": t, "transitions": [S*S], "emissions": [S*16], "likelihood": l};
 * "type" must come first (:214-218), transitions and emissions are mandatory (:240-245), any other key aborts. */
static const char *json_number_array(const char *s, const char *end, double *out, int64_t n, const char *what) {
    s = json_skip(s, end);
    if (s >= end || *s != '[') die("cpecan_hip: hmm json: expected an array for %s", what);
    s++;
    for (int64_t i = 0; i < n; i++) {
        s = json_skip(s, end);
        char *stop = NULL;
        out[i] = strtod(s, &stop);
        if (stop == s || stop > end) die("cpecan_hip: hmm json: %s holds fewer than %lld numbers", what, (long long)n);
        s = stop;
    }
    s = json_skip(s, end);
    if (s >= end || *s != ']') die("cpecan_hip: hmm json: %s holds more than %lld numbers", what, (long long)n);
    return s + 1;
}
Hmm *hmm_jsonParse(char *buf, size_t r) {
    const char *s = buf, *end = buf + r;
    s = json_skip(s, end);
    if (s >= end || *s != '{') die("cpecan_hip: hmm json: expected an object");
    s++;
    Hmm *hmm = NULL;
    int gotEmissions = 0, gotTransitions = 0;
    for (;;) {
        s = json_skip(s, end);
        if (s >= end) die("cpecan_hip: hmm json: unterminated object");
        if (*s == '}') break;
        if (*s != '"') die("cpecan_hip: hmm json: expected a key");
        const char *key = ++s;
        while (s < end && *s != '"') s++;
        if (s >= end) die("cpecan_hip: hmm json: unterminated key");
        const size_t keyLen = (size_t)(s - key);
        s = json_skip(s + 1, end);
#define KEY(name) (keyLen == strlen(name) && strncmp(key, name, keyLen) == 0)
        if (!hmm) {
            if (!KEY("type")) die("cpecan_hip: ERROR: Unrecognised key in polish params json: %.*s", (int)keyLen, key);
            char *stop = NULL;
            const long long type = strtoll(s, &stop, 10);
            if (stop == s) die("cpecan_hip: hmm json: type without a value");
            s = stop;
            hmm = hmm_constructEmpty(0, (StateMachineType)type);
        } else if (KEY("transitions")) {
            s = json_number_array(s, end, hmm->transitions, hmm->stateNumber * hmm->stateNumber, "transitions");
            gotTransitions = 1;
        } else if (KEY("emissions")) {
            s = json_number_array(s, end, hmm->emissions, hmm->stateNumber * SYMBOL_NUMBER_NO_N * SYMBOL_NUMBER_NO_N, "emissions");
            gotEmissions = 1;
        } else if (KEY("likelihood")) {
            char *stop = NULL;
            hmm->likelihood = strtod(s, &stop);
            if (stop == s) die("cpecan_hip: hmm json: likelihood without a value");
            s = stop;
        } else {
            die("cpecan_hip: ERROR: Unrecognised key in hmm json: %.*s", (int)keyLen, key);
        }
#undef KEY
    }
    if (!hmm) die("cpecan_hip: ERROR: too few tokens to parse in hmm json");
    if (!gotEmissions) die("cpecan_hip: ERROR: Did not find emissions specified in json HMM");
    if (!gotTransitions) die("cpecan_hip: ERROR: Did not find transitions specified in json HMM");
    return hmm;
}

/* ---------------- DpDiagonal / DpMatrix / cell_* / diagonalCalculation* (inc/pairwiseAligner.h:186-237) ----------------
 * The containers the reference's unit tests link (tests/pairwiseAlignerTest.c:155-324), with the reference's semantics
 * (impl/pairwiseAligner.c:448-586).  They are host memory; every piece of DP arithmetic on them -- a cell's transition
 * list, a diagonal's sweep, the posterior's exp -- is evaluated on the GPU through cpecan_ref_cells (one lane, the
 * reference's order of operations): the library holds no CPU implementation of the recurrences.  The fold of a dot
 * product is logAdd, which the reference itself exports as a host utility (:287-307) and so does this layer. */
struct _dpDiagonal {
    Diagonal diagonal;
    int64_t stateNumber;
    double *cells;
};
struct _dpMatrix {
    DpDiagonal **diagonals;
    int64_t diagonalNumber;
    int64_t activeDiagonals;
    int64_t stateNumber;
};

char *diagonal_getString(Diagonal diagonal) { /* impl/pairwiseAligner.c:75-78; the caller frees */
    char *out = malloc(128);
    if (!out) die("cpecan_hip: out of memory");
    snprintf(out, 128, "Diagonal, xay: %lld xmyL %lld, xmyR: %lld", (long long)diagonal.xay, (long long)diagonal.xmyL,
             (long long)diagonal.xmyR);
    return out;
}

DpDiagonal *dpDiagonal_construct(Diagonal diagonal, int64_t stateNumber) { /* :454-461 */
    DpDiagonal *d = malloc(sizeof *d);
    const int64_t w = diagonal_getWidth(diagonal);
    if (!d || w < 0) die("cpecan_hip: dpDiagonal_construct: out of memory or negative width");
    d->diagonal = diagonal;
    d->stateNumber = stateNumber;
    d->cells = malloc(sizeof(double) * (size_t)(stateNumber * w > 0 ? stateNumber * w : 1));
    if (!d->cells) die("cpecan_hip: out of memory");
    return d;
}
DpDiagonal *dpDiagonal_clone(DpDiagonal *diagonal) {
    DpDiagonal *c2 = dpDiagonal_construct(diagonal->diagonal, diagonal->stateNumber);
    memcpy(c2->cells, diagonal->cells, sizeof(double) * (size_t)(diagonal_getWidth(diagonal->diagonal) * diagonal->stateNumber));
    return c2;
}
bool dpDiagonal_equals(DpDiagonal *d1, DpDiagonal *d2) { /* :469-482 */
    if (!diagonal_equals(d1->diagonal, d2->diagonal) || d1->stateNumber != d2->stateNumber) return 0;
    for (int64_t i = 0; i < diagonal_getWidth(d1->diagonal) * d1->stateNumber; i++)
        if (d1->cells[i] != d2->cells[i]) return 0;
    return 1;
}
void dpDiagonal_destruct(DpDiagonal *dpDiagonal) {
    free(dpDiagonal->cells);
    free(dpDiagonal);
}
double *dpDiagonal_getCell(DpDiagonal *dpDiagonal, int64_t xmy) { /* NULL outside the band: how band edges reach the cell functions, :489-495 */
    if (xmy < dpDiagonal->diagonal.xmyL || xmy > dpDiagonal->diagonal.xmyR) return NULL;
    return &dpDiagonal->cells[((xmy - dpDiagonal->diagonal.xmyL) / 2) * dpDiagonal->stateNumber];
}
void dpDiagonal_zeroValues(DpDiagonal *diagonal) {
    for (int64_t i = 0; i < diagonal_getWidth(diagonal->diagonal) * diagonal->stateNumber; i++) diagonal->cells[i] = LOG_ZERO;
}
void dpDiagonal_initialiseValues(DpDiagonal *diagonal, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t)) {
    for (int64_t i = diagonal->diagonal.xmyL; i <= diagonal->diagonal.xmyR; i += 2) {
        double *cell = dpDiagonal_getCell(diagonal, i);
        for (int64_t j = 0; j < diagonal->stateNumber; j++) cell[j] = getStateValue(sM, j);
    }
}
double cell_dotProduct(double *cell1, double *cell2, int64_t stateNumber) { /* :402-408 */
    double totalProb = cell1[0] + cell2[0];
    for (int64_t i = 1; i < stateNumber; i++) totalProb = logAdd(totalProb, cell1[i] + cell2[i]);
    return totalProb;
}
double cell_dotProduct2(double *cell, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t)) { /* :410-416 */
    double totalProb = cell[0] + getStateValue(sM, 0);
    for (int64_t i = 1; i < sM->stateNumber; i++) totalProb = logAdd(totalProb, cell[i] + getStateValue(sM, i));
    return totalProb;
}
double dpDiagonal_dotProduct(DpDiagonal *diagonal1, DpDiagonal *diagonal2) { /* :513-523 */
    double totalProbability = LOG_ZERO;
    for (int64_t xmy = diagonal1->diagonal.xmyL; xmy <= diagonal1->diagonal.xmyR; xmy += 2)
        totalProbability = logAdd(totalProbability, cell_dotProduct(dpDiagonal_getCell(diagonal1, xmy), dpDiagonal_getCell(diagonal2, xmy),
                                                                    diagonal1->stateNumber));
    return totalProbability;
}

DpMatrix *dpMatrix_construct(int64_t diagonalNumber, int64_t stateNumber) { /* :540-548 */
    DpMatrix *m = malloc(sizeof *m);
    if (!m || diagonalNumber < 0) die("cpecan_hip: dpMatrix_construct: out of memory or negative size");
    m->diagonalNumber = diagonalNumber;
    m->diagonals = calloc((size_t)diagonalNumber + 1, sizeof(DpDiagonal *));
    if (!m->diagonals) die("cpecan_hip: out of memory");
    m->activeDiagonals = 0;
    m->stateNumber = stateNumber;
    return m;
}
void dpMatrix_destruct(DpMatrix *dpMatrix) {
    free(dpMatrix->diagonals);
    free(dpMatrix);
}
DpDiagonal *dpMatrix_getDiagonal(DpMatrix *dpMatrix, int64_t xay) {
    if (xay < 0 || xay > dpMatrix->diagonalNumber) return NULL;
    return dpMatrix->diagonals[xay];
}
int64_t dpMatrix_getActiveDiagonalNumber(DpMatrix *dpMatrix) { return dpMatrix->activeDiagonals; }
DpDiagonal *dpMatrix_createDiagonal(DpMatrix *dpMatrix, Diagonal diagonal) { /* :567-576 */
    if (diagonal.xay < 0 || diagonal.xay > dpMatrix->diagonalNumber || dpMatrix->diagonals[diagonal.xay] != NULL)
        die("cpecan_hip: dpMatrix_createDiagonal: diagonal %lld out of range or already there", (long long)diagonal.xay);
    DpDiagonal *d = dpDiagonal_construct(diagonal, dpMatrix->stateNumber);
    dpMatrix->diagonals[diagonal.xay] = d;
    dpMatrix->activeDiagonals++;
    return d;
}
void dpMatrix_deleteDiagonal(DpMatrix *dpMatrix, int64_t xay) { /* :578-587 */
    if (xay < 0 || xay > dpMatrix->diagonalNumber) die("cpecan_hip: dpMatrix_deleteDiagonal: diagonal %lld out of range", (long long)xay);
    if (dpMatrix->diagonals[xay] != NULL) {
        dpMatrix->activeDiagonals--;
        dpDiagonal_destruct(dpMatrix->diagonals[xay]);
        dpMatrix->diagonals[xay] = NULL;
    }
}

/* a flat buffer of cells for cpecan_ref_cells: every distinct cell pointer of a call gets `S` doubles of it */
typedef struct {
    double *buf;
    double **where; /* the host cell each slot mirrors */
    int64_t nCells, cap;
    int64_t S;
} CellPack;
static int32_t pack_cell(CellPack *pk, double *cell) {
    if (!cell) return -1;
    for (int64_t i = pk->nCells - 1; i >= 0 && i >= pk->nCells - 8; i--) /* a neighbour shared with the cell before */
        if (pk->where[i] == cell) return (int32_t)(i * pk->S);
    if (pk->nCells == pk->cap) {
        pk->cap = pk->cap ? 2 * pk->cap : 64;
        pk->buf = realloc(pk->buf, sizeof(double) * (size_t)(pk->cap * pk->S));
        pk->where = realloc(pk->where, sizeof(double *) * (size_t)pk->cap);
        if (!pk->buf || !pk->where) die("cpecan_hip: out of memory");
    }
    pk->where[pk->nCells] = cell;
    memcpy(pk->buf + pk->nCells * pk->S, cell, sizeof(double) * (size_t)pk->S);
    return (int32_t)(pk->nCells++ * pk->S);
}
static void unpack_cells(CellPack *pk) {
    for (int64_t i = 0; i < pk->nCells; i++) memcpy(pk->where[i], pk->buf + i * pk->S, sizeof(double) * (size_t)pk->S);
    free(pk->buf);
    free(pk->where);
}
static void run_cells(StateMachine *sM, int mode, cpecan_cell_op *ops, int64_t n, CellPack *pk, const char *what) {
    check(cpecan_ref_cells(flat_or_die(sM), mode, ops, n, pk->buf, pk->nCells * pk->S, 0.0), what);
    unpack_cells(pk);
}
static void one_cell(StateMachine *sM, int mode, double *current, double *lower, double *middle, double *upper, Symbol cX, Symbol cY,
                     const char *what) {
    CellPack pk = {NULL, NULL, 0, 0, sM->stateNumber};
    cpecan_cell_op op;
    op.cur = pack_cell(&pk, current);
    op.lower = pack_cell(&pk, lower);
    op.middle = pack_cell(&pk, middle);
    op.upper = pack_cell(&pk, upper);
    op.cX = (int32_t)cX;
    op.cY = (int32_t)cY;
    run_cells(sM, mode, &op, 1, &pk, what);
}
void cell_calculateForward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, Symbol cX, Symbol cY,
                           void *extraArgs) { /* :387-390 */
    (void)extraArgs;
    one_cell(sM, CPECAN_CELLS_FORWARD, current, lower, middle, upper, cX, cY, "cell_calculateForward");
}
void cell_calculateBackward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, Symbol cX, Symbol cY,
                            void *extraArgs) { /* :397-400 */
    (void)extraArgs;
    one_cell(sM, CPECAN_CELLS_BACKWARD, current, lower, middle, upper, cX, cY, "cell_calculateBackward");
}
/* diagonalCalculation, :609-624: the cells of dpDiagonal in ascending x-y against dpDiagonalM1 / dpDiagonalM2 */
static void diagonal_calculation(StateMachine *sM, int mode, DpDiagonal *dpDiagonal, DpDiagonal *m1, DpDiagonal *m2,
                                 const SymbolString sX, const SymbolString sY, const char *what) {
    const Diagonal dg = dpDiagonal->diagonal;
    const int64_t w = diagonal_getWidth(dg);
    if (w <= 0) return;
    CellPack pk = {NULL, NULL, 0, 0, sM->stateNumber};
    cpecan_cell_op *ops = malloc(sizeof *ops * (size_t)w);
    if (!ops) die("cpecan_hip: out of memory");
    int64_t k = 0;
    for (int64_t xmy = dg.xmyL; xmy <= dg.xmyR; xmy += 2, k++) {
        const int64_t x = diagonal_getXCoordinate(dg.xay, xmy), y = diagonal_getYCoordinate(dg.xay, xmy);
        ops[k].cX = (int32_t)(x > 0 ? sX.sequence[x - 1] : n); /* getXCharacter, :597-607 */
        ops[k].cY = (int32_t)(y > 0 ? sY.sequence[y - 1] : n);
        ops[k].cur = pack_cell(&pk, dpDiagonal_getCell(dpDiagonal, xmy));
        ops[k].lower = pack_cell(&pk, m1 ? dpDiagonal_getCell(m1, xmy - 1) : NULL);
        ops[k].middle = pack_cell(&pk, m2 ? dpDiagonal_getCell(m2, xmy) : NULL);
        ops[k].upper = pack_cell(&pk, m1 ? dpDiagonal_getCell(m1, xmy + 1) : NULL);
    }
    run_cells(sM, mode, ops, k, &pk, what);
    free(ops);
}
void diagonalCalculationForward(StateMachine *sM, int64_t xay, DpMatrix *dpMatrix, const SymbolString sX, const SymbolString sY) {
    diagonal_calculation(sM, CPECAN_CELLS_FORWARD, dpMatrix_getDiagonal(dpMatrix, xay), dpMatrix_getDiagonal(dpMatrix, xay - 1),
                         dpMatrix_getDiagonal(dpMatrix, xay - 2), sX, sY, "diagonalCalculationForward");
}
void diagonalCalculationBackward(StateMachine *sM, int64_t xay, DpMatrix *dpMatrix, const SymbolString sX, const SymbolString sY) {
    diagonal_calculation(sM, CPECAN_CELLS_BACKWARD, dpMatrix_getDiagonal(dpMatrix, xay), dpMatrix_getDiagonal(dpMatrix, xay - 1),
                         dpMatrix_getDiagonal(dpMatrix, xay - 2), sX, sY, "diagonalCalculationBackward");
}
double diagonalCalculationTotalProbability(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                           const SymbolString sX, const SymbolString sY) { /* :636-653 */
    DpDiagonal *forwardDiagonal = dpMatrix_getDiagonal(forwardDpMatrix, xay);
    DpDiagonal *backDiagonal = dpMatrix_getDiagonal(backwardDpMatrix, xay);
    double totalProbability = dpDiagonal_dotProduct(forwardDiagonal, backDiagonal);
    forwardDiagonal = dpMatrix_getDiagonal(forwardDpMatrix, xay - 1);
    backDiagonal = dpMatrix_getDiagonal(backwardDpMatrix, xay + 1);
    if (backDiagonal != NULL && forwardDiagonal != NULL) { /* matches straddling the diagonal */
        DpDiagonal *matchDiagonal = dpDiagonal_clone(backDiagonal);
        dpDiagonal_zeroValues(matchDiagonal);
        diagonal_calculation(sM, CPECAN_CELLS_FORWARD, matchDiagonal, NULL, forwardDiagonal, sX, sY, "diagonalCalculationTotalProbability");
        totalProbability = logAdd(totalProbability, dpDiagonal_dotProduct(matchDiagonal, backDiagonal));
        dpDiagonal_destruct(matchDiagonal);
    }
    return totalProbability;
}

/* The reference's per-diagonal emitters (impl/pairwiseAligner.c:666-689, :691-733, :735-746).  Passed to
 * getPosteriorProbsWithBanding they are TOKENS: recognised by address and routed to the device emitters of the sweep
 * kernels (the diagonals of that engine live in LDS, no DpMatrix exists).  Called directly with DpMatrix rows -- as the
 * reference's test_diagonalDPCalculations does -- the match emitter works on those rows: exp() of every cell on the GPU
 * (cpecan_ref_cells, mode posterior), threshold / clamp / floor as addPosteriorProb (:655-664). */
static void posterior_list(StateMachine *sM, DpDiagonal *fD, DpDiagonal *bD, int64_t state, double totalProbability,
                           PairwiseAlignmentParameters *p, stList *out, int needX, int needY) {
    const Diagonal dg = fD->diagonal;
    const int64_t w = diagonal_getWidth(dg);
    if (w <= 0) return;
    double *buf = malloc(sizeof(double) * 3 * (size_t)w);
    cpecan_cell_op *ops = malloc(sizeof *ops * (size_t)w);
    if (!buf || !ops) die("cpecan_hip: out of memory");
    int64_t k = 0;
    for (int64_t xmy = dg.xmyL; xmy <= dg.xmyR; xmy += 2, k++) {
        buf[3 * k] = dpDiagonal_getCell(fD, xmy)[state];
        buf[3 * k + 1] = dpDiagonal_getCell(bD, xmy)[state];
        buf[3 * k + 2] = 0.0;
        ops[k].cur = (int32_t)(3 * k);
        ops[k].lower = (int32_t)(3 * k + 1);
        ops[k].middle = -1;
        ops[k].upper = (int32_t)(3 * k + 2);
        ops[k].cX = ops[k].cY = 0;
    }
    check(cpecan_ref_cells(flat_or_die(sM), CPECAN_CELLS_POSTERIOR, ops, k, buf, 3 * k, totalProbability), "posterior probabilities");
    k = 0;
    for (int64_t xmy = dg.xmyL; xmy <= dg.xmyR; xmy += 2, k++) {
        const int64_t x = diagonal_getXCoordinate(dg.xay, xmy), y = diagonal_getYCoordinate(dg.xay, xmy);
        if ((needX && x <= 0) || (needY && y <= 0)) continue;
        double pp = buf[3 * k + 2];
        if (pp >= p->threshold) { /* addPosteriorProb */
            if (pp > 1.0) pp = 1.0;
            stList_append(out, stIntTuple_construct3((int64_t)floor(pp * PAIR_ALIGNMENT_PROB_1), x - 1, y - 1));
        }
    }
    free(buf);
    free(ops);
}
void diagonalCalculationPosteriorMatchProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                            const SymbolString sX, const SymbolString sY, double totalProbability,
                                            PairwiseAlignmentParameters *p, void *extraArgs) {
    (void)sX; (void)sY;
    if (!forwardDpMatrix || !backwardDpMatrix) die("cpecan_hip: diagonalCalculationPosteriorMatchProbs: no DpMatrix given");
    posterior_list(sM, dpMatrix_getDiagonal(forwardDpMatrix, xay), dpMatrix_getDiagonal(backwardDpMatrix, xay), sM->matchState,
                   totalProbability, p, (stList *)((void **)extraArgs)[0], 1, 1);
}
void diagonalCalculationPosteriorProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                       const SymbolString sX, const SymbolString sY, double totalProbability,
                                       PairwiseAlignmentParameters *p, void *extraArgs) { /* :691-733 */
    (void)sX; (void)sY;
    if (!forwardDpMatrix || !backwardDpMatrix) die("cpecan_hip: diagonalCalculationPosteriorProbs: no DpMatrix given");
    DpDiagonal *fD = dpMatrix_getDiagonal(forwardDpMatrix, xay), *bD = dpMatrix_getDiagonal(backwardDpMatrix, xay);
    posterior_list(sM, fD, bD, sM->matchState, totalProbability, p, (stList *)((void **)extraArgs)[0], 1, 1);
    posterior_list(sM, fD, bD, sM->gapXState, totalProbability, p, (stList *)((void **)extraArgs)[2], 1, 0);
    posterior_list(sM, fD, bD, sM->gapYState, totalProbability, p, (stList *)((void **)extraArgs)[4], 0, 1);
}
void diagonalCalculationExpectations(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                     const SymbolString sX, const SymbolString sY, double totalProbability,
                                     PairwiseAlignmentParameters *p, void *extraArgs) {
    (void)sM; (void)xay; (void)forwardDpMatrix; (void)backwardDpMatrix; (void)sX; (void)sY; (void)totalProbability; (void)p; (void)extraArgs;
    die("cpecan_hip: diagonalCalculationExpectations is an emitter token for getPosteriorProbsWithBanding / getExpectationsUsingAnchors; "
        "the expectation sums are accumulated by the device emitter only");
}

static char *chars_of(const SymbolString s) {
    char *out = malloc((size_t)s.length + 1);
    if (!out) die("cpecan_hip: out of memory");
    for (int64_t i = 0; i < s.length; i++) out[i] = symbol_convertSymbolToChar(s.sequence[i]);
    out[s.length] = 0;
    return out;
}
/* appends the triples in the order in which the reference's emitter appends them: traceback after traceback as the
 * forward sweep advances, within a traceback the diagonals DEScending, within a diagonal x-y AScending
 * (impl/pairwiseAligner.c:813-841, :673-688) -- the exact reverse of the list order of getAlignedPairsUsingAnchors,
 * whose wrapper pops each traceback's pairs from the end (:1411-1418) */
static void append_reversed(stList *to, const int32_t *tr, int64_t n) {
    for (int64_t i = n - 1; i >= 0; i--) stList_append(to, stIntTuple_construct3(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2]));
}

/* getPosteriorProbsWithBanding, impl/pairwiseAligner.c:756-877 (inc/pairwiseAligner.h:245-248): one region, no splitting. */
void getPosteriorProbsWithBanding(StateMachine *sM, stList *anchorPairs, const SymbolString sX, const SymbolString sY,
                                  PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd,
                                  void (*diagonalPosteriorProbFn)(StateMachine *, int64_t, DpMatrix *, DpMatrix *, const SymbolString,
                                                                  const SymbolString, double, PairwiseAlignmentParameters *, void *),
                                  void *extraArgs) {
    cpecan_params q;
    flatten_params(p, &q);
    q.splitMatrixBiggerThanThis = INT64_MAX / 4; /* this function is what the splitting wrapper calls per region */
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    char *cX = chars_of(sX), *cY = chars_of(sY);
    if (diagonalPosteriorProbFn == diagonalCalculationPosteriorMatchProbs) {
        int32_t *tr = NULL;
        int64_t cnt = 0;
        check(cpecan_get_aligned_pairs_using_anchors(flat_or_die(sM), cX, cY, anchors, n, &q, alignmentHasRaggedLeftEnd,
                                                     alignmentHasRaggedRightEnd, &tr, &cnt), "getPosteriorProbsWithBanding");
        append_reversed((stList *)((void **)extraArgs)[0], tr, cnt);
        cpecan_free(tr);
    } else if (diagonalPosteriorProbFn == diagonalCalculationPosteriorProbs) {
        int32_t *t0 = NULL, *t1 = NULL, *t2 = NULL;
        int64_t c0 = 0, c1 = 0, c2 = 0;
        check(cpecan_get_aligned_pairs_with_indels_using_anchors(flat_or_die(sM), cX, cY, anchors, n, &q, alignmentHasRaggedLeftEnd,
                                                                 alignmentHasRaggedRightEnd, &t0, &c0, &t1, &c1, &t2, &c2),
              "getPosteriorProbsWithBanding");
        /* the indel emitter's lists sit at extraArgs[0], [2] and [4] (:697-699) */
        append_reversed((stList *)((void **)extraArgs)[0], t0, c0);
        append_reversed((stList *)((void **)extraArgs)[2], t1, c1);
        append_reversed((stList *)((void **)extraArgs)[4], t2, c2);
        cpecan_free(t0);
        cpecan_free(t1);
        cpecan_free(t2);
    } else if (diagonalPosteriorProbFn == diagonalCalculationExpectations) {
        stList *one = NULL; /* the anchors are already flat: go through the batch directly */
        (void)one;
        cpecan_batch *b = NULL;
        check(cpecan_batch_create(&b, flat_or_die(sM), &q, CPECAN_EMIT_EXPECT, cpecan_current_device()), "cpecan_batch_create");
        if (cpecan_batch_add(b, cX, sX.length, cY, sY.length, anchors, n, alignmentHasRaggedLeftEnd, alignmentHasRaggedRightEnd) < 0)
            die("cpecan_hip: invalid anchors");
        check(cpecan_batch_upload(b), "cpecan_batch_upload");
        check(cpecan_batch_run(b, NULL), "cpecan_batch_run");
        check(cpecan_batch_download(b), "cpecan_batch_download");
        cpecan_hmm acc;
        to_flat((Hmm *)extraArgs, &acc);
        check(cpecan_batch_expectations(b, &acc), "cpecan_batch_expectations");
        from_flat(&acc, (Hmm *)extraArgs);
        cpecan_batch_destroy(b);
    } else {
        /* a foreign per-diagonal callback would have to be fed DpMatrix rows diagonal by diagonal from the host: that is
         * the CPU algorithm, which this library does not contain */
        die("cpecan_hip: getPosteriorProbsWithBanding: the emitter must be one of diagonalCalculationPosteriorMatchProbs, "
            "diagonalCalculationPosteriorProbs, diagonalCalculationExpectations (routed to the device emitters)");
    }
    free(cX);
    free(cY);
    free(anchors);
}


/* getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps, impl/pairwiseAligner.c:1273-1326 (inc/pairwiseAligner.h:264):
 * the rectangles of getSplitPoints, each aligned as a region of its own (ragged on the sides where it was cut) with its
 * anchors rebased, and coordinateCorrectionFn(x1, y1, extraArgs) called after each -- in the reference that callback moves
 * the region's pairs from the scratch lists of extraArgs into the caller's lists (:1411-1432).  Here every rectangle is
 * one problem of ONE batch (one upload, one launch per size class, one download); the emitter's lists are then filled
 * and the callback made rectangle by rectangle, in the reference's order. */
void getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(
    StateMachine *sM, stList *anchorPairs, const char *sX, const char *sY, int64_t lX, int64_t lY, PairwiseAlignmentParameters *p,
    bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd,
    void (*diagonalPosteriorProbFn)(StateMachine *, int64_t, DpMatrix *, DpMatrix *, const SymbolString, const SymbolString, double,
                                    PairwiseAlignmentParameters *, void *),
    void (*coordinateCorrectionFn)(), void *extraArgs) {
    int emit;
    if (diagonalPosteriorProbFn == diagonalCalculationPosteriorMatchProbs) emit = CPECAN_EMIT_MATCH;
    else if (diagonalPosteriorProbFn == diagonalCalculationPosteriorProbs) emit = CPECAN_EMIT_INDEL;
    else if (diagonalPosteriorProbFn == diagonalCalculationExpectations) emit = CPECAN_EMIT_EXPECT;
    else {
        die("cpecan_hip: getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps: the emitter must be one of "
            "diagonalCalculationPosteriorMatchProbs, diagonalCalculationPosteriorProbs, diagonalCalculationExpectations");
        return;
    }
    cpecan_params q;
    flatten_params(p, &q);
    const int64_t maxMatrixSize = q.splitMatrixBiggerThanThis;
    q.splitMatrixBiggerThanThis = INT64_MAX / 4; /* the rectangles are cut here, as the reference cuts them */
    int64_t n;
    int64_t *anchors = flatten_anchors(anchorPairs, &n);
    int64_t *rect = malloc(sizeof(int64_t) * 4 * (size_t)(n + 2));
    if (!rect) die("cpecan_hip: out of memory");
    const int64_t nRect = cpecan_split_points(anchors, n, lX, lY, maxMatrixSize, alignmentHasRaggedLeftEnd, alignmentHasRaggedRightEnd, rect);
    if (nRect < 0) die("cpecan_hip: getSplitPoints: invalid anchors");
    cpecan_batch *b = NULL;
    check(cpecan_batch_create(&b, flat_or_die(sM), &q, emit, cpecan_current_device()), "cpecan_batch_create");
    int64_t j = 0;
    int64_t *sub = malloc(sizeof(int64_t) * 3 * (size_t)(n ? n : 1));
    if (!sub) die("cpecan_hip: out of memory");
    for (int64_t i = 0; i < nRect; i++) {
        const int64_t x1 = rect[4 * i], y1 = rect[4 * i + 1], x2 = rect[4 * i + 2], y2 = rect[4 * i + 3];
        int64_t m = 0;
        while (j < n && anchors[3 * j] + anchors[3 * j + 1] < x2 + y2) { /* :1299-1311 */
            sub[3 * m] = anchors[3 * j] - x1;
            sub[3 * m + 1] = anchors[3 * j + 1] - y1;
            sub[3 * m + 2] = anchors[3 * j + 2];
            m++;
            j++;
        }
        if (cpecan_batch_add(b, sX + x1, x2 - x1, sY + y1, y2 - y1, sub, m, alignmentHasRaggedLeftEnd || i > 0,
                             alignmentHasRaggedRightEnd || i < nRect - 1) < 0)
            die("cpecan_hip: %s", cpecan_last_error());
    }
    if (nRect > 0) {
        check(cpecan_batch_upload(b), "cpecan_batch_upload");
        check(cpecan_batch_run(b, NULL), "cpecan_batch_run");
        check(cpecan_batch_download(b), "cpecan_batch_download");
    }
    if (emit == CPECAN_EMIT_EXPECT && nRect > 0) {
        cpecan_hmm acc;
        to_flat((Hmm *)extraArgs, &acc);
        check(cpecan_batch_expectations(b, &acc), "cpecan_batch_expectations");
        from_flat(&acc, (Hmm *)extraArgs);
    }
    for (int64_t i = 0; i < nRect; i++) {
        const int nLists = emit == CPECAN_EMIT_MATCH ? 1 : (emit == CPECAN_EMIT_INDEL ? 3 : 0);
        for (int l = 0; l < nLists; l++) {
            const int32_t *tr = NULL;
            int64_t cnt = 0;
            check(cpecan_batch_result(b, i, l, &tr, &cnt), "cpecan_batch_result");
            append_reversed((stList *)((void **)extraArgs)[2 * l], tr, cnt); /* the emitter's own order, :813-841 */
        }
        if (coordinateCorrectionFn != NULL) ((void (*)(int64_t, int64_t, void *))coordinateCorrectionFn)(rect[4 * i], rect[4 * i + 1], extraArgs);
    }
    cpecan_batch_destroy(b);
    free(sub);
    free(rect);
    free(anchors);
}

/* getIndelProbabilities / reweightAlignedPairs, impl/pairwiseAligner.c:1519-1548 (inc/pairwiseAligner.h:272-276): the two
 * halves of reweightAlignedPairs2 for callers that hold the per-base arrays themselves -- integer list bookkeeping on the
 * caller's lists; the fused form (reweightAlignedPairs2 above, cpecan_batch_set_post) runs on the device. */
int64_t *getIndelProbabilities(stList *alignedPairs, int64_t seqLength, bool xIfTrueElseY) {
    int64_t *indelProbs = malloc(sizeof(int64_t) * (size_t)(seqLength > 0 ? seqLength : 1));
    if (!indelProbs) die("cpecan_hip: out of memory");
    for (int64_t i = 0; i < seqLength; i++) indelProbs[i] = PAIR_ALIGNMENT_PROB_1;
    for (int64_t i = 0; i < stList_length(alignedPairs); i++) {
        stIntTuple *tp = stList_get(alignedPairs, i);
        indelProbs[stIntTuple_get(tp, xIfTrueElseY ? 1 : 2)] -= stIntTuple_get(tp, 0);
    }
    for (int64_t i = 0; i < seqLength; i++)
        if (indelProbs[i] < 0) indelProbs[i] = 0;
    return indelProbs;
}
stList *reweightAlignedPairs(stList *alignedPairs, int64_t *indelProbsX, int64_t *indelProbsY, double gapGamma) {
    stList *out = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int64_t i = 0; i < stList_length(alignedPairs); i++) {
        stIntTuple *aPair = stList_get(alignedPairs, i);
        const int64_t x = stIntTuple_get(aPair, 1), y = stIntTuple_get(aPair, 2);
        /* int64 - double * int64, truncated on the assignment: the reference's arithmetic (:1543) */
        const int64_t updatedWeight = (int64_t)(stIntTuple_get(aPair, 0) - gapGamma * (indelProbsX[x] + indelProbsY[y]));
        stList_append(out, stIntTuple_construct3(updatedWeight, x, y));
    }
    stList_destruct(alignedPairs); /* "destroys input aligned pairs in the process" */
    return out;
}
int64_t getNumberOfMatchingAlignedPairs(char *subSeqX, char *subSeqY, stList *alignedPairs) { /* :1562-1570 */
    int64_t matches = 0;
    for (int64_t i = 0; i < stList_length(alignedPairs); i++) {
        stIntTuple *aPair = stList_get(alignedPairs, i);
        const int64_t x = stIntTuple_get(aPair, 1), y = stIntTuple_get(aPair, 2);
        const int cx = toupper((unsigned char)subSeqX[x]), cy = toupper((unsigned char)subSeqY[y]);
        matches += cx == cy && cx != 'N';
    }
    return matches;
}
