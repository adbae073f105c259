/* C-level test of the reference-named drop-in layer (include/cpecan_dropin.h).  Mirrors the fixed-input parts of the
 * reference's tests/pairwiseAlignerTest.c: test_bands (:69), test_getSplitPoints (:578), test_hmm (:997),
 * test_symbol (:146), test_diagonalDPCalculations pair set (:242) and the SURVEY 8c known answers.
 * usage: test_dropin cpu | gpu */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cpecan_dropin.h"

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                    \
        }                                                                  \
    } while (0)

static int diag_is(Diagonal d, int64_t xay, int64_t l, int64_t r) { return d.xay == xay && d.xmyL == l && d.xmyR == r; }

static void test_bands(void) {
    stList *anchors = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList_append(anchors, stIntTuple_construct2(1, 0));
    stList_append(anchors, stIntTuple_construct2(2, 1));
    stList_append(anchors, stIntTuple_construct2(3, 3));
    Band *band = band_construct(anchors, 6, 5, 2);
    BandIterator *it = bandIterator_construct(band);
    const int64_t gold[12][3] = {{0, 0, 0}, {1, -1, 1}, {2, -2, 2}, {3, -1, 3}, {4, -2, 4}, {5, -1, 3},
                                 {6, -2, 4}, {7, -3, 3}, {8, -2, 2}, {9, -1, 3}, {10, 0, 2}, {11, 1, 1}};
    for (int i = 0; i < 12; i++) CHECK(diag_is(bandIterator_getNext(it), gold[i][0], gold[i][1], gold[i][2]));
    CHECK(diag_is(bandIterator_getNext(it), 11, 1, 1)); /* saturates */
    for (int i = 11; i >= 0; i--) CHECK(diag_is(bandIterator_getPrevious(it), gold[i][0], gold[i][1], gold[i][2]));
    CHECK(diag_is(bandIterator_getPrevious(it), 0, 0, 0));
    bandIterator_destruct(it);
    band_destruct(band);
    stList_destruct(anchors);
}

static int rect_is(stList *l, int64_t i, int64_t x1, int64_t y1, int64_t x2, int64_t y2) {
    stIntTuple *t4 = stList_get(l, i);
    return stIntTuple_get(t4, 0) == x1 && stIntTuple_get(t4, 1) == y1 && stIntTuple_get(t4, 2) == x2 && stIntTuple_get(t4, 3) == y2;
}

static void test_split_points(void) {
    const int64_t size = 2000 * 2000;
    stList *anchors = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList *s = getSplitPoints(anchors, 3000, 1000, size, 0, 0);
    CHECK(stList_length(s) == 1 && rect_is(s, 0, 0, 0, 3000, 1000));
    stList_destruct(s);
    s = getSplitPoints(anchors, 20000, 25000, size, 1, 1);
    CHECK(stList_length(s) == 0);
    stList_destruct(s);
    s = getSplitPoints(anchors, 20000, 25000, size, 0, 0);
    CHECK(stList_length(s) == 2 && rect_is(s, 0, 0, 0, 2000, 2000) && rect_is(s, 1, 18000, 23000, 20000, 25000));
    stList_destruct(s);
    const int64_t pts[8][2] = {{2000, 2000}, {4002, 4001}, {5000, 5000}, {8000, 6000}, {9000, 9000}, {10000, 14000}, {15000, 15000}, {16000, 16000}};
    for (int i = 0; i < 8; i++) stList_append(anchors, stIntTuple_construct2(pts[i][0], pts[i][1]));
    s = getSplitPoints(anchors, 20000, 25000, size, 0, 0);
    CHECK(stList_length(s) == 5);
    CHECK(rect_is(s, 0, 0, 0, 3001, 3001) && rect_is(s, 1, 3002, 3001, 9500, 11001) && rect_is(s, 2, 9501, 12000, 12001, 14500));
    CHECK(rect_is(s, 3, 13000, 14501, 18000, 18001) && rect_is(s, 4, 18001, 23000, 20000, 25000));
    stList_destruct(s);
    stList_destruct(anchors);
}

static void test_hmm(StateMachineType type) {
    Hmm *h = hmm_constructEmpty(0.0, type);
    const int64_t S = h->stateNumber;
    for (int64_t f = 0; f < S; f++)
        for (int64_t to = 0; to < S; to++) hmm_addToTransitionExpectation(h, f, to, (double)(f * S + to));
    for (int64_t s = 0; s < S; s++)
        for (int x = 0; x < 4; x++)
            for (int y = 0; y < 4; y++) hmm_addToEmissionsExpectation(h, s, (Symbol)x, (Symbol)y, (double)(s * 16 + x * 4 + y));
    const char *path = "./cpecan_dropin_test.hmm";
    FILE *fh = fopen(path, "w");
    hmm_write(h, fh);
    fclose(fh);
    hmm_destruct(h);
    h = hmm_loadFromFile(path);
    remove(path);
    CHECK(h->type == type && h->stateNumber == S);
    for (int64_t f = 0; f < S; f++)
        for (int64_t to = 0; to < S; to++) CHECK(hmm_getTransition(h, f, to) == (double)(f * S + to));
    hmm_normalise(h);
    for (int64_t f = 0; f < S; f++) {
        const double z = (double)(f * S * S + (S * (S - 1)) / 2);
        for (int64_t to = 0; to < S; to++) CHECK(hmm_getTransition(h, f, to) == (double)(f * S + to) / z);
    }
    hmm_destruct(h);
}

static void test_symbols(void) {
    const char *s = "AcGTntNCG";
    const Symbol want[9] = {a, c, g, t, n, t, n, c, g};
    for (int i = 0; i < 9; i++) CHECK(symbol_convertCharToSymbol(s[i]) == want[i]);
    CHECK(symbol_convertSymbolToChar(g) == 'G' && symbol_convertSymbolToChar(n) == 'N');
}

static void test_models(void) {
    StateMachine *sM = stateMachine5_construct(fiveState);
    CHECK(sM->stateNumber == 5 && sM->matchState == 0 && sM->gapXState == 1 && sM->gapYState == 2);
    CHECK(sM->startStateProb(sM, 0) == 0.0 && isinf(sM->startStateProb(sM, 1)));
    CHECK(sM->endStateProb(sM, 0) == -0.030064059121770816 && sM->endStateProb(sM, 3) == -5.673280173170473);
    CHECK(sM->raggedStartStateProb(sM, 3) == 0.0 && isinf(sM->raggedStartStateProb(sM, 0)));
    CHECK(sM->raggedEndStateProb(sM, 4) == -0.003442492794189331);
    stateMachine_destruct(sM);
    sM = stateMachine3_construct(threeState);
    CHECK(sM->stateNumber == 3 && sM->raggedEndStateProb(sM, 0) == (-4.21256642 + -4.21256642) / 2.0);
    stateMachine_destruct(sM);
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    CHECK(p->threshold == 0.01 && p->minDiagsBetweenTraceBack == 1000 && p->traceBackDiagonals == 40 &&
          p->diagonalExpansion == 20 && p->splitMatrixBiggerThanThis == 9000000 && p->gapGamma == 0.5f);
    pairwiseAlignmentBandingParameters_destruct(p);
}

/* pairwiseAlignmentParameters_jsonParse (impl/pairwiseAligner.c:1354-1408) and
 * convertPairwiseForwardStrandAlignmentToAnchorPairs (:979-1003) */
static void test_params_json_and_anchor_conversion(void) {
    char js[] = "{ \"threshold\": 0.2, \"diagonalExpansion\":6,\n \"gapGamma\" : 0.25, \"dynamicAnchorExpansion\": true,"
                " \"splitMatrixBiggerThanThis\": 100 , \"alignAmbiguityCharacters\": false}";
    PairwiseAlignmentParameters *p = pairwiseAlignmentParameters_jsonParse(js, strlen(js));
    CHECK(p->threshold == 0.2 && p->diagonalExpansion == 6 && p->gapGamma == 0.25f && p->dynamicAnchorExpansion);
    CHECK(p->splitMatrixBiggerThanThis == 100 && !p->alignAmbiguityCharacters);
    CHECK(p->minDiagsBetweenTraceBack == 1000 && p->traceBackDiagonals == 40 && p->constraintDiagonalTrim == 14); /* defaults kept */
    pairwiseAlignmentBandingParameters_destruct(p);
    char empty[] = "{}";
    p = pairwiseAlignmentParameters_jsonParse(empty, strlen(empty));
    CHECK(p->threshold == 0.01 && p->diagonalExpansion == 20);
    pairwiseAlignmentBandingParameters_destruct(p);

    struct AlignmentOperation ops[4] = {{PAIRWISE_MATCH, 5, 0}, {PAIRWISE_INDEL_X, 2, 0}, {PAIRWISE_MATCH, 3, 0}, {PAIRWISE_INDEL_Y, 4, 0}};
    void *opPtrs[4] = {&ops[0], &ops[1], &ops[2], &ops[3]};
    struct List opList = {4, 4, opPtrs, NULL};
    struct PairwiseAlignment pA = {"x", 10, 20, 1, "y", 100, 112, 1, 0.0, &opList};
    stList *anchors = convertPairwiseForwardStrandAlignmentToAnchorPairs(&pA, 1, 8);
    /* the first run keeps columns 1..3, the second (behind a 2-base gap in y... in x) its middle column */
    const int64_t wantX[] = {11, 12, 13, 18}, wantY[] = {101, 102, 103, 106};
    CHECK(stList_length(anchors) == 4);
    for (int i = 0; i < 4 && i < stList_length(anchors); i++) {
        stIntTuple *t = stList_get(anchors, i);
        CHECK(stIntTuple_get(t, 0) == wantX[i] && stIntTuple_get(t, 1) == wantY[i] && stIntTuple_get(t, 2) == 8);
    }
    stList_destruct(anchors);
}

static void test_known_answers_gpu(void) {
    StateMachine *sM5 = stateMachine5_construct(fiveState), *sM3 = stateMachine3_construct(threeState);
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    stList *anchors = stList_construct();
    p->threshold = 0.2;
    stList *pairs = getAlignedPairsUsingAnchors(sM5, "AGCG", "AGTTCG", anchors, p, 0, 0);
    const int64_t gold[4][3] = {{9944673, 0, 0}, {9259684, 1, 1}, {8665179, 2, 4}, {9893294, 3, 5}};
    CHECK(stList_length(pairs) == 4);
    for (int i = 0; i < 4 && i < stList_length(pairs); i++) {
        stIntTuple *tp = stList_get(pairs, i);
        CHECK(stIntTuple_get(tp, 0) == gold[i][0] && stIntTuple_get(tp, 1) == gold[i][1] && stIntTuple_get(tp, 2) == gold[i][2]);
    }
    stList_destruct(pairs);
    p->threshold = 0.01;
    char sx[] = "AGCG", sy[] = "AGTTCG";
    CHECK(fabs(computeForwardProbability(sx, sy, anchors, p, sM5, 0, 0) + 17.519321161239) < 1e-11);
    CHECK(fabs(computeForwardProbability(sx, sy, anchors, p, sM3, 0, 0) + 17.381039440328) < 1e-11);
    CHECK(fabs(computeForwardProbability(sx, sy, anchors, p, sM3, 1, 1) + 20.651524037167) < 1e-11);
    Hmm *h = hmm_constructEmpty(0.0, fiveState);
    getExpectationsUsingAnchors(sM5, h, "AGCG", "AGTTCG", anchors, p, 0, 0);
    /* (the events are exp2f of an fp32 argument: ~1e-7 relative each, against north_star's 1e-5) */
    CHECK(fabs(h->likelihood + 175.193211612) < 1e-8 && fabs(hmm_getTransition(h, 0, 0) - 3.010391804) < 1e-7);
    CHECK(fabs(hmm_getTransition(h, 0, 1) - 0.000751913) < 1e-8 && fabs(hmm_getEmissionsExpectation(h, 0, a, a) - 0.994467322) < 1e-7);
    hmm_destruct(h);
    stList *m, *gx, *gy;
    getAlignedPairsWithIndelsUsingAnchors(sM3, "ACGTACGTTTACG", "ACGTCGTTTAACG", anchors, p, &m, &gx, &gy, 0, 0);
    CHECK(stList_length(m) > 8 && stList_length(gx) > 0 && stList_length(gy) > 0);
    stList_destruct(m);
    stList_destruct(gx);
    stList_destruct(gy);
    stList_destruct(anchors);
    pairwiseAlignmentBandingParameters_destruct(p);
    stateMachine_destruct(sM5);
    /* the anchorless entry point is the same call with no anchors up to anchorMatrixBiggerThanThis (:1164) */
    PairwiseAlignmentParameters *q = pairwiseAlignmentBandingParameters_construct();
    q->threshold = 0.2;
    stList *viaShort = getAlignedPairs(sM5, "AGCG", "AGTTCG", q, 0, 0);
    CHECK(stList_length(viaShort) == 4);
    if (stList_length(viaShort) == 4) CHECK(stIntTuple_get(stList_get(viaShort, 2), 0) == 8665179);
    stList_destruct(viaShort);
    pairwiseAlignmentBandingParameters_destruct(q);
    stateMachine_destruct(sM3);
}

/* tests/pairwiseAlignerTest.c:944-995 (test_leftShiftAlignment), through the reference-named GPU entry point; then the
 * other list consumers on a tiny hand-checked case (impl/pairwiseAligner.c:1519-1597, :1628, :1767). */
static void test_consumers_gpu(void) {
    char *seqX = "GATTTACATC", *seqY = "GATTACAATCTG";
    const int64_t ax[] = {0, 1, 2, 3, 5, 6, 7, 8, 9}, ay[] = {0, 1, 2, 3, 4, 5, 6, 8, 11};
    const int64_t sx[] = {0, 1, 3, 4, 5, 6, 7, 8, 9}, sy[] = {0, 1, 2, 3, 4, 5, 6, 10, 11};
    stList *pairs = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int i = 0; i < 9; i++) stList_append(pairs, stIntTuple_construct3(1, ax[i], ay[i]));
    stList *shifted = leftShiftAlignment(pairs, seqX, seqY);
    CHECK(stList_length(shifted) == 9);
    for (int i = 0; i < 9 && i < stList_length(shifted); i++) {
        stIntTuple *t = stList_get(shifted, i);
        CHECK(stIntTuple_get(t, 1) == sx[i] && stIntTuple_get(t, 2) == sy[i]);
    }
    stList_destruct(shifted);
    stList_destruct(pairs);

    /* reweighting worked by hand: unaligned mass X[0] = 1e6, Y[0] = 4e6, Y[1] = 7e6, gapGamma 0.5 */
    stList *l = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList_append(l, stIntTuple_construct3(6000000, 0, 0));
    stList_append(l, stIntTuple_construct3(3000000, 0, 1));
    CHECK(fabs(scoreByPosteriorProbability(1, 2, l) - 100.0 * 2 * 9000000 / (3.0 * PAIR_ALIGNMENT_PROB_1)) < 1e-12);
    CHECK(fabs(scoreByPosteriorProbabilityIgnoringGaps(l) - 100.0 * 9000000 / (2.0 * PAIR_ALIGNMENT_PROB_1)) < 1e-12);
    l = reweightAlignedPairs2(l, 1, 2, 0.5);
    CHECK(stList_length(l) == 2);
    CHECK(stIntTuple_get(stList_get(l, 0), 0) == 3500000 && stIntTuple_get(stList_get(l, 1), 0) == -1000000);
    stList_destruct(l);

    /* identity scores (:1562-1580) and the ordered filter (impl/multipleAligner.c:945) on a hand-checked case: the
     * diagonal chain outweighs the heavy off-diagonal pair; the survivors come back in reverse input order */
    stList *o = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList_append(o, stIntTuple_construct3(6000000, 0, 0));
    stList_append(o, stIntTuple_construct3(9000000, 0, 2));
    stList_append(o, stIntTuple_construct3(5000000, 1, 1));
    stList_append(o, stIntTuple_construct3(4000000, 2, 2));
    CHECK(fabs(scoreByIdentity("ACn", "AGN", 3, 3, o) - 100.0 * 2 * 1 / 6.0) < 1e-12);
    CHECK(fabs(scoreByIdentityIgnoringGaps("ACn", "AGN", o) - 100.0 * 1 / 4.0) < 1e-12);
    o = filterPairwiseAlignmentToMakePairsOrdered(o, "ACG", "ACG", 0.1f);
    CHECK(stList_length(o) == 3);
    if (stList_length(o) == 3) {
        CHECK(stIntTuple_get(stList_get(o, 0), 1) == 2 && stIntTuple_get(stList_get(o, 0), 0) == 4000000);
        CHECK(stIntTuple_get(stList_get(o, 1), 1) == 1 && stIntTuple_get(stList_get(o, 2), 1) == 0);
    }
    stList_destruct(o);

    /* MEA of two crossing pairs without gap mass keeps the heavier one; getShiftedMEAAlignment returns a chain */
    stList *a = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList_append(a, stIntTuple_construct3(4000000, 1, 0));
    stList_append(a, stIntTuple_construct3(7000000, 0, 1));
    stList *none = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    double score = -1;
    stList *mea = getMaximalExpectedAccuracyPairwiseAlignment(a, none, none, 2, 2, &score, p);
    CHECK(stList_length(mea) == 1 && stIntTuple_get(stList_get(mea, 0), 0) == 7000000 && score == 7000000.0);
    stList_destruct(mea);
    stList_destruct(a);
    StateMachine *sM = stateMachine5_construct(fiveState);
    stList *chain = getShiftedMEAAlignment("AGCGTTACGT", "AGCGTACGT", none, p, sM, 0, 0, &score);
    CHECK(stList_length(chain) >= 8);
    for (int64_t i = 1; i < stList_length(chain); i++) {
        stIntTuple *u = stList_get(chain, i - 1), *v = stList_get(chain, i);
        CHECK(stIntTuple_get(u, 1) < stIntTuple_get(v, 1) && stIntTuple_get(u, 2) < stIntTuple_get(v, 2));
    }
    stList_destruct(chain);
    stList_destruct(none);
    stateMachine_destruct(sM);
    pairwiseAlignmentBandingParameters_destruct(p);
}

/* ---- the rest of the reference header's hot-path surface (inc/pairwiseAligner.h:23,122,128,167,245-248;
 * inc/stateMachine.h:71,91) ---- */
static int thrown = 0;
static char thrownId[64];
void cpecan_dropin_throw(const char *exceptionId, const char *message) { /* overrides the library's weak default */
    (void)message;
    thrown++;
    strncpy(thrownId, exceptionId, sizeof thrownId - 1);
}

static void test_diagonal_logadd_dynamic_band_hmm_json(void) {
    /* test_diagonal, tests/pairwiseAlignerTest.c:17-59 */
    Diagonal d = diagonal_construct(30, -10, 30);
    CHECK(diagonal_getXay(d) == 30 && diagonal_getMinXmy(d) == -10 && diagonal_getMaxXmy(d) == 30 && diagonal_getWidth(d) == 21);
    CHECK(diagonal_getXCoordinate(30, -10) == 10 && diagonal_getYCoordinate(30, -10) == 20);
    CHECK(diagonal_equals(d, diagonal_construct(30, -10, 30)) && !diagonal_equals(d, diagonal_construct(0, 0, 0)));
    CHECK(thrown == 0);
    (void)diagonal_construct(10, 5, 5); /* parity violation: the reference throws PAIRWISE_ALIGNMENT_EXCEPTION (:31-35) */
    CHECK(thrown == 1 && strcmp(thrownId, PAIRWISE_ALIGNMENT_EXCEPTION_ID) == 0 && strcmp(thrownId, "PAIRWISE_ALIGNMENT_EXCEPTION") == 0);
    (void)diagonal_construct(10, 6, 4); /* xmyL > xmyR */
    CHECK(thrown == 2);

    /* test_logAdd, :134-144: within 0.001 of the exact value in linear space; and the documented special cases */
    unsigned long long st = 12345;
    for (int i = 0; i < 100000; i++) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const double x = (double)(st >> 11) / 9007199254740992.0;
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const double y = (double)(st >> 11) / 9007199254740992.0;
        const double z = x + y;
        CHECK(fabs(exp(logAdd(log(x), log(y))) - z) < 0.001);
    }
    CHECK(logAdd(LOG_ZERO, -3.5) == -3.5 && logAdd(-3.5, LOG_ZERO) == -3.5 && logAdd(LOG_ZERO, LOG_ZERO) == LOG_ZERO);
    CHECK(logAdd(-10.0, -2.5) == -2.5 && logAdd(-2.5, -10.0) == -2.5);            /* difference 7.5: the larger one */
    CHECK(logAdd(-1.0, -1.0) == (double)0.693203116424741f + -1.0);              /* x == y: second branch, lookup(0) + y */
    CHECK(logAdd(-1.0, -2.0) == logAdd(-2.0, -1.0));

    /* band_constructDynamic (:128-181): anchors carry their own expansion; with one common expansion it is band_construct */
    stList *fixed = stList_construct3(0, (void (*)(void *))stIntTuple_destruct), *dyn = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    const int64_t pts[3][2] = {{1, 0}, {2, 1}, {3, 3}};
    for (int i = 0; i < 3; i++) {
        stList_append(fixed, stIntTuple_construct2(pts[i][0], pts[i][1]));
        stList_append(dyn, stIntTuple_construct3(pts[i][0], pts[i][1], 2));
    }
    Band *bf = band_construct(fixed, 6, 5, 2), *bd = band_constructDynamic(dyn, 6, 5);
    BandIterator *itf = bandIterator_construct(bf), *itd = bandIterator_construct(bd);
    for (int i = 0; i < 12; i++) CHECK(diagonal_equals(bandIterator_getNext(itf), bandIterator_getNext(itd)));
    bandIterator_destruct(itf);
    bandIterator_destruct(itd);
    band_destruct(bf);
    band_destruct(bd);
    stList_destruct(fixed);
    stList_destruct(dyn);

    /* hmm_randomise (stateMachine.c:114-131): a normalised HMM of values in (0, 1) */
    Hmm *h = hmm_constructEmpty(0.0, fiveState);
    hmm_randomise(h);
    for (int64_t from = 0; from < 5; from++) {
        double row = 0.0;
        for (int64_t to = 0; to < 5; to++) {
            CHECK(hmm_getTransition(h, from, to) > 0.0 && hmm_getTransition(h, from, to) < 1.0);
            row += hmm_getTransition(h, from, to);
        }
        CHECK(fabs(row - 1.0) < 1e-9);
        double e = 0.0;
        for (int x = 0; x < 4; x++)
            for (int y = 0; y < 4; y++) e += hmm_getEmissionsExpectation(h, from, (Symbol)x, (Symbol)y);
        CHECK(fabs(e - 1.0) < 1e-9);
    }
    hmm_destruct(h);

    /* hmm_jsonParse (stateMachine.c:204-253) */
    char js[4096];
    int at = snprintf(js, sizeof js, "{\"type\": 2, \"transitions\": [0.9, 0.05, 0.05, 0.4, 0.6, 0, 0.4, 0, 0.6], \"emissions\": [");
    for (int i = 0; i < 48; i++) at += snprintf(js + at, sizeof js - (size_t)at, "%s%g", i ? ", " : "", 0.0625 + 0.001 * i);
    at += snprintf(js + at, sizeof js - (size_t)at, "], \"likelihood\": -12.5}");
    Hmm *hj = hmm_jsonParse(js, (size_t)at);
    CHECK(hj->type == threeState && hj->stateNumber == 3 && hj->likelihood == -12.5);
    CHECK(hmm_getTransition(hj, 0, 0) == 0.9 && hmm_getTransition(hj, 1, 1) == 0.6 && hmm_getTransition(hj, 2, 0) == 0.4);
    CHECK(hmm_getEmissionsExpectation(hj, 0, a, a) == 0.0625 && hmm_getEmissionsExpectation(hj, 2, t, t) == 0.0625 + 0.001 * 47);
    hmm_destruct(hj);

    SymbolString ss = symbolString_construct("AcGTntNCG", 9);
    const Symbol gold[9] = {a, c, g, t, n, t, n, c, g};
    for (int i = 0; i < 9; i++) CHECK(ss.sequence[i] == gold[i]);
    CHECK(ss.length == 9);
    free(ss.sequence);
}

/* getPosteriorProbsWithBanding with the reference's own emitters, as tests/pairwiseAlignerTest.c:403-438 calls it */
static void test_getPosteriorProbsWithBanding_gpu(void) {
    const char *sx = "AGCG", *sy = "AGTTCG";
    SymbolString sX = symbolString_construct(sx, 4), sY = symbolString_construct(sy, 6);
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->threshold = 0.2;
    StateMachine *sM = stateMachine5_construct(fiveState);
    stList *anchors = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList *alignedPairs = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    void *extraArgs[1] = {alignedPairs};
    getPosteriorProbsWithBanding(sM, anchors, sX, sY, p, 0, 0, diagonalCalculationPosteriorMatchProbs, extraArgs);
    /* the four pairs of test_diagonalDPCalculations (:311-322) with the SURVEY 8c scores, in the EMITTER's order: one
     * traceback from the last diagonal, diagonals descending */
    const int64_t gold[4][3] = {{9893294, 3, 5}, {8665179, 2, 4}, {9259684, 1, 1}, {9944673, 0, 0}};
    CHECK(stList_length(alignedPairs) == 4);
    for (int i = 0; i < 4 && i < stList_length(alignedPairs); i++) {
        stIntTuple *tp = stList_get(alignedPairs, i);
        CHECK(llabs(stIntTuple_get(tp, 0) - gold[i][0]) <= 1 && stIntTuple_get(tp, 1) == gold[i][1] && stIntTuple_get(tp, 2) == gold[i][2]);
    }
    /* ... which is getAlignedPairsUsingAnchors' list reversed (its wrapper pops the pairs of each traceback, :1411-1418) */
    stList *viaApi = getAlignedPairsUsingAnchors(sM, sx, sy, anchors, p, 0, 0);
    CHECK(stList_length(viaApi) == stList_length(alignedPairs));
    for (int64_t i = 0; i < stList_length(viaApi) && i < stList_length(alignedPairs); i++) {
        stIntTuple *u = stList_get(viaApi, i), *v = stList_get(alignedPairs, stList_length(alignedPairs) - 1 - i);
        CHECK(stIntTuple_get(u, 0) == stIntTuple_get(v, 0) && stIntTuple_get(u, 1) == stIntTuple_get(v, 1) && stIntTuple_get(u, 2) == stIntTuple_get(v, 2));
    }
    stList_destruct(viaApi);
    stList_destruct(alignedPairs);

    /* the indel emitter: lists at extraArgs[0], [2], [4] (:697-699) */
    stList *m0 = stList_construct3(0, (void (*)(void *))stIntTuple_destruct), *gx = stList_construct3(0, (void (*)(void *))stIntTuple_destruct),
           *gy = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    void *extra5[5] = {m0, NULL, gx, NULL, gy};
    getPosteriorProbsWithBanding(sM, anchors, sX, sY, p, 0, 0, diagonalCalculationPosteriorProbs, extra5);
    CHECK(stList_length(m0) == 4 && stList_length(gy) >= 1);
    stList_destruct(m0);
    stList_destruct(gx);
    stList_destruct(gy);

    /* the expectation emitter: extraArgs is the Hmm (SURVEY 8c known answers) */
    Hmm *h = hmm_constructEmpty(0.0, fiveState);
    getPosteriorProbsWithBanding(sM, anchors, sX, sY, p, 0, 0, diagonalCalculationExpectations, h);
    /* (an event is exp2f of an fp32 argument, a window's sums fp32: ~1e-7 relative against north_star's 1e-5 -- the
     * tolerance of tests/test_gpu_parity.py::test_expectations_known_answers; the likelihood is fp64 throughout) */
    CHECK(fabs(h->likelihood + 175.193211612) < 1e-8 && fabs(hmm_getTransition(h, 0, 0) - 3.010391804) < 1e-7);
    hmm_destruct(h);

    /* dynamic anchor expansion through the engine: every anchor with its own band */
    stList *dyn = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList_append(dyn, stIntTuple_construct3(1, 1, 2));
    p->dynamicAnchorExpansion = 1;
    stList *out = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    void *extra1[1] = {out};
    getPosteriorProbsWithBanding(sM, dyn, sX, sY, p, 0, 0, diagonalCalculationPosteriorMatchProbs, extra1);
    CHECK(stList_length(out) >= 3);
    stList_destruct(out);
    stList_destruct(dyn);
    stList_destruct(anchors);
    stateMachine_destruct(sM);
    pairwiseAlignmentBandingParameters_destruct(p);
    free(sX.sequence);
    free(sY.sequence);
}


/* test_dpDiagonal / test_dpMatrix, tests/pairwiseAlignerTest.c:184-240: the containers (host memory, no GPU) */
static void test_dp_containers(void) {
    StateMachine *sM = stateMachine5_construct(fiveState);
    Diagonal diagonal = diagonal_construct(3, -1, 1);
    DpDiagonal *dpDiagonal = dpDiagonal_construct(diagonal, sM->stateNumber);
    double *c1 = dpDiagonal_getCell(dpDiagonal, -1), *c2 = dpDiagonal_getCell(dpDiagonal, 1);
    CHECK(c1 != NULL && c2 != NULL);
    CHECK(dpDiagonal_getCell(dpDiagonal, 3) == NULL && dpDiagonal_getCell(dpDiagonal, -3) == NULL);
    dpDiagonal_initialiseValues(dpDiagonal, sM, sM->endStateProb);
    double totalProb = LOG_ZERO;
    for (int64_t i = 0; i < sM->stateNumber; i++) {
        CHECK(c1[i] == sM->endStateProb(sM, i) && c2[i] == sM->endStateProb(sM, i));
        totalProb = logAdd(totalProb, 2 * c1[i]);
        totalProb = logAdd(totalProb, 2 * c2[i]);
    }
    DpDiagonal *dpDiagonal2 = dpDiagonal_clone(dpDiagonal);
    CHECK(dpDiagonal_equals(dpDiagonal, dpDiagonal2));
    CHECK(fabs(totalProb - dpDiagonal_dotProduct(dpDiagonal, dpDiagonal2)) < 0.001);
    dpDiagonal_zeroValues(dpDiagonal2);
    CHECK(!dpDiagonal_equals(dpDiagonal, dpDiagonal2) && isinf(dpDiagonal_getCell(dpDiagonal2, 1)[0]));
    dpDiagonal_destruct(dpDiagonal);
    dpDiagonal_destruct(dpDiagonal2);

    const int64_t lX = 3, lY = 2;
    DpMatrix *dpMatrix = dpMatrix_construct(lX + lY, 5);
    CHECK(dpMatrix_getActiveDiagonalNumber(dpMatrix) == 0);
    for (int64_t i = -1; i <= lX + lY + 10; i++) CHECK(dpMatrix_getDiagonal(dpMatrix, i) == NULL);
    for (int64_t i = 0; i <= lX + lY; i++) {
        DpDiagonal *d = dpMatrix_createDiagonal(dpMatrix, diagonal_construct(i, -i, i));
        CHECK(d == dpMatrix_getDiagonal(dpMatrix, i));
        CHECK(dpMatrix_getActiveDiagonalNumber(dpMatrix) == i + 1);
    }
    for (int64_t i = lX + lY; i >= 0; i--) {
        dpMatrix_deleteDiagonal(dpMatrix, i);
        CHECK(dpMatrix_getDiagonal(dpMatrix, i) == NULL);
        CHECK(dpMatrix_getActiveDiagonalNumber(dpMatrix) == i);
    }
    dpMatrix_destruct(dpMatrix);
    char *str = diagonal_getString(diagonal_construct(30, -10, 30));
    CHECK(strcmp(str, "Diagonal, xay: 30 xmyL -10, xmyR: 30") == 0);
    free(str);
    stateMachine_destruct(sM);
}

/* the list helpers of inc/pairwiseAligner.h:272-287: the two halves of reweightAlignedPairs2 and the identity count */
static void test_list_helpers(void) {
    stList *pairs = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    stList_append(pairs, stIntTuple_construct3(9000000, 0, 0));
    stList_append(pairs, stIntTuple_construct3(4000000, 1, 1));
    stList_append(pairs, stIntTuple_construct3(7000000, 1, 2));
    stList_append(pairs, stIntTuple_construct3(2000000, 3, 2));
    int64_t *ix = getIndelProbabilities(pairs, 4, 1), *iy = getIndelProbabilities(pairs, 3, 0);
    CHECK(ix[0] == 1000000 && ix[1] == 0 /* clamped: 1e7 - 1.1e7 */ && ix[2] == 10000000 && ix[3] == 8000000);
    CHECK(iy[0] == 1000000 && iy[1] == 6000000 && iy[2] == 1000000);
    char sx[] = "ACnT", sy[] = "aCN";
    CHECK(getNumberOfMatchingAlignedPairs(sx, sy, pairs) == 2); /* (0,0) A/a and (1,1) C/C; (1,2) C/N no; (3,2) T/N no */
    stList *rw = reweightAlignedPairs(pairs, ix, iy, 0.5); /* consumes pairs */
    CHECK(stList_length(rw) == 4);
    const int64_t want[4] = {9000000 - 1000000, 4000000 - 3000000, 7000000 - 500000, 2000000 - 4500000};
    for (int i = 0; i < 4; i++) CHECK(stIntTuple_get(stList_get(rw, i), 0) == want[i]);
    stList_destruct(rw);
    free(ix);
    free(iy);
}

/* test_cell, tests/pairwiseAlignerTest.c:155-182: forward and backward through one cell and its three neighbours give the
 * same total probability (cell arithmetic on the GPU through cpecan_ref_cells) */
static void test_cell_gpu(StateMachine *sM) {
    double lowerF[5], middleF[5], upperF[5], currentF[5], lowerB[5], middleB[5], upperB[5], currentB[5];
    for (int64_t i = 0; i < sM->stateNumber; i++) {
        middleF[i] = sM->startStateProb(sM, i);
        middleB[i] = lowerF[i] = lowerB[i] = upperF[i] = upperB[i] = currentF[i] = LOG_ZERO;
        currentB[i] = sM->endStateProb(sM, i);
    }
    const Symbol cX = a, cY = t;
    cell_calculateForward(sM, lowerF, NULL, NULL, middleF, cX, cY, NULL);
    cell_calculateForward(sM, upperF, middleF, NULL, NULL, cX, cY, NULL);
    cell_calculateForward(sM, currentF, lowerF, middleF, upperF, cX, cY, NULL);
    cell_calculateBackward(sM, currentB, lowerB, middleB, upperB, cX, cY, NULL);
    cell_calculateBackward(sM, upperB, middleB, NULL, NULL, cX, cY, NULL);
    cell_calculateBackward(sM, lowerB, NULL, NULL, middleB, cX, cY, NULL);
    const double totalProbForward = cell_dotProduct2(currentF, sM, sM->endStateProb);
    const double totalProbBackward = cell_dotProduct2(middleB, sM, sM->startStateProb);
    CHECK(isfinite(totalProbForward) && fabs(totalProbForward - totalProbBackward) < 0.00001);
}

/* test_diagonalDPCalculations, tests/pairwiseAlignerTest.c:242-324, statement for statement: a complete matrix for
 * AGCG / AGTTCG diagonal by diagonal through the DpMatrix API, forward == backward == every diagonal's total, and the four
 * pairs.  SURVEY 8c recorded the reference's numbers for this input: total -17.519321161239 and the four scores. */
static void test_diagonalDPCalculations_gpu(void) {
    const char *sX = "AGCG", *sY = "AGTTCG";
    const int64_t lX = 4, lY = 6;
    SymbolString sX2 = symbolString_construct(sX, lX), sY2 = symbolString_construct(sY, lY);
    StateMachine *sM = stateMachine5_construct(fiveState);
    DpMatrix *fwd = dpMatrix_construct(lX + lY, sM->stateNumber), *bwd = dpMatrix_construct(lX + lY, sM->stateNumber);
    stList *anchorPairs = stList_construct();
    Band *band = band_construct(anchorPairs, lX, lY, 2);
    BandIterator *bandIt = bandIterator_construct(band);
    for (int64_t i = 0; i <= lX + lY; i++) {
        Diagonal d = bandIterator_getNext(bandIt);
        dpDiagonal_zeroValues(dpMatrix_createDiagonal(bwd, d));
        dpDiagonal_zeroValues(dpMatrix_createDiagonal(fwd, d));
    }
    dpDiagonal_initialiseValues(dpMatrix_getDiagonal(fwd, 0), sM, sM->startStateProb);
    dpDiagonal_initialiseValues(dpMatrix_getDiagonal(bwd, lX + lY), sM, sM->endStateProb);
    for (int64_t i = 1; i <= lX + lY; i++) diagonalCalculationForward(sM, i, fwd, sX2, sY2);
    for (int64_t i = lX + lY; i > 0; i--) diagonalCalculationBackward(sM, i, bwd, sX2, sY2);
    const double totalProbForward = cell_dotProduct2(dpDiagonal_getCell(dpMatrix_getDiagonal(fwd, lX + lY), lX - lY), sM, sM->endStateProb);
    const double totalProbBackward = cell_dotProduct2(dpDiagonal_getCell(dpMatrix_getDiagonal(bwd, 0), 0), sM, sM->startStateProb);
    CHECK(fabs(totalProbForward - totalProbBackward) < 0.001);
    CHECK(fabs(totalProbForward - (-17.519321161239)) < 1e-9); /* computeForwardProbability of the same input, SURVEY 8c */
    for (int64_t i = 0; i <= lX + lY; i++)
        CHECK(fabs(totalProbForward - diagonalCalculationTotalProbability(sM, i, fwd, bwd, sX2, sY2)) < 0.01);
    stList *alignedPairs = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    void *extraArgs[1] = {alignedPairs};
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->threshold = 0.2;
    for (int64_t i = 1; i <= lX + lY; i++) diagonalCalculationPosteriorMatchProbs(sM, i, fwd, bwd, sX2, sY2, totalProbForward, p, extraArgs);
    CHECK(stList_length(alignedPairs) == 4);
    const int64_t want[4][3] = {{9944673, 0, 0}, {9259684, 1, 1}, {8665179, 2, 4}, {9893294, 3, 5}}; /* SURVEY 8c */
    for (int64_t i = 0; i < stList_length(alignedPairs) && i < 4; i++) {
        stIntTuple *pair = stList_get(alignedPairs, i);
        CHECK(stIntTuple_get(pair, 1) == want[i][1] && stIntTuple_get(pair, 2) == want[i][2]);
        CHECK(llabs((long long)(stIntTuple_get(pair, 0) - want[i][0])) <= 100); /* total of the last diagonal instead of per-diagonal totals: ~1e-5 */
    }
    /* and against the engine itself on the same input */
    stList *engine = getAlignedPairsUsingAnchors(sM, sX, sY, anchorPairs, p, 0, 0);
    CHECK(stList_length(engine) == 4);
    stList_destruct(engine);
    stList_destruct(alignedPairs);
    for (int64_t i = 0; i <= lX + lY; i++) {
        dpMatrix_deleteDiagonal(fwd, i);
        dpMatrix_deleteDiagonal(bwd, i);
    }
    dpMatrix_destruct(fwd);
    dpMatrix_destruct(bwd);
    bandIterator_destruct(bandIt);
    band_destruct(band);
    stList_destruct(anchorPairs);
    pairwiseAlignmentBandingParameters_destruct(p);
    stateMachine_destruct(sM);
    free(sX2.sequence);
    free(sY2.sequence);
}

/* getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps (inc/pairwiseAligner.h:264) driven exactly as
 * getAlignedPairsUsingAnchors drives it in the reference (impl/pairwiseAligner.c:1411-1444): scratch list + caller's list
 * in extraArgs, a coordinate-correction callback that shifts and pops -- must equal getAlignedPairsUsingAnchors. */
static int correctionCalls = 0;
static int64_t consumed = 0; /* the minimal stList has no pop: the callback remembers how far it has read the scratch list */
static void pop_and_shift(int64_t offsetX, int64_t offsetY, void *extraArgs) {
    stList *sub = ((void **)extraArgs)[0], *all = ((void **)extraArgs)[1];
    correctionCalls++;
    for (int64_t k = stList_length(sub) - 1; k >= consumed; k--) { /* the reference pops from the end (:1415-1417) */
        stIntTuple *i = stList_get(sub, k);
        stList_append(all, stIntTuple_construct3(stIntTuple_get(i, 0), stIntTuple_get(i, 1) + offsetX, stIntTuple_get(i, 2) + offsetY));
    }
    consumed = stList_length(sub);
}
static void test_splitting_wrapper_gpu(void) {
    /* two similar 600-base sequences with a 40-base insertion in Y in the middle: anchors around the gap, split at 10 x 10 */
    enum { L = 600 };
    static char sx[L + 1], sy[L + 41];
    const char bases[4] = {'A', 'C', 'G', 'T'};
    uint64_t z = 12345;
    for (int i = 0; i < L; i++) {
        z = z * 6364136223846793005ull + 1442695040888963407ull;
        sx[i] = bases[(z >> 33) & 3];
    }
    sx[L] = 0;
    memcpy(sy, sx, 300);
    for (int i = 0; i < 40; i++) sy[300 + i] = bases[(i * 7 + 1) & 3];
    memcpy(sy + 340, sx + 300, 300);
    sy[L + 40] = 0;
    stList *anchors = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    for (int x = 20; x < 290; x += 30) stList_append(anchors, stIntTuple_construct3(x, x, 10));
    for (int x = 310; x < 590; x += 30) stList_append(anchors, stIntTuple_construct3(x, x + 40, 10));
    StateMachine *sM = stateMachine5_construct(fiveState);
    PairwiseAlignmentParameters *p = pairwiseAlignmentBandingParameters_construct();
    p->splitMatrixBiggerThanThis = 100;
    p->diagonalExpansion = 10;
    stList *want = getAlignedPairsUsingAnchors(sM, sx, sy, anchors, p, 0, 0);
    stList *splits = getSplitPoints(anchors, L, L + 40, 100, 0, 0);
    CHECK(stList_length(splits) >= 2);
    stList *sub = stList_construct();
    stList *all = stList_construct3(0, (void (*)(void *))stIntTuple_destruct);
    void *extraArgs[2] = {sub, all};
    correctionCalls = 0;
    getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(sM, anchors, sx, sy, L, L + 40, p, 0, 0, diagonalCalculationPosteriorMatchProbs,
                                                               NULL, extraArgs);
    /* without a correction callback every rectangle's pairs stay in the scratch list, in rectangle coordinates */
    CHECK(stList_length(sub) == stList_length(want) && correctionCalls == 0);
    /* with the callback: called once per rectangle with the rectangle's origin, and the caller's list comes out as
     * getAlignedPairsUsingAnchors builds it -- same triples, same order */
    stList *sub2 = stList_construct();
    void *extraArgs2[2] = {sub2, all};
    consumed = 0;
    getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(sM, anchors, sx, sy, L, L + 40, p, 0, 0, diagonalCalculationPosteriorMatchProbs,
                                                               (void (*)())pop_and_shift, extraArgs2);
    CHECK(correctionCalls == stList_length(splits));
    /* anchors 30 apart and split at 100 cells: only ~10 x 10 boxes around the anchors are aligned, one rectangle each */
    CHECK(stList_length(all) == stList_length(want) && stList_length(want) > 100);
    if (stList_length(all) != stList_length(want) || stList_length(want) <= 100)
        fprintf(stderr, "splitting wrapper: %lld pairs through the wrapper, %lld from getAlignedPairsUsingAnchors, %lld rectangles\n",
                (long long)stList_length(all), (long long)stList_length(want), (long long)stList_length(splits));
    for (int64_t i = 0; i < stList_length(all) && i < stList_length(want); i++)
        for (int f = 0; f < 3; f++) CHECK(stIntTuple_get(stList_get(all, i), f) == stIntTuple_get(stList_get(want, i), f));
    for (int64_t i = 0; i < stList_length(sub); i++) stIntTuple_destruct(stList_get(sub, i));
    for (int64_t i = 0; i < stList_length(sub2); i++) stIntTuple_destruct(stList_get(sub2, i));
    stList_destruct(sub);
    stList_destruct(sub2);
    stList_destruct(all);
    stList_destruct(splits);
    stList_destruct(want);
    stList_destruct(anchors);
    pairwiseAlignmentBandingParameters_destruct(p);
    stateMachine_destruct(sM);
}

/* sM->cellCalculate (inc/stateMachine.h:47-50): the vtable entry hands the ordered transition list of
 * stateMachine5_cellCalculate / stateMachine3_cellCalculate (impl/stateMachine.c:450-480, :689-714) to the caller's
 * callback.  (a) the list itself, recorded: order, states, emission and transition log-probabilities, absent neighbours
 * skipped; (b) with a forward fold as the callback -- to[t] = logAdd(to[t], from[f] + (eP + tP)), doTransitionForward of
 * impl/pairwiseAligner.c:382-385 -- the cell equals what cell_calculateForward computes (on the GPU). */
typedef struct {
    int n;
    int64_t from[16], to[16];
    double eP[16], tP[16];
    double *src[16];
} TransitionLog;
static void record_transition(double *from, double *to, int64_t f, int64_t t, double eP, double tP, void *extra) {
    TransitionLog *log = extra;
    (void)to;
    if (log->n < 16) {
        log->src[log->n] = from;
        log->from[log->n] = f;
        log->to[log->n] = t;
        log->eP[log->n] = eP;
        log->tP[log->n] = tP;
    }
    log->n++;
}
static void fold_forward(double *from, double *to, int64_t f, int64_t t, double eP, double tP, void *extra) {
    (void)extra;
    to[t] = logAdd(to[t], from[f] + (eP + tP));
}
static void test_cellCalculate_vtable(int gpu) {
    StateMachine *sM5 = stateMachine5_construct(fiveState), *sM3 = stateMachine3_construct(threeState);
    double lower[5] = {-1, -2, -3, -4, -5}, middle[5] = {-1.5, -2.5, -3.5, -4.5, -5.5}, upper[5] = {-0.5, -6, -7, -8, -9}, cur[5];
    TransitionLog log;
    memset(&log, 0, sizeof log);
    sM5->cellCalculate(sM5, cur, lower, middle, upper, c, g, record_transition, &log);
    CHECK(log.n == 13);
    const int64_t from5[13] = {0, 1, 0, 3, 0, 1, 2, 3, 4, 0, 2, 0, 4}, to5[13] = {1, 1, 3, 3, 0, 0, 0, 0, 0, 2, 2, 4, 4};
    for (int i = 0; i < 13 && i < log.n; i++) {
        CHECK(log.from[i] == from5[i] && log.to[i] == to5[i]);
        CHECK(log.src[i] == (i < 4 ? lower : (i < 9 ? middle : upper)));
    }
    CHECK(log.eP[0] == log.eP[3] && log.eP[4] == log.eP[8] && log.eP[9] == log.eP[12] && log.eP[0] != log.eP[4]);
    memset(&log, 0, sizeof log);
    sM5->cellCalculate(sM5, cur, NULL, middle, NULL, a, n, record_transition, &log); /* band edge: only the middle block; an N */
    CHECK(log.n == 5 && log.eP[0] == -2.772588722 && log.src[0] == middle);
    memset(&log, 0, sizeof log);
    sM3->cellCalculate(sM3, cur, lower, middle, upper, t, t, record_transition, &log);
    CHECK(log.n == 9);
    const int64_t from3[9] = {0, 1, 2, 0, 1, 2, 0, 2, 1}, to3[9] = {1, 1, 1, 0, 0, 0, 2, 2, 2};
    for (int i = 0; i < 9 && i < log.n; i++) CHECK(log.from[i] == from3[i] && log.to[i] == to3[i]);
    memset(&log, 0, sizeof log);
    sM3->cellCalculate(sM3, cur, lower, NULL, upper, n, a, record_transition, &log);
    CHECK(log.n == 6 && log.eP[0] == -1.386294361); /* gapX emission of an N */
    if (gpu) {
        StateMachine *both[2] = {sM5, sM3};
        for (int k = 0; k < 2; k++) {
            StateMachine *sM = both[k];
            double viaCallback[5], viaGpu[5];
            for (int64_t i = 0; i < sM->stateNumber; i++) viaCallback[i] = viaGpu[i] = LOG_ZERO;
            sM->cellCalculate(sM, viaCallback, lower, middle, upper, g, t, fold_forward, NULL);
            cell_calculateForward(sM, viaGpu, lower, middle, upper, g, t, NULL);
            for (int64_t i = 0; i < sM->stateNumber; i++) CHECK(fabs(viaCallback[i] - viaGpu[i]) < 1e-12);
        }
    }
    stateMachine_destruct(sM5);
    stateMachine_destruct(sM3);
}

int main(int argc, char **argv) {
    const int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    test_bands();
    test_split_points();
    test_params_json_and_anchor_conversion();
    test_hmm(fiveState);
    test_hmm(fiveStateAsymmetric);
    test_hmm(threeState);
    test_hmm(threeStateAsymmetric);
    test_symbols();
    test_models();
    test_diagonal_logadd_dynamic_band_hmm_json();
    test_dp_containers();
    test_list_helpers();
    test_cellCalculate_vtable(gpu);
    if (gpu) test_known_answers_gpu();
    if (gpu) test_getPosteriorProbsWithBanding_gpu();
    if (gpu) test_consumers_gpu();
    if (gpu) {
        StateMachine *sM5 = stateMachine5_construct(fiveState), *sM3 = stateMachine3_construct(threeState);
        test_cell_gpu(sM5);
        test_cell_gpu(sM3);
        stateMachine_destruct(sM5);
        stateMachine_destruct(sM3);
        test_diagonalDPCalculations_gpu();
        test_splitting_wrapper_gpu();
    }
    printf("%s: %d failure(s)\n", gpu ? "gpu" : "cpu", failures);
    return failures != 0;
}
