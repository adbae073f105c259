/*
 * cpecan_dropin.h -- the reference's own symbol names for the hot path, layered on include/cpecan_hip.h.
 *
 * A C caller of cPecan (impl/multipleAligner.c:660, cPecanAlign.c:123, cPecanRealign.c:532,537) can include this
 * header instead of inc/pairwiseAligner.h + inc/stateMachine.h and link libcpecan_hip.so: same names, same argument
 * meaning, same struct layouts for the types callers touch.
 *
 *   struct _stateMachine      inc/stateMachine.h:37-55      (vtable layout kept; see cellCalculate note below)
 *   Hmm                       inc/stateMachine.h:61-67
 *   PairwiseAlignmentParameters inc/pairwiseAligner.h:28-41
 *   Diagonal                  inc/pairwiseAligner.h:116-120
 *
 * sonLib: the reference takes stList / stIntTuple from sonLib, which is not vendored.  This library carries a
 * minimal implementation of exactly the container calls the API needs, exported as WEAK symbols so that a real
 * sonLib linked into the same program takes precedence.
 *
 * Not provided: the lastz path of getAlignedPairs / getExpectations (beyond anchorMatrixBiggerThanThis the reference shells
 * out to lastz, impl/pairwiseAligner.c:1032-1042).  The primitives the reference's unit tests link (inc/pairwiseAligner.h:
 * 186-237: DpDiagonal, DpMatrix, cell_calculateForward / Backward, cell_dotProduct[2], diagonalCalculationForward /
 * Backward / TotalProbability, the per-diagonal emitters on DpMatrix rows) ARE provided: the containers are host memory
 * with the reference's semantics, their DP arithmetic runs on the GPU (cpecan_ref_cells, one lane, the reference's order
 * of operations -- the library holds no CPU implementation of the recurrences).  sM->cellCalculate, which takes an
 * arbitrary per-transition callback, hands that callback the model's ordered transition list (states, emission and
 * transition log-probabilities, absent neighbours skipped: impl/stateMachine.c:450-480, :689-714) -- the vtable's contract,
 * no arithmetic of its own (round 4; it aborted before); getPosteriorProbsWithBanding accepts the reference's three
 * emitters (recognised by address, see below) and refuses foreign diagonal callbacks.
 */
#ifndef CPECAN_DROPIN_H_
#define CPECAN_DROPIN_H_

#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#include "cpecan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- minimal sonLib subset (weak) ---- */
typedef struct _stList stList;
typedef struct _stIntTuple stIntTuple;
stList *stList_construct(void);
stList *stList_construct3(int64_t size, void (*destructElement)(void *));
void stList_destruct(stList *list);
int64_t stList_length(stList *list);
void *stList_get(stList *list, int64_t index);
void stList_append(stList *list, void *item);
stIntTuple *stIntTuple_construct2(int64_t a, int64_t b);
stIntTuple *stIntTuple_construct3(int64_t a, int64_t b, int64_t c);
void stIntTuple_destruct(stIntTuple *t);
int64_t stIntTuple_get(stIntTuple *t, int64_t index);
int64_t stIntTuple_length(stIntTuple *t);

/* ---- inc/stateMachine.h ---- */
#define SYMBOL_NUMBER 5
#define SYMBOL_NUMBER_NO_N 4
typedef enum { a = 0, c = 1, g = 2, t = 3, n = 4 } Symbol;
typedef enum { fiveState = 0, fiveStateAsymmetric = 1, threeState = 2, threeStateAsymmetric = 3 } StateMachineType;

typedef struct _stateMachine StateMachine;
struct _stateMachine {
    StateMachineType type;
    int64_t stateNumber;
    int64_t matchState;
    int64_t gapXState;
    int64_t gapYState;
    double (*startStateProb)(StateMachine *sM, int64_t state);
    double (*endStateProb)(StateMachine *sM, int64_t state);
    double (*raggedEndStateProb)(StateMachine *sM, int64_t state);
    double (*raggedStartStateProb)(StateMachine *sM, int64_t state);
    void (*cellCalculate)(StateMachine *sM, double *current, double *lower, double *middle, double *upper, Symbol cX,
                          Symbol cY, void (*doTransition)(double *, double *, int64_t, int64_t, double, double, void *),
                          void *extraArgs);
};

typedef struct _hmm {
    StateMachineType type;
    double *transitions;
    double *emissions;
    double likelihood;
    int64_t stateNumber;
} Hmm;

Hmm *hmm_constructEmpty(double pseudoExpectation, StateMachineType type);
void hmm_destruct(Hmm *hmm);
void hmm_write(Hmm *hmm, FILE *fileHandle);
void hmm_addToTransitionExpectation(Hmm *hmm, int64_t from, int64_t to, double p);
double hmm_getTransition(Hmm *hmm, int64_t from, int64_t to);
void hmm_setTransition(Hmm *hmm, int64_t from, int64_t to, double p);
void hmm_addToEmissionsExpectation(Hmm *hmm, int64_t state, Symbol x, Symbol y, double p);
double hmm_getEmissionsExpectation(Hmm *hmm, int64_t state, Symbol x, Symbol y);
void hmm_setEmissionsExpectation(Hmm *hmm, int64_t state, Symbol x, Symbol y, double p);
Hmm *hmm_loadFromFile(const char *fileName);
void hmm_randomise(Hmm *hmm);              /* inc/stateMachine.h:71 (impl/stateMachine.c:114); st_random is a weak symbol */
Hmm *hmm_jsonParse(char *buf, size_t r);   /* inc/stateMachine.h:91 (impl/stateMachine.c:204) */
double st_random(void);                     /* sonLib's; weak here */
void hmm_normalise(Hmm *hmm);
StateMachine *hmm_getStateMachine(Hmm *hmm);
StateMachine *stateMachine5_construct(StateMachineType type);
StateMachine *stateMachine3_construct(StateMachineType type);
void stateMachine_destruct(StateMachine *stateMachine);
/* the flattened parameters behind a StateMachine built by this library (NULL for a foreign vtable) */
const cpecan_model *stateMachine_flat(StateMachine *sM);

/* ---- inc/pairwiseAligner.h ---- */
#define PAIR_ALIGNMENT_PROB_1 10000000
typedef struct _pairwiseAlignmentBandingParameters {
    double threshold;
    int64_t minDiagsBetweenTraceBack;
    int64_t traceBackDiagonals;
    int64_t diagonalExpansion;
    int64_t constraintDiagonalTrim;
    int64_t anchorMatrixBiggerThanThis;
    int64_t repeatMaskMatrixBiggerThanThis;
    int64_t splitMatrixBiggerThanThis;
    bool alignAmbiguityCharacters;
    float gapGamma;
    bool dynamicAnchorExpansion;
} PairwiseAlignmentParameters;

PairwiseAlignmentParameters *pairwiseAlignmentBandingParameters_construct(void);
void pairwiseAlignmentBandingParameters_destruct(PairwiseAlignmentParameters *p);
/* inc/pairwiseAligner.h:51 (impl/pairwiseAligner.c:1354-1408): a flat JSON object of the eleven field names; fields that
 * are absent keep their defaults; an unknown key aborts, as st_errAbort does in the reference. */
PairwiseAlignmentParameters *pairwiseAlignmentParameters_jsonParse(char *buf, size_t r);

/* sonLib's pairwise alignment types (pairwiseAlignment.h, commonC.h), as far as the reference reads them
 * (impl/pairwiseAligner.c:979-1003, cPecanRealign.c:49-96); define CPECAN_HAVE_SONLIB_ALIGNMENT to use sonLib's own. */
#ifndef CPECAN_HAVE_SONLIB_ALIGNMENT
#define PAIRWISE_MATCH 0
#define PAIRWISE_INDEL_X 1
#define PAIRWISE_INDEL_Y 2
struct List {
    int64_t length;
    int64_t maxLength;
    void **list;
    void (*destructElement)(void *);
};
struct AlignmentOperation {
    int64_t opType;
    int64_t length;
    float score;
};
struct PairwiseAlignment {
    char *contig1;
    int64_t start1;
    int64_t end1;
    int64_t strand1;
    char *contig2;
    int64_t start2;
    int64_t end2;
    int64_t strand2;
    double score;
    struct List *operationList;
};
#endif
/* inc/pairwiseAligner.h:73 (impl/pairwiseAligner.c:979-1003): both strands must be forward (asserted there). */
stList *convertPairwiseForwardStrandAlignmentToAnchorPairs(struct PairwiseAlignment *pA, int64_t trim, int64_t diagonalExpansion);
/* inc/pairwiseAligner.h (impl/pairwiseAligner.c:1095-1135): a new list; the input is left alone, as in the reference */
stList *filterToRemoveOverlap(stList *sortedOverlappingPairs);

stList *getAlignedPairsUsingAnchors(StateMachine *sM, const char *sX, const char *sY, stList *anchorPairs,
                                    PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd,
                                    bool alignmentHasRaggedRightEnd);
void getAlignedPairsWithIndelsUsingAnchors(StateMachine *sM, const char *sX, const char *sY, stList *anchorPairs,
                                           PairwiseAlignmentParameters *p, stList **alignedPairs, stList **gapXPairs,
                                           stList **gapYPairs, bool alignmentHasRaggedLeftEnd,
                                           bool alignmentHasRaggedRightEnd);
void getExpectationsUsingAnchors(StateMachine *sM, Hmm *hmmExpectations, const char *sX, const char *sY,
                                 stList *anchorPairs, PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd,
                                 bool alignmentHasRaggedRightEnd);
double computeForwardProbability(char *seqX, char *seqY, stList *anchorPairs, PairwiseAlignmentParameters *p,
                                 StateMachine *sM, bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
/* The entry points that find their own anchors (impl/pairwiseAligner.c:1481-1513).  The reference anchors with lastz only
 * when lX * lY > p->anchorMatrixBiggerThanThis (:1164); up to that size these are the functions above with no anchors,
 * and that is what is provided.  Beyond it they abort: the lastz anchoring (:959-1196) is not part of this library --
 * pass anchors to the *UsingAnchors functions. */
stList *getAlignedPairs(StateMachine *sM, const char *sX, const char *sY, PairwiseAlignmentParameters *p,
                        bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);
void getAlignedPairsWithIndels(StateMachine *sM, const char *sX, const char *sY, PairwiseAlignmentParameters *p,
                               stList **alignedPairs, stList **gapXPairs, stList **gapYPairs, bool alignmentHasRaggedLeftEnd,
                               bool alignmentHasRaggedRightEnd);
void getExpectations(StateMachine *sM, Hmm *hmmExpectations, const char *sX, const char *sY, PairwiseAlignmentParameters *p,
                     bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd);

/* Consumers of the posterior lists, evaluated on the GPU (inc/pairwiseAligner.h:86-98, :272-290;
 * impl/pairwiseAligner.c:1519-1790).  reweightAlignedPairs2 consumes its input list unless gapGamma <= 0. */
stList *reweightAlignedPairs2(stList *alignedPairs, int64_t seqLengthX, int64_t seqLengthY, double gapGamma);
double scoreByPosteriorProbability(int64_t lX, int64_t lY, stList *alignedPairs);
double scoreByPosteriorProbabilityIgnoringGaps(stList *alignedPairs);
double scoreByIdentity(char *subSeqX, char *subSeqY, int64_t lX, int64_t lY, stList *alignedPairs);
double scoreByIdentityIgnoringGaps(char *subSeqX, char *subSeqY, stList *alignedPairs);
/* inc/multipleAligner.h (impl/multipleAligner.c:945): consumes its input list.  The reference perturbs every weight by
 * st_random() * 0.00001 before comparing (multipleAligner.c:145); this one does not. */
stList *filterPairwiseAlignmentToMakePairsOrdered(stList *alignedPairs, const char *seqX, const char *seqY, float matchGamma);
stList *getMaximalExpectedAccuracyPairwiseAlignment(stList *alignedPairs, stList *gapXPairs, stList *gapYPairs,
                                                    int64_t seqXLength, int64_t seqYLength, double *alignmentScore,
                                                    PairwiseAlignmentParameters *p);
stList *leftShiftAlignment(stList *alignedPairs, char *seqX, char *seqY);
stList *getShiftedMEAAlignment(char *seqX, char *seqY, stList *anchorAlignment, PairwiseAlignmentParameters *p,
                               StateMachine *sM, bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd,
                               double *alignmentScore);

typedef struct _diagonal {
    int64_t xay;
    int64_t xmyL;
    int64_t xmyR;
} Diagonal;
/* inc/pairwiseAligner.h:23: the id of the exception diagonal_construct raises (impl/pairwiseAligner.c:29-35).  sonLib's
 * stExcept is not vendored: the raise goes through the WEAK hook cpecan_dropin_throw, which a sonLib program overrides
 * with stThrowNew(id, "%s", message); the default prints the message and aborts (an uncaught exception). */
extern const char *PAIRWISE_ALIGNMENT_EXCEPTION_ID;
void cpecan_dropin_throw(const char *exceptionId, const char *message);
Diagonal diagonal_construct(int64_t xay, int64_t xmyL, int64_t xmyR); /* inc/pairwiseAligner.h:122 */
int64_t diagonal_getXay(Diagonal diagonal);
int64_t diagonal_getMinXmy(Diagonal diagonal);
int64_t diagonal_getMaxXmy(Diagonal diagonal);
int64_t diagonal_getWidth(Diagonal diagonal);
int64_t diagonal_getXCoordinate(int64_t xay, int64_t xmy);
int64_t diagonal_getYCoordinate(int64_t xay, int64_t xmy);
int64_t diagonal_equals(Diagonal diagonal1, Diagonal diagonal2);
char *diagonal_getString(Diagonal diagonal); /* inc/pairwiseAligner.h:138; malloc'ed, the caller frees */

typedef struct _band Band;
Band *band_construct(stList *anchorPairs, int64_t lX, int64_t lY, int64_t expansion);
Band *band_constructDynamic(stList *anchorPairs, int64_t lX, int64_t lY); /* impl/pairwiseAligner.c:128 (anchors carry their expansion) */
void band_destruct(Band *band);
typedef struct _bandIterator BandIterator;
BandIterator *bandIterator_construct(Band *band);
void bandIterator_destruct(BandIterator *bandIterator);
BandIterator *bandIterator_clone(BandIterator *bandIterator);
Diagonal bandIterator_getNext(BandIterator *bandIterator);
Diagonal bandIterator_getPrevious(BandIterator *bandIterator);

Symbol symbol_convertCharToSymbol(char i);
char symbol_convertSymbolToChar(Symbol i);
Symbol *symbol_convertStringToSymbols(const char *s, int64_t sL); /* inc/pairwiseAligner.h:175 */
typedef struct _symbolString {
    Symbol *sequence;
    int64_t length;
} SymbolString;
SymbolString symbolString_construct(const char *sequence, int64_t length); /* inc/pairwiseAligner.h:182 */

/* inc/pairwiseAligner.h:165-167 */
#define LOG_ZERO (-INFINITY)
double logAdd(double x, double y);

/* inc/pairwiseAligner.h:245-248: the banded engine itself, one region, with the reference's emitter-callback signature.
 * The per-diagonal emitters of the reference read DpMatrix rows; the engine keeps its DP diagonals on the GPU, so as
 * arguments of getPosteriorProbsWithBanding the three emitters below are TOKENS: recognised by address and routed to the
 * device emitter (extraArgs as in the reference: {alignedPairs} / {alignedPairs, _, gapXPairs, _, gapYPairs} / Hmm*); any
 * other callback is refused (abort with a message).  Lists are appended in the reference emitter's own order.  Called
 * directly on DpMatrix rows (as the reference's test_diagonalDPCalculations does) the two posterior emitters work on those
 * rows, their exp() on the GPU. */
typedef struct _dpMatrix DpMatrix;
void diagonalCalculationPosteriorMatchProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                            const SymbolString sX, const SymbolString sY, double totalProbability,
                                            PairwiseAlignmentParameters *p, void *extraArgs);
void diagonalCalculationPosteriorProbs(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                       const SymbolString sX, const SymbolString sY, double totalProbability,
                                       PairwiseAlignmentParameters *p, void *extraArgs);
void diagonalCalculationExpectations(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                     const SymbolString sX, const SymbolString sY, double totalProbability,
                                     PairwiseAlignmentParameters *p, void *extraArgs);
void getPosteriorProbsWithBanding(StateMachine *sM, stList *anchorPairs, const SymbolString sX, const SymbolString sY,
                                  PairwiseAlignmentParameters *p, bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd,
                                  void (*diagonalPosteriorProbFn)(StateMachine *, int64_t, DpMatrix *, DpMatrix *, const SymbolString,
                                                                  const SymbolString, double, PairwiseAlignmentParameters *, void *),
                                  void *extraArgs);
stList *getSplitPoints(stList *anchorPairs, int64_t lX, int64_t lY, int64_t maxMatrixSize, bool alignmentHasRaggedLeftEnd,
                       bool alignmentHasRaggedRightEnd);
/* inc/pairwiseAligner.h:264 (impl/pairwiseAligner.c:1273-1326): the rectangles of getSplitPoints as the problems of ONE
 * batch; per rectangle the emitter's lists (extraArgs[0], [2], [4], or the Hmm) are filled and coordinateCorrectionFn
 * (offsetX, offsetY, extraArgs) is called, in the reference's order.  Emitters as for getPosteriorProbsWithBanding. */
void getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(
    StateMachine *sM, stList *anchorPairs, const char *sX, const char *sY, int64_t lX, int64_t lY, PairwiseAlignmentParameters *p,
    bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd,
    void (*diagonalPosteriorProbFn)(StateMachine *, int64_t, DpMatrix *, DpMatrix *, const SymbolString, const SymbolString, double,
                                    PairwiseAlignmentParameters *, void *),
    void (*coordinateCorrectionFn)(), void *extraArgs);

/* inc/pairwiseAligner.h:186-237: the containers and cell-level calculations the reference's unit tests link
 * (impl/pairwiseAligner.c:382-416, :448-653); see the note at the top of this header */
void cell_calculateForward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, Symbol cX, Symbol cY, void *extraArgs);
void cell_calculateBackward(StateMachine *sM, double *current, double *lower, double *middle, double *upper, Symbol cX, Symbol cY, void *extraArgs);
double cell_dotProduct(double *cell1, double *cell2, int64_t stateNumber);
double cell_dotProduct2(double *cell1, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t));
typedef struct _dpDiagonal DpDiagonal;
DpDiagonal *dpDiagonal_construct(Diagonal diagonal, int64_t stateNumber);
DpDiagonal *dpDiagonal_clone(DpDiagonal *diagonal);
bool dpDiagonal_equals(DpDiagonal *diagonal1, DpDiagonal *diagonal2);
void dpDiagonal_destruct(DpDiagonal *dpDiagonal);
double *dpDiagonal_getCell(DpDiagonal *dpDiagonal, int64_t xmy);
double dpDiagonal_dotProduct(DpDiagonal *diagonal1, DpDiagonal *diagonal2);
void dpDiagonal_zeroValues(DpDiagonal *diagonal);
void dpDiagonal_initialiseValues(DpDiagonal *diagonal, StateMachine *sM, double (*getStateValue)(StateMachine *, int64_t));
DpMatrix *dpMatrix_construct(int64_t diagonalNumber, int64_t stateNumber);
void dpMatrix_destruct(DpMatrix *dpMatrix);
DpDiagonal *dpMatrix_getDiagonal(DpMatrix *dpMatrix, int64_t xay);
int64_t dpMatrix_getActiveDiagonalNumber(DpMatrix *dpMatrix);
DpDiagonal *dpMatrix_createDiagonal(DpMatrix *dpMatrix, Diagonal diagonal);
void dpMatrix_deleteDiagonal(DpMatrix *dpMatrix, int64_t xay);
void diagonalCalculationForward(StateMachine *sM, int64_t xay, DpMatrix *dpMatrix, const SymbolString sX, const SymbolString sY);
void diagonalCalculationBackward(StateMachine *sM, int64_t xay, DpMatrix *dpMatrix, const SymbolString sX, const SymbolString sY);
double diagonalCalculationTotalProbability(StateMachine *sM, int64_t xay, DpMatrix *forwardDpMatrix, DpMatrix *backwardDpMatrix,
                                           const SymbolString sX, const SymbolString sY);

/* inc/pairwiseAligner.h:272-276, :287 (impl/pairwiseAligner.c:1519-1548, :1562-1570) */
int64_t *getIndelProbabilities(stList *alignedPairs, int64_t seqLength, bool xIfTrueElseY);
stList *reweightAlignedPairs(stList *alignedPairs, int64_t *indelProbsX, int64_t *indelProbsY, double gapGamma);
int64_t getNumberOfMatchingAlignedPairs(char *subSeqX, char *subSeqY, stList *alignedPairs);

#ifdef __cplusplus
}
#endif
#endif
