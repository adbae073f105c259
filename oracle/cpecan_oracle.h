/*
 * cpecan_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A from-scratch plain-C restatement of the banded pair-HMM forward/backward/posterior
 * path of benedictpaten/cPecan (impl/pairwiseAligner.c:20-949,1206-1513 and
 * impl/stateMachine.c:23-112,255-823).  It exists only to CHECK the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (cpecan_amd/csrc) never links, includes or calls anything here.
 *
 * Parity pin: the reference cannot be compiled in this image (it needs the
 * un-vendored sibling library sonLib, include.mk:2-9), so this oracle is pinned by
 * (a) every fixed-input fixture of the reference's own tests
 *     (tests/pairwiseAlignerTest.c: test_diagonal :17, test_bands :69, test_symbol :146,
 *      test_cell :155, test_diagonalDPCalculations :242, test_getSplitPoints :578,
 *      test_hmm :997) and
 * (b) the known answers the survey recorded from the reference.
 * See tests/test_oracle_golden.py.
 *
 * All arrays are flat; all functions are re-entrant.
 */
#ifndef CPECAN_ORACLE_H_
#define CPECAN_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_STATES 5
#define ORC_MAX_TRANSITIONS 16
#define ORC_PROB_1 10000000 /* inc/pairwiseAligner.h:26 */

/* inc/stateMachine.h:28-33 */
enum { ORC_FIVE_STATE = 0, ORC_FIVE_STATE_ASYM = 1, ORC_THREE_STATE = 2, ORC_THREE_STATE_ASYM = 3 };

/* One anti-diagonal of the band; inc/pairwiseAligner.h:116-120 */
typedef struct {
    int64_t xay;  /* x + y */
    int64_t xmyL; /* smallest x - y on the diagonal */
    int64_t xmyR; /* largest x - y on the diagonal */
} OrcDiagonal;

/* A transition of the pair-HMM as it appears in the per-cell ordered list
 * (impl/stateMachine.c:450-480 and :689-714). block: 0 = "lower" neighbour (x-1,y),
 * 1 = "middle" (x-1,y-1), 2 = "upper" (x,y-1). */
typedef struct {
    int32_t block, from, to;
    double tP;
} OrcTransition;

typedef struct {
    int32_t type;
    int32_t S;                                /* number of states */
    int32_t matchState, gapXState, gapYState; /* impl/stateMachine.c:511-513 */
    int32_t nTransitions;
    OrcTransition tr[ORC_MAX_TRANSITIONS];
    double matchEm[25]; /* [cX*5+cY], row/col 4 = N  (impl/stateMachine.c:359-366) */
    double gapXEm[5];   /* [cX], entry 4 = N         (impl/stateMachine.c:351-357) */
    double gapYEm[5];
    double start[ORC_MAX_STATES], raggedStart[ORC_MAX_STATES];
    double end[ORC_MAX_STATES], raggedEnd[ORC_MAX_STATES];
} OrcModel;

/* Expectation accumulator / model file contents; inc/stateMachine.h:61-67 */
typedef struct {
    int32_t type;
    int32_t S;
    double T[ORC_MAX_STATES * ORC_MAX_STATES]; /* [from*S+to] */
    double E[ORC_MAX_STATES * 16];             /* [state*16 + x*4 + y] */
    double likelihood;
} OrcHmm;

/* The subset of PairwiseAlignmentParameters the DP reads; inc/pairwiseAligner.h:28-41,
 * defaults impl/pairwiseAligner.c:1334-1348 */
typedef struct {
    double threshold;
    int64_t minDiagsBetweenTraceBack;
    int64_t traceBackDiagonals;
    int64_t diagonalExpansion;
    int64_t splitMatrixBiggerThanThis;
    int32_t dynamicAnchorExpansion;
} OrcParams;

/* Optional per-problem debug record filled by the banded driver (single region only). */
typedef struct {
    int64_t nDiagonals;   /* lX+lY+1 */
    int64_t nCells;       /* sum of band widths */
    int64_t nTracebacks;  /* number of traceback segments */
    int64_t *cellOffset;  /* [nDiagonals+1] prefix sum of widths, malloc'd by the oracle */
    double *totalUsed;    /* [nDiagonals] total log-prob used when emitting diagonal d (NaN if none) */
    double *fbMatch;      /* [nCells] F.match + B.match at emit time (NaN where not emitted) */
    double *forward;      /* [nCells*S] forward values, AoS [cell][state] */
} OrcTrace;

void orc_trace_free(OrcTrace *t);

/* ---- primitives ---- */
double orc_logAdd(double x, double y);                /* impl/pairwiseAligner.c:287-307 */
int32_t orc_symbol(char c);                           /* impl/pairwiseAligner.c:317-334 */
int orc_diagonal_valid(int64_t xay, int64_t xmyL, int64_t xmyR); /* :30-35 */
void orc_params_default(OrcParams *p);

/* Band for (anchors, lX, lY); anchors = n triples (x, y, expansion), 0-based sequence coords.
 * out must hold lX+lY+1 diagonals.  Returns 0, or -1 on an invalid diagonal. */
int orc_band(const int64_t *anchors, int64_t n, int64_t lX, int64_t lY, int64_t expansion, int dynamic,
             OrcDiagonal *out);

/* Split rectangles (x1,y1,x2,y2); out must hold 4*(n+2) int64. Returns the count. */
int64_t orc_split_points(const int64_t *anchors, int64_t n, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                         int raggedLeft, int raggedRight, int64_t *out);

/* ---- model ---- */
void orc_model_default(OrcModel *m, int32_t type); /* stateMachine5/3_construct */
int orc_model_from_hmm(OrcModel *m, const OrcHmm *h); /* hmm_getStateMachine */
void orc_hmm_init(OrcHmm *h, int32_t type, double pseudo);
void orc_hmm_normalise(OrcHmm *h);

/* ---- one DP cell (exported for the reference's test_cell) ---- */
void orc_cell_forward(const OrcModel *m, double *current, const double *lower, const double *middle,
                      const double *upper, int32_t cX, int32_t cY);
void orc_cell_backward(const OrcModel *m, const double *current, double *lower, double *middle, double *upper,
                       int32_t cX, int32_t cY);

/* ---- whole-API level (mirrors getAlignedPairsUsingAnchors & co.) ----
 * Output triples are (score, x, y) int64, in the reference's list order. The returned
 * buffers are malloc'd; release with orc_free(). */
int64_t orc_aligned_pairs(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors, int64_t n,
                          const OrcParams *p, int raggedLeft, int raggedRight, int64_t **outTriples);
void orc_aligned_pairs_with_indels(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors,
                                   int64_t n, const OrcParams *p, int raggedLeft, int raggedRight,
                                   int64_t **match, int64_t *nMatch, int64_t **gapX, int64_t *nGapX,
                                   int64_t **gapY, int64_t *nGapY);
void orc_expectations(const OrcModel *m, OrcHmm *acc, const char *sX, const char *sY, const int64_t *anchors,
                      int64_t n, const OrcParams *p, int raggedLeft, int raggedRight);
double orc_forward_prob(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors, int64_t n,
                        const OrcParams *p, int raggedLeft, int raggedRight);

/* Single region (no splitting), with trace; same output order as orc_aligned_pairs. */
int64_t orc_aligned_pairs_traced(const OrcModel *m, const char *sX, const char *sY, const int64_t *anchors,
                                 int64_t n, const OrcParams *p, int raggedLeft, int raggedRight,
                                 int64_t **outTriples, OrcTrace *trace);

/* Number of band cells (sum of widths over all split regions) -- the bench metric's unit. */
int64_t orc_band_cells(const char *sX, const char *sY, const int64_t *anchors, int64_t n, const OrcParams *p,
                       int raggedLeft, int raggedRight);

/* Batch driver for the CPU baseline: runs orc_aligned_pairs on nPairs problems packed as
 * concatenated strings (NUL-terminated, offsets in seqOff[2*i], seqOff[2*i+1]) and anchors
 * (anchorOff[i]..anchorOff[i+1] triples). Uses OpenMP threads if nThreads>1 and built with
 * -fopenmp. Returns total pairs emitted; *cells receives band cells processed. */
int64_t orc_batch_aligned_pairs(const OrcModel *m, const char *seqBlob, const int64_t *seqOff,
                                const int64_t *anchors, const int64_t *anchorOff, int64_t nPairs,
                                const OrcParams *p, int raggedLeft, int raggedRight, int nThreads,
                                int64_t *cells);

/* orc_expectations over a batch (same blob layout as orc_batch_aligned_pairs), OpenMP over problems; the counts are
 * ADDED to *acc.  Returns the band cells processed. */
int64_t orc_batch_expectations(const OrcModel *m, OrcHmm *acc, const char *seqBlob, const int64_t *seqOff,
                               const int64_t *anchors, const int64_t *anchorOff, int64_t nPairs, const OrcParams *p,
                               int raggedLeft, int raggedRight, int nThreads);

/* ---- consumers of the posterior lists.  Triples are (score, x, y) int64. ----
 * Pinning: orc_left_shift_alignment is pinned by the reference's test vector (tests/pairwiseAlignerTest.c:944-995).
 * The reference has no test of reweightAlignedPairs2 or of the MEA chain: those two restatements are checked by
 * hand-worked cases and an independent chain DP only -- PARITY UNPINNED for them. */
/* reweightAlignedPairs2, impl/pairwiseAligner.c:1519-1558: in place; gapGamma <= 0 leaves the list unchanged. */
void orc_reweight_aligned_pairs(int64_t *triples, int64_t n, int64_t lX, int64_t lY, double gapGamma);
/* scoreByPosteriorProbability / scoreByPosteriorProbabilityIgnoringGaps, :1578-1597 */
double orc_score_by_posterior(int64_t lX, int64_t lY, const int64_t *triples, int64_t n);
double orc_score_by_posterior_ignoring_gaps(const int64_t *triples, int64_t n);
/* getMaximalExpectedAccuracyPairwiseAlignment, :1603-1724.  gapGamma is the float of PairwiseAlignmentParameters
 * (inc/pairwiseAligner.h:38): parts of the score arithmetic run in float, as in the reference.  out holds n triples;
 * returns the number written. */
int64_t orc_mea_alignment(const int64_t *pairs, int64_t n, const int64_t *gapX, int64_t nGapX, const int64_t *gapY,
                          int64_t nGapY, int64_t lX, int64_t lY, float gapGamma, int64_t *out, double *alignmentScore);
/* leftShiftAlignment, :1726-1762.  out holds n + min(lX, lY) + 1 triples; returns the number written. */
int64_t orc_left_shift_alignment(const int64_t *pairs, int64_t n, const char *sX, const char *sY, int64_t *out);
/* scoreByIdentity / scoreByIdentityIgnoringGaps, :1562-1580 */
double orc_score_by_identity(const char *sX, const char *sY, int64_t lX, int64_t lY, const int64_t *triples, int64_t n);
double orc_score_by_identity_ignoring_gaps(const char *sX, const char *sY, const int64_t *triples, int64_t n);
/* filterPairwiseAlignmentToMakePairsOrdered, impl/multipleAligner.c:945-972 (the two-sequence case of
 * pairwiseAlignColumns, :358-492): the heaviest chain of pairs with weight >= matchGamma, as a filter of the input list;
 * the survivors come out in reverse input order.  The reference adds st_random() * 0.00001 to every weight (:145), so
 * its own output is not a function of its input; this restatement leaves the jitter out -- PARITY UNPINNED (the
 * reference's test of pairwiseAlignColumns, tests/multipleAlignerTest.c, checks properties only, which the tests here
 * repeat).  The pairs must be distinct cells.  out holds n triples; returns the number written. */
int64_t orc_filter_pairs_ordered(const int64_t *pairs, int64_t n, int64_t lX, int64_t lY, double matchGamma, int64_t *out);
/* filterToRemoveOverlap, impl/pairwiseAligner.c:1095-1135 (input sorted by x, then y); out holds n triples */
int64_t orc_filter_to_remove_overlap(const int64_t *pairs, int64_t n, int64_t *out);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
