/*
 * cpecan_hip.h -- C ABI of libcpecan_hip.so: the MI355X-native replacement for cPecan's banded
 * pair-HMM forward/backward/posterior path (reference: impl/pairwiseAligner.c:756-949 driven
 * through :1273-1513, model in impl/stateMachine.c:255-823).
 *
 * Plain pointers and sizes only.  Two layers:
 *
 *  1. the BATCH API (cpecan_batch_*): N independent alignment problems are packed, shipped to one
 *     GPU, run by hand-written HIP kernels, and returned as packed (score,x,y) int32 triples in the
 *     reference's list order.  This is what a batched caller (cPecanRealign-style loop,
 *     cPecanRealign.c:509-537) should use; it is additive to the reference API.
 *
 *  2. single-problem entry points (cpecan_get_aligned_pairs_using_anchors, ...): one call == a batch
 *     of one; they take the same arguments as the reference functions they replace with stList /
 *     StateMachine flattened to arrays / PODs.  include/cpecan_dropin.h layers the reference's own
 *     symbol names (getAlignedPairsUsingAnchors, stateMachine5_construct, ...) on top of these.
 *
 * Every function returns 0 on success or a negative CPECAN_E* code; nothing here falls back to a
 * CPU implementation: without a usable GPU the calls fail with CPECAN_ENODEVICE.
 */
#ifndef CPECAN_HIP_H_
#define CPECAN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPECAN_PROB_1 10000000 /* PAIR_ALIGNMENT_PROB_1, inc/pairwiseAligner.h:26 */

enum {
    CPECAN_OK = 0,
    CPECAN_EINVAL = -1,    /* bad argument (the reference would assert or stThrow, pairwiseAligner.c:31,761-765) */
    CPECAN_ENODEVICE = -2, /* no HIP device / kernels unavailable */
    CPECAN_EHIP = -3,      /* a HIP runtime call failed; see cpecan_last_error() */
    CPECAN_ENOMEM = -4,
    CPECAN_ESTATE = -5     /* call sequence error (e.g. results requested before run) */
};

/* StateMachineType, inc/stateMachine.h:28-33 */
enum { CPECAN_FIVE_STATE = 0, CPECAN_FIVE_STATE_ASYM = 1, CPECAN_THREE_STATE = 2, CPECAN_THREE_STATE_ASYM = 3 };

/* What the traceback emits per diagonal (the reference passes a callback, inc/pairwiseAligner.h:245-248):
 * MATCH  = diagonalCalculationPosteriorMatchProbs  (impl/pairwiseAligner.c:666)
 * INDEL  = diagonalCalculationPosteriorProbs       (impl/pairwiseAligner.c:691)
 * EXPECT = diagonalCalculationExpectations         (impl/pairwiseAligner.c:735)
 * FORWARD = no traceback: getForwardProbWithBanding     (impl/pairwiseAligner.c:879) */
enum { CPECAN_EMIT_MATCH = 0, CPECAN_EMIT_INDEL = 1, CPECAN_EMIT_EXPECT = 2, CPECAN_EMIT_FORWARD = 3 };

/* Flattened StateMachine5 / StateMachine3 (impl/stateMachine.c:377-399, 631-646): log-space
 * transition and emission parameters. Three-state models use the "short" fields and ignore "long". */
typedef struct cpecan_model {
    int32_t type; /* CPECAN_FIVE_STATE ... */
    int32_t reserved;
    double matchContinue;
    double matchFromShortGapX, matchFromShortGapY, matchFromLongGapX, matchFromLongGapY;
    double gapShortOpenX, gapShortOpenY, gapShortExtendX, gapShortExtendY, gapShortSwitchToX, gapShortSwitchToY;
    double gapLongOpenX, gapLongOpenY, gapLongExtendX, gapLongExtendY, gapLongSwitchToX, gapLongSwitchToY;
    double emissionMatch[16]; /* [x*4+y] */
    double emissionGapX[4];
    double emissionGapY[4];
} cpecan_model;

/* Hmm, inc/stateMachine.h:61-67, with fixed-size arrays. */
typedef struct cpecan_hmm {
    int32_t type;
    int32_t stateNumber;
    double transitions[25]; /* [from*stateNumber+to] */
    double emissions[80];   /* [state*16+x*4+y] */
    double likelihood;
} cpecan_hmm;

/* The fields of PairwiseAlignmentParameters the DP reads (inc/pairwiseAligner.h:28-41). */
typedef struct cpecan_params {
    double threshold;
    int64_t minDiagsBetweenTraceBack;
    int64_t traceBackDiagonals;
    int64_t diagonalExpansion;
    int64_t splitMatrixBiggerThanThis;
    int32_t dynamicAnchorExpansion;
    int32_t reserved;
} cpecan_params;

typedef struct cpecan_batch cpecan_batch;

/* Per-run statistics (for bench.py and the roofline line). */
typedef struct cpecan_stats {
    int64_t problems;      /* alignment problems added */
    int64_t regions;       /* DP sub-problems after splitting by large gaps */
    int64_t cells;         /* sum of band widths over all regions == the metric's unit */
    int64_t diagonals;     /* sum of (lX+lY+1) over regions */
    int64_t pairs;         /* triples emitted (after the last run) */
    int64_t deviceBytes;   /* device memory held by the batch */
    double kernelMs;       /* HIP-event time of the last run's DP kernel launch(es) */
    double h2dMs, d2hMs;   /* last upload / download */
    int32_t launches;      /* kernel launches in the last run (1 unless output overflow forced a re-run) */
    int32_t wavesPerLaunch;
    /* How the widest size class of the batch runs (CPECAN_FORM_*): one wave per region, or its tracebacks as queue items
     * of a second launch / of the same launch; | CPECAN_FORM_ABS: sweeps over absolute positions.  The split forms keep
     * the forward values of whole regions (68 GB at BASELINE config B) and are only taken while the batch stays under
     * CPECAN_SPLIT_BUDGET_FRAC (default 0.45) of the device's memory, so that two pipelined batches fit. */
    int32_t launchForm;
    int32_t reserved;
} cpecan_stats;

enum { CPECAN_FORM_WHOLE = 0, CPECAN_FORM_SPLIT = 1, CPECAN_FORM_FUSED = 2, CPECAN_FORM_ABS = 4 };

/* ---- model / parameter helpers (stateMachine.c / pairwiseAligner.c defaults) ---- */
int cpecan_model_default(cpecan_model *m, int32_t type);          /* stateMachine5/3_construct, stateMachine.c:482,716 */
int cpecan_model_from_hmm(cpecan_model *m, const cpecan_hmm *h);  /* hmm_getStateMachine, stateMachine.c:797 */
int cpecan_hmm_init(cpecan_hmm *h, int32_t type, double pseudoExpectation); /* hmm_constructEmpty :23 */
int cpecan_hmm_normalise(cpecan_hmm *h);                          /* hmm_normalise :88 */
int cpecan_hmm_write(const cpecan_hmm *h, const char *path);      /* hmm_write :133 */
int cpecan_hmm_load(cpecan_hmm *h, const char *path);             /* hmm_loadFromFile :145 */
int cpecan_params_default(cpecan_params *p);                      /* pairwiseAlignmentBandingParameters_construct :1334 */

/* ---- integer geometry, exported because the reference exports and unit-tests it ---- */
/* band_construct / band_constructDynamic (pairwiseAligner.c:128-234): out[3*d..] = xay,xmyL,xmyR for d in 0..lX+lY. */
int cpecan_band(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY, int64_t expansion, int dynamic,
                int64_t *out);
/* getSplitPoints (pairwiseAligner.c:1230): out holds 4*(nAnchors+2) values; returns the count (>=0) or <0. */
int64_t cpecan_split_points(const int64_t *anchors, int64_t nAnchors, int64_t lX, int64_t lY, int64_t maxMatrixSize,
                            int raggedLeft, int raggedRight, int64_t *out);

/* Anchors from an existing pairwise alignment, as cPecanRealign builds them (cPecanRealign.c:525-529):
 * convertPairwiseForwardStrandAlignmentToAnchorPairs (pairwiseAligner.c:979-1003) -- every column of a match operation
 * except `trim` columns at either end of the operation becomes an anchor (x, y, expansion) -- followed, when sX and sY
 * are given, by the exact-match filter (matchFn, cPecanRealign.c:277-281: keep x,y with equal letters, case-insensitive,
 * not N).  ops: nOps pairs (type, length), type CPECAN_OP_MATCH / _INDEL_X (consumes X only) / _INDEL_Y (Y only);
 * start1/start2: first X/Y coordinate of the alignment.  anchors receives up to the sum of the match lengths triples.
 * Returns the number of anchors, or < 0. */
enum { CPECAN_OP_MATCH = 0, CPECAN_OP_INDEL_X = 1, CPECAN_OP_INDEL_Y = 2 };
int64_t cpecan_anchors_from_alignment(const int64_t *ops, int64_t nOps, int64_t start1, int64_t start2, int64_t trim,
                                      int64_t expansion, const char *sX, int64_t lX, const char *sY, int64_t lY,
                                      int64_t *anchors);
/* The same anchors as runs, for cpecan_batch_add_many_runs: the kept columns that follow each other on a matrix diagonal
 * -- a mismatched column or the end of the operation ends a run -- as quadruples (x, y, length, expansion).  runs receives
 * at most cap quadruples; returns the number of runs (which may exceed cap), or < 0. */
int64_t cpecan_anchor_runs_from_alignment(const int64_t *ops, int64_t nOps, int64_t start1, int64_t start2, int64_t trim,
                                          int64_t expansion, const char *sX, int64_t lX, const char *sY, int64_t lY,
                                          int64_t *runs, int64_t cap);

/* filterToRemoveOverlap (pairwiseAligner.c:1095-1135), the step between a sorted list of blast / lastz pairs and an
 * anchor list: a pair (x, y, expansion) is kept when every earlier pair is strictly smaller and every later pair strictly
 * larger in both coordinates.  pairs: n triples sorted by x, then y; out: room for n triples.  Returns the number kept. */
int64_t cpecan_filter_to_remove_overlap(const int64_t *pairs, int64_t n, int64_t *out);

/* The reference's cell-level primitives (inc/pairwiseAligner.h:186-237: cell_calculateForward / Backward,
 * diagonalCalculationForward / Backward, the posterior of :683-685), which its unit tests link, evaluated on the caller's
 * current device: `n` operations applied in order to the cells held in `cells` (nDoubles doubles, changed in place).
 * An operation names its cells by offsets in doubles, -1 = NULL (the reference's out-of-band neighbour).
 * mode 0: forward, current[to] = logAdd(current[to], from[..] + (eP + tP)) over the transition list
 * (impl/pairwiseAligner.c:382-385; impl/stateMachine.c:450-480, :689-714); mode 1: backward, the same list scattered into
 * the neighbours (:392-395); mode 2: cells[upper] = exp(cells[cur] + cells[lower] - total).  Not a hot-path entry point:
 * one lane, blocking copies. */
typedef struct cpecan_cell_op {
    int32_t cur, lower, middle, upper;
    int32_t cX, cY; /* symbols 0..4 (a c g t n) of the current cell */
} cpecan_cell_op;
enum { CPECAN_CELLS_FORWARD = 0, CPECAN_CELLS_BACKWARD = 1, CPECAN_CELLS_POSTERIOR = 2 };
int cpecan_ref_cells(const cpecan_model *model, int mode, const cpecan_cell_op *ops, int64_t n, double *cells,
                     int64_t nDoubles, double total);

/* ---- device ---- */
int cpecan_device_count(void);
/* The calling thread's current HIP device (hipGetDevice).  Device rule of this library: a batch works on the device it
 * was created for and every entry point leaves the caller's current device as it found it; the single-problem entry
 * points (cpecan_get_aligned_pairs_using_anchors ... and the reference-named layer in cpecan_dropin.h) run on the
 * caller's CURRENT device, so in a one-process-per-GPU job they follow hipSetDevice / torch.cuda.set_device. */
int cpecan_current_device(void);
const char *cpecan_last_error(void);
/* Device and pinned-host blocks of destroyed batches are recycled, not freed (hipFree waits for the whole device): up to
 * CPECAN_CACHE_MB of idle device blocks per device (default: half of the device's memory) and CPECAN_HOST_CACHE_MB of
 * idle host blocks (default 16 GiB).  The last live batch to leave a device already gives back every idle block above
 * CPECAN_CACHE_KEEP_MB (default 256); this call gives back ALL idle blocks of `device` (-1: every device) and all idle
 * host blocks, for a process that shares the GPU with another allocator (torch, RCCL).  Returns the bytes released. */
int64_t cpecan_cache_trim(int device);

/* ---- batch API ---- */
/* Creates an empty batch bound to HIP device `device`. `emit` selects the emitter for every problem. */
int cpecan_batch_create(cpecan_batch **out, const cpecan_model *model, const cpecan_params *params, int emit,
                        int device);
void cpecan_batch_destroy(cpecan_batch *b);

/* Adds one alignment problem: same inputs as getAlignedPairsUsingAnchors (pairwiseAligner.c:1431):
 * sequences (need not be NUL-terminated; any byte that is not ACGTacgt is N), anchors as nAnchors
 * triples (x, y, expansion) strictly increasing in x and y, and the two ragged-end flags.
 * The problem is cut into DP regions exactly as pairwiseAligner.c:1273-1326 does. Returns the problem index. */
int64_t cpecan_batch_add(cpecan_batch *b, const char *sX, int64_t lX, const char *sY, int64_t lY,
                         const int64_t *anchors, int64_t nAnchors, int raggedLeft, int raggedRight);

/* The same for n problems at once: they are cut, converted and copied in parallel.  Returns the index of the first. */
typedef struct cpecan_problem {
    const char *sX;
    int64_t lX;
    const char *sY;
    int64_t lY;
    const int64_t *anchors; /* nAnchors triples (x, y, expansion) */
    int64_t nAnchors;
    int32_t raggedLeft, raggedRight;
} cpecan_problem;
int64_t cpecan_batch_add_many(cpecan_batch *b, const cpecan_problem *problems, int64_t n);

/* The same with the anchors as RUNS: run (x, y, length, expansion) stands for the `length` anchors (x + i, y + i, expansion),
 * i = 0 .. length - 1.  This is what the realign flow holds before it makes one anchor per aligned column out of the match
 * operations of a cigar (convertPairwiseForwardStrandAlignmentToAnchorPairs, pairwiseAligner.c:979-1003) and drops the
 * mismatched columns (cPecanRealign.c:525-529): a batch of BASELINE config 4 is 1.4 GB of per-column triples and 0.13 GB of
 * runs.  Runs strictly increase in x and y from one to the next; results are those of cpecan_batch_add_many on the
 * expanded anchors, bit for bit.  Returns the index of the first problem. */
typedef struct cpecan_problem_runs {
    const char *sX;
    int64_t lX;
    const char *sY;
    int64_t lY;
    const int64_t *runs; /* nRuns quadruples (x, y, length, expansion) */
    int64_t nRuns;
    int32_t raggedLeft, raggedRight;
} cpecan_problem_runs;
int64_t cpecan_batch_add_many_runs(cpecan_batch *b, const cpecan_problem_runs *problems, int64_t n);
/* Run-length form of an anchor list: out receives at most cap quadruples; returns the number of runs the list has (which
 * may exceed cap: nothing beyond cap is written), or < 0.  Consecutive anchors join a run when both coordinates step by
 * one and the expansion is the same. */
int64_t cpecan_anchor_runs(const int64_t *anchors, int64_t nAnchors, int64_t *out, int64_t cap);

/* Freezes the batch: builds band tables and traceback schedules on the host, allocates device
 * memory and copies the packed inputs to the GPU. */
int cpecan_batch_upload(cpecan_batch *b);

/* Launches the DP on `stream` (a hipStream_t passed as void*, NULL = the null stream) with inputs
 * already resident; asynchronous. May be called repeatedly (bench). */
int cpecan_batch_run(cpecan_batch *b, void *stream);

/* Waits for the run, copies results to the host and orders them as the reference's lists. */
int cpecan_batch_download(cpecan_batch *b);
/* The same on a helper thread of the batch's own: _begin returns at once, _end waits for the helper and returns what
 * cpecan_batch_download would have returned.  Between the two the caller may do anything that does not touch THIS batch
 * -- typically pack, plan and upload the next one (a stream of batches).  One download at a
 * time per batch; cpecan_batch_destroy waits for a download that was begun and never ended. */
int cpecan_batch_download_begin(cpecan_batch *b);
int cpecan_batch_download_end(cpecan_batch *b);

/* Results for problem i after download. Triples are (score, x, y) int32 in the reference's list order
 * (per region: traceback segments descending, diagonals ascending, x-y descending).
 * which: 0 = aligned pairs, 1 = gapX pairs, 2 = gapY pairs (1,2 only for CPECAN_EMIT_INDEL). */
int cpecan_batch_result(const cpecan_batch *b, int64_t problem, int which, const int32_t **triples, int64_t *n);

/* For CPECAN_EMIT_EXPECT: adds the batch's expectation counts into *acc (transitions, emissions, likelihood),
 * like getExpectationsUsingAnchors (pairwiseAligner.c:1500) called once per problem on the same Hmm. */
int cpecan_batch_expectations(const cpecan_batch *b, cpecan_hmm *acc);

/* For CPECAN_EMIT_FORWARD: total forward log-probability of problem i (computeForwardProbability, pairwiseAligner.c:936).
 * Problems of a FORWARD batch are never split into regions and always use the static band (:894). */
int cpecan_batch_forward_prob(const cpecan_batch *b, int64_t problem, double *logProb);

int cpecan_batch_stats(const cpecan_batch *b, cpecan_stats *s);

/* ---- consumers of the posterior lists, run on the device between the sweep and the download ----
 * REWEIGHT   reweightAlignedPairs2 (pairwiseAligner.c:1550) on every problem's aligned pairs (list 0), in place
 * MEA        getMaximalExpectedAccuracyPairwiseAlignment (:1628) from lists 0..2 (needs CPECAN_EMIT_INDEL); the alignment
 *            is result list 3; gapGamma is used as the float of PairwiseAlignmentParameters
 * LEFT_SHIFT leftShiftAlignment (:1726) of the MEA alignment (list 3 then holds the shifted alignment): MEA | LEFT_SHIFT
 *            == getShiftedMEAAlignment (:1767)
 * ORDERED    filterPairwiseAlignmentToMakePairsOrdered (impl/multipleAligner.c:945) on list 0 (after REWEIGHT, if set --
 *            REWEIGHT | ORDERED is the realign step of cPecanRealign.c:552-553): the heaviest chain of the pairs whose
 *            weight / PAIR_ALIGNMENT_PROB_1 reaches matchGamma, as result list 3, in the reference's output order
 *            (reverse input order).  The reference adds st_random() * 0.00001 to every weight before it compares them
 *            (multipleAligner.c:145); that jitter is left out here, so chains whose weights tie within it may differ.
 * REWEIGHT and MEA are alternatives in the reference's callers and exclude each other here; so do ORDERED and MEA. */
enum { CPECAN_POST_REWEIGHT = 1, CPECAN_POST_MEA = 2, CPECAN_POST_LEFT_SHIFT = 4, CPECAN_POST_ORDERED = 8 };
/* Selects what cpecan_batch_download does to the lists before they leave the device. */
int cpecan_batch_set_post(cpecan_batch *b, int flags, double gapGamma);
/* matchGamma of ORDERED (default 0.85f, cPecanRealign.c:355); a float, widened to double as the reference passes it. */
int cpecan_batch_set_match_gamma(cpecan_batch *b, float matchGamma);
/* After download: scoreByPosteriorProbability (:1587) and scoreByPosteriorProbabilityIgnoringGaps (:1591) of the final
 * list -- list 0 as downloaded (reweighted if REWEIGHT was set), or the ordered alignment with ORDERED, as
 * cPecanRealign.c:556-563 scores it -- and the MEA alignment score (0 without MEA). NULL = not wanted. */
int cpecan_batch_scores(const cpecan_batch *b, int64_t problem, double *byPosterior, double *byPosteriorIgnoringGaps,
                        double *meaScore);
/* scoreByIdentity (:1572) and scoreByIdentityIgnoringGaps (:1577) of the same final list; needs a consumer stage to
 * have been selected (any flag). */
int cpecan_batch_identity_scores(const cpecan_batch *b, int64_t problem, double *byIdentity, double *byIdentityIgnoringGaps);

/* Debug / test hook: for single-region problem i, copies the per-cell forward+backward match sums
 * (fb[cell] = F.match + B.match at emit time) and the total log-probability used for each diagonal.
 * Buffers must hold `cells` and `diagonals` doubles; needs the batch to have been created with
 * cpecan_batch_set_debug(b,1) before upload. */
int cpecan_batch_set_debug(cpecan_batch *b, int on);
int cpecan_batch_debug_fetch(const cpecan_batch *b, int64_t problem, double *fbMatch, int64_t cells,
                             double *totalUsed, int64_t diagonals);

/* ---- single-problem convenience (a batch of one) ---- */
/* getAlignedPairsUsingAnchors (pairwiseAligner.c:1431): *triples is malloc'd (free with cpecan_free). */
int cpecan_get_aligned_pairs_using_anchors(const cpecan_model *m, const char *sX, const char *sY,
                                           const int64_t *anchors, int64_t nAnchors, const cpecan_params *p,
                                           int raggedLeft, int raggedRight, int32_t **triples, int64_t *n);
/* getAlignedPairsWithIndelsUsingAnchors (pairwiseAligner.c:1451): three malloc'd triple lists; gap lists may hold -1
 * as the coordinate of the sequence that is gapped. */
int cpecan_get_aligned_pairs_with_indels_using_anchors(const cpecan_model *m, const char *sX, const char *sY,
                                                       const int64_t *anchors, int64_t nAnchors, const cpecan_params *p,
                                                       int raggedLeft, int raggedRight, int32_t **match, int64_t *nMatch,
                                                       int32_t **gapX, int64_t *nGapX, int32_t **gapY, int64_t *nGapY);
/* computeForwardProbability (pairwiseAligner.c:936). */
int cpecan_compute_forward_probability(const cpecan_model *m, const char *sX, const char *sY, const int64_t *anchors,
                                       int64_t nAnchors, const cpecan_params *p, int raggedLeft, int raggedRight,
                                       double *logProb);
/* ---- the consumers on lists held by the caller (one problem; uploaded, processed on the GPU, copied back) ---- */
/* reweightAlignedPairs2 (:1550): triples (score, x, y) rewritten in place. */
int cpecan_reweight_aligned_pairs(int32_t *triples, int64_t n, int64_t lX, int64_t lY, double gapGamma);
/* scoreByPosteriorProbability / ...IgnoringGaps (:1587-1597). */
int cpecan_posterior_scores(const int32_t *triples, int64_t n, int64_t lX, int64_t lY, double *byPosterior,
                            double *byPosteriorIgnoringGaps);
/* scoreByIdentity / scoreByIdentityIgnoringGaps (:1572-1580). */
int cpecan_identity_scores(const int32_t *triples, int64_t n, const char *sX, const char *sY, double *byIdentity,
                           double *byIdentityIgnoringGaps);
/* filterPairwiseAlignmentToMakePairsOrdered (impl/multipleAligner.c:945), without the reference's random jitter (see
 * CPECAN_POST_ORDERED); the pairs must be distinct cells; *out is malloc'd (cpecan_free). */
int cpecan_filter_pairs_ordered(const int32_t *pairs, int64_t n, int64_t lX, int64_t lY, float matchGamma, int32_t **out,
                                int64_t *nOut);
/* getMaximalExpectedAccuracyPairwiseAlignment (:1628): *out is malloc'd (cpecan_free). */
int cpecan_mea_alignment(const int32_t *pairs, int64_t n, const int32_t *gapX, int64_t nGapX, const int32_t *gapY,
                         int64_t nGapY, int64_t lX, int64_t lY, float gapGamma, int32_t **out, int64_t *nOut,
                         double *alignmentScore);
/* leftShiftAlignment (:1726): *out is malloc'd (cpecan_free). */
int cpecan_left_shift_alignment(const int32_t *pairs, int64_t n, const char *sX, const char *sY, int32_t **out,
                                int64_t *nOut);
/* getShiftedMEAAlignment (:1767). */
int cpecan_get_shifted_mea_alignment(const cpecan_model *m, const char *sX, const char *sY, const int64_t *anchors,
                                     int64_t nAnchors, const cpecan_params *p, float gapGamma, int raggedLeft,
                                     int raggedRight, int32_t **out, int64_t *nOut, double *alignmentScore);
void cpecan_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* CPECAN_HIP_H_ */
