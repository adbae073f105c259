/*
 * cpecan_internal.h -- structures shared by the C host code (cpecan_host.c) and the HIP
 * translation unit (cpecan_kernels.hip).  Not part of the public ABI.
 */
#ifndef CPECAN_INTERNAL_H_
#define CPECAN_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>

#include "cpecan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CPK_MAX_STATES 5
#define CPK_SYM_N 4
#define CPK_WAVE 64
#define CPK_REFRESH_PERIOD 10 /* total probability is refreshed every 10th emitted diagonal, pairwiseAligner.c:830 */

/* One anti-diagonal of a region's band, as the kernels read it (one 16-byte scalar load). */
typedef struct {
    int32_t xmyL;    /* smallest x-y on the diagonal */
    int32_t width;   /* number of cells */
    int32_t ringOff; /* first cell of this diagonal inside the region's forward ring (in cells) */
    int32_t cellOff; /* number of band cells on earlier diagonals of the region */
} CpkDiag;

/* One traceback segment (pairwiseAligner.c:791-862): the forward sweep runs up to dTop, then the
 * backward sweep runs from dTop down to tbPrev+1 and diagonals tbPrev+1..tbFrom are emitted. */
typedef struct {
    int32_t tbPrev;   /* tracedBackTo on entry */
    int32_t dTop;     /* diagonal the traceback starts from */
    int32_t tbFrom;   /* tracedBackFrom: highest emitted diagonal */
    int32_t atEnd;    /* dTop == lX+lY */
    int32_t nRefresh; /* number of total-probability refresh points in the segment */
    int32_t outOff;   /* the segment's own part of the region's output slice (per list), used when the tracebacks of a */
    int32_t outCap;   /* region run as separate queue items (split classes): first triple and capacity */
    int32_t emitCells; /* band cells on the emitted diagonals tbPrev+1..tbFrom: no list of the segment can be longer */
} CpkSegment;

/* One traceback work item of a split class: segment `seg` (index within its region) of device region `region`. */
typedef struct {
    int32_t region, seg;
} CpkItem;

/* One DP region (a whole problem, or one rectangle of getSplitPoints). */
typedef struct {
    int64_t seqXOff, seqYOff; /* byte offsets of the padded symbol strings (symbol[0] = N, symbol[i] = base i-1) */
    int64_t diagOff;          /* index of the region's first CpkDiag */
    int64_t segOff;           /* index of the region's first CpkSegment */
    int64_t outOff;           /* first output triple slot of the region (per output list) */
    int64_t dbgCellOff;       /* debug: first cell in the debug fb array */
    int64_t dbgDiagOff;       /* debug: first diagonal in the debug total array */
    int64_t anchorOff;        /* first anchor triple of the region (coordinates relative to the region) */
    int64_t ringBase;         /* split classes: first cell of the region's own forward ring (it holds every segment) */
    int64_t cells;            /* band cells of the region */
    int32_t lX, lY;
    int32_t nSeg;
    int32_t outCap;           /* capacity in triples (per output list) */
    int32_t raggedLeft, raggedRight;
    int32_t maxWidth;
    int32_t ringCap;          /* cells of forward ring the region needs: its longest live span + its widest diagonal */
    int32_t nAnchors;
    int32_t split;            /* set by the device layer: the region's tracebacks run as separate queue items */
    int32_t absOk;            /* planning: the band's edges move by one x-y step per diagonal (absolute-position sweeps allowed) */
} CpkRegion;

/* Model constants as the kernels use them: per-state priors, named transitions and padded emissions. */
typedef struct {
    int32_t nStates;
    int32_t type;
    double start[CPK_MAX_STATES], raggedStart[CPK_MAX_STATES];
    double end[CPK_MAX_STATES], raggedEnd[CPK_MAX_STATES];
    /* transitions, log space */
    double matchContinue;
    double matchFromShortX, matchFromShortY, matchFromLongX, matchFromLongY;
    double shortOpenX, shortOpenY, shortExtendX, shortExtendY, shortSwitchToX, shortSwitchToY;
    double longOpenX, longOpenY, longExtendX, longExtendY;
    /* emissions with the N row/column filled in (stateMachine.c:351-366) */
    double matchEm[25]; /* [cX*5+cY] */
    double gapXEm[5], gapYEm[5];
    double threshold;
} CpkModel;

#define CPK_WIDE_CLASSES 8
/* Per-launch geometry computed on the host. */
typedef struct {
    int32_t nRegions;
    int32_t nStates;
    int32_t emit;
    int32_t maxWidth;     /* widest diagonal in the batch */
    int32_t rollStride;   /* doubles per state row of a rolling buffer (maxWidth + the -inf guard at position 0) */
    int32_t maxRefresh;   /* most refresh points in any segment */
    int32_t useGlobalRoll;/* slow path: rolling buffers and symbol strings stay in global memory (too big for LDS) */
    int32_t seqLdsBytes;  /* fast path: bytes of LDS for the two padded symbol strings of the largest region */
    int32_t debug;
    int32_t fusedSpin;    /* one-launch form: polls of an item for its region's forward values before it gives up (reported, re-run in two launches) */
    int32_t expInSweep;   /* expectation emitter: the events are formed inside the traceback (every diagonal of the class fits one 64-lane group) */
    int32_t reserved0;
    int64_t ringCells;    /* forward ring capacity per slot, in cells */
    int64_t fbCells;      /* posterior-candidate scratch per slot, in cells (most emitted cells of one segment) */
    int64_t refreshCells; /* c/m scratch per slot = maxWidth * maxRefresh (each) */
    int64_t rollDoubles;  /* global rolling buffer per slot, doubles (only when useGlobalRoll) */
    /* Narrow regions come first in the device order, by class: every diagonal at most 8 / 16 / 32 cells wide.  Class k
     * is run by the packed kernel with groups of 8 << k lanes (64 / (8 << k) regions per wave); the fields above then
     * describe the remaining (wide) regions only and these the narrow classes.  nPacked[k] == 0: no launch. */
    int32_t nPacked[3];
    int32_t pMaxRefresh[3];
    int64_t pRingCells[3], pFbCells[3];
    /* The wide regions follow, again by class: widest diagonal at most 128 / 192 / 256 / 384 / 512 cells, wider but
     * still inside the 64 KiB LDS budget, or wider still (global-memory rolling buffers).  Every class is one launch of the sweep kernel with LDS and per-wave scratch sized
     * for ITS largest region, so that one very wide region neither takes the LDS that decides how many waves the others
     * get nor multiplies everybody's scratch.  The scalar fields above are filled per launch from these. */
    int32_t nWide[CPK_WIDE_CLASSES];
    int32_t wMaxWidth[CPK_WIDE_CLASSES], wMaxRefresh[CPK_WIDE_CLASSES], wSeqLdsBytes[CPK_WIDE_CLASSES];
    int32_t wWinLdsBytes[CPK_WIDE_CLASSES]; /* the same for a class that stages symbol windows per traceback segment (absolute positions) */
    int64_t wRingCells[CPK_WIDE_CLASSES], wFbCells[CPK_WIDE_CLASSES];
} CpkGeometry;

/* One run of consecutive triples to move into the compact, list-ordered result buffer: the triples of one traceback
 * segment of one region, shifted by the region's offset inside its problem (pairwiseAligner.c:1411-1418). */
typedef struct {
    int64_t src;     /* first triple in the kernel's output (list offset included) */
    int64_t dst;     /* first triple in the compact buffer */
    int32_t len;
    int32_t dx, dy;  /* added to x and y */
    int32_t pad;
} CpkChunk;

/* Device-side mirror of a frozen batch; owned by the HIP TU. */
typedef struct CpkDevice CpkDevice;

/* One operation of cpk_ref_cells (cpk_cells.inl): offsets, in doubles, of the cells in the flat buffer; -1 = NULL.
 * Layout shared with include/cpecan_hip.h's cpecan_cell_op. */
typedef struct {
    int32_t cur, lower, middle, upper;
    int32_t cX, cY;
} CpkCellOp;

/* ---- implemented in cpecan_kernels.hip ---- */
/* The reference's cell-level primitives on the current device: `n` operations applied in order by one lane to the
 * `nDoubles` doubles of buf (copied up, changed in place, copied back).  mode 0 forward, 1 backward, 2 posterior. */
int cpk_ref_cells(int device, const CpkModel *model, int mode, const CpkCellOp *ops, int64_t n, double *buf, int64_t nDoubles,
                  double total);
/* Host blocks of a batch: pinned and recycled when a HIP device is present, plain malloc below 256 KB or without a GPU. */
void *cpk_host_alloc(size_t bytes);
void cpk_host_free(void *p);
void *cpk_host_grow(void *p, size_t usedBytes, size_t newBytes); /* NULL on failure: p is still valid then */
int cpk_device_count(void);
int64_t cpk_cache_trim(int device); /* idle device blocks of `device` (-1: all) and idle host blocks -> driver / OS; bytes */
int cpk_current_device(void); /* the calling thread's current HIP device (0 when there is none) */
const char *cpk_last_error(void);
int cpk_device_create(CpkDevice **out, int device);
void cpk_device_destroy(CpkDevice *dev);
/* Copies the packed inputs to the GPU and sizes every scratch buffer. */
/* The per-diagonal table (nDiags entries) is built on the device from the anchors (cpecan_band.inl). */
/* Regions of a class that is split (see CpkItem) get ringCap / ringBase / split set here, in the caller's array. */
int cpk_device_upload(CpkDevice *dev, const CpkGeometry *geo, const CpkModel *model, CpkRegion *regions,
                      const int32_t *anchors /* cpk_anchor_t, cpecan_band.inl */, int anchorStride, int64_t nAnchors,
                      const int32_t *runs /* or: (x, y, length, first anchor) per run, expanded on the device; anchors NULL */, int64_t nRuns,
                      int64_t nDiags,
                      int64_t expansion, int dynamic,
                      const CpkSegment *segs, int64_t nSegs, const uint8_t *symbols, int64_t nSymbolBytes,
                      int64_t outTriplesPerList, int nLists, int64_t dbgCells, int64_t dbgDiags, double *h2dMs);
double cpk_device_h2d_ms(CpkDevice *dev); /* the upload's copy time; waits for the copies */
int cpk_device_update_regions(CpkDevice *dev, const CpkRegion *regions, const CpkSegment *segs, int64_t outTriplesPerList);
int cpk_device_run(CpkDevice *dev, void *stream);
int cpk_device_form(const CpkDevice *dev); /* CPECAN_FORM_* of the batch's last (widest) size class */
/* Once more on the stream of the last run (after an output overflow); kernel times of a batch's launches add up. */
int cpk_device_rerun(CpkDevice *dev);
/* Blocks until the run is complete and copies back the per-region counts and per-segment start offsets (and the
 * expectation sums / forward probabilities). The triples stay on the device: see cpk_device_gather. */
int cpk_device_download(CpkDevice *dev, int32_t *counts /* [nLists][nRegions] */,
                        int32_t *segStarts /* [nLists][nSegsTotal] */, int32_t *segCounts /* [nLists][nSegsTotal]: split regions */,
                        double *expect /* [106] */, double *kernelMs, double *d2hMs);
/* Moves the chunks into one compact buffer on the device (reference list order, region offsets applied). */
int cpk_device_gather(CpkDevice *dev, const CpkChunk *chunks, int64_t nChunks, int64_t total);
/* Copies the compact buffer -- `total` triples, nothing else -- to hostOut. */
int cpk_device_fetch(CpkDevice *dev, int32_t *hostOut, int64_t total, double *d2hMs);

/* ---- consumers of the posterior lists, one descriptor per alignment problem ---- */
typedef struct {
    int64_t off[3];          /* first triple of the aligned / gapX / gapY list in the triple buffer */
    int32_t n[3];            /* their lengths */
    int32_t lX, lY;
    int32_t pad;
    int64_t seqOff;          /* first of lX + lY scratch slots (unaligned mass, cumulative gap mass) */
    int64_t chainOff;        /* first of n[0] + 1 chain-DP slots */
    int64_t charX, charY;    /* raw upper-case sequences (left shift) */
    int64_t meaOut;          /* first of n[0] output triples */
    int64_t shiftOut;        /* first of n[0] + min(lX, lY) + 1 output triples */
} CpkPostProblem;

#define CPK_POST_SCORES 5
typedef struct {
    int flags;               /* CPECAN_POST_* */
    double gapGamma;
    double matchGamma;       /* ORDERED: the float of cPecanRealign.c:355 widened to double */
    int64_t nProblems;
    const CpkPostProblem *problems;
    const uint8_t *chars;    /* host: raw upper-case sequences, or NULL (then no left shift, no identity scores) */
    int64_t nChars;
    int64_t seqSlots, chainSlots, meaCap, shiftCap; /* totals over the problems */
    /* outputs (host) */
    double *scores;          /* [nProblems][CPK_POST_SCORES]: byPosterior, byPosteriorIgnoringGaps, MEA alignment score,
                              * byIdentity, byIdentityIgnoringGaps (the last two only when chars are given) */
    int32_t *counts;         /* [nProblems][2]: MEA or ordered pairs, left-shifted pairs */
    int32_t *mea;            /* [meaCap*3] or NULL */
    int32_t *shift;          /* [shiftCap*3] or NULL */
} CpkPostJob;

/* Runs the job on the batch's compact result buffer (after cpk_device_gather, before cpk_device_fetch). */
int cpk_device_post(CpkDevice *dev, const CpkPostJob *job);
/* Runs the job on lists given by the host: `triples` (total*3 int32) is uploaded, processed and copied back. */
int cpk_post_lists(int device, int32_t *triples, int64_t total, const CpkPostJob *job);
int cpk_device_debug_fetch(CpkDevice *dev, double *fb, int64_t cells, double *totals, int64_t diags);
int64_t cpk_device_bytes(const CpkDevice *dev);
int cpk_device_waves(const CpkDevice *dev);
void cpk_set_error(const char *fmt, ...);
int cpk_host_threads(void); /* threads of the host's parallel loops (cpecan_host.c) */

#ifdef __cplusplus
}
#endif
#endif
