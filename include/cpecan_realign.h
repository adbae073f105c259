/*
 * cpecan_realign.h -- the batch front end of cPecanRealign on top of cpecan_hip.h: cigar and fasta
 * text in, realigned cigars (or expectation counts) out, every alignment of a call in ONE GPU batch.
 *
 * Reference: cPecanRealign.c:354-624 (main loop), :49-96 (convertAlignedPairsToPairwiseAlignment), :98-230 (hasLongIndel,
 * splitPairwiseAlignment), :232-277 (rebase, getSubSequence, addToSequencesHash), :314-348 (scoreAnchorPairs).
 * Text formats: cigarRead / cigarWrite / fastaReadToFunction / stString_reverseComplementString live in sonLib
 * (benedictpaten/sonLib, a sibling checkout the reference's include.mk points at; not vendored in the reference, no pinned
 * version).  Their published formats are restated here:
 *   cigar line  "cigar: <contig2> <start2> <end2> <strand2> <contig1> <start1> <end1> <strand1> <score>( <op> <length>)*"
 *               op M = match, D = bases of contig1 (X) only, I = bases of contig2 (Y) only; strand '+' or '-'; on '-' the
 *               start is larger than the end; score printed with %f
 *   fasta       '>' header lines, sequence lines concatenated with white space removed; the key of a sequence is the first
 *               white-space delimited token of its header (cPecanRealign.c:245-275)
 */
#ifndef CPECAN_REALIGN_H_
#define CPECAN_REALIGN_H_

#include "cpecan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* struct PairwiseAlignment of sonLib, flattened.  contig1 is sequence X of the aligner, contig2 is Y. */
typedef struct cpecan_cigar {
    char *contig1, *contig2; /* malloc'd */
    int64_t start1, end1, start2, end2;
    int32_t strand1, strand2; /* 1 = '+' */
    double score;
    int64_t nOps;
    int64_t *ops; /* nOps pairs (CPECAN_OP_*, length), malloc'd */
} cpecan_cigar;

/* cigarRead on one line of text.  Returns CPECAN_EINVAL for a line that is not a cigar or whose operations do not add up
 * to its coordinates (checkPairwiseAlignment). */
int cpecan_cigar_parse(const char *line, cpecan_cigar *out);
/* cigarWrite(fileHandle, pA, 0) without the newline.  Returns the length of the text; writes at most cap bytes incl. NUL. */
int64_t cpecan_cigar_format(const cpecan_cigar *c, char *buf, int64_t cap);
void cpecan_cigar_clear(cpecan_cigar *c);
/* Frees an array returned by cpecan_realigner_realign or cpecan_cigar_split. */
void cpecan_cigars_free(cpecan_cigar *cigars, int64_t n);
/* convertAlignedPairsToPairwiseAlignment (cPecanRealign.c:49-96): xy holds n pairs (x, y) in increasing order; the result
 * spans [0, length1) x [0, length2) on the forward strands, unaligned ends included as leading / trailing indels. */
int cpecan_cigar_from_aligned_pairs(const char *contig1, const char *contig2, double score, int64_t length1, int64_t length2,
                                    const int64_t *xy, int64_t n, cpecan_cigar *out);
/* splitPairwiseAlignment (cPecanRealign.c:117-230): cuts c at every run of indels longer than maxIndelLength; the runs
 * that are cut and any indels at either end are dropped.  *out: malloc'd array of *nOut cigars (cpecan_cigars_free). */
int cpecan_cigar_split(const cpecan_cigar *c, int64_t maxIndelLength, cpecan_cigar **out, int64_t *nOut);

/* The options of cPecanRealign's command line with its defaults (cPecanRealign.c:354-370). */
typedef struct cpecan_realign_options {
    cpecan_params params;           /* diagonalExpansion 4 (-r), splitMatrixBiggerThanThis 10 (-o takes the square root) */
    int64_t constraintDiagonalTrim; /* -t, 0 */
    float gapGamma;                 /* -l, 0.5 */
    float matchGamma;               /* -L, 0.85 */
    int32_t rescoreOriginalAlignment;           /* -x */
    int32_t rescoreByIdentity;                  /* -i */
    int32_t rescoreByPosteriorProb;             /* -j */
    int32_t rescoreByIdentityIgnoringGaps;      /* -k */
    int32_t rescoreByPosteriorProbIgnoringGaps; /* -m */
    int64_t splitIndelsLongerThanThis;          /* -s, -1 = do not split */
} cpecan_realign_options;
void cpecan_realign_options_default(cpecan_realign_options *o);

typedef struct cpecan_realigner cpecan_realigner;
int cpecan_realigner_create(cpecan_realigner **out, const cpecan_model *model, const cpecan_realign_options *o, int device);
void cpecan_realigner_destroy(cpecan_realigner *r);
/* addToSequencesHash (cPecanRealign.c:245-275): a repeated key keeps the longer sequence. */
int cpecan_realigner_add_sequence(cpecan_realigner *r, const char *header, const char *seq, int64_t length);
/* fastaReadToFunction(file, addToSequencesHash).  Returns the number of records read, or < 0. */
int64_t cpecan_realigner_read_fasta(cpecan_realigner *r, const char *path);
/* Tab separated "x y probability" files as --outputPosteriorProbs (the final pairs) and --outputAllPosteriorProbs (every
 * pair of the banded alignment).  The reference reopens the file with "w" for every cigar, so what is left is the last
 * cigar's pairs; that is what is written here.  NULL = none. */
int cpecan_realigner_set_posterior_files(cpecan_realigner *r, const char *finalPairsPath, const char *allPairsPath);

/* Several GPUs from ONE process.  The reference fans a realignment out as one cPecanRealign process per shard
 * of the cigar file and sums the shards' expectation files (cPecanEm.py:168-188); with a device list the cigars of every
 * realign / expectations call are cut into nDevices contiguous shards of about equal band cells
 * (cpecan_realign_shard_bounds), shard k runs as its own batch on devices[k] from a host thread of its own, and the
 * results are joined in input order -- expectation counts summed on the host in shard order.  A device may be listed more
 * than once (two shards on one GPU); nDevices 0 or 1: one batch on the device given to cpecan_realigner_create. */
int cpecan_realigner_set_devices(cpecan_realigner *r, const int *devices, int nDevices);
/* bounds[0..nShards]: shard k is cigars [bounds[k], bounds[k+1]); a cigar costs (|end1 - start1| + |end2 - start2| + 1) *
 * (diagonalExpansion + 1), shard k ends at the first cigar where the running cost reaches k / nShards of the total. */
int cpecan_realign_shard_bounds(const cpecan_cigar *in, int64_t n, int64_t diagonalExpansion, int nShards, int64_t *bounds);

/* The realign loop (cPecanRealign.c:509-600) over n cigars as one batch.  *out: malloc'd array of *nOut cigars in input
 * order (more than n when splitIndelsLongerThanThis cuts some), to be released with cpecan_cigars_free. */
int cpecan_realigner_realign(cpecan_realigner *r, const cpecan_cigar *in, int64_t n, cpecan_cigar **out, int64_t *nOut);
/* --outputExpectations (cPecanRealign.c:530-534): adds the expectation counts of the n alignments to *acc, which the
 * caller made with cpecan_hmm_init(acc, type, 0.000000000001) (:497) and writes with cpecan_hmm_write (:612). */
int cpecan_realigner_expectations(cpecan_realigner *r, const cpecan_cigar *in, int64_t n, cpecan_hmm *acc);

#ifdef __cplusplus
}
#endif
#endif /* CPECAN_REALIGN_H_ */
