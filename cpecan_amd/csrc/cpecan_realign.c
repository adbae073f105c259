/*
 * cpecan_realign.c -- the batch front end of cPecanRealign over the GPU batch API (include/cpecan_realign.h).
 *
 * Host code only: text formats, coordinate bookkeeping and the per-cigar loop of cPecanRealign.c:509-600, with the
 * aligner calls of that loop (getAlignedPairsUsingAnchors / getExpectationsUsingAnchors, reweightAlignedPairs2,
 * filterPairwiseAlignmentToMakePairsOrdered, the score functions) replaced by ONE cpecan_batch per call.  Nothing here
 * computes a posterior: without a GPU every entry point that needs one fails with CPECAN_ENODEVICE.
 */
#define _POSIX_C_SOURCE 200809L
#include "cpecan_realign.h"

#include <ctype.h>
#include <omp.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cpecan_internal.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------
 * cigars
 * ---------------------------------------------------------------------------------------------- */
void cpecan_cigar_clear(cpecan_cigar *c) {
    if (!c) return;
    free(c->contig1);
    free(c->contig2);
    free(c->ops);
    memset(c, 0, sizeof *c);
}

void cpecan_cigars_free(cpecan_cigar *cigars, int64_t n) {
    if (!cigars) return;
    for (int64_t i = 0; i < n; i++) cpecan_cigar_clear(&cigars[i]);
    free(cigars);
}

static char *copy_span(const char *s, size_t n) {
    char *t = malloc(n + 1);
    if (t) {
        memcpy(t, s, n);
        t[n] = 0;
    }
    return t;
}

static const char *skip_space(const char *s) {
    while (*s && isspace((unsigned char)*s)) s++;
    return s;
}

/* next white-space delimited token; returns its length, 0 at the end of the line */
static size_t next_token(const char **s, const char **tok) {
    const char *p = skip_space(*s);
    *tok = p;
    while (*p && !isspace((unsigned char)*p)) p++;
    *s = p;
    return (size_t)(p - *tok);
}

static int parse_i64(const char *tok, size_t n, int64_t *v) {
    size_t i = 0;
    const int neg = n > 0 && tok[0] == '-';
    if (neg || (n > 0 && tok[0] == '+')) i++;
    if (i >= n || n - i > 18) return 0; /* 18 digits always fit */
    int64_t acc = 0;
    for (; i < n; i++) {
        if (tok[i] < '0' || tok[i] > '9') return 0;
        acc = acc * 10 + (tok[i] - '0');
    }
    *v = neg ? -acc : acc;
    return 1;
}

/* checkPairwiseAlignment (sonLib): strands and coordinates agree, the operations add up to the two spans */
static int cigar_consistent(const cpecan_cigar *c) {
    int64_t span1 = 0, span2 = 0;
    for (int64_t i = 0; i < c->nOps; i++) {
        const int64_t type = c->ops[2 * i], len = c->ops[2 * i + 1];
        if (len < 0) return 0;
        if (type != CPECAN_OP_INDEL_Y) span1 += len;
        if (type != CPECAN_OP_INDEL_X) span2 += len;
    }
    const int64_t d1 = c->strand1 ? c->end1 - c->start1 : c->start1 - c->end1;
    const int64_t d2 = c->strand2 ? c->end2 - c->start2 : c->start2 - c->end2;
    return c->start1 >= 0 && c->end1 >= 0 && c->start2 >= 0 && c->end2 >= 0 && d1 == span1 && d2 == span2;
}

int cpecan_cigar_parse(const char *line, cpecan_cigar *out) {
    if (!line || !out) return CPECAN_EINVAL;
    memset(out, 0, sizeof *out);
    const char *s = line, *tok;
    size_t n = next_token(&s, &tok);
    if (n != 6 || strncmp(tok, "cigar:", 6) != 0) {
        cpk_set_error("not a cigar line");
        return CPECAN_EINVAL;
    }
    int ok = 1;
    /* the query (contig2, sequence Y) comes first, then the target (contig1, sequence X) */
    for (int side = 2; side >= 1 && ok; side--) {
        n = next_token(&s, &tok);
        char *name = n ? copy_span(tok, n) : NULL;
        int64_t start = 0, end = 0;
        ok = name != NULL;
        n = next_token(&s, &tok);
        ok = ok && parse_i64(tok, n, &start);
        n = next_token(&s, &tok);
        ok = ok && parse_i64(tok, n, &end);
        n = next_token(&s, &tok);
        ok = ok && n == 1 && (tok[0] == '+' || tok[0] == '-');
        const int32_t strand = ok && tok[0] == '+';
        if (side == 2) {
            out->contig2 = name;
            out->start2 = start;
            out->end2 = end;
            out->strand2 = strand;
        } else {
            out->contig1 = name;
            out->start1 = start;
            out->end1 = end;
            out->strand1 = strand;
        }
    }
    if (ok) {
        n = next_token(&s, &tok);
        char buf[64];
        ok = n > 0 && n < sizeof buf;
        if (ok) {
            memcpy(buf, tok, n);
            buf[n] = 0;
            char *end;
            out->score = strtod(buf, &end);
            ok = *end == 0;
        }
    }
    int64_t cap = 0;
    while (ok) {
        n = next_token(&s, &tok);
        if (n == 0) break;
        int64_t type = -1, len = 0;
        if (n == 1 && tok[0] == 'M') type = CPECAN_OP_MATCH;
        if (n == 1 && tok[0] == 'D') type = CPECAN_OP_INDEL_X;
        if (n == 1 && tok[0] == 'I') type = CPECAN_OP_INDEL_Y;
        n = next_token(&s, &tok);
        ok = type >= 0 && parse_i64(tok, n, &len);
        if (ok && out->nOps == cap) {
            cap = cap ? 2 * cap : 16;
            int64_t *grown = realloc(out->ops, sizeof(int64_t) * 2 * (size_t)cap);
            ok = grown != NULL;
            if (ok) out->ops = grown;
        }
        if (ok) {
            out->ops[2 * out->nOps] = type;
            out->ops[2 * out->nOps + 1] = len;
            out->nOps++;
        }
    }
    if (ok && !cigar_consistent(out)) {
        cpk_set_error("cigar operations do not add up to its coordinates");
        ok = 0;
    } else if (!ok) {
        cpk_set_error("malformed cigar line");
    }
    if (!ok) {
        cpecan_cigar_clear(out);
        return CPECAN_EINVAL;
    }
    return CPECAN_OK;
}

/* decimal text of v at dst (room for 21 bytes); returns the length */
static int put_i64(char *dst, int64_t v) {
    char tmp[24];
    int n = 0, neg = v < 0;
    uint64_t u = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    int at = 0;
    if (neg) dst[at++] = '-';
    while (n) dst[at++] = tmp[--n];
    return at;
}

int64_t cpecan_cigar_format(const cpecan_cigar *c, char *buf, int64_t cap) {
    if (!c || !c->contig1 || !c->contig2 || cap < 0) return CPECAN_EINVAL;
    static const char opChar[3] = {'M', 'D', 'I'};
    char *dst = cap > 0 ? buf : NULL;
    const int w = snprintf(dst, dst ? (size_t)cap : 0, "cigar: %s %lld %lld %c %s %lld %lld %c %f", c->contig2,
                           (long long)c->start2, (long long)c->end2, c->strand2 ? '+' : '-', c->contig1, (long long)c->start1,
                           (long long)c->end1, c->strand1 ? '+' : '-', c->score);
    if (w < 0) return CPECAN_EINVAL;
    int64_t at = w;
    for (int64_t i = 0; i < c->nOps; i++) { /* " <op> <length>": by hand, an alignment has hundreds of these */
        char piece[32];
        int n = 0;
        piece[n++] = ' ';
        piece[n++] = opChar[c->ops[2 * i]];
        piece[n++] = ' ';
        n += put_i64(piece + n, c->ops[2 * i + 1]);
        if (at + n < cap) memcpy(buf + at, piece, (size_t)n);
        else if (at < cap - 1) memcpy(buf + at, piece, (size_t)(cap - 1 - at));
        at += n;
    }
    if (cap > 0) buf[at < cap ? at : cap - 1] = 0;
    return at;
}

static int cigar_set(cpecan_cigar *c, const char *contig1, int64_t start1, int64_t end1, int strand1, const char *contig2,
                     int64_t start2, int64_t end2, int strand2, double score) {
    memset(c, 0, sizeof *c);
    c->contig1 = copy_span(contig1, strlen(contig1));
    c->contig2 = copy_span(contig2, strlen(contig2));
    c->start1 = start1;
    c->end1 = end1;
    c->strand1 = strand1;
    c->start2 = start2;
    c->end2 = end2;
    c->strand2 = strand2;
    c->score = score;
    return c->contig1 && c->contig2 ? CPECAN_OK : CPECAN_ENOMEM;
}

static int cigar_push(cpecan_cigar *c, int64_t *cap, int64_t type, int64_t len) {
    if (c->nOps == *cap) {
        *cap = *cap ? 2 * *cap : 16;
        int64_t *grown = realloc(c->ops, sizeof(int64_t) * 2 * (size_t)*cap);
        if (!grown) return CPECAN_ENOMEM;
        c->ops = grown;
    }
    c->ops[2 * c->nOps] = type;
    c->ops[2 * c->nOps + 1] = len;
    c->nOps++;
    return CPECAN_OK;
}

/* rebasePairwiseAlignmentCoordinates, cPecanRealign.c:232-243 */
static void rebase(int64_t *start, int64_t *end, int32_t *strand, int64_t shift, int flip) {
    *start += shift;
    *end += shift;
    if (flip) {
        *strand = !*strand;
        const int64_t t = *end;
        *end = *start;
        *start = t;
    }
}

/* One more op of the cigar being built; a match directly after a match extends it (two diagonal runs that touch). */
static int cigar_emit(cpecan_cigar *c, int64_t *cap, int64_t type, int64_t length) {
    if (length <= 0) return CPECAN_OK;
    if (type == CPECAN_OP_MATCH && c->nOps > 0 && c->ops[2 * (c->nOps - 1)] == CPECAN_OP_MATCH) {
        c->ops[2 * (c->nOps - 1) + 1] += length;
        return CPECAN_OK;
    }
    return cigar_push(c, cap, type, length);
}

/* The cigar of a chain of aligned pairs (what convertAlignedPairsToPairwiseAlignment, cPecanRealign.c:49-96, produces):
 * xy holds n (x, y) pairs.  The chain is read as maximal DIAGONAL RUNS -- pairs (x0 + k, y0 + k), k = 0 .. run-1 --
 * each one match op; the bases of X that no op has covered when a run starts become one X-indel, then those of Y one
 * Y-indel, and the same once more for what is left behind the last run.  A pair that does not lie beyond everything
 * covered so far in BOTH sequences is passed over (the reference's input is an ordered chain, where there is none). */
static int cigar_from_pairs(cpecan_cigar *c, const char *contig1, const char *contig2, double score, int64_t length1,
                            int64_t length2, const int64_t *xy, int64_t n) {
    int rc = cigar_set(c, contig1, 0, length1, 1, contig2, 0, length2, 1, score);
    int64_t cap = 0, covered1 = 0, covered2 = 0; /* bases of each sequence that the ops so far account for */
    for (int64_t i = 0; rc == CPECAN_OK && i < n;) {
        const int64_t x0 = xy[2 * i], y0 = xy[2 * i + 1];
        if (x0 < covered1 || y0 < covered2 || x0 >= length1 || y0 >= length2) {
            i++;
            continue;
        }
        int64_t run = 1;
        while (i + run < n && xy[2 * (i + run)] == x0 + run && xy[2 * (i + run) + 1] == y0 + run) run++;
        rc = cigar_emit(c, &cap, CPECAN_OP_INDEL_X, x0 - covered1);
        if (rc == CPECAN_OK) rc = cigar_emit(c, &cap, CPECAN_OP_INDEL_Y, y0 - covered2);
        if (rc == CPECAN_OK) rc = cigar_emit(c, &cap, CPECAN_OP_MATCH, run);
        covered1 = x0 + run;
        covered2 = y0 + run;
        i += run;
    }
    if (rc == CPECAN_OK) rc = cigar_emit(c, &cap, CPECAN_OP_INDEL_X, length1 - covered1);
    if (rc == CPECAN_OK) rc = cigar_emit(c, &cap, CPECAN_OP_INDEL_Y, length2 - covered2);
    return rc;
}

/* splitPairwiseAlignment, cPecanRealign.c:117-230: cuts pA at every run of indels longer than maxIndelLength; the runs
 * that are cut, and any run of indels at either end, are dropped.  Appends to (*out)[*nOut..]. */
static int cigar_split(const cpecan_cigar *pA, int64_t maxIndelLength, cpecan_cigar **out, int64_t *nOut, int64_t *capOut) {
    cpecan_cigar cur;
    int64_t curCap = 0, run = 0, pendFirst = -1; /* pendFirst: first op of the pending indel run (indelOpList) */
    int64_t pos1 = pA->start1, pos2 = pA->start2, curStart1 = pA->start1, curStart2 = pA->start2, curEnd1 = 0, curEnd2 = 0;
    int rc = cigar_set(&cur, pA->contig1, 0, 0, pA->strand1, pA->contig2, 0, 0, pA->strand2, pA->score);
    for (int64_t i = 0; rc == CPECAN_OK && i <= pA->nOps; i++) {
        const int last = i == pA->nOps;
        const int64_t type = last ? CPECAN_OP_MATCH : pA->ops[2 * i], len = last ? 0 : pA->ops[2 * i + 1];
        if (type != CPECAN_OP_MATCH) {
            run += len;
            if (pendFirst < 0) pendFirst = i;
            if (type == CPECAN_OP_INDEL_X) pos1 += pA->strand1 ? len : -len;
            else pos2 += pA->strand2 ? len : -len;
            continue;
        }
        const int flush = last ? cur.nOps != 0 : (run > maxIndelLength && cur.nOps != 0);
        if (flush) { /* close the alignment so far (:140-156, :196-203) */
            if (*nOut == *capOut) {
                *capOut = *capOut ? 2 * *capOut : 16;
                cpecan_cigar *grown = realloc(*out, sizeof(cpecan_cigar) * (size_t)*capOut);
                if (!grown) {
                    rc = CPECAN_ENOMEM;
                    break;
                }
                *out = grown;
            }
            cur.start1 = curStart1;
            cur.end1 = curEnd1;
            cur.start2 = curStart2;
            cur.end2 = curEnd2;
            (*out)[(*nOut)++] = cur;
            curCap = 0;
            memset(&cur, 0, sizeof cur);
            if (last) break;
            rc = cigar_set(&cur, pA->contig1, 0, 0, pA->strand1, pA->contig2, 0, 0, pA->strand2, pA->score);
            if (rc != CPECAN_OK) break;
        }
        if (last) break;
        if (flush || cur.nOps == 0) { /* the pending run is dropped: the next piece starts behind it (:152-167) */
            curStart1 = pos1;
            curStart2 = pos2;
        } else if (pendFirst >= 0) { /* the pending run is kept (:172-177) */
            for (int64_t j = pendFirst; rc == CPECAN_OK && j < i; j++) rc = cigar_push(&cur, &curCap, pA->ops[2 * j], pA->ops[2 * j + 1]);
        }
        run = 0;
        pendFirst = -1;
        pos1 += pA->strand1 ? len : -len;
        pos2 += pA->strand2 ? len : -len;
        curEnd1 = pos1;
        curEnd2 = pos2;
        if (rc == CPECAN_OK) rc = cigar_push(&cur, &curCap, CPECAN_OP_MATCH, len);
    }
    cpecan_cigar_clear(&cur);
    return rc;
}

int cpecan_cigar_from_aligned_pairs(const char *contig1, const char *contig2, double score, int64_t length1, int64_t length2,
                                    const int64_t *xy, int64_t n, cpecan_cigar *out) {
    if (!contig1 || !contig2 || length1 < 0 || length2 < 0 || n < 0 || (n > 0 && !xy) || !out) return CPECAN_EINVAL;
    for (int64_t i = 0; i < n; i++)
        if (xy[2 * i] < 0 || xy[2 * i] >= length1 || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= length2) return CPECAN_EINVAL;
    const int rc = cigar_from_pairs(out, contig1, contig2, score, length1, length2, xy, n);
    if (rc != CPECAN_OK) cpecan_cigar_clear(out);
    return rc;
}

int cpecan_cigar_split(const cpecan_cigar *c, int64_t maxIndelLength, cpecan_cigar **out, int64_t *nOut) {
    if (!c || !c->contig1 || !c->contig2 || maxIndelLength < 0 || !out || !nOut || !cigar_consistent(c)) return CPECAN_EINVAL;
    *out = NULL;
    *nOut = 0;
    int64_t cap = 0;
    const int rc = cigar_split(c, maxIndelLength, out, nOut, &cap);
    if (rc != CPECAN_OK) {
        cpecan_cigars_free(*out, *nOut);
        *out = NULL;
        *nOut = 0;
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * the realigner
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    char *key, *seq;
    int64_t length;
} SeqEntry;

#define CPK_REALIGN_MAX_DEVICES 64
struct cpecan_realigner {
    cpecan_model model;
    cpecan_realign_options opt;
    int device;
    int devices[CPK_REALIGN_MAX_DEVICES]; /* cpecan_realigner_set_devices: the shards of a call, one per entry */
    int nDevices;                         /* 0 or 1: everything on `device` */
    SeqEntry *seqs; /* open addressing, capacity a power of two */
    int64_t nSeqs, capSeqs;
    char *finalPairsPath, *allPairsPath;
};

void cpecan_realign_options_default(cpecan_realign_options *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    cpecan_params_default(&o->params);
    o->params.diagonalExpansion = 4;          /* cPecanRealign.c:358 */
    o->params.splitMatrixBiggerThanThis = 10; /* :357 */
    o->constraintDiagonalTrim = 0;            /* :356 */
    o->gapGamma = 0.5f;                       /* pairwiseAlignmentBandingParameters_construct, pairwiseAligner.c:1345 */
    o->matchGamma = 0.85f;                    /* :355 */
    o->splitIndelsLongerThanThis = -1;        /* :365 */
}

int cpecan_realigner_create(cpecan_realigner **out, const cpecan_model *model, const cpecan_realign_options *o, int device) {
    if (!out || !model || !o) return CPECAN_EINVAL;
    if (!(o->gapGamma >= 0.0f) || !(o->matchGamma >= 0.0f) || o->constraintDiagonalTrim < 0 ||
        o->params.diagonalExpansion < 0 || o->params.diagonalExpansion % 2 != 0 || o->params.splitMatrixBiggerThanThis < 0 ||
        o->splitIndelsLongerThanThis < -1) { /* the asserts of cPecanRealign.c:408-460 */
        cpk_set_error("bad realign option");
        return CPECAN_EINVAL;
    }
    cpecan_realigner *r = calloc(1, sizeof *r);
    if (!r) return CPECAN_ENOMEM;
    r->model = *model;
    r->opt = *o;
    r->device = device;
    *out = r;
    return CPECAN_OK;
}

void cpecan_realigner_destroy(cpecan_realigner *r) {
    if (!r) return;
    for (int64_t i = 0; i < r->capSeqs; i++) {
        free(r->seqs[i].key);
        free(r->seqs[i].seq);
    }
    free(r->seqs);
    free(r->finalPairsPath);
    free(r->allPairsPath);
    free(r);
}

static uint64_t hash_key(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ull; /* FNV-1a */
    for (size_t i = 0; i < n; i++) h = (h ^ (unsigned char)s[i]) * 1099511628211ull;
    return h;
}

static SeqEntry *seq_slot(SeqEntry *table, int64_t cap, const char *key, size_t n) {
    for (uint64_t at = hash_key(key, n) & (uint64_t)(cap - 1);; at = (at + 1) & (uint64_t)(cap - 1)) {
        SeqEntry *e = &table[at];
        if (!e->key || (strlen(e->key) == n && memcmp(e->key, key, n) == 0)) return e;
    }
}

static const SeqEntry *seq_find(const cpecan_realigner *r, const char *key) {
    if (!r->capSeqs) return NULL;
    const SeqEntry *e = seq_slot(r->seqs, r->capSeqs, key, strlen(key));
    return e->key ? e : NULL;
}

int cpecan_realigner_add_sequence(cpecan_realigner *r, const char *header, const char *seq, int64_t length) {
    if (!r || !header || (!seq && length > 0) || length < 0) return CPECAN_EINVAL;
    const char *p = header, *tok;
    const size_t n = next_token(&p, &tok); /* the first token of the header is the name (cPecanRealign.c:246-247) */
    if (n == 0) {
        cpk_set_error("fasta header without a name");
        return CPECAN_EINVAL;
    }
    if (2 * (r->nSeqs + 1) > r->capSeqs) {
        const int64_t cap = r->capSeqs ? 2 * r->capSeqs : 64;
        SeqEntry *grown = calloc((size_t)cap, sizeof(SeqEntry));
        if (!grown) return CPECAN_ENOMEM;
        for (int64_t i = 0; i < r->capSeqs; i++)
            if (r->seqs[i].key) *seq_slot(grown, cap, r->seqs[i].key, strlen(r->seqs[i].key)) = r->seqs[i];
        free(r->seqs);
        r->seqs = grown;
        r->capSeqs = cap;
    }
    SeqEntry *e = seq_slot(r->seqs, r->capSeqs, tok, n);
    if (e->key && length <= e->length) return CPECAN_OK; /* a repeat that is no longer: keep the first (:248-265) */
    char *copy = copy_span(seq ? seq : "", (size_t)length);
    if (!copy) return CPECAN_ENOMEM;
    if (!e->key) {
        e->key = copy_span(tok, n);
        if (!e->key) {
            free(copy);
            return CPECAN_ENOMEM;
        }
        r->nSeqs++;
    }
    free(e->seq);
    e->seq = copy;
    e->length = length;
    return CPECAN_OK;
}

int64_t cpecan_realigner_read_fasta(cpecan_realigner *r, const char *path) {
    if (!r || !path) return CPECAN_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) {
        cpk_set_error("cannot open %s", path);
        return CPECAN_EINVAL;
    }
    char *line = NULL, *header = NULL, *seq = NULL;
    size_t lineCap = 0;
    int64_t seqLen = 0, seqCap = 0, records = 0;
    int rc = CPECAN_OK;
    ssize_t got;
    while (rc == CPECAN_OK) {
        got = getline(&line, &lineCap, f);
        if (got < 0 || line[0] == '>') {
            if (header) {
                rc = cpecan_realigner_add_sequence(r, header, seq, seqLen);
                records++;
                free(header);
                header = NULL;
            }
            if (got < 0) break;
            while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
            header = copy_span(line + 1, (size_t)got - 1);
            if (!header) rc = CPECAN_ENOMEM;
            seqLen = 0;
            continue;
        }
        if (!header) continue; /* text before the first record */
        if (seqLen + got + 1 > seqCap) {
            seqCap = 2 * (seqLen + got + 1);
            char *grown = realloc(seq, (size_t)seqCap);
            if (!grown) {
                rc = CPECAN_ENOMEM;
                break;
            }
            seq = grown;
        }
        for (ssize_t i = 0; i < got; i++)
            if (!isspace((unsigned char)line[i])) seq[seqLen++] = line[i];
    }
    free(header);
    free(line);
    free(seq);
    fclose(f);
    return rc == CPECAN_OK ? records : rc;
}

static int set_path(char **dst, const char *src) {
    free(*dst);
    *dst = NULL;
    if (!src) return CPECAN_OK;
    *dst = copy_span(src, strlen(src));
    return *dst ? CPECAN_OK : CPECAN_ENOMEM;
}

int cpecan_realigner_set_posterior_files(cpecan_realigner *r, const char *finalPairsPath, const char *allPairsPath) {
    if (!r) return CPECAN_EINVAL;
    int rc = set_path(&r->finalPairsPath, finalPairsPath);
    if (rc == CPECAN_OK) rc = set_path(&r->allPairsPath, allPairsPath);
    return rc;
}

/* stString_reverseComplementString of sonLib: the four bases and the IUPAC codes are complemented, case kept */
static char complement(char c) {
    static const char from[] = "ACGTRYKMBVDHacgtrykmbvdh", to[] = "TGCAYRMKVBHDtgcayrmkvbhd";
    const char *p = c ? strchr(from, c) : NULL;
    return p ? to[p - from] : c;
}

/* getSubSequence, cPecanRealign.c:245-253 */
static char *sub_sequence(const SeqEntry *e, int64_t start, int64_t end, int strand) {
    const int64_t lo = strand ? start : end, hi = strand ? end : start;
    if (lo < 0 || hi < lo || hi > e->length) return NULL;
    char *s = copy_span(e->seq + lo, (size_t)(hi - lo));
    if (s && !strand) {
        const int64_t n = hi - lo;
        for (int64_t a = 0, b = n - 1; a <= b; a++, b--) {
            const char ca = complement(s[a]), cb = complement(s[b]);
            s[a] = cb;
            s[b] = ca;
        }
    }
    return s;
}

typedef struct {
    char *subX, *subY;
    int64_t lX, lY;
    int64_t shift1, shift2; /* coordinateShift1/2 (:516-517) */
    int flip1, flip2;
    int64_t *anchors;   /* every match column (x, y, expansion) */
    int64_t nAnchors;
    int64_t *filtered;  /* the exact-match ones (:529), as nFiltered runs (x, y, length, expansion) */
    int64_t nFiltered;
} Item;

static void item_clear(Item *it) {
    free(it->subX);
    free(it->subY);
    free(it->anchors);
    free(it->filtered);
    memset(it, 0, sizeof *it);
}

/* cPecanRealign.c:511-529: sub-sequences on the forward strand from 0, anchors from the cigar's operations */
static int item_prepare(const cpecan_realigner *r, const cpecan_cigar *pA, Item *it) {
    memset(it, 0, sizeof *it);
    if (!pA->contig1 || !pA->contig2 || !cigar_consistent(pA)) {
        cpk_set_error("inconsistent pairwise alignment");
        return CPECAN_EINVAL;
    }
    const SeqEntry *eX = seq_find(r, pA->contig1), *eY = seq_find(r, pA->contig2);
    if (!eX || !eY) {
        cpk_set_error("no sequence named %s", eX ? pA->contig2 : pA->contig1);
        return CPECAN_EINVAL;
    }
    it->flip1 = !pA->strand1;
    it->flip2 = !pA->strand2;
    it->shift1 = pA->strand1 ? pA->start1 : pA->end1;
    it->shift2 = pA->strand2 ? pA->start2 : pA->end2;
    it->subX = sub_sequence(eX, pA->start1, pA->end1, pA->strand1);
    it->subY = sub_sequence(eY, pA->start2, pA->end2, pA->strand2);
    if (!it->subX || !it->subY) {
        cpk_set_error("cigar coordinates outside sequence %s", it->subX ? pA->contig2 : pA->contig1);
        return CPECAN_EINVAL;
    }
    it->lX = (int64_t)strlen(it->subX);
    it->lY = (int64_t)strlen(it->subY);
    int64_t matches = 0;
    for (int64_t i = 0; i < pA->nOps; i++)
        if (pA->ops[2 * i] == CPECAN_OP_MATCH) matches += pA->ops[2 * i + 1];
    /* the exact-match anchors (:525-529) as runs of diagonal neighbours: 32 bytes per run instead of 24 per column */
    it->nFiltered = cpecan_anchor_runs_from_alignment(pA->ops, pA->nOps, 0, 0, r->opt.constraintDiagonalTrim,
                                                      r->opt.params.diagonalExpansion, it->subX, it->lX, it->subY, it->lY, NULL, 0);
    if (it->nFiltered < 0) return CPECAN_EINVAL;
    it->filtered = malloc(sizeof(int64_t) * 4 * (size_t)(it->nFiltered ? it->nFiltered : 1));
    if (!it->filtered) return CPECAN_ENOMEM;
    if (r->opt.rescoreOriginalAlignment) { /* scoreAnchorPairs looks the unfiltered anchors up (:548) */
        it->anchors = malloc(sizeof(int64_t) * 3 * (size_t)(matches ? matches : 1));
        if (!it->anchors) return CPECAN_ENOMEM;
        it->nAnchors = cpecan_anchors_from_alignment(pA->ops, pA->nOps, 0, 0, r->opt.constraintDiagonalTrim,
                                                     r->opt.params.diagonalExpansion, NULL, 0, NULL, 0, it->anchors);
    }
    it->nFiltered = cpecan_anchor_runs_from_alignment(pA->ops, pA->nOps, 0, 0, r->opt.constraintDiagonalTrim,
                                                      r->opt.params.diagonalExpansion, it->subX, it->lX, it->subY, it->lY,
                                                      it->filtered, it->nFiltered);
    if (it->nAnchors < 0 || it->nFiltered < 0) return CPECAN_EINVAL;
    return CPECAN_OK;
}

/* transformCoordinate + writePosteriorProbs, cPecanRealign.c:290-312 */
static int write_pairs(const char *path, const int32_t *triples, int64_t n, const Item *it) {
    FILE *f = fopen(path, "w");
    if (!f) {
        cpk_set_error("cannot write %s", path);
        return CPECAN_EINVAL;
    }
    for (int64_t i = 0; i < n; i++) {
        const int64_t x = triples[3 * i + 1], y = triples[3 * i + 2];
        fprintf(f, "%lld\t%lld\t%f\n", (long long)(it->shift1 + (it->flip1 ? it->lX - 1 - x : x)),
                (long long)(it->shift2 + (it->flip2 ? it->lY - 1 - y : y)), (double)triples[3 * i] / CPECAN_PROB_1);
    }
    fclose(f);
    return CPECAN_OK;
}

static int cmp_xy(const void *a, const void *b) {
    const int64_t *p = a, *q = b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    return p[1] < q[1] ? -1 : (p[1] > q[1] ? 1 : 0);
}

/* scoreAnchorPairs, cPecanRealign.c:314-348: the aligned pairs that are anchors, in list order, then the anchors the
 * aligner did not return, with score 0.  Anchors increase strictly in x, so a per-x table is the sorted set. */
static int score_anchor_pairs(const Item *it, const int32_t *pairs, int64_t n, int32_t **out, int64_t *nOut) {
    int64_t *yOfX = malloc(sizeof(int64_t) * (size_t)(it->lX ? it->lX : 1));
    int32_t *t = malloc(sizeof(int32_t) * 3 * (size_t)(it->nAnchors ? it->nAnchors : 1));
    if (!yOfX || !t) {
        free(yOfX);
        free(t);
        return CPECAN_ENOMEM;
    }
    for (int64_t x = 0; x < it->lX; x++) yOfX[x] = -1;
    for (int64_t i = 0; i < it->nAnchors; i++) yOfX[it->anchors[3 * i]] = it->anchors[3 * i + 1];
    int64_t at = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t x = pairs[3 * i + 1], y = pairs[3 * i + 2];
        if (yOfX[x] == y) {
            t[3 * at] = pairs[3 * i];
            t[3 * at + 1] = x;
            t[3 * at + 2] = y;
            at++;
            yOfX[x] = -1;
        }
    }
    for (int64_t x = 0; x < it->lX; x++)
        if (yOfX[x] >= 0) {
            t[3 * at] = 0;
            t[3 * at + 1] = (int32_t)x;
            t[3 * at + 2] = (int32_t)yOfX[x];
            at++;
        }
    free(yOfX);
    *out = t;
    *nOut = at;
    return CPECAN_OK;
}

/* the four score functions (pairwiseAligner.c:1562-1597) of a list held on the host: only the rescoreOriginalAlignment
 * path needs them, every other list is scored on the device by the batch's consumer stage */
static void host_scores(const Item *it, const int32_t *t, int64_t n, double s[4]) {
    double total = 0.0;
    int64_t matches = 0;
    for (int64_t i = 0; i < n; i++) {
        total += t[3 * i];
        const int a = toupper((unsigned char)it->subX[t[3 * i + 1]]), b = toupper((unsigned char)it->subY[t[3 * i + 2]]);
        matches += a == b && a != 'N';
    }
    const int64_t L = it->lX + it->lY;
    s[0] = 100.0 * (L == 0 ? 0 : (2.0 * total) / (double)(L * CPECAN_PROB_1));
    s[1] = 100.0 * total / ((double)n * CPECAN_PROB_1);
    s[2] = 100.0 * (L == 0 ? 0 : (2.0 * matches) / (double)L);
    s[3] = 100.0 * matches / (double)n;
}

static __thread double g_stage[3]; /* upload, run, download of the last run_batch (CPECAN_REALIGN_TIMING only) */
static int run_batch(cpecan_batch *b) {
    const double t0 = now_ms();
    int rc = cpecan_batch_upload(b);
    const double t1 = now_ms();
    if (rc == CPECAN_OK) rc = cpecan_batch_run(b, NULL);
    const double t2 = now_ms();
    if (rc == CPECAN_OK) rc = cpecan_batch_download(b);
    g_stage[0] = t1 - t0;
    g_stage[1] = t2 - t1;
    g_stage[2] = now_ms() - t2;
    return rc;
}

/* cPecanRealign.c:511-529 for every cigar, in parallel (the cigars are independent).  An error message belongs to the
 * thread that set it, so the first failing cigar is prepared once more on the calling thread. */
static int prepare_range(const cpecan_realigner *r, const cpecan_cigar *in, int64_t from, int64_t to, Item *items) {
    int64_t firstBad = to;
#pragma omp parallel for schedule(dynamic, 16) num_threads(cpk_host_threads())
    for (int64_t i = from; i < to; i++)
        if (item_prepare(r, &in[i], &items[i]) != CPECAN_OK) {
#pragma omp critical(cpk_realign)
            if (i < firstBad) firstBad = i;
        }
    if (firstBad == to) return CPECAN_OK;
    item_clear(&items[firstBad]);
    const int rc = item_prepare(r, &in[firstBad], &items[firstBad]);
    return rc != CPECAN_OK ? rc : CPECAN_ESTATE;
}

/* Prepares the cigars a slice at a time and adds each slice to the batch (both ends ragged, :537); the anchor lists of a
 * slice are released as soon as the batch holds them, so the next slice reuses their memory. */
static int prepare_and_add(const cpecan_realigner *r, const cpecan_cigar *in, int64_t n, Item *items, cpecan_batch *b) {
    const int64_t slice = 4096;
    for (int64_t from = 0; from < n; from += slice) {
        const int64_t to = from + slice < n ? from + slice : n;
        int rc = prepare_range(r, in, from, to, items);
        cpecan_problem_runs *probs = rc == CPECAN_OK ? malloc(sizeof(cpecan_problem_runs) * (size_t)(to - from)) : NULL;
        if (rc == CPECAN_OK && !probs) rc = CPECAN_ENOMEM;
        for (int64_t i = from; rc == CPECAN_OK && i < to; i++) {
            const cpecan_problem_runs p = {items[i].subX, items[i].lX, items[i].subY, items[i].lY, items[i].filtered,
                                           items[i].nFiltered, 1, 1};
            probs[i - from] = p;
        }
        if (rc == CPECAN_OK) {
            const int64_t first = cpecan_batch_add_many_runs(b, probs, to - from);
            rc = first < 0 ? (int)first : CPECAN_OK;
        }
        free(probs);
        for (int64_t i = from; i < to; i++) {
            free(items[i].filtered);
            items[i].filtered = NULL;
        }
        if (rc != CPECAN_OK) return rc;
    }
    return CPECAN_OK;
}

/* The list a cigar is rebuilt from, and its four scores: the ordered alignment with the scores of the batch's consumer
 * stage, or -- rescoreOriginalAlignment -- the input's own columns as the aligner scored them (:547-564). */
static int final_list(const cpecan_realigner *r, const cpecan_batch *b, const Item *it, int64_t i, const int32_t **list,
                      int32_t **owned, int64_t *nList, double s[4]) {
    *owned = NULL;
    if (r->opt.rescoreOriginalAlignment) {
        const int32_t *all;
        int64_t nAll;
        int rc = cpecan_batch_result(b, i, 0, &all, &nAll);
        if (rc == CPECAN_OK) rc = score_anchor_pairs(it, all, nAll, owned, nList);
        if (rc != CPECAN_OK) return rc;
        *list = *owned;
        host_scores(it, *list, *nList, s);
        return CPECAN_OK;
    }
    int rc = cpecan_batch_result(b, i, 3, list, nList);
    if (rc == CPECAN_OK) rc = cpecan_batch_scores(b, i, &s[0], &s[1], NULL);
    if (rc == CPECAN_OK) rc = cpecan_batch_identity_scores(b, i, &s[2], &s[3]);
    return rc;
}

typedef struct {
    cpecan_cigar *pieces;
    int64_t n, cap;
} OutSlot;

/* cPecanRealign.c:556-591 for one cigar: score, cigar from the sorted pairs, rebase, optional split */
static int build_output(const cpecan_realigner *r, const cpecan_batch *b, const cpecan_cigar *pA, const Item *it, int64_t i,
                        OutSlot *slot) {
    const cpecan_realign_options *o = &r->opt;
    const int32_t *list = NULL;
    int32_t *owned = NULL;
    int64_t nList = 0;
    double s[4] = {0, 0, 0, 0};
    int rc = final_list(r, b, it, i, &list, &owned, &nList, s);
    if (rc != CPECAN_OK) return rc;
    double score = pA->score; /* :556-564 */
    if (o->rescoreByPosteriorProb) score = s[0];
    else if (o->rescoreByPosteriorProbIgnoringGaps) score = s[1];
    else if (o->rescoreByIdentity) score = s[2];
    else if (o->rescoreByIdentityIgnoringGaps) score = s[3];
    int64_t *xy = malloc(sizeof(int64_t) * 2 * (size_t)(nList ? nList : 1));
    cpecan_cigar rPA;
    memset(&rPA, 0, sizeof rPA);
    if (!xy) rc = CPECAN_ENOMEM;
    if (rc == CPECAN_OK) {
        int sorted = 1;
        for (int64_t k = 0; k < nList; k++) {
            xy[2 * k] = list[3 * k + 1];
            xy[2 * k + 1] = list[3 * k + 2];
            if (k > 0 && cmp_xy(&xy[2 * k - 2], &xy[2 * k]) > 0) sorted = 0;
        }
        if (!sorted) qsort(xy, (size_t)nList, 2 * sizeof(int64_t), cmp_xy); /* :573 */
        rc = cigar_from_pairs(&rPA, pA->contig1, pA->contig2, score, it->lX, it->lY, xy, nList);
    }
    free(xy);
    free(owned);
    if (rc == CPECAN_OK) {
        rebase(&rPA.start1, &rPA.end1, &rPA.strand1, it->shift1, it->flip1); /* :578-581 */
        rebase(&rPA.start2, &rPA.end2, &rPA.strand2, it->shift2, it->flip2);
        if (!cigar_consistent(&rPA)) {
            cpk_set_error("internal: realigned cigar is inconsistent");
            rc = CPECAN_ESTATE;
        }
    }
    if (rc == CPECAN_OK && o->splitIndelsLongerThanThis != -1) {
        rc = cigar_split(&rPA, o->splitIndelsLongerThanThis, &slot->pieces, &slot->n, &slot->cap);
        cpecan_cigar_clear(&rPA);
    } else if (rc == CPECAN_OK) {
        slot->pieces = malloc(sizeof(cpecan_cigar));
        if (slot->pieces) {
            slot->pieces[0] = rPA;
            slot->n = slot->cap = 1;
        } else {
            rc = CPECAN_ENOMEM;
            cpecan_cigar_clear(&rPA);
        }
    } else {
        cpecan_cigar_clear(&rPA);
    }
    return rc;
}

static int realign_on_device(const cpecan_realigner *r, const cpecan_cigar *in, int64_t n, cpecan_cigar **out, int64_t *nOut) {
    *out = NULL;
    *nOut = 0;
    const cpecan_realign_options *o = &r->opt;
    Item *items = calloc((size_t)(n ? n : 1), sizeof(Item));
    OutSlot *slots = calloc((size_t)(n ? n : 1), sizeof(OutSlot));
    cpecan_cigar *res = NULL;
    int64_t nRes = 0;
    cpecan_batch *b = NULL;
    int rc = items && slots ? CPECAN_OK : CPECAN_ENOMEM;
    const int timing = getenv("CPECAN_REALIGN_TIMING") != NULL; /* stage times of this call on stderr */
    const double t0 = now_ms();
    if (rc == CPECAN_OK) rc = cpecan_batch_create(&b, &r->model, &o->params, CPECAN_EMIT_MATCH, r->device);
    if (rc == CPECAN_OK && !o->rescoreOriginalAlignment) { /* :552-553 as the batch's consumer stage */
        rc = cpecan_batch_set_post(b, CPECAN_POST_REWEIGHT | CPECAN_POST_ORDERED, (double)o->gapGamma);
        if (rc == CPECAN_OK) rc = cpecan_batch_set_match_gamma(b, o->matchGamma);
    }
    if (rc == CPECAN_OK) rc = prepare_and_add(r, in, n, items, b);
    const double t1 = now_ms();
    if (rc == CPECAN_OK && n > 0) rc = run_batch(b);
    const double t2 = now_ms();
    if (rc == CPECAN_OK) {
        int64_t firstBad = n;
#pragma omp parallel for schedule(dynamic, 64) num_threads(cpk_host_threads())
        for (int64_t i = 0; i < n; i++)
            if (build_output(r, b, &in[i], &items[i], i, &slots[i]) != CPECAN_OK) {
#pragma omp critical(cpk_realign)
                if (i < firstBad) firstBad = i;
            }
        if (firstBad < n) { /* once more on this thread, for its error message */
            cpecan_cigars_free(slots[firstBad].pieces, slots[firstBad].n);
            memset(&slots[firstBad], 0, sizeof(OutSlot));
            rc = build_output(r, b, &in[firstBad], &items[firstBad], firstBad, &slots[firstBad]);
            if (rc == CPECAN_OK) rc = CPECAN_ESTATE;
        }
    }
    if (rc == CPECAN_OK) {
        for (int64_t i = 0; i < n; i++) nRes += slots[i].n;
        res = malloc(sizeof(cpecan_cigar) * (size_t)(nRes ? nRes : 1));
        if (!res) rc = CPECAN_ENOMEM;
        for (int64_t i = 0, at = 0; rc == CPECAN_OK && i < n; i++) { /* ownership of the pieces moves to res */
            if (slots[i].n) memcpy(res + at, slots[i].pieces, sizeof(cpecan_cigar) * (size_t)slots[i].n);
            at += slots[i].n;
            free(slots[i].pieces);
            memset(&slots[i], 0, sizeof(OutSlot));
        }
    }
    if (rc == CPECAN_OK && n > 0 && r->finalPairsPath) { /* the last cigar's final pairs (:566-570) */
        const int32_t *list = NULL;
        int32_t *owned = NULL;
        int64_t nList = 0;
        double s[4];
        rc = final_list(r, b, &items[n - 1], n - 1, &list, &owned, &nList, s);
        if (rc == CPECAN_OK) rc = write_pairs(r->finalPairsPath, list, nList, &items[n - 1]);
        free(owned);
    }
    if (rc == CPECAN_OK && n > 0 && r->allPairsPath) { /* every pair of the last alignment, before reweighting (:541-545) */
        cpecan_batch *raw = NULL;
        Item *it = &items[n - 1];
        item_clear(it);
        rc = item_prepare(r, &in[n - 1], it);
        if (rc == CPECAN_OK) rc = cpecan_batch_create(&raw, &r->model, &o->params, CPECAN_EMIT_MATCH, r->device);
        if (rc == CPECAN_OK) {
            const cpecan_problem_runs pr = {it->subX, it->lX, it->subY, it->lY, it->filtered, it->nFiltered, 1, 1};
            const int64_t idx = cpecan_batch_add_many_runs(raw, &pr, 1);
            rc = idx < 0 ? (int)idx : run_batch(raw);
        }
        const int32_t *all = NULL;
        int64_t nAll = 0;
        if (rc == CPECAN_OK) rc = cpecan_batch_result(raw, 0, 0, &all, &nAll);
        if (rc == CPECAN_OK) rc = write_pairs(r->allPairsPath, all, nAll, it);
        cpecan_batch_destroy(raw);
    }
    if (timing && rc == CPECAN_OK && n > 0) {
        cpecan_stats st;
        cpecan_batch_stats(b, &st);
        fprintf(stderr,
                "cpecan_realign: %lld cigars, %lld regions, %lld cells, %lld pairs: prepare+add %.1f ms, batch %.1f ms "
                "(plan+upload %.1f [h2d copy %.1f], launch %.1f, wait+consumers+download %.1f [DP kernel %.1f, d2h copy %.1f]), "
                "cigars %.1f ms\n",
                (long long)n, (long long)st.regions, (long long)st.cells, (long long)st.pairs, t1 - t0, t2 - t1, g_stage[0],
                st.h2dMs, g_stage[1], g_stage[2], st.kernelMs, st.d2hMs, now_ms() - t2);
    }
    cpecan_batch_destroy(b);
    for (int64_t i = 0; items && i < n; i++) item_clear(&items[i]);
    free(items);
    for (int64_t i = 0; slots && i < n; i++) cpecan_cigars_free(slots[i].pieces, slots[i].n);
    free(slots);
    if (rc != CPECAN_OK) {
        cpecan_cigars_free(res, nRes);
        return rc;
    }
    *out = res;
    *nOut = nRes;
    return CPECAN_OK;
}

static int expectations_on_device(const cpecan_realigner *r, const cpecan_cigar *in, int64_t n, cpecan_hmm *acc) {
    if (n == 0) return CPECAN_OK;
    cpecan_batch *b = NULL;
    int rc = cpecan_batch_create(&b, &r->model, &r->opt.params, CPECAN_EMIT_EXPECT, r->device);
    for (int64_t i = 0; rc == CPECAN_OK && i < n; i++) {
        Item it;
        rc = item_prepare(r, &in[i], &it);
        if (rc == CPECAN_OK) {
            const cpecan_problem_runs pr = {it.subX, it.lX, it.subY, it.lY, it.filtered, it.nFiltered, 1, 1};
            const int64_t idx = cpecan_batch_add_many_runs(b, &pr, 1); /* :532 */
            rc = idx < 0 ? (int)idx : CPECAN_OK;
        }
        item_clear(&it);
    }
    if (rc == CPECAN_OK) rc = run_batch(b);
    if (rc == CPECAN_OK) rc = cpecan_batch_expectations(b, acc);
    cpecan_batch_destroy(b);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Several GPUs from one process.  The reference fans a realignment out as one
 * cPecanRealign process per shard of the cigar file and sums the shards' expectation files (cPecanEm.py:168-188); here
 * the cigars of ONE call are cut into contiguous shards of about equal band cells, shard k runs as its own batch on
 * devices[k] from a host thread of its own, and the results are joined in input order -- the expectation counts summed on
 * the host in shard order (106 doubles: no collective needed inside one process).
 * ---------------------------------------------------------------------------------------------- */
int cpecan_realign_shard_bounds(const cpecan_cigar *in, int64_t n, int64_t diagonalExpansion, int nShards, int64_t *bounds) {
    if ((!in && n > 0) || n < 0 || nShards < 1 || !bounds || diagonalExpansion < 0) return CPECAN_EINVAL;
    /* band cells of a cigar, near enough: diagonals times band width (cpecan_amd/dist.py: cigar_cost) */
    long double total = 0;
    for (int64_t i = 0; i < n; i++)
        total += (long double)(llabs((long long)(in[i].end1 - in[i].start1)) + llabs((long long)(in[i].end2 - in[i].start2)) + 1) *
                 (long double)(diagonalExpansion + 1);
    bounds[0] = 0;
    long double run = 0;
    int64_t at = 0;
    for (int k = 1; k < nShards; k++) { /* the first cigar at which the running cost reaches k / nShards of the total */
        const long double want = total * (long double)k / (long double)nShards;
        while (at < n && run < want) {
            run += (long double)(llabs((long long)(in[at].end1 - in[at].start1)) + llabs((long long)(in[at].end2 - in[at].start2)) + 1) *
                   (long double)(diagonalExpansion + 1);
            at++;
        }
        bounds[k] = at;
    }
    bounds[nShards] = n;
    return CPECAN_OK;
}

int cpecan_realigner_set_devices(cpecan_realigner *r, const int *devices, int nDevices) {
    if (!r || nDevices < 0 || nDevices > CPK_REALIGN_MAX_DEVICES || (nDevices > 0 && !devices)) return CPECAN_EINVAL;
    for (int k = 0; k < nDevices; k++)
        if (devices[k] < 0) return CPECAN_EINVAL;
    for (int k = 0; k < nDevices; k++) r->devices[k] = devices[k];
    r->nDevices = nDevices;
    if (nDevices > 0) r->device = devices[0];
    return CPECAN_OK;
}

typedef struct {
    cpecan_realigner shard; /* a shallow copy bound to one device: the sequences are shared, read only */
    const cpecan_cigar *in;
    int64_t n;
    int expect;
    int threads;
    cpecan_cigar *out;
    int64_t nOut;
    cpecan_hmm acc;
    int rc;
    char err[512];
} ShardJob;

static void *shard_main(void *arg) {
    ShardJob *j = arg;
    omp_set_num_threads(j->threads); /* this thread's parallel loops: its share of the host's cores */
    j->rc = j->expect ? expectations_on_device(&j->shard, j->in, j->n, &j->acc)
                      : realign_on_device(&j->shard, j->in, j->n, &j->out, &j->nOut);
    if (j->rc != CPECAN_OK) {
        strncpy(j->err, cpk_last_error(), sizeof j->err - 1);
        j->err[sizeof j->err - 1] = 0;
    }
    return NULL;
}

/* runs the shards of a call on their devices; jobs[k].out / .acc hold the results */
static int run_shards(cpecan_realigner *r, const cpecan_cigar *in, int64_t n, int expect, int32_t hmmType, ShardJob *jobs) {
    const int K = r->nDevices;
    int64_t bounds[CPK_REALIGN_MAX_DEVICES + 1];
    int rc = cpecan_realign_shard_bounds(in, n, r->opt.params.diagonalExpansion, K, bounds);
    if (rc != CPECAN_OK) return rc;
    int threads = cpk_host_threads() / K;
    if (threads < 1) threads = 1;
    pthread_t tid[CPK_REALIGN_MAX_DEVICES];
    int started = 0;
    for (int k = 0; k < K; k++) {
        ShardJob *j = &jobs[k];
        memset(j, 0, sizeof *j);
        j->shard = *r;
        j->shard.device = r->devices[k];
        j->shard.nDevices = 0;
        if (bounds[k + 1] != n || bounds[k] == n) /* the posterior files are the LAST cigar's (cPecanRealign.c:541-570) */
            j->shard.finalPairsPath = j->shard.allPairsPath = NULL;
        j->in = in + bounds[k];
        j->n = bounds[k + 1] - bounds[k];
        j->expect = expect;
        j->threads = threads;
        if (expect && (rc = cpecan_hmm_init(&j->acc, hmmType, 0.0)) != CPECAN_OK) break;
        if (pthread_create(&tid[k], NULL, shard_main, j) != 0) {
            cpk_set_error("cannot start the thread of shard %d", k);
            rc = CPECAN_ENOMEM;
            break;
        }
        started++;
    }
    for (int k = 0; k < started; k++) pthread_join(tid[k], NULL);
    for (int k = 0; k < started && rc == CPECAN_OK; k++)
        if (jobs[k].rc != CPECAN_OK) {
            rc = jobs[k].rc;
            cpk_set_error("shard %d (device %d): %s", k, r->devices[k], jobs[k].err);
        }
    if (rc != CPECAN_OK)
        for (int k = 0; k < started; k++) {
            cpecan_cigars_free(jobs[k].out, jobs[k].nOut);
            jobs[k].out = NULL;
            jobs[k].nOut = 0;
        }
    return rc;
}

int cpecan_realigner_realign(cpecan_realigner *r, const cpecan_cigar *in, int64_t n, cpecan_cigar **out, int64_t *nOut) {
    if (!r || (!in && n > 0) || n < 0 || !out || !nOut) return CPECAN_EINVAL;
    if (r->nDevices <= 1 || n < 2) return realign_on_device(r, in, n, out, nOut);
    *out = NULL;
    *nOut = 0;
    ShardJob *jobs = malloc(sizeof(ShardJob) * (size_t)r->nDevices);
    if (!jobs) return CPECAN_ENOMEM;
    int rc = run_shards(r, in, n, 0, 0, jobs);
    if (rc == CPECAN_OK) {
        int64_t total = 0;
        for (int k = 0; k < r->nDevices; k++) total += jobs[k].nOut;
        cpecan_cigar *res = malloc(sizeof(cpecan_cigar) * (size_t)(total ? total : 1));
        if (!res) {
            rc = CPECAN_ENOMEM;
            for (int k = 0; k < r->nDevices; k++) cpecan_cigars_free(jobs[k].out, jobs[k].nOut);
        } else {
            int64_t at = 0;
            for (int k = 0; k < r->nDevices; k++) { /* shard order is input order: ownership of the cigars moves to res */
                if (jobs[k].nOut) memcpy(res + at, jobs[k].out, sizeof(cpecan_cigar) * (size_t)jobs[k].nOut);
                at += jobs[k].nOut;
                free(jobs[k].out);
            }
            *out = res;
            *nOut = total;
        }
    }
    free(jobs);
    return rc;
}

int cpecan_realigner_expectations(cpecan_realigner *r, const cpecan_cigar *in, int64_t n, cpecan_hmm *acc) {
    if (!r || (!in && n > 0) || n < 0 || !acc) return CPECAN_EINVAL;
    if (r->nDevices <= 1 || n < 2) return expectations_on_device(r, in, n, acc);
    ShardJob *jobs = malloc(sizeof(ShardJob) * (size_t)r->nDevices);
    if (!jobs) return CPECAN_ENOMEM;
    int rc = run_shards(r, in, n, 1, acc->type, jobs);
    for (int k = 0; rc == CPECAN_OK && k < r->nDevices; k++) { /* the sum of cPecanEm.py:184-188, in shard order */
        for (int i = 0; i < 25; i++) acc->transitions[i] += jobs[k].acc.transitions[i];
        for (int i = 0; i < 80; i++) acc->emissions[i] += jobs[k].acc.emissions[i];
        acc->likelihood += jobs[k].acc.likelihood;
    }
    free(jobs);
    return rc;
}
