// cpk_post.inl -- consumers of the posterior lists: reweighting, posterior scores, MEA chain, left shift.
// Part of the single HIP translation unit cpecan_kernels.hip (included there, in this order); not compiled on its own.

// ------------------------------------------------------------------------------------------------
// Consumers of the posterior lists.  Integer / order-defined arithmetic: bit-exact.
// ------------------------------------------------------------------------------------------------
constexpr int kPostReweight = 1, kPostMea = 2, kPostLeftShift = 4, kPostOrdered = 8;  // == CPECAN_POST_*
constexpr int kPostScores = CPK_POST_SCORES;  // doubles per problem: see CpkPostJob.scores
constexpr int kPostTile = 2048;  // words of LDS of the wave-per-problem consumers: column counters (ordered filter; longer sequences count in global memory), tiles of the walks back along the predecessors

// reweightAlignedPairs2 (impl/pairwiseAligner.c:1519-1558) + scoreByPosteriorProbability[IgnoringGaps] (:1578-1597).
// One workgroup per problem.  mass[] = PROB_1 minus the listed mass of every base of X then Y, floored at 0 when read
// (:1529-1533); a pair keeps  score - gapGamma * (massX + massY), evaluated in double and truncated towards zero (:1543).
__global__ void __launch_bounds__(256) cpecan_post_reweight(const CpkPostProblem *problems, int32_t *triples, int32_t *mass,
                                                            double gapGamma, int reweight, double *scores) {
    const CpkPostProblem pb = problems[blockIdx.x];
    int32_t *t = triples + 3 * pb.off[0];
    const int n = pb.n[0];
    __shared__ long long partial[256];
    long long sum = 0;  // exact: |score| <= 1e7 * (1 + 2 gapGamma), n < 2^31
    if (reweight && gapGamma > 0.0) {  // :1551
        int32_t *mx = mass + pb.seqOff, *my = mx + pb.lX;
        for (int i = threadIdx.x; i < pb.lX + pb.lY; i += blockDim.x) mx[i] = CPECAN_PROB_1;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            atomicSub(&mx[t[3 * i + 1]], t[3 * i]);
            atomicSub(&my[t[3 * i + 2]], t[3 * i]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const long long ux = mx[t[3 * i + 1]], uy = my[t[3 * i + 2]];
            const long long unaligned = (ux < 0 ? 0 : ux) + (uy < 0 ? 0 : uy);
            const long long w = (long long)((double)(long long)t[3 * i] - gapGamma * (double)unaligned);
            t[3 * i] = (int32_t)w;
            sum += w;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) sum += t[3 * i];
    }
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) partial[threadIdx.x] += partial[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double total = (double)partial[0];  // the reference adds int64 scores into a double: exact below 2^53
        const long long L = (long long)pb.lX + pb.lY;
        scores[kPostScores * blockIdx.x + 0] = 100.0 * (L == 0 ? 0 : (2.0 * total) / (double)(L * CPECAN_PROB_1));
        scores[kPostScores * blockIdx.x + 1] = 100.0 * total / ((double)n * CPECAN_PROB_1);
    }
}

// getIndelProb (:1621-1625): gap mass of `length` bases starting at `start`
__device__ __forceinline__ long long gap_mass(const long long *cum, long long start, long long length) {
    return length == 0 ? 0 : cum[start + length - 1] - (start > 0 ? cum[start - 1] : 0);
}

// getMaximalExpectedAccuracyPairwiseAlignment (:1628-1724), one LANE per problem: the chain DP walks the pairs in
// list order with a data-dependent walk back.  gapGamma is the float of PairwiseAlignmentParameters, so
// `int64 * gapGamma` and `int64 + that` are float arithmetic, `int64 + double + float` is double truncated to int64.
__global__ void __launch_bounds__(64) cpecan_post_mea(const CpkPostProblem *problems, int64_t nProblems,
                                                      const int32_t *triples, long long *cum, double *best, int32_t *prev,
                                                      uint8_t *record, float gapGamma, int32_t *meaOut, int32_t *counts,
                                                      double *scores) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nProblems) return;
    const CpkPostProblem pb = problems[p];
    const int32_t *pairs = triples + 3 * pb.off[0], *gx = triples + 3 * pb.off[1], *gy = triples + 3 * pb.off[2];
    const int n = pb.n[0];
    const long long lX = pb.lX, lY = pb.lY;
    long long *cx = cum + pb.seqOff, *cy = cx + pb.lX;  // getCumulativeGapProbs (:1603-1619)
    for (long long i = 0; i < lX + lY; i++) cx[i] = 0;
    for (int i = 0; i < pb.n[1]; i++) cx[gx[3 * i + 1]] += gx[3 * i];
    for (int i = 0; i < pb.n[2]; i++) cy[gy[3 * i + 2]] += gy[3 * i];
    for (long long i = 1; i < lX; i++) cx[i] += cx[i - 1];
    for (long long i = 1; i < lY; i++) cy[i] += cy[i - 1];
    double *bs = best + pb.chainOff;
    int32_t *pv = prev + pb.chainOff;
    uint8_t *rec = record + pb.chainOff;
    double top = 0;
    for (int i = 0; i <= n; i++) {
        long long w, x, y;
        if (i == n) {  // sentinel behind both sequences (:1652-1654)
            w = 0;
            x = lX;
            y = lY;
        } else {
            w = pairs[3 * i];
            x = pairs[3 * i + 1];
            y = pairs[3 * i + 2];
        }
        double score = (double)((float)w + (float)(gap_mass(cx, 0, x) + gap_mass(cy, 0, y)) * gapGamma);  // :1660-1661
        int from = -1;
        for (int j = i - 1; j >= 0; j--) {
            const long long x2 = pairs[3 * j + 1], y2 = pairs[3 * j + 2];
            if (x2 < x && y2 < y) {
                const float g = (float)(gap_mass(cx, x2 + 1, x - x2 - 1) + gap_mass(cy, y2 + 1, y - y2 - 1)) * gapGamma;
                const long long sc = (long long)(((double)w + bs[j]) + (double)g);  // :1673-1675
                if ((double)sc > score) {
                    score = (double)sc;
                    from = j;
                }
                if (rec[j]) break;  // :1685
            }
        }
        pv[i] = from;
        bs[i] = score;
        const float tail = (float)((x < lX ? gap_mass(cx, x + 1, lX - x - 1) : 0) + (y < lY ? gap_mass(cy, y + 1, lY - y - 1) : 0)) * gapGamma;
        const double sc = score + (double)tail;  // :1695-1696
        rec[i] = 0;
        if (sc >= top) {
            top = sc;
            rec[i] = 1;
        }
    }
    int count = 0;
    for (int i = pv[n]; i >= 0; i = pv[i]) count++;
    int32_t *out = meaOut + 3 * pb.meaOut;
    int at = count;
    for (int i = pv[n]; i >= 0; i = pv[i]) {  // back to front == built reversed, then flipped (:1714)
        at--;
        out[3 * at] = pairs[3 * i];
        out[3 * at + 1] = pairs[3 * i + 1];
        out[3 * at + 2] = pairs[3 * i + 2];
    }
    counts[2 * p] = count;
    scores[kPostScores * p + 2] = top;
}

// The same chain with one WAVE per problem (round 3).  The lane-per-problem kernel above walks back through global memory
// one pair at a time, every lane of a wave on a list of its own: 6.07 s for 10 000 config-4 pairs beside 30 ms of DP
// (tools/mea_bench.py).  Here:
//  * the gap masses become three 64-bit words per pair, computed by the wave in parallel from the cumulative arrays (two
//    prefix sums): G = mass of the gaps in front of the pair (cum_x[x-1] + cum_y[y-1]), H = mass up to and including its
//    own row and column (cum_x[x] + cum_y[y]), T = mass behind it; the gap between pair j and a later pair i is then
//    G_i - H_j (:1621-1625 are differences of the same cumulative sums; integer arithmetic, exact);
//  * the walk back of pair i runs 64 earlier pairs at a time, lane t on pair i-1-t: the most recent 64 pairs (coordinates,
//    H, chain score, record flag) live in registers and move up one lane per pair, older ones are read from global memory,
//    coalesced.  A walk ends at the first dominated pair that is a record (:1685): the lanes up to it take part, the
//    reference's "first strictly greater" is the first lane holding the maximum;
//  * scores, predecessors and record flags of the last 64 pairs leave as coalesced stores once per 64 pairs, and the walk
//    back along the predecessors goes through LDS a tile at a time (cf. cpecan_post_ordered_wave).
// Same float / double / int64 expressions per pair as above: identical alignments and scores.
__global__ void __launch_bounds__(64) cpecan_post_mea_wave(const CpkPostProblem *problems, const int32_t *triples, long long *cum,
                                                           double *best, int32_t *prev, uint8_t *record, long long *gapG,
                                                           long long *gapH, long long *gapT, float gapGamma, int32_t *meaOut,
                                                           int32_t *counts, double *scores) {
    __shared__ int32_t tile[kPostTile];
    const int lane = threadIdx.x;
    const CpkPostProblem pb = problems[blockIdx.x];
    const int32_t *pairs = triples + 3 * pb.off[0], *gx = triples + 3 * pb.off[1], *gy = triples + 3 * pb.off[2];
    const int n = pb.n[0];
    const int lX = pb.lX, lY = pb.lY;
    long long *cx = cum + pb.seqOff, *cy = cx + lX;  // getCumulativeGapProbs (:1603-1619)
    double *bs = best + pb.chainOff;
    int32_t *pv = prev + pb.chainOff;
    uint8_t *rec = record + pb.chainOff;
    long long *G = gapG + pb.chainOff, *H = gapH + pb.chainOff, *T = gapT + pb.chainOff;
    for (int i = lane; i < lX + lY; i += CPK_WAVE) cx[i] = 0;
    __syncthreads();
    for (int i = lane; i < pb.n[1]; i += CPK_WAVE)
        atomicAdd(reinterpret_cast<unsigned long long *>(&cx[gx[3 * i + 1]]), (unsigned long long)(long long)gx[3 * i]);
    for (int i = lane; i < pb.n[2]; i += CPK_WAVE)
        atomicAdd(reinterpret_cast<unsigned long long *>(&cy[gy[3 * i + 2]]), (unsigned long long)(long long)gy[3 * i]);
    __syncthreads();
    for (int a = 0; a < 2; a++) {  // inclusive prefix sums of both arrays
        long long *c = a == 0 ? cx : cy;
        const int len = a == 0 ? lX : lY;
        long long carry = 0;
        for (int i0 = 0; i0 < len; i0 += CPK_WAVE) {
            const int i = i0 + lane;
            long long v = i < len ? c[i] : 0;
#pragma unroll
            for (int off = 1; off < CPK_WAVE; off <<= 1) {
                const long long t = __shfl_up(v, off);
                if (lane >= off) v += t;
            }
            if (i < len) c[i] = carry + v;
            carry += __shfl(v, CPK_WAVE - 1);
        }
    }
    __syncthreads();
    const long long endX = lX > 0 ? cx[lX - 1] : 0, endY = lY > 0 ? cy[lY - 1] : 0;
    for (int i = lane; i <= n; i += CPK_WAVE) {
        if (i == n) {  // the sentinel behind both sequences (:1652-1654)
            G[i] = endX + endY;
            H[i] = 0;
            T[i] = 0;
        } else {
            const int x = pairs[3 * i + 1], y = pairs[3 * i + 2];
            G[i] = (x > 0 ? cx[x - 1] : 0) + (y > 0 ? cy[y - 1] : 0);
            H[i] = cx[x] + cy[y];
            T[i] = (endX - cx[x]) + (endY - cy[y]);
        }
    }
    __syncthreads();
    // ---- the chain (:1656-1702)
    double top = 0;
    // the 64 pairs in front of pair i, lane t <-> pair i-1-t (rj < 0: no such pair)
    int rx = 0, ry = 0, rrec = 0, rj = -1;
    long long rH = 0;
    double rbs = 0.0;
    // results of the current chunk of 64 pairs, lane (i - base): stored once per chunk
    double oBs = 0.0;
    int oPv = 0, oRec = 0;
    for (int base = 0; base <= n; base += CPK_WAVE) {
        const int cnt = n + 1 - base < CPK_WAVE ? n + 1 - base : CPK_WAVE;
        // the chunk's pairs: lane k holds pair base + k
        int cw = 0, cx2 = lX, cy2 = lY;
        long long cG, cH, cT;
        {
            const int i = base + lane <= n ? base + lane : n;
            if (i < n) {
                cw = pairs[3 * i];
                cx2 = pairs[3 * i + 1];
                cy2 = pairs[3 * i + 2];
            }
            cG = G[i];
            cH = H[i];
            cT = T[i];
            asm volatile("" ::"v"(cw), "v"(cx2), "v"(cy2), "v"(cG), "v"(cH), "v"(cT));
        }
        for (int k = 0; k < cnt; k++) {
            const int i = base + k;
            const int w = __builtin_amdgcn_readlane(cw, k), x = __builtin_amdgcn_readlane(cx2, k), y = __builtin_amdgcn_readlane(cy2, k);
            const long long Gi = __shfl(cG, k), Hi = __shfl(cH, k), Ti = __shfl(cT, k);
            double score = (double)((float)w + (float)Gi * gapGamma);  // :1660-1661
            int from = -1;
            // The walk back, 64 pairs at a time: the ones in registers first, then older ones from memory, two chunks of
            // loads in flight.  Every lane keeps the best of ITS pairs (strictly greater replaces: the more recent pair
            // stays on a tie); the wave's maximum is formed once, behind the walk.
            double bestV = -__builtin_huge_val();
            int bestJ = -1;
            auto take = [&](int j, int x2, int y2, int rc, long long Hj, double bj) {  // returns true when the walk ends in this chunk
                const bool dom = j >= 0 && x2 < x && y2 < y;
                const unsigned long long stops = __ballot(dom && rc != 0);  // :1685
                const int stopLane = stops ? __builtin_ctzll(stops) : CPK_WAVE;
                const float g = (float)(Gi - Hj) * gapGamma;
                const long long sc = (long long)(((double)w + bj) + (double)g);  // :1673-1675
                if (dom && lane <= stopLane && (double)sc > bestV) {
                    bestV = (double)sc;
                    bestJ = j;
                }
                return stops != 0;
            };
            struct Older {
                int x2, y2, rc, j;
                long long Hj;
                double bj;
            };
            auto fetch = [&](int c) {  // pairs i-1-64c-lane
                Older o;
                o.j = i - 1 - (c * CPK_WAVE + lane);
                const int jj = o.j >= 0 ? o.j : 0;
                o.x2 = pairs[3 * jj + 1];
                o.y2 = pairs[3 * jj + 2];
                o.rc = rec[jj];
                o.Hj = H[jj];
                o.bj = bs[jj];
                return o;
            };
            if (!take(rj, rx, ry, rrec, rH, rbs) && i - 1 - CPK_WAVE >= 0) {
                Older o1 = fetch(1), o2 = fetch(2);  // (a chunk past the front of the list reads pair 0 and takes nothing)
                for (int c = 1;; c += 2) {
                    asm volatile("" ::"v"(o1.x2), "v"(o1.y2), "v"(o1.rc), "v"(o1.Hj), "v"(o1.bj));
                    if (take(o1.j, o1.x2, o1.y2, o1.rc, o1.Hj, o1.bj) || i - 1 - (c + 1) * CPK_WAVE < 0) break;
                    o1 = fetch(c + 2);
                    asm volatile("" ::"v"(o2.x2), "v"(o2.y2), "v"(o2.rc), "v"(o2.Hj), "v"(o2.bj));
                    if (take(o2.j, o2.x2, o2.y2, o2.rc, o2.Hj, o2.bj) || i - 1 - (c + 2) * CPK_WAVE < 0) break;
                    o2 = fetch(c + 3);
                }
            }
            {
                double m = bestV;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
                if (m > score) {  // the first pair of the walk -- the most recent one -- that holds the maximum
                    score = m;
                    from = wave_max_i32(bestV == m ? bestJ : -1);
                }
            }
            const float tail = (float)Ti * gapGamma;
            const double scEnd = score + (double)tail;  // :1695-1696
            int r = 0;
            if (scEnd >= top) {
                top = scEnd;
                r = 1;
            }
            if (lane == k) {
                oBs = score;
                oPv = from;
                oRec = r;
            }
            // pair i joins the front of the register window
            rx = __shfl_up(rx, 1);
            ry = __shfl_up(ry, 1);
            rrec = __shfl_up(rrec, 1);
            rj = __shfl_up(rj, 1);
            rH = __shfl_up(rH, 1);
            rbs = __shfl_up(rbs, 1);
            if (lane == 0) {
                rx = x;
                ry = y;
                rrec = r;
                rj = i;
                rH = Hi;
                rbs = score;
            }
        }
        if (lane < cnt) {
            bs[base + lane] = oBs;
            pv[base + lane] = oPv;
            rec[base + lane] = (uint8_t)oRec;
        }
    }
    __syncthreads();
    // ---- the alignment (:1704-1714): the chain from the sentinel's predecessor, marked in place (p -> -3 - p <= -2)
    for (int k = n; k >= 0;) {
        const int t0 = k - (kPostTile - 1) > 0 ? k - (kPostTile - 1) : 0;
        for (int j = lane; j <= k - t0; j += CPK_WAVE) tile[j] = pv[t0 + j];
        __syncthreads();
        const int hi = k;
        bool first = k == n;
        while (k >= t0) {
            const int p = tile[k - t0];
            if (!first && lane == 0) tile[k - t0] = -3 - p;  // (the sentinel itself is not part of the alignment)
            first = false;
            k = p;
        }
        __syncthreads();
        for (int j = lane; j <= hi - t0; j += CPK_WAVE) pv[t0 + j] = tile[j];
        __syncthreads();
    }
    int32_t *out = meaOut + 3 * pb.meaOut;
    int count = 0;
    for (int i0 = 0; i0 < n; i0 += CPK_WAVE) {  // back to front == built reversed, then flipped (:1714): ascending pairs
        const int i = i0 + lane;
        const bool sel = i < n && pv[i] <= -2;
        const unsigned long long mask = __ballot(sel);
        if (sel) {
            const int at = count + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            out[3 * at] = pairs[3 * i];
            out[3 * at + 1] = pairs[3 * i + 1];
            out[3 * at + 2] = pairs[3 * i + 2];
        }
        count += __popcll(mask);
    }
    if (lane == 0) {
        counts[2 * blockIdx.x] = count;
        scores[kPostScores * blockIdx.x + 2] = top;
    }
}

// LEFT_SHIFT without MEA: list 0 is the chain to shift; put it where the MEA stage would have put its alignment.
__global__ void __launch_bounds__(256) cpecan_post_copy_chain(const CpkPostProblem *problems, const int32_t *triples,
                                                              int32_t *meaOut, int32_t *counts) {
    const CpkPostProblem pb = problems[blockIdx.x];
    const int32_t *src = triples + 3 * pb.off[0];
    int32_t *dst = meaOut + 3 * pb.meaOut;
    for (int i = threadIdx.x; i < 3 * pb.n[0]; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) counts[2 * blockIdx.x] = pb.n[0];
}

// leftShiftAlignment (:1726-1762), one lane per problem, on the MEA alignment.  chars: raw upper-case sequences.
__global__ void __launch_bounds__(64) cpecan_post_left_shift(const CpkPostProblem *problems, int64_t nProblems,
                                                             const int32_t *mea, const uint8_t *chars, int32_t *shiftOut,
                                                             int32_t *counts) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nProblems) return;
    const CpkPostProblem pb = problems[p];
    const int32_t *pairs = mea + 3 * pb.meaOut;
    const int n = counts[2 * p];
    const uint8_t *sX = chars + pb.charX, *sY = chars + pb.charY;
    int32_t *out = shiftOut + 3 * pb.shiftOut;
    int count = 0;
    int x = pb.lX, y = pb.lY;
    for (int i = n - 1; i >= 0; i--) {
        const int w = pairs[3 * i], x2 = pairs[3 * i + 1], y2 = pairs[3 * i + 2];
        while ((x - x2 > 1 || y - y2 > 1) && sX[x - 1] == sY[y - 1]) {  // :1737-1744
            out[3 * count] = w;
            out[3 * count + 1] = x - 1;
            out[3 * count + 2] = y - 1;
            count++;
            x--;
            y--;
            if (x2 == x || y2 == y) break;
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
    const int first = n > 0 ? pairs[0] : 1;  // :1754
    while (x > 0 && y > 0 && sX[x - 1] == sY[y - 1]) {
        out[3 * count] = first;
        out[3 * count + 1] = x - 1;
        out[3 * count + 2] = y - 1;
        count++;
        x--;
        y--;
    }
    for (int a = 0, b = count - 1; a < b; a++, b--)  // :1759
        for (int f = 0; f < 3; f++) {
            const int32_t t = out[3 * a + f];
            out[3 * a + f] = out[3 * b + f];
            out[3 * b + f] = t;
        }
    counts[2 * p + 1] = count;
}

// filterPairwiseAlignmentToMakePairsOrdered (impl/multipleAligner.c:945-972), one lane per problem: the heaviest chain
// of the pairs whose weight reaches matchGamma, as a filter of list 0.  With two sequences the reference's
// pairwiseAlignColumns (:358-492) keeps a frontier of chain ends sorted by y with strictly increasing scores; the
// predecessor it gives a pair is the frontier entry with the largest y below it (:397), and an entry is dropped for
// one that is maximal under (score, then smaller y, then later column) (:418-427).
// That frontier is kept here as what it is: a STAIRCASE, the chain ends that no other end dominates, sorted by y with
// strictly growing scores (stairY / stairI).  An alignment is close to monotone, so nearly every pair finds its
// predecessor in the last entry and is appended behind it -- two or three memory operations of this lane, where the
// Fenwick tree over y of rounds 1-2 (prefix maxima, nothing ever deleted) walked ~2 log2(lY) nodes and their scores for
// every pair: 38.6 ms for the 50 000 cigars of the realign benchmark, more than the DP kernels.
// Pairs of one X column are all scored before any of them is inserted (:389-409).
// The reference's st_random() * 0.00001 per weight (:145) is left out.
// Hand-off between the lanes of ONE wave through memory (cpecan_post_ordered_wave: lane 0, or one lane per entry,
// stores; all 64 lanes load the same words later).  The hardware completes a wave's accesses to one address in order;
// the fence makes the compiler keep that order too -- no load is hoisted over, and no store sunk below it (ADVICE r3:
// without it, load-PRE over `if (writer) stairY[pos] = y; ... reloadLast()` could hand the other lanes the old value and
// the wave's uniform control flow would diverge).  Wavefront scope: no instruction is emitted for it.
__device__ __forceinline__ void wave_handoff_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct OrderedStairs {
    const int32_t *pairs;
    const double *best;
    int32_t *stairY, *stairI;  // y and pair index of the entries, ascending y (at most one entry per y: lY words each)
    int len;
    int lastY, lastI;  // the last entry, in registers: the usual pair needs nothing else of the staircase
    double lastS;
    bool writer = true;  // false: the lanes of a wave run the staircase in step and only lane 0 stores (cpecan_post_ordered_wave)
    // (A value loaded on a rare path is CONSUMED there -- the empty asm: where such a path joins the usual one hipcc
    // otherwise waits with vmcnt(0) for a register that may have been loaded, i.e. on EVERY pair, and in the wave kernel
    // that wait includes the acknowledgements of the stores of the pair before: measured 1.4 us a pair.)
    __device__ __forceinline__ void reloadLast() {
        lastY = len > 0 ? stairY[len - 1] : -1;
        lastI = len > 0 ? stairI[len - 1] : -1;
        lastS = len > 0 ? best[lastI] : 0.0;
        asm volatile("" ::"v"(lastY), "v"(lastI), "v"(lastS));
    }
    // number of entries with y' < y
    __device__ __forceinline__ int below(int y) const {
        if (len == 0 || lastY < y) return len;  // the usual case: the pair extends the alignment
        wave_handoff_fence();  // the entries read below may have been stored by another lane of this wave
        int lo = 0, hi = len - 1;                         // stairY[hi] >= y
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (stairY[mid] < y) lo = mid + 1;
            else hi = mid;
        }
        asm volatile("" ::"v"(lo));
        return lo;
    }
    // best chain end with y' < y: the last entry below y (the highest score; of equal scores only the smaller y is kept)
    __device__ __forceinline__ int query(int y) const {
        const int pos = below(y);
        if (pos == 0) return -1;
        if (pos == len) return lastI;
        const int i = stairI[pos - 1];
        asm volatile("" ::"v"(i));
        return i;
    }
    // pair i (end y, chain score si, of the current column -- later than every entry's) becomes a chain end
    __device__ __forceinline__ void insert(int y, int i, double si) {
        const int pos = below(y);
        // an end at a smaller y with at least this score is preferred wherever both are candidates
        if (pos > 0) {
            double before = lastS;
            if (pos != len) {
                before = best[stairI[pos - 1]];
                asm volatile("" ::"v"(before));
            }
            if (before >= si) return;
        }
        if (pos == len) {  // appended behind everything: no entry to compare with, nothing to move
            if (writer) {
                stairY[len] = y;
                stairI[len] = i;
            }
            len++;
            lastY = y;
            lastI = i;
            lastS = si;
            return;
        }
        int q = pos;
        if (q < len && stairY[q] == y) {  // the same y: the higher score, then the later column (this pair)
            if (best[stairI[q]] > si) return;
            q++;
        }
        while (q < len && best[stairI[q]] <= si) q++;  // ends at a larger y that do not beat this score are dominated
        // entries [pos, q) go, the pair takes position pos, the tail [q, len) follows it
        const int shift = pos + 1 - q;
        if (shift > 0) {
            for (int k = len - 1; k >= q; k--) {
                const int ty = stairY[k], ti = stairI[k];
                if (writer) {
                    stairY[k + 1] = ty;
                    stairI[k + 1] = ti;
                }
            }
        } else if (shift < 0) {
            for (int k = q; k < len; k++) {
                const int ty = stairY[k], ti = stairI[k];
                if (writer) {
                    stairY[k + shift] = ty;
                    stairI[k + shift] = ti;
                }
            }
        }
        if (writer) {
            stairY[pos] = y;
            stairI[pos] = i;
        }
        len += shift;
        wave_handoff_fence();  // the writer lane's stores above, every lane's loads below
        reloadLast();
    }
};

__global__ void __launch_bounds__(64) cpecan_post_ordered(const CpkPostProblem *problems, int64_t nProblems,
                                                          const int32_t *triples, int32_t *seqScratch, double *best,
                                                          int32_t *prev, int32_t *next, uint8_t *chosen, double matchGamma,
                                                          int32_t *out, int32_t *counts) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nProblems) return;
    const CpkPostProblem pb = problems[p];
    const int32_t *pairs = triples + 3 * pb.off[0];
    const int n = pb.n[0], lX = pb.lX, lY = pb.lY;
    // seqScratch: lX + 2 lY words of this problem (post_layout): first pair of each X column, then the staircase
    int32_t *head = seqScratch + pb.seqOff;
    double *bs = best + pb.chainOff;
    int32_t *pv = prev + pb.chainOff, *nx = next + pb.chainOff;
    uint8_t *ch = chosen + pb.chainOff;
    for (int i = 0; i < lX; i++) head[i] = -1;
    for (int i = n - 1; i >= 0; i--) {
        const int x = pairs[3 * i + 1];
        nx[i] = head[x];
        head[x] = i;
        ch[i] = 0;
    }
    OrderedStairs st{pairs, bs, head + lX, head + lX + lY, 0, -1, -1, 0.0};
    for (int x = 0; x < lX; x++) {
        bool any = false;
        for (int i = head[x]; i >= 0; i = nx[i]) {
            const double w = (double)pairs[3 * i] / (double)CPECAN_PROB_1;
            if (w >= matchGamma && w > 0.0) {  // :393
                const int from = st.query(pairs[3 * i + 2]);  // y' < y
                pv[i] = from;
                bs[i] = (from < 0 ? 0.0 : bs[from]) + w * 1.0;  // :404
                ch[i] = 2;
                any = true;
            }
        }
        if (!any) continue;
        for (int i = head[x]; i >= 0; i = nx[i])
            if (ch[i] == 2) st.insert(pairs[3 * i + 2], i, bs[i]);
    }
    const int last = st.lastI;  // the best chain end over every y (-1: no pair reached matchGamma)
    for (int i = last; i >= 0; i = pv[i]) ch[i] = 1;  // :437-475
    int32_t *o = out + 3 * pb.meaOut;
    int count = 0;
    for (int i = n - 1; i >= 0; i--)  // the list conversions of :621-651 and :582 reverse the list three times
        if (ch[i] == 1) {
            o[3 * count] = pairs[3 * i];
            o[3 * count + 1] = pairs[3 * i + 1];
            o[3 * count + 2] = pairs[3 * i + 2];
            count++;
        }
    counts[2 * p] = count;
}

// The same filter with one WAVE per problem (round 3).  One lane per problem left every step of the chain a round trip to
// global memory of a lane of its own (64 scattered accesses per wave instruction): 21.7 ms for the 50 000 cigars of the
// realign benchmark, as long as the DP kernels.  Here the parallel parts are the wave's -- the pairs that reach
// matchGamma are counted per X column (LDS atomics), the columns' starts come from a prefix sum, the pairs are scattered
// into column order (kept in list order inside a column), the chosen pairs are compacted -- and the chain itself, which
// is a sequence, runs on values the wave already holds: 64 sorted pairs at a time in registers (read by lane index), the
// last staircase entry in registers, the scores of a column's pairs in LDS; scores, predecessors and staircase entries
// go to global memory as stores nobody waits for.  Every lane runs the chain in step with the same values and lane 0
// alone stores; a wave's memory operations complete in order, so what lane 0 stored is what every lane reads back.
// The walk back along the predecessors goes through LDS a tile of kPostTile places at a time (a predecessor is always an
// earlier place).  Same arithmetic and tie-breaks as cpecan_post_ordered: identical lists.
__global__ void __launch_bounds__(64) cpecan_post_ordered_wave(const CpkPostProblem *problems, const int32_t *triples,
                                                               int32_t *seqScratch, double *best, int32_t *prev, int32_t *sortI,
                                                               int32_t *sortX, int32_t *sortY, double *sortW, uint8_t *chosen,
                                                               double matchGamma, int32_t *out, int32_t *counts) {
    __shared__ int32_t tile[kPostTile];
    __shared__ double colS[CPK_WAVE];  // scores and y of the pairs of the column being scored (a ring: longer columns re-read memory)
    __shared__ int32_t colY[CPK_WAVE];
    const int lane = threadIdx.x;
    const CpkPostProblem pb = problems[blockIdx.x];
    const int32_t *pairs = triples + 3 * pb.off[0];
    const int n = pb.n[0], lX = pb.lX, lY = pb.lY;
    int32_t *gcol = seqScratch + pb.seqOff;  // lX words: pairs per X column, then the columns' fill pointers
    int32_t *stairY = gcol + lX, *stairI = stairY + lY;
    int32_t *col = lX <= kPostTile ? tile : gcol;
    double *bs = best + pb.chainOff;
    int32_t *pv = prev + pb.chainOff;
    int32_t *sI = sortI + pb.chainOff, *sX = sortX + pb.chainOff, *sY = sortY + pb.chainOff;
    double *sW = sortW + pb.chainOff;  // the pair's weight as the chain adds it: score / PROB_1 (:393), divided once here, by a lane of its own
    uint8_t *ch = chosen + pb.chainOff;
    auto qualifies = [&](int w) {  // :393
        const double wd = (double)w / (double)CPECAN_PROB_1;
        return wd >= matchGamma && wd > 0.0;
    };
    for (int x = lane; x < lX; x += CPK_WAVE) col[x] = 0;
    for (int i = lane; i < n; i += CPK_WAVE) ch[i] = 0;
    __syncthreads();
    for (int i = lane; i < n; i += CPK_WAVE)
        if (qualifies(pairs[3 * i])) atomicAdd(&col[pairs[3 * i + 1]], 1);
    __syncthreads();
    int m = 0;  // pairs that take part
    for (int x0 = 0; x0 < lX; x0 += CPK_WAVE) {  // exclusive prefix sum over the columns
        const int x = x0 + lane;
        const int c = x < lX ? col[x] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < CPK_WAVE; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (x < lX) col[x] = m + incl - c;
        m += __shfl(incl, CPK_WAVE - 1);
    }
    __syncthreads();
    for (int i = lane; i < n; i += CPK_WAVE) {
        const int w = pairs[3 * i];
        if (qualifies(w)) {
            const int x = pairs[3 * i + 1];
            const int pos = atomicAdd(&col[x], 1);
            sI[pos] = i;
            sX[pos] = x;
            sY[pos] = pairs[3 * i + 2];
            sW[pos] = (double)w / (double)CPECAN_PROB_1;
        }
    }
    __syncthreads();
    // inside a column the pairs go back into list order (the atomics hand out places in any order): the lane of a
    // column's first place sorts it by insertion -- columns hold a pair or two
    for (int k0 = 0; k0 < m; k0 += CPK_WAVE) {
        const int k = k0 + lane;
        if (k < m && (k == 0 || sX[k - 1] != sX[k])) {
            const int x = sX[k];
            int e = k + 1;
            while (e < m && sX[e] == x) e++;
            for (int a2 = k + 1; a2 < e; a2++) {
                const int i = sI[a2], y = sY[a2];
                const double w = sW[a2];
                int b = a2 - 1;
                while (b >= k && sI[b] > i) {
                    sI[b + 1] = sI[b];
                    sY[b + 1] = sY[b];
                    sW[b + 1] = sW[b];
                    b--;
                }
                sI[b + 1] = i;
                sY[b + 1] = y;
                sW[b + 1] = w;
            }
        }
    }
    __syncthreads();
    // ---- the chain (:389-427), column by column: every pair of a column is scored before any of them is inserted.
    // Nothing in the usual step touches memory: the 64 sorted pairs of the current chunk are in registers (lane j <-> place
    // base + j), their scores and predecessors are collected in registers the same way (v_writelane) and leave as one
    // coalesced store per chunk, appended staircase entries likewise 64 at a time, a column's scores wait for its
    // inserts in registers (one pair) or LDS.  A step that needs the staircase in memory (a pair that does not extend
    // the alignment: binary search, entries dropped or moved) first flushes what the registers hold.
    // (With a store per pair the loop ran at 1.4 us a pair: hipcc waits with vmcnt(0) wherever a rarely loaded value may
    // join the usual path, and that wait includes the acknowledgement of the store just issued.)
    OrderedStairs st{pairs, bs, stairY, stairI, 0, -1, -1, 0.0, lane == 0};
    int cx = 0, cy = 0, cwLo = 0, cwHi = 0;  // the chunk's pairs
    int pvR = 0, bsLo = 0, bsHi = 0;    // ... their predecessors and scores
    int syR = 0, siR = 0, sBase = 0;    // staircase entries [sBase, st.len) that are not in memory yet: lane j <-> entry sBase + j
    int base = 0, done = 0;             // the chunk's first place; places of it scored so far
    auto flushStairs = [&]() {
        if (sBase + lane < st.len) {
            stairY[sBase + lane] = syR;
            stairI[sBase + lane] = siR;
        }
        sBase = st.len;
    };
    auto flushChunk = [&]() {
        if (lane < done) {
            pv[base + lane] = pvR;
            bs[base + lane] = __hiloint2double(bsHi, bsLo);
        }
    };
    auto scoreOf = [&](int from) {  // chain score of an earlier place
        if (from >= base) {
            const int j = __builtin_amdgcn_readfirstlane(from - base);
            return __hiloint2double(__builtin_amdgcn_readlane(bsHi, j), __builtin_amdgcn_readlane(bsLo, j));
        }
        const double v = bs[from];
        asm volatile("" ::"v"(v));
        return v;
    };
    auto insert = [&](int y, int k, double sc) {
        if (st.len == 0 || st.lastY < y) {  // the pair extends the alignment: appended, or dominated by the last end
            if (st.len > 0 && st.lastS >= sc) return;
            if (st.len - sBase == CPK_WAVE) flushStairs();
            const int j = __builtin_amdgcn_readfirstlane(st.len - sBase);
            syR = (lane == j ? y : syR);
            siR = (lane == j ? k : siR);
            st.len++;
            st.lastY = y;
            st.lastI = k;
            st.lastS = sc;
            return;
        }
        flushStairs();
        flushChunk();
        wave_handoff_fence();  // st.insert reads entries and scores other lanes have just stored
        st.insert(y, k, sc);
        sBase = st.len;
    };
    int colX = -1, colK = 0, colN = 0, firstY = 0;  // the column being scored: its x, first place, pairs so far
    double firstS = 0.0;
    auto finishColumn = [&]() {
        if (colN == 1) {
            insert(firstY, colK, firstS);
        } else if (colN > 1) {
            wave_handoff_fence();  // lane 0 filled colY / colS, every lane reads them
            const bool ring = colN <= CPK_WAVE;  // else the ring has wrapped: re-read what the chunk stores hold
            if (!ring) {
                flushChunk();
                __syncthreads();
            }
            for (int q = 0; q < colN; q++) {
                const int yq = ring ? colY[q] : sY[colK + q];
                const double sq = ring ? colS[q] : bs[colK + q];
                insert(yq, colK + q, sq);
            }
        }
    };
    for (base = 0; base < m; base += CPK_WAVE) {
        {
            const int k = base + lane < m ? base + lane : m - 1;
            cx = sX[k];
            cy = sY[k];
            const double wk = sW[k];
            cwLo = __double2loint(wk);
            cwHi = __double2hiint(wk);
            asm volatile("" ::"v"(cx), "v"(cy), "v"(cwLo), "v"(cwHi));
        }
        const int cnt = m - base < CPK_WAVE ? m - base : CPK_WAVE;
        for (done = 0; done < cnt; done++) {
            const int x = __builtin_amdgcn_readlane(cx, done), y = __builtin_amdgcn_readlane(cy, done);
            const double w = __hiloint2double(__builtin_amdgcn_readlane(cwHi, done), __builtin_amdgcn_readlane(cwLo, done));
            if (x != colX) {
                finishColumn();
                colX = x;
                colK = base + done;
                colN = 0;
            }
            int from;  // best chain end with y' < y
            if (st.len == 0) {
                from = -1;
            } else if (st.lastY < y) {
                from = st.lastI;
            } else {
                flushStairs();
                wave_handoff_fence();
                from = st.query(y);
            }
            const double sFrom = from < 0 ? 0.0 : (from == st.lastI ? st.lastS : scoreOf(from));
            const double sc = sFrom + w * 1.0;  // :404
            pvR = (lane == done ? from : pvR);
            bsLo = (lane == done ? __double2loint(sc) : bsLo);
            bsHi = (lane == done ? __double2hiint(sc) : bsHi);
            if (colN == 0) {  // (most columns hold one pair: it stays in registers)
                firstY = y;
                firstS = sc;
            } else if (lane == 0) {
                if (colN == 1) {
                    colY[0] = firstY;
                    colS[0] = firstS;
                }
                colY[colN & (CPK_WAVE - 1)] = y;
                colS[colN & (CPK_WAVE - 1)] = sc;
            }
            colN++;
        }
        done = cnt;
        flushChunk();
        wave_handoff_fence();  // scoreOf() of a later chunk reads these scores back, any lane's
    }
    done = 0;  // (nothing of a chunk is pending any more)
    finishColumn();
    flushStairs();
    __syncthreads();
    // ---- the chain's pairs (:437-475): walked back from the best end and marked in place (predecessor p -> -3 - p <= -2)
    for (int k = st.lastI; k >= 0;) {
        const int t0 = k - (kPostTile - 1) > 0 ? k - (kPostTile - 1) : 0;
        for (int j = lane; j <= k - t0; j += CPK_WAVE) tile[j] = pv[t0 + j];
        __syncthreads();
        const int top = k;
        while (k >= t0) {
            const int p = tile[k - t0];
            if (lane == 0) tile[k - t0] = -3 - p;
            k = p;
        }
        __syncthreads();
        for (int j = lane; j <= top - t0; j += CPK_WAVE) pv[t0 + j] = tile[j];
        __syncthreads();
    }
    __syncthreads();
    for (int k = lane; k < m; k += CPK_WAVE)
        if (pv[k] <= -2) ch[sI[k]] = 1;
    __syncthreads();
    int32_t *o = out + 3 * pb.meaOut;
    int count = 0;
    for (int top = n; top > 0; top -= CPK_WAVE) {  // the list conversions of :621-651 and :582 reverse the list three times
        const int i = top - 1 - lane;
        const bool sel = i >= 0 && ch[i] == 1;
        const unsigned long long mask = __ballot(sel);
        if (sel) {
            const int at2 = count + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            o[3 * at2] = pairs[3 * i];
            o[3 * at2 + 1] = pairs[3 * i + 1];
            o[3 * at2 + 2] = pairs[3 * i + 2];
        }
        count += __popcll(mask);
    }
    if (lane == 0) counts[2 * blockIdx.x] = count;
}

// Scores of the final list (list 0, or the ordered alignment when fromOut): scoreByPosteriorProbability[IgnoringGaps]
// (:1578-1597) when posterior != 0, scoreByIdentity[IgnoringGaps] (:1562-1580) when chars are given.  One workgroup per
// problem.  chars are upper-case already.
__global__ void __launch_bounds__(256) cpecan_post_list_scores(const CpkPostProblem *problems, const int32_t *triples,
                                                               const int32_t *out, const int32_t *counts, int fromOut,
                                                               int posterior, const uint8_t *chars, double *scores) {
    const CpkPostProblem pb = problems[blockIdx.x];
    const int32_t *t = fromOut ? out + 3 * pb.meaOut : triples + 3 * pb.off[0];
    const int n = fromOut ? counts[2 * blockIdx.x] : pb.n[0];
    __shared__ long long partial[2][256];
    long long sum = 0, matches = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        sum += t[3 * i];
        if (chars) {
            const uint8_t a = chars[pb.charX + t[3 * i + 1]], b = chars[pb.charY + t[3 * i + 2]];
            matches += (a == b) & (a != 'N');
        }
    }
    partial[0][threadIdx.x] = sum;
    partial[1][threadIdx.x] = matches;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            partial[0][threadIdx.x] += partial[0][threadIdx.x + off];
            partial[1][threadIdx.x] += partial[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long L = (long long)pb.lX + pb.lY;
        double *s = scores + kPostScores * blockIdx.x;
        if (posterior) {
            const double total = (double)partial[0][0];
            s[0] = 100.0 * (L == 0 ? 0 : (2.0 * total) / (double)(L * CPECAN_PROB_1));
            s[1] = 100.0 * total / ((double)n * CPECAN_PROB_1);
        }
        if (chars) {
            const long long m = partial[1][0];
            s[3] = 100.0 * (L == 0 ? 0 : (2.0 * (double)m) / (double)L);
            s[4] = 100.0 * (double)m / (double)n;
        }
    }
}
