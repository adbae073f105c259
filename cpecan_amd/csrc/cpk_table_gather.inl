// cpk_table_gather.inl -- device-side planning and result plumbing: the per-diagonal band table, the list-order gather.
// Part of the single HIP translation unit cpecan_kernels.hip (included there, in this order); not compiled on its own.

// The per-diagonal table the sweeps read, built on the device: one thread per region walks its band with the host's
// own iterator (cpecan_band.inl; the host has already validated the anchors with it) and writes
// {x-y of the first cell, width, position in the region's forward ring, cells on earlier diagonals}.  The ring
// position follows the rule the kernels rely on: diagonals are laid end to end and never straddle the ring's end.
// (In a pipeline of batches this kernel of batch k+1 runs on that batch's own stream; its 56 allocated VGPRs do not fit
// in the 48 a resident sweep of batch k leaves free per lane, so it starts when that sweep drains: ~3 ms per config-B batch.)
// ---- positions for the absolute-position sweeps (Sweep::forwardStreamAbs / tracebackAbs, cpk_sweep.inl) ----
// Those sweeps keep the rolling rows indexed by a cell's matrix diagonal instead of its rank on the anti-diagonal:
// position p = (xmy - B) >> 1 for an even base B that changes rarely, so that a cell's three neighbours sit at constant
// offsets (d even: lower p-1, upper p; d odd: lower p, upper p+1; middle p) and out-of-band neighbours read the -inf a
// position holds unless a band cell wrote it.  Two things need care, and both are decided HERE, per diagonal and per
// sweep direction, so that the kernels only follow a flag:
//  * the positions of three consecutive diagonals must fit the rows: [1, P - 2] with P = maxWidth + kAbsSlack positions
//    (0 and P - 1 stay -inf);
//  * a position vacated by the band keeps its last value.  Inside one rectangle of the band an in-band cell never has an
//    out-of-band neighbour that was in the band before (such neighbours lie left of / below the rectangle); across
//    rectangles it can -- exactly when an edge of the band turns back over positions it has left: the low edge after
//    rising, the high edge after falling (with a fixed expansion both edges move by exactly 1 per diagonal).  At such a
//    turn, or when an edge jumps, the sweep re-bases: it moves the two live diagonals to the new positions and wipes
//    everything else (Sweep::absRebase).
// dpos[d] = posF | flagF << 15 | posB << 16 | flagB << 31: position of the diagonal's first cell under the base of the
// forward (ascending) / backward (descending) sweep when it computes d, and whether the base changed in front of d.
#ifndef CPK_ABS_SLACK
#define CPK_ABS_SLACK 6
#endif
constexpr int kAbsSlack = CPK_ABS_SLACK;  // three consecutive diagonals span at most maxWidth + 3 positions, one -inf position at either end, one to spare
__device__ __forceinline__ int abs_chain_step(int lo, int hi, int lo1, int hi1, int lo2, int hi2, int have, int P, int &B) {
    // lo/hi: x-y range of the diagonal about to be computed; lo1/hi1, lo2/hi2: the one / two diagonals before it in sweep
    // order (have = how many of them exist).  Returns the position of the first cell, bit 15 set when the base moved.
    int needLo = lo, needHi = hi;
    bool danger = have == 0;
    if (have >= 1) {
        needLo = lo1 < needLo ? lo1 : needLo;
        needHi = hi1 > needHi ? hi1 : needHi;
        const int dl = lo - lo1, dh = hi - hi1;
        danger = danger || (dl != 1 && dl != -1) || (dh != 1 && dh != -1);
        if (have >= 2) {
            needLo = lo2 < needLo ? lo2 : needLo;
            needHi = hi2 > needHi ? hi2 : needHi;
            danger = danger || (lo < lo1 && lo1 > lo2) || (hi > hi1 && hi1 < hi2);  // an edge turns back over vacated positions
        }
    }
    const bool fits = ((needLo - B) >> 1) >= 1 && ((needHi - B) >> 1) <= P - 2;
    int flag = 0;
    if (danger || !fits) {
        const int span = ((needHi - needLo) >> 1) + 2;  // positions the three diagonals cover, mixed parity included
        int slack = (P - 2) - span;
        if (slack < 0) slack = 0;
        B = needLo - 2 * (1 + slack / 2);
        B -= B & 1;  // even (two's complement: rounds towards -infinity)
        flag = 1;
    }
    return ((lo - B) >> 1) | (flag << 15);
}

// Anchors handed over as runs of diagonal neighbours (round 4: a realign-style batch keeps its runs on the host, 16 bytes each
// instead of 8 per column): run (x, y, length, first) becomes the anchors first .. first + length - 1 the builders below read.
__global__ void __launch_bounds__(256) cpecan_expand_runs(const int4 *runs, int64_t nRuns, cpk_anchor_t *anchors) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nRuns; i += (int64_t)gridDim.x * blockDim.x) {
        const int4 r = runs[i];
        int2 *out = reinterpret_cast<int2 *>(anchors) + (size_t)(unsigned)r.w;
        for (int q = 0; q < r.z; q++) out[q] = int2{r.x + q, r.y + q};
    }
}

__global__ void __launch_bounds__(64) cpecan_build_diag_table(const CpkRegion *regions, int nRegions, const cpk_anchor_t *anchors, int anchorStride,
                                                              const CpkSegment *segs, int S, CpkDiag *diags, int32_t *dpos, int64_t expansion, int dynamic,
                                                              int skipSplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRegions) return;
    const CpkRegion rg = regions[i];
    if (rg.split && skipSplit) return;  // its table is cpecan_build_diag_table_wave's (one wave per region, below)
    CpkDiag *table = diags + rg.diagOff;
    const int64_t N = (int64_t)rg.lX + rg.lY;
    CpkBandIter it;
    cpk_band_init(&it, anchors + (size_t)anchorStride * rg.anchorOff, anchorStride, rg.nAnchors, rg.lX, rg.lY, expansion, dynamic);
    int32_t cells = 0, pos = 0;
    // A split region's ring (rg.split: match emitter, a ring of the region's own that never wraps) counts DOUBLES and
    // gives a diagonal S doubles per cell only where the sweep stores every state -- the predicate of the forward loop in
    // cpecan_pairhmm_sweep: diagonal 0, the refresh diagonals of the segment that emits it, the two diagonals below a
    // segment's top -- and one double per cell (the match row) elsewhere.  The other rings count cells of S doubles.
    const CpkSegment *sg = segs + rg.segOff;
    int emitSeg = 0, covSeg = 0;
    int emitFrom = rg.nSeg > 0 ? sg[0].tbFrom : 0, covTop = rg.nSeg > 0 ? sg[0].dTop : 0;
    auto put = [&](int64_t d, int32_t lo, int32_t w) {
        CpkDiag e;
        e.xmyL = lo;
        e.width = w;
        e.cellOff = cells;
        if (rg.split) {
            while (d > emitFrom && emitSeg + 1 < rg.nSeg) emitFrom = sg[++emitSeg].tbFrom;
            while (d > covTop && covSeg + 1 < rg.nSeg) covTop = sg[++covSeg].dTop;
            const bool all = d == 0 || (emitFrom - (int)d) % CPK_REFRESH_PERIOD == 0 || d >= covTop - 1;
            e.ringOff = pos;
            // every diagonal starts on an even double and its match row is padded to one: the one-launch form stores
            // pairs of match cells, and a cell's other states, as aligned 16-byte writes (Sweep::ringPut)
            const int32_t we = (w + 1) & ~1;
            pos += all ? we + w * (S - 1) : we;
        } else {
            if (pos + w > rg.ringCap) pos = 0;
            e.ringOff = pos;
            pos += w;
        }
        table[d] = e;
        cells += w;
    };
    const int32_t E = (int32_t)expansion;
    // The walk is a chain of dependent anchor loads -- every interval of the band starts with the next anchor, and with an
    // anchor per matching column (realign-style input) that is every other diagonal: ~1 us each, 9 ms for the longest
    // region of BASELINE config 4, between the sweeps of a pipeline of batches.  So the anchors are requested kAhead
    // intervals ahead and handed to the iterator from registers.
    constexpr int kAhead = 8;
    int32_t ax[kAhead], ay[kAhead];  // anchors it.used .. it.used + kAhead - 1 (clamped to the last one)
    const cpk_anchor_t *ra = anchors + (size_t)anchorStride * rg.anchorOff;
    const int64_t nA = rg.nAnchors;
    const bool queued = !dynamic && anchorStride == 2 && nA > 0;
    if (queued) {
#pragma unroll
        for (int q = 0; q < kAhead; q++) {
            const int64_t at = q < nA ? q : nA - 1;
            ax[q] = ra[2 * at];
            ay[q] = ra[2 * at + 1];
        }
    }
    // cpk_band_advance with the next anchor taken from the queue
    auto advance = [&]() {
        if (!queued) return cpk_band_advance(&it);
        it.pX = it.qX;
        it.pY = it.qY;
        it.qX = it.lX;
        it.qY = it.lY;
        if (it.used < it.n) {
            it.qX = (int64_t)ax[0] + 1;
            it.qY = (int64_t)ay[0] + 1;
            it.used++;
#pragma unroll
            for (int q = 0; q + 1 < kAhead; q++) {
                ax[q] = ax[q + 1];
                ay[q] = ay[q + 1];
            }
            const int64_t at = it.used + kAhead - 1 < nA ? it.used + kAhead - 1 : nA - 1;
            ax[kAhead - 1] = ra[2 * at];
            ay[kAhead - 1] = ra[2 * at + 1];
        }
        it.qSum = it.qX + it.qY;
        const int64_t h = it.e / 2;
        it.xLo = cpk_clamp(it.pX - h, it.lX);
        it.yHi = cpk_clamp(it.qY + h, it.lY);
        it.xHi = cpk_clamp(it.qX + h, it.lX);
        it.yLo = cpk_clamp(it.pY - h, it.lY);
        return 0;
    };
    for (int64_t d = 0; d <= N;) {
        // Between two anchors of a run of diagonal neighbours the two diagonals are known without the rectangle arithmetic
        // (cpk_band_in_run): a region of BASELINE config 4 is such runs nearly everywhere.
        if (cpk_band_in_run(&it, d)) {
            const int32_t xmy0 = (int32_t)(it.pX - it.pY);
            put(d, xmy0 - E - 1, E + 2);
            put(d + 1, xmy0 - E, E + 1);
            advance();
            d += 2;
            continue;
        }
        // cpk_band_next, with the iterator's advance from the queue
        const int64_t a2 = it.xLo > d - it.yHi ? it.xLo : d - it.yHi;
        const int64_t b2 = it.xHi < d - it.yLo ? it.xHi : d - it.yLo;
        put(d, (int32_t)(2 * a2 - d), (int32_t)(b2 - a2 + 1));
        if (it.qSum == d) advance();
        d++;
    }
    if (!dpos || dynamic) return;
    // the position chains of the absolute-position sweeps (see above); the entries just written are read back
    int32_t *pt = dpos + rg.diagOff;
    const int P = rg.maxWidth + kAbsSlack;
    {
        int B = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
        for (int64_t d = 0; d <= N; d++) {
            const int lo = table[d].xmyL, hi = lo + 2 * (table[d].width - 1);
            pt[d] = abs_chain_step(lo, hi, lo1, hi1, lo2, hi2, d >= 2 ? 2 : (int)d, P, B);
            lo2 = lo1; hi2 = hi1; lo1 = lo; hi1 = hi;
        }
    }
    {
        int B = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
        for (int64_t d = N; d >= 0; d--) {
            const int lo = table[d].xmyL, hi = lo + 2 * (table[d].width - 1);
            const int v = abs_chain_step(lo, hi, lo1, hi1, lo2, hi2, N - d >= 2 ? 2 : (int)(N - d), P, B);
            pt[d] = (pt[d] & 0xffff) | (v << 16);
            lo2 = lo1; hi2 = hi1; lo1 = lo; hi1 = hi;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same table for the regions of a SPLIT class, one WAVE per region (round 4).  One thread per region is a chain of
// ~N dependent iterations -- 7-9 ms for a 2 kb pair's 4001 diagonals and its two position chains whatever the batch, and
// ~160 waves on a 256-CU chip: it was the longest stage of a 1250-pair batch's pipeline next to its 12.6 ms sweep, the whole
// of config A's gap between kernel and end-to-end rate, and a third of a single call's latency.  A split region's ring
// never wraps, so everything the table holds is a prefix sum or a local property:
//   * lane l takes the diagonals [l * chunk, (l + 1) * chunk); it finds the band's interval at its first diagonal by a
//     binary search over the anchors' x + y (strictly increasing) and sets the iterator there;
//   * pass A sums its chunk's cells and ring doubles, a wave scan gives every lane its offsets, pass B walks the chunk again
//     and writes the entries;
//   * the position chains (see above) re-base -- forget everything -- at every `danger` diagonal, and whether a diagonal is
//     one depends on it and its two neighbours only: a lane scans back (forward chain) or ahead (backward chain) from its
//     chunk to the nearest one, runs the chain from there without writing, and writes from its chunk's first diagonal on.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool abs_danger(int lo, int hi, int lo1, int hi1, int lo2, int hi2, int have) {
    if (have == 0) return true;
    const int dl = lo - lo1, dh = hi - hi1;
    bool danger = (dl != 1 && dl != -1) || (dh != 1 && dh != -1);
    if (have >= 2) danger = danger || (lo < lo1 && lo1 > lo2) || (hi > hi1 && hi1 < hi2);
    return danger;
}

__global__ void __launch_bounds__(64) cpecan_build_diag_table_wave(const CpkRegion *regions, int regionBase, const cpk_anchor_t *anchors,
                                                                   int anchorStride, const CpkSegment *segs, int S, CpkDiag *diags,
                                                                   int32_t *dpos, int64_t expansion, int dynamic) {
    const int lane = threadIdx.x;
    const CpkRegion rg = regions[regionBase + blockIdx.x];
    CpkDiag *table = diags + rg.diagOff;
    const int N = rg.lX + rg.lY, nD = N + 1;
    const int chunk = (nD + CPK_WAVE - 1) / CPK_WAVE;
    const int d0 = lane * chunk < nD ? lane * chunk : nD, d1 = d0 + chunk < nD ? d0 + chunk : nD;
    const cpk_anchor_t *ra = anchors + (size_t)anchorStride * rg.anchorOff;
    const int nA = rg.nAnchors;
    CpkBandIter it;
    cpk_band_init(&it, ra, anchorStride, nA, rg.lX, rg.lY, expansion, dynamic);
    if (d0 > 0 && d0 < nD) {
        // the interval of diagonal d0: (p -> q), q the first anchor whose own diagonal (x + y + 2 in matrix coordinates) is
        // not below d0, p the anchor before it (the virtual ones (0, 0) / (lX, lY) at either end)
        int loI = 0, hiI = nA;  // first index with sum >= d0 lies in [loI, hiI]
        while (loI < hiI) {
            const int mid = (loI + hiI) >> 1;
            if (ra[(size_t)anchorStride * mid] + ra[(size_t)anchorStride * mid + 1] + 2 < d0) loI = mid + 1;
            else hiI = mid;
        }
        const int j = loI;
        it.pX = j > 0 ? (int64_t)ra[(size_t)anchorStride * (j - 1)] + 1 : 0;
        it.pY = j > 0 ? (int64_t)ra[(size_t)anchorStride * (j - 1) + 1] + 1 : 0;
        it.qX = j < nA ? (int64_t)ra[(size_t)anchorStride * j] + 1 : rg.lX;
        it.qY = j < nA ? (int64_t)ra[(size_t)anchorStride * j + 1] + 1 : rg.lY;
        it.used = j < nA ? j + 1 : nA;
        if (dynamic) it.e = j < nA ? ra[(size_t)anchorStride * j + 2] : (nA > 0 ? ra[(size_t)anchorStride * (nA - 1) + 2] : 0);
        it.qSum = it.qX + it.qY;
        const int64_t h = it.e / 2;
        it.xLo = cpk_clamp(it.pX - h, it.lX);
        it.yHi = cpk_clamp(it.qY + h, it.lY);
        it.xHi = cpk_clamp(it.qX + h, it.lX);
        it.yLo = cpk_clamp(it.pY - h, it.lY);
    }
    // the segments in force at d0 (the serial builder's emitSeg / covSeg)
    const CpkSegment *sg = segs + rg.segOff;
    int emitSeg = 0, covSeg = 0;
    while (emitSeg + 1 < rg.nSeg && d0 > sg[emitSeg].tbFrom) emitSeg++;
    while (covSeg + 1 < rg.nSeg && d0 > sg[covSeg].dTop) covSeg++;
    const CpkBandIter it0 = it;
    const int emitSeg0 = emitSeg, covSeg0 = covSeg;
    int32_t cells = 0, pos = 0;
    for (int pass = 0; pass < 2; pass++) {
        it = it0;
        emitSeg = emitSeg0;
        covSeg = covSeg0;
        int emitFrom = rg.nSeg > 0 ? sg[emitSeg].tbFrom : 0, covTop = rg.nSeg > 0 ? sg[covSeg].dTop : 0;
        for (int d = d0; d < d1; d++) {
            int64_t lo = 0, hi = 0;
            cpk_band_next(&it, d, &lo, &hi);
            const int32_t w = (int32_t)((hi - lo) / 2 + 1);
            while (d > emitFrom && emitSeg + 1 < rg.nSeg) emitFrom = sg[++emitSeg].tbFrom;
            while (d > covTop && covSeg + 1 < rg.nSeg) covTop = sg[++covSeg].dTop;
            const bool all = d == 0 || (emitFrom - d) % CPK_REFRESH_PERIOD == 0 || d >= covTop - 1;
            const int32_t we = (w + 1) & ~1;
            if (pass == 1) table[d] = CpkDiag{(int32_t)lo, w, pos, cells};
            cells += w;
            pos += all ? we + w * (S - 1) : we;
        }
        if (pass == 0) {  // exclusive scan of the chunks' sums: where this lane's chunk starts
            int32_t incC = cells, incP = pos;
#pragma unroll
            for (int off = 1; off < CPK_WAVE; off <<= 1) {
                const int32_t tc = __shfl_up(incC, off), tp = __shfl_up(incP, off);
                if (lane >= off) {
                    incC += tc;
                    incP += tp;
                }
            }
            cells = incC - cells;
            pos = incP - pos;
        }
    }
    if (!dpos || dynamic) return;
    __syncthreads();  // (one wave: every lane's entries are written and visible)
    int32_t *pt = dpos + rg.diagOff;
    const int P = rg.maxWidth + kAbsSlack;
    auto loOf = [&](int d) { return table[d].xmyL; };
    auto hiOf = [&](int d) { return table[d].xmyL + 2 * (table[d].width - 1); };
    if (d0 < d1) {
        // forward chain: from the nearest danger diagonal at or below d0
        int s0 = d0;
        for (; s0 > 0; s0--) {
            const int have = s0 >= 2 ? 2 : s0;
            if (abs_danger(loOf(s0), hiOf(s0), loOf(s0 - 1), hiOf(s0 - 1), have >= 2 ? loOf(s0 - 2) : 0, have >= 2 ? hiOf(s0 - 2) : 0, have)) break;
        }
        int B = 0;
        int lo1 = s0 >= 1 ? loOf(s0 - 1) : 0, hi1 = s0 >= 1 ? hiOf(s0 - 1) : 0, lo2 = s0 >= 2 ? loOf(s0 - 2) : 0, hi2 = s0 >= 2 ? hiOf(s0 - 2) : 0;
        for (int d = s0; d < d1; d++) {
            const int lo = loOf(d), hi = hiOf(d);
            const int v = abs_chain_step(lo, hi, lo1, hi1, lo2, hi2, d >= 2 ? 2 : d, P, B);
            if (d >= d0) pt[d] = v;
            lo2 = lo1; hi2 = hi1; lo1 = lo; hi1 = hi;
        }
        // backward chain (descending diagonals): from the nearest danger diagonal at or above the chunk's last one
        int s1 = d1 - 1;
        for (; s1 < N; s1++) {
            const int have = N - s1 >= 2 ? 2 : N - s1;
            if (abs_danger(loOf(s1), hiOf(s1), loOf(s1 + 1), hiOf(s1 + 1), have >= 2 ? loOf(s1 + 2) : 0, have >= 2 ? hiOf(s1 + 2) : 0, have)) break;
        }
        B = 0;
        lo1 = s1 + 1 <= N ? loOf(s1 + 1) : 0; hi1 = s1 + 1 <= N ? hiOf(s1 + 1) : 0;
        lo2 = s1 + 2 <= N ? loOf(s1 + 2) : 0; hi2 = s1 + 2 <= N ? hiOf(s1 + 2) : 0;
        for (int d = s1; d >= d0; d--) {
            const int lo = loOf(d), hi = hiOf(d);
            const int v = abs_chain_step(lo, hi, lo1, hi1, lo2, hi2, N - d >= 2 ? 2 : N - d, P, B);
            if (d < d1) pt[d] = (pt[d] & 0xffff) | (v << 16);
            lo2 = lo1; hi2 = hi1; lo1 = lo; hi1 = hi;
        }
    }
}

// Result compaction: the sweep leaves every region's triples in its own slice, segments in processing order.  One
// workgroup per chunk (a region's segment) copies it to its place in the compact buffer -- problems in order, regions in
// order, segments DEScending (the reference prepends each traceback's pairs, pairwiseAligner.c:1415-1417) -- and adds
// the region offset.  The host then fetches exactly the emitted triples instead of the slices' capacity.
__global__ void __launch_bounds__(256) cpecan_gather_lists(const CpkChunk *chunks, int64_t nChunks,
                                                           const int32_t *triples, int32_t *out) {
    for (int64_t c = blockIdx.x; c < nChunks; c += gridDim.x) {
        const CpkChunk ch = chunks[c];
        const int32_t *src = triples + 3 * ch.src;
        int32_t *dst = out + 3 * ch.dst;
        for (int i = threadIdx.x; i < 3 * ch.len; i += blockDim.x) {
            const int f = i % 3;
            dst[i] = src[i] + (f == 1 ? ch.dx : (f == 2 ? ch.dy : 0));
        }
    }
}
