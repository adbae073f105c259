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

__global__ void __launch_bounds__(64) cpecan_build_diag_table(const CpkRegion *regions, int nRegions, const cpk_anchor_t *anchors, int anchorStride,
                                                              const CpkSegment *segs, int S, CpkDiag *diags, int32_t *dpos, int64_t expansion, int dynamic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRegions) return;
    const CpkRegion rg = regions[i];
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
