// cpk_table_gather.inl -- device-side planning and result plumbing: the per-diagonal band table, the list-order gather.
// Part of the single HIP translation unit cpecan_kernels.hip (included there, in this order); not compiled on its own.

// The per-diagonal table the sweeps read, built on the device: one thread per region walks its band with the host's
// own iterator (cpecan_band.inl; the host has already validated the anchors with it) and writes
// {x-y of the first cell, width, position in the region's forward ring, cells on earlier diagonals}.  The ring
// position follows the rule the kernels rely on: diagonals are laid end to end and never straddle the ring's end.
// (In a pipeline of batches this kernel of batch k+1 runs on that batch's own stream; its 56 allocated VGPRs do not fit
// in the 48 a resident sweep of batch k leaves free per lane, so it starts when that sweep drains: ~3 ms per config-B batch.)
__global__ void __launch_bounds__(64) cpecan_build_diag_table(const CpkRegion *regions, int nRegions, const cpk_anchor_t *anchors, int anchorStride,
                                                              const CpkSegment *segs, int S, CpkDiag *diags, int64_t expansion, int dynamic) {
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
    for (int64_t d = 0; d <= N; d++) {
        int64_t lo = 0, hi = 0;
        cpk_band_next(&it, d, &lo, &hi);
        const int32_t w = (int32_t)((hi - lo) / 2 + 1);
        CpkDiag e;
        e.xmyL = (int32_t)lo;
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
