// cpk_packed.inl -- the packed kernel: 64/GW narrow-band regions per wave.
// Part of the single HIP translation unit cpecan_kernels.hip (included there, in this order); not compiled on its own.

// ------------------------------------------------------------------------------------------------
// Packed kernel for narrow bands (realign-style work: diagonalExpansion 4-10, diagonals of 5-30 cells).
// One wave per region leaves most lanes idle there and pays the per-diagonal bookkeeping for a handful of cells.
// Here a wave runs G = 64 / GW regions at once: lane = (group g, cell c), every diagonal of a narrow region is one
// group of at most GW cells, so there is no loop over groups and no in-place ordering problem.  Everything that is
// wave-uniform in the sweep kernel (diagonal counter, table entries, neighbour shifts, row pointers, traceback
// schedule) is per lane here, identical within a group.  The groups of a wave move in lock-step through the same
// phases -- forward sweep of segment i, traceback of segment i, totals, emission -- each over its own diagonals; a
// group that has nothing to do in a phase idles (regions are handed out sorted by size, so neighbours are alike).
// Arithmetic: the sweep kernel's own cell functions (fwdCellsSym / bwdCellsSym), same order, bit-identical results.
// Match, indel and expectation emitters; symbols are staged per chunk of 64 diagonals into two LDS windows (one byte per symbol).
// ------------------------------------------------------------------------------------------------
template <int GW>
__device__ __forceinline__ float group_max_f32(float v) {
#pragma unroll
    for (int off = GW / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// LDS of the packed kernel behind the shared tables, per group: rolling buffers | 64 table entries | two symbol
// windows | candidate staging ring
constexpr int kPackChunk = 64;                  // diagonals per staged chunk
__host__ __device__ constexpr int pack_win_bytes(int gw) { return (kPackChunk + gw + 15) / 8 * 8; }  // a symbol window
__host__ __device__ constexpr int pack_rows_bytes(int S, int gw) { return (8 * (2 * S + 1) * (gw + 1) + 15) / 16 * 16; }
__host__ __device__ constexpr int pack_group_bytes(int S, int gw) {  // a multiple of 16: entries and candidates are 16-byte items
    return pack_rows_bytes(S, gw) + 16 * kPackChunk + 2 * pack_win_bytes(gw) + 16 * 2 * gw;
}

#ifndef CPK_PACKED_WAVES
#define CPK_PACKED_WAVES 2  // waves per SIMD the packed kernel's registers are allocated for (3: 73-84 spilled VGPRs, measured slower)
#endif
// DYN: bands with per-anchor expansions (band_constructDynamic, pairwiseAligner.c:184-234).  Their edges may move BACK where
// an anchor brings a larger expansion, so the symbols a chunk of 64 diagonals needs do not start at its first diagonal's:
// the windows start at the minimum over the chunk's diagonals, and a cell outside a window (a chunk whose band wanders
// further than the window holds: never with a fixed expansion) reads its symbol from global memory.
// MODE (round 4, match emitter): kModeWhole -- a group takes a region through its forward sweep and its tracebacks, segment
// by segment, over a ring of the group's own; kModeForward / kModeTrace -- a SPLIT class (cf. cpecan_pairhmm_sweep): the
// first launch sweeps every region forward into a ring of the region's own (never wraps; the layout of the sweep kernel's
// split regions: the match row of every diagonal, every state only where a traceback reads it back), the second takes
// one queue item per (region, traceback segment), longest first, G items to a wave.  Why: a realign-style batch holds
// regions of 100 to 10 000 diagonals, and ONE group walking a 10 000-diagonal region through 10 000 forward and 10 400
// backward steps is 20 ms whatever else the chip does -- BASELINE config 4 took 20.2 ms with 6 000 pairs and 29.3 with
// 50 000 (profiles/r04_config4_chain_bound.txt).  Split, the chain is the forward steps alone and the tracebacks of a
// long region run side by side.
template <int S, int GW, int EMIT, bool DYN = false, int MODE = kModeWhole>  // EMIT: CPECAN_EMIT_MATCH, _INDEL or _EXPECT
__global__ void __launch_bounds__(CPK_WAVE) __attribute__((amdgpu_waves_per_eu(CPK_PACKED_WAVES, CPK_PACKED_WAVES)))
cpecan_pairhmm_packed(const KArgs a) {
    static_assert(MODE == kModeWhole || MODE == kModeForward || MODE == kModeTrace, "packed kernel: whole regions, or the two launches of a split class");
    static_assert(MODE == kModeWhole || (EMIT == CPECAN_EMIT_MATCH && !DYN), "split classes: match emitter, fixed expansion");
    constexpr bool kSplit = MODE != kModeWhole;
    constexpr int G = CPK_WAVE / GW;
    constexpr int R = 2 * S + 1;
    constexpr int kRowDoubles = R * (GW + 1);
    constexpr int kWin = pack_win_bytes(GW);
    constexpr int kStageP = 2 * GW;  // candidate staging slots per group (flushed GW at a time)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int g = lane / GW, c = lane % GW;
    const CpkModel &m = *a.model;

    fill_cubics(lds);
    const Cubic *lg = reinterpret_cast<const Cubic *>(lds);
    double *wt = lds + kLdsCubics + kLdsEm;
    fill_weights<S>(wt, m, a.kc, lane);
    constexpr bool kExpect = EMIT == CPECAN_EMIT_EXPECT;
    // Indel emitter (round 4; diagonalCalculationPosteriorProbs, pairwiseAligner.c:691-733): three lists -- match, gapX,
    // gapY -- of one segment.  Round 3 built it with three candidate lists live in the traceback: 55 spilled registers,
    // not kept.  Here the traceback parks B of the emitted cells in global memory, as it does for the expectation step,
    // and a pass of its own behind the totals forms the three posteriors of every emitted cell from the ring and those
    // values, diagonal by diagonal in the lists' order: no candidates, nothing more alive in the traceback.
    constexpr bool kIndel = EMIT == CPECAN_EMIT_INDEL;
    constexpr bool kKeepB = kExpect || kIndel;
    constexpr int NL = kIndel ? 3 : 1;
    double *eLds = lds + kLdsCubics + kLdsEm + kLdsWeights;  // expectation emitter: emission sums of this wave, four copies
    if (kExpect)
        for (int i = lane; i < kExpectCopies * 80; i += CPK_WAVE) eLds[i] = 0.0;
    constexpr int kNT = S == 5 ? 13 : 9;
    double tAcc[kNT];  // transition sums of this lane, one per transition in list order
#pragma unroll
    for (int i = 0; i < kNT; i++) tAcc[i] = 0.0;
    double likelihood = 0.0;
    uint8_t *mine = reinterpret_cast<uint8_t *>(lds + kLdsCubics + kLdsEm + kLdsWeights + (kExpect ? kExpectCopies * 80 : 0)) +
                    (size_t)g * pack_group_bytes(S, GW);
    double *rows = reinterpret_cast<double *>(mine);                                   // rolling buffers
    int4 *ebuf = reinterpret_cast<int4 *>(mine + pack_rows_bytes(S, GW));              // table entries of the chunk
    uint8_t *xwin = mine + pack_rows_bytes(S, GW) + 16 * kPackChunk, *ywin = xwin + kWin;  // symbols of the chunk
    Candidate *stage = reinterpret_cast<Candidate *>(ywin + kWin);                     // candidate ring
    __syncthreads();

    // the cell functions only need the tables; every position-dependent input is passed per call
    // (RDBL = kSplit: a split region's ring counts doubles and pads the match row, Sweep::ringIdx)
    using SW = Sweep<S, false, 2 * S + 1, false, kSplit>;
    SW sw{a, a.kc, DiagCache{nullptr, 0, 0, lane, 0, 0, 0, 0}, nullptr, nullptr, rows, lds + kLdsCubics, wt, lg,
          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, GW + 1, lane, lane * R, 0, CpkDiag{}, CpkDiag{}};
    const unsigned long long groupBits = (GW == 64 ? ~0ull : ((1ull << GW) - 1ull)) << (g * GW);
    const unsigned long long belowMe = groupBits & ((1ull << lane) - 1ull);
    const unsigned long long aboveMe = groupBits & ~((2ull << lane) - 1ull);  // (2 << 63 wraps to 0: nothing above lane 63)
    const float logThr = (float)log(a.kc.threshold);
    const double thr = a.kc.threshold;

    for (;;) {
        const unsigned int ticket = atomicAdd(a.queue, lane == 0 ? (unsigned)G : 0u);
        const int tk = __builtin_amdgcn_readfirstlane((int)ticket);
        if (tk >= a.regionCount) break;
        const bool have = tk + g < a.regionCount;
        // kModeTrace: the queue holds (region, segment) items, longest first (a.regionCount of them); else regions
        const int itemSeg = MODE == kModeTrace ? a.items[have ? tk + g : tk].seg : 0;
        const int r = MODE == kModeTrace ? a.items[have ? tk + g : tk].region : a.regionBase + (have ? tk + g : tk);
        const CpkRegion rg = a.regions[r];
        const int N = have ? rg.lX + rg.lY : 0;
        const int nSeg = (have && N > 0) ? (MODE == kModeTrace ? 1 : rg.nSeg) : 0;
        const CpkDiag *table = a.diags + rg.diagOff;
        const CpkSegment *segs = a.segs + rg.segOff + itemSeg;
        const uint8_t *gx = a.symbols + rg.seqXOff, *gy = a.symbols + rg.seqYOff;  // padded: index p = base p-1, N at both ends
        const size_t sub = (size_t)blockIdx.x * G + g;  // scratch sub-slot of this group
        double *ring = kSplit ? a.ring + (size_t)rg.ringBase : a.ring + sub * (size_t)a.geo.ringCells * S;
        Candidate *cand = a.cand + sub * (size_t)a.geo.fbCells;
        double *cbuf = a.cbuf + sub * (size_t)a.geo.refreshCells, *mbuf = a.mbuf + sub * (size_t)a.geo.refreshCells;
        double *totals = a.totals + sub * (size_t)a.geo.maxRefresh;
        double *bring = kKeepB ? a.bring + sub * (size_t)a.geo.fbCells * S : nullptr;  // B of the segment's emitted cells
        int32_t *out = a.triples + 3 * rg.outOff;
        int count = 0;
        int countL[NL];  // indel emitter: triples of the region so far, per list
#pragma unroll
        for (int l = 0; l < NL; l++) countL[l] = 0;

        for (int i = c; i < kRowDoubles; i += GW) rows[i] = NEG_INF;  // position 0 stays the -inf guard
        auto fbuf1 = [&](int d) { return rows + R + (d & 1) * S; };
        auto bM1 = [&](int d) { return rows + R + (d + 3) % 3; };
        auto bG1 = [&](int d) { return rows + R + 2 + (d & 1) * (S - 1); };
        auto ringAt = [&](const CpkDiag &e) { return ring + (size_t)e.ringOff * (kSplit ? 1 : S); };
        auto unpack = [](const int4 &t) { return CpkDiag{t.x, t.y, t.z, t.w}; };
        // the symbol of padded position p of X / Y for a cell of the staged chunk (windows from x0 / y0)
        auto symAtX = [&](int p, int x0) -> int {
            const int q = p - x0;
            if (!DYN || (unsigned)q < (unsigned)kWin) return xwin[q];
            return gx[p < 0 ? 0 : (p > rg.lX + 1 ? rg.lX + 1 : p)];
        };
        auto symAtY = [&](int p, int y0) -> int {
            const int q = p - y0;
            if (!DYN || (unsigned)q < (unsigned)kWin) return ywin[q];
            return gy[p < 0 ? 0 : (p > rg.lY + 1 ? rg.lY + 1 : p)];
        };
        // Stages the table entries of `cnt` (<= 64) diagonals first, first + step, ... into ebuf and the X / Y symbols
        // their cells use (shifted by `shift`: the backward step reads the symbols of (x+1, y+1)) into the two windows.
        // One global round trip per 64 diagonals instead of three per diagonal.
        // Every load of a stage is issued before the first one is waited for (unrolled loops over registers, addresses
        // clamped instead of predicated): as loops of load-then-store the compiler waited for each in turn, ~30 global
        // round trips in a row per chunk.
        auto stage_chunk = [&](bool on, int first, int step, int cnt, int shift, int &x0, int &y0) {
            int xMin = 0x7fffffff, yMin = 0x7fffffff;  // DYN: the lowest x and y of the chunk's cells
            {
                int4 t[kPackChunk / GW];
#pragma unroll
                for (int j = 0; j < kPackChunk / GW; j++) {
                    const int i = c + j * GW;
                    int dd = first + step * (i < cnt ? i : (cnt > 0 ? cnt - 1 : 0));
                    dd = dd < 0 ? 0 : (dd > N ? N : dd);
                    t[j] = *reinterpret_cast<const int4 *>(table + dd);  // N = 0 for a group without a region: entry 0 of a valid table
                    if (DYN) {
                        const int xl = (dd + t[j].x) >> 1;
                        xMin = xl < xMin ? xl : xMin;
                        const int yl = dd - (xl + t[j].y - 1);
                        yMin = yl < yMin ? yl : yMin;
                    }
                }
#pragma unroll
                for (int j = 0; j < kPackChunk / GW; j++) ebuf[c + j * GW] = on ? t[j] : int4{0, 1, 0, 0};
                if (DYN) {
#pragma unroll
                    for (int off = GW / 2; off > 0; off >>= 1) {
                        const int ox = __shfl_xor(xMin, off), oy = __shfl_xor(yMin, off);
                        xMin = ox < xMin ? ox : xMin;
                        yMin = oy < yMin ? oy : yMin;
                    }
                }
            }
            // both ends of the chunk bound the coordinates in between (x and y never decrease with the diagonal)
            int dA = step > 0 ? first : first - (cnt - 1), dB = step > 0 ? first + (cnt - 1) : first;
            dA = dA < 0 ? 0 : dA;  // the entries above are clamped the same way (look-ahead below diagonal 0)
            dB = dB > N ? N : dB;
            const CpkDiag eA = unpack(ebuf[step > 0 ? 0 : (cnt > 0 ? cnt - 1 : 0)]);
            const CpkDiag eB = unpack(ebuf[step > 0 ? (cnt > 0 ? cnt - 1 : 0) : 0]);
            const int xloA = (dA + eA.xmyL) >> 1, xloB = (dB + eB.xmyL) >> 1;
            x0 = (DYN ? xMin : xloA) + shift;
            y0 = (DYN ? yMin : dA - (xloA + eA.width - 1)) + shift;
            const int x1 = DYN ? rg.lX + 1 : xloB + eB.width - 1 + shift, y1 = DYN ? rg.lY + 1 : dB - xloB + shift;
            constexpr int kWinIter = (kWin + GW - 1) / GW;
            uint8_t bx[kWinIter], by[kWinIter];
#pragma unroll
            for (int j = 0; j < kWinIter; j++) {
                const int px = x0 + c + j * GW, py = y0 + c + j * GW;
                bx[j] = gx[px < 0 ? 0 : (px > rg.lX + 1 ? rg.lX + 1 : px)];  // the padded strings hold indices 0 .. l+1
                by[j] = gy[py < 0 ? 0 : (py > rg.lY + 1 ? rg.lY + 1 : py)];
            }
#pragma unroll
            for (int j = 0; j < kWinIter; j++) {
                const int i = c + j * GW;
                const int px = x0 + i, py = y0 + i;
                if (i < kWin) {
                    xwin[i] = (on && cnt > 0 && px >= 0 && px <= x1 && px <= rg.lX + 1) ? bx[j] : (uint8_t)CPK_SYM_N;
                    ywin[i] = (on && cnt > 0 && py >= 0 && py <= y1 && py <= rg.lY + 1) ? by[j] : (uint8_t)CPK_SYM_N;
                }
            }
        };

        CpkDiag e1{}, e2{};  // entries of d-1 and d-2 of the forward sweep
        if (nSeg > 0 && MODE != kModeTrace) {
            const int4 t0 = *reinterpret_cast<const int4 *>(table);
            e1 = e2 = unpack(t0);
            const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
            if (c < S) {  // diagonal 0: the single cell (0,0) holds the start prior (pairwiseAligner.c:776-777)
                fbuf1(0)[c] = startPrior[c];
                ringAt(e1)[SW::ringIdx(1, c, 0)] = startPrior[c];
            }
        }
        int d = 1;
        // split classes: which states of F[d] go to the ring -- the table builder's rule (cpk_table_gather.inl, `all`; the
        // forward loop of cpecan_pairhmm_sweep): every state on the refresh diagonals of the segment that EMITS d (the
        // first with tbFrom >= d) and on the two diagonals below the top of the segment that covers d; else the match row
        int emitSeg = 0, emitFrom = (kSplit && nSeg > 0) ? segs[0].tbFrom : 0;
        const int maxSeg = wave_max_i32(nSeg);
        for (int si = 0; si < maxSeg; si++) {
            const bool segOn = si < nSeg;
            CpkSegment sg{};
            if (segOn) sg = segs[si];
            // ---------------- forward sweep up to dTop (pairwiseAligner.c:609-629) ----------------
            while (MODE != kModeTrace && __ballot(segOn && d <= sg.dTop)) {
                const bool more = segOn && d <= sg.dTop;
                const int cnt = more ? (sg.dTop - d + 1 < kPackChunk ? sg.dTop - d + 1 : kPackChunk) : 0;
                int x0, y0;
                stage_chunk(more, d, 1, cnt, 0, x0, y0);
                for (int i = 0; i < kPackChunk; i++) {
                    if (!__ballot(i < cnt)) break;
                    const bool act = i < cnt;
                    const CpkDiag e = act ? unpack(ebuf[i]) : e1;
                    const int W = e.width;
                    const bool on = act && c < W;
                    typename SW::FwdCtx fc;
                    fc.d = d;
                    fc.xlo = (d + e.xmyL) >> 1;
                    fc.dlR = ((e.xmyL - 1 - e1.xmyL) >> 1) * R;
                    fc.w1R = e1.width * R;
                    fc.dmR = ((e.xmyL - e2.xmyL) >> 1) * R;
                    fc.w2R = d >= 2 ? e2.width * R : 0;
                    fc.p1 = fbuf1(d - 1);
                    fc.p2 = fbuf1(d - 2);
                    const int x = fc.xlo + c, y = d - x;
                    const int cX[1] = {on ? symAtX(x, x0) : CPK_SYM_N}, cY[1] = {on ? symAtY(y, y0) : CPK_SYM_N};
                    const int kR[1] = {c * R};
                    double v[1][S];
                    sw.template fwdCellsSym<1>(fc, cX, cY, kR, v);
                    bool all = true;
                    if (kSplit && act) {
                        while (d > emitFrom && emitSeg + 1 < nSeg) emitFrom = segs[++emitSeg].tbFrom;
                        all = (emitFrom - d) % CPK_REFRESH_PERIOD == 0 || d >= sg.dTop - 1;
                    }
                    if (on) {
                        double *cur = fbuf1(d);
                        double *o = ringAt(e);
#pragma unroll
                        for (int s = 0; s < S; s++) cur[s + c * R] = v[0][s];
                        o[SW::ringIdx(W, 0, c)] = v[0][0];
                        if (all) {
#pragma unroll
                            for (int s = 1; s < S; s++) o[SW::ringIdx(W, s, c)] = v[0][s];
                        }
                    }
                    if (act) {
                        e2 = e1;
                        e1 = e;
                        d++;
                    }
                }
            }
            if (MODE == kModeForward) continue;  // the tracebacks of a split class are the items of the next launch
            // ---------------- traceback of the segment (pairwiseAligner.c:796-862) ----------------
            const double *endPrior = (segOn && sg.atEnd && rg.raggedRight) ? m.raggedEnd : m.end;
            double ep[S];
#pragma unroll
            for (int s = 0; s < S; s++) ep[s] = endPrior[s];
            const int J = sg.nRefresh;
            int nCand = 0, pend = 0, head = 0;  // candidates in HBM; staged in LDS; ring position of the oldest staged
            auto flush = [&](int n) {           // the n <= GW oldest staged candidates of this group -> cand[nCand ..]
                if (c < n) cand[nCand + c] = stage[(head + c) & (kStageP - 1)];
                head = (head + n) & (kStageP - 1);
                pend -= n;
                nCand += n;
            };
            float lastMax = -__builtin_huge_valf();
            // Values of the two series wait a block (kAhead steps) in registers and go to global memory at the top of the
            // next one, behind its wait for the requested F values and in front of the next request: that wait counts every
            // store issued since as well, and a write acknowledgement is the slowest thing there is to wait for.  A group
            // has at most one refresh point and one diagonal above one in a block (period 10 > kAhead).
            double pendM = 0.0, pendC = 0.0;
            int pendMj = -1, pendCj = -1;  // series index, -1: nothing waits
            auto issueStores = [&]() {
                if (pendMj >= 0) mbuf[(size_t)c * J + pendMj] = pendM;
                if (pendCj >= 0) cbuf[(size_t)c * J + pendCj] = pendC;
                pendMj = pendCj = -1;
                if (!kKeepB && __ballot(pend >= GW)) {
                    if (pend >= GW) flush(GW);
                }
            };
            const int bBase = (kKeepB && segOn) ? table[sg.tbPrev + 1].cellOff : 0;
            int d2 = segOn ? sg.dTop : 0;
            CpkDiag eb{}, ea{};  // entries of d2+1, d2+2
            while (__ballot(segOn && d2 > sg.tbPrev)) {
                const bool more = segOn && d2 > sg.tbPrev;
                const int cnt = more ? (d2 - sg.tbPrev < kPackChunk ? d2 - sg.tbPrev : kPackChunk) : 0;
                int x0, y0;
                stage_chunk(more, d2, -1, cnt, 1, x0, y0);
                // F.match of this lane's cell (and, for a refresh diagonal, the other states of that cell) come from the ring
                // in HBM.  They are requested a BLOCK of kAhead diagonals ahead: at the top of a block the values of the
                // block are taken over -- the one place the wave waits for memory -- and the requests of the next block go
                // out; the kAhead steps in between issue no load.  hipcc cannot do better than `s_waitcnt vmcnt(0)` for a
                // load once stores may be in flight as well (gfx9 counts both in vmcnt and lets them complete out of
                // order), and these steps store candidates, cbuf and mbuf values all the time: with a request per step,
                // every step waited for the request of the step before (waves waiting 55 % of their cycles, round 2).
                // The loads are unconditional: clamped to a valid cell of a valid diagonal.
                constexpr int kAhead = 4;
                static_assert(kPackChunk % kAhead == 0, "the chunk loop is unrolled by the block size");
                double fqN[kAhead], rfN[S];
                // block [i0, i0 + kAhead): F.match of every step, and the rows of the (at most one: period 10) refresh step
                auto request = [&](int i0, int dAt) {
                    int iRef = i0;  // without a refresh step in the block: any valid entry
#pragma unroll
                    for (int j = 0; j < kAhead; j++) {
                        const int i = i0 + j;
                        const CpkDiag en = unpack(ebuf[i < kPackChunk ? i : kPackChunk - 1]);  // past the chunk: its last diagonal again
                        fqN[j] = ld_self(ringAt(en) + SW::ringIdx(en.width, 0, c < en.width ? c : en.width - 1));
                        const int dd = dAt - j;
                        if (i < cnt && dd <= sg.tbFrom && (sg.tbFrom - dd) % CPK_REFRESH_PERIOD == 0) iRef = i;
                    }
                    const CpkDiag er = unpack(ebuf[iRef < kPackChunk ? iRef : kPackChunk - 1]);
#pragma unroll
                    for (int s2 = 1; s2 < S; s2++)
                        rfN[s2] = ld_self(ringAt(er) + SW::ringIdx(er.width, s2, c < er.width ? c : er.width - 1));
                };
                rfN[0] = 0.0;
                request(0, d2);
                bool chunkDone = false;
                for (int i0 = 0; i0 < kPackChunk && !chunkDone; i0 += kAhead) {
                  if (!__ballot(i0 < cnt)) break;
                  double fq[kAhead], rf[S];
#pragma unroll
                  for (int j = 0; j < kAhead; j++) {
                      asm volatile("" : "+v"(fqN[j]));
                      fq[j] = fqN[j];
                  }
                  rf[0] = 0.0;
#pragma unroll
                  for (int s2 = 1; s2 < S; s2++) {
                      asm volatile("" : "+v"(rfN[s2]));
                      rf[s2] = rfN[s2];
                  }
                  issueStores();
                  if (i0 + kAhead < kPackChunk) request(i0 + kAhead, d2 - kAhead);  // a group's d2 falls by one per active step
#pragma unroll
                  for (int j = 0; j < kAhead; j++) {
                    const int i = i0 + j;
                    if (!__ballot(i < cnt)) {
                        chunkDone = true;
                        break;
                    }
                    const bool act = i < cnt;
                    const CpkDiag e = act ? unpack(ebuf[i]) : CpkDiag{0, 1, 0, 0};
                    const int W = e.width;
                    const bool on = act && c < W;
                    const bool emit = act && d2 <= sg.tbFrom;
                    const int sinceFrom = sg.tbFrom - d2;
                    const bool refresh = emit && sinceFrom % CPK_REFRESH_PERIOD == 0;
                    const double f0 = fq[j];
                    const bool seeded = d2 == sg.dTop;
                    const int jr = sinceFrom / CPK_REFRESH_PERIOD;
                    // the fb values of the diagonal above a refresh point are its straddle series (see Sweep::traceback)
                    const bool feeds = act && d2 - 1 > sg.tbPrev && d2 - 1 <= sg.tbFrom &&
                                       (sg.tbFrom - (d2 - 1)) % CPK_REFRESH_PERIOD == 0;
                    const int jrNext = (sg.tbFrom - (d2 - 1)) / CPK_REFRESH_PERIOD;
                    typename SW::BwdCtx bc;
                    bc.d2 = d2;
                    bc.xlo = (d2 + e.xmyL) >> 1;
                    bc.dbR = ((e.xmyL - 1 - eb.xmyL) >> 1) * R;
                    bc.wBR = seeded ? 0 : eb.width * R;
                    bc.daR = ((e.xmyL - ea.xmyL) >> 1) * R;
                    bc.wAR = (!seeded && d2 + 2 <= sg.dTop) ? ea.width * R : 0;
                    bc.pb = bG1(d2 + 1);
                    bc.pa = bM1(d2 + 2);
                    const int x = bc.xlo + c, y = d2 - x;
                    // symbols of the source cells (x+1, .) and (., y+1): the windows are staged one to the right
                    const int cX1[1] = {on ? symAtX(x + 1, x0) : CPK_SYM_N}, cY1[1] = {on ? symAtY(y + 1, y0) : CPK_SYM_N};
                    const int kR[1] = {c * R};
                    double v[1][S];
                    sw.template bwdCellsSym<1>(bc, cX1, cY1, kR, v);
                    if (seeded) {  // every cell of the top diagonal gets the end-state prior (:798-799)
#pragma unroll
                        for (int s = 0; s < S; s++) v[0][s] = ep[s];
                    }
                    if (on) {
                        bM1(d2)[c * R] = v[0][0];
                        double *curG = bG1(d2);
#pragma unroll
                        for (int s = 1; s < S; s++) curG[s + c * R] = v[0][s];
                    }
                    const double fbv = f0 + v[0][0];
                    if (feeds && on) {
                        pendM = fbv;
                        pendMj = jrNext;
                    }
                    if (kKeepB && emit && on) {  // kept for the expectation step / the indel emitter's pass
                        double *bo = bring + (size_t)(e.cellOff - bBase + c) * S;
#pragma unroll
                        for (int s = 0; s < S; s++) bo[s] = v[0][s];
                    }
                    if (!kKeepB) {
                        const float keepFrom = lastMax + logThr - kCandMargin;
                        const bool keep = on && emit && x > 0 && y > 0 && (float)fbv >= keepFrom;
                        const unsigned long long mask = __ballot(keep);
                        if (keep) {
                            Candidate cd;
                            cd.fb = fbv;
                            cd.x = x;
                            cd.y = y;
                            stage[(head + pend + __popcll(mask & belowMe)) & (kStageP - 1)] = cd;
                        }
                        pend += __popcll(mask & groupBits);
                        if (__ballot(pend > kStageP - GW)) {  // the next step's candidates might not fit: now (else at the top of the next block)
                            if (pend > kStageP - GW) flush(GW);
                        }
                    }
                    if (__ballot(refresh)) {
                        // cell dot product over the states (cell_dotProduct :402-408): this lane holds its cell's B values
                        double t = fbv;
                        float fbf = -__builtin_huge_valf();
                        if (refresh && on) {
#pragma unroll
                            for (int s2 = 1; s2 < S; s2++)
                                t = logadd(lg, t, rf[s2] + v[0][s2]);
                            pendC = t;
                            pendCj = jr;
                            if (!kKeepB && x > 0 && y > 0) fbf = (float)fbv;
                        }
                        const float diagMax = group_max_f32<GW>(fbf);
                        if (refresh) lastMax = fmaxf(diagMax, lastMax - 1.0f);
                    }
                    if (act) {
                        ea = eb;
                        eb = e;
                        d2--;
                    }
                  }
                }
            }
            issueStores();
            if (!kKeepB && __ballot(pend > 0)) flush(pend);
            roll_fence<true>();  // candidate / cbuf / mbuf stores of the group's lanes are visible to each other
            // ---------------- totals at the refresh points (:636-653): lane c takes points c, c + GW, ... ----------------
            for (int j0 = 0; __ballot(segOn && j0 + c < J); j0 += GW) {
                const int j = j0 + c;
                if (segOn && j < J) {
                    const int rr = sg.tbFrom - CPK_REFRESH_PERIOD * j;
                    const int Wc = table[rr].width;
                    const int Wm = rr + 1 <= sg.dTop ? table[rr + 1].width : 0;
                    // eight values of each series are requested together, then folded in order (the two series side by side);
                    // padding with -inf leaves a fold unchanged because logAdd(x, -inf) returns x exactly
                    double ts[2] = {NEG_INF, NEG_INF};
                    for (int k0 = 0; k0 < Wc || k0 < Wm; k0 += 8) {
                        double xs[8], ys[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const int k = k0 + q < GW ? k0 + q : GW - 1;  // inside the group's slice whatever Wc, Wm
                            xs[q] = ld_self(cbuf + (size_t)k * J + j);
                            ys[q] = ld_self(mbuf + (size_t)k * J + j);
                        }
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const double xy[2] = {k0 + q < Wc ? xs[q] : NEG_INF, k0 + q < Wm ? ys[q] : NEG_INF};
                            logadd_n<2>(lg, ts, xy);
                        }
                    }
                    double total = ts[0];
                    if (rr + 1 <= sg.dTop) total = logadd(lg, total, ts[1]);
                    totals[j] = total;
                }
            }
            roll_fence<true>();
            if (kExpect) {
                // ---------------- expectation step (:735-746, :418-432), see Sweep::expectations ----------------
                // Per emitted diagonal d2 and cell: p = exp(F_nbr[from] + B[to] + (eP + tP) - total) for every transition
                // into the cell, neighbours from F[d2-1] / F[d2-2] (the latter is gone at d2 == tbPrev+1, :843-845).
                int e2d = segOn ? sg.tbFrom : 0;
                while (__ballot(segOn && e2d > sg.tbPrev)) {
                    const bool more = segOn && e2d > sg.tbPrev;
                    // 62 diagonals per chunk: the two entries below each diagonal are staged with it
                    const int cnt = more ? (e2d - sg.tbPrev < kPackChunk - 2 ? e2d - sg.tbPrev : kPackChunk - 2) : 0;
                    int x0, y0;
                    stage_chunk(more, e2d, -1, more ? cnt + 2 : 0, 0, x0, y0);
                    for (int i = 0; i < kPackChunk - 2; i++) {
                        if (!__ballot(i < cnt)) break;
                        const bool act = i < cnt;
                        const CpkDiag e = act ? unpack(ebuf[i]) : CpkDiag{0, 1, 0, 0};
                        const CpkDiag g1 = unpack(ebuf[i + 1]), g2 = unpack(ebuf[i + 2]);  // entries of d2-1, d2-2
                        const int W = e.width;
                        const bool on = act && c < W;
                        const bool haveM2 = e2d - 2 >= sg.tbPrev;
                        const int xlo = (e2d + e.xmyL) >> 1;
                        const int dl = (e.xmyL - 1 - g1.xmyL) >> 1, dm = (e.xmyL - g2.xmyL) >> 1;
                        const int w1 = g1.width, w2 = haveM2 ? g2.width : 0;
                        const int kL = c + dl, kU = c + dl + 1, kM = c + dm;
                        const bool okL = (unsigned)kL < (unsigned)w1, okU = (unsigned)kU < (unsigned)w1,
                                   okM = (unsigned)kM < (unsigned)w2;
                        const int qL = okL ? kL : 0, qU = okU ? kU : 0, qM = okM ? kM : 0;
                        if (on) {
                            const double total = ld_self(totals + (sg.tbFrom - e2d) / CPK_REFRESH_PERIOD);
                            if (c == 0) likelihood += total;  // once per diagonal (:743)
                            const double *f1 = ringAt(g1), *f2 = ringAt(g2);
                            const double *bo = bring + (size_t)(e.cellOff - bBase + c) * S;
                            double v[S], fL[S], fU[S], fM[S];
#pragma unroll
                            for (int s = 0; s < S; s++) {
                                const bool needL = S == 3 || s == 0 || s == 1 || s == 3, needU = S == 3 || s == 0 || s == 2 || s == 4;
                                v[s] = ld_self(bo + s);
                                fL[s] = (needL && okL) ? ld_self(f1 + SW::ringIdx(w1, s, qL)) : NEG_INF;
                                fU[s] = (needU && okU) ? ld_self(f1 + SW::ringIdx(w1, s, qU)) : NEG_INF;
                                fM[s] = okM ? ld_self(f2 + SW::ringIdx(w2, s, qM)) : NEG_INF;
                            }
                            const int x = xlo + c, y = e2d - x;
                            const int cX = symAtX(x, x0), cY = symAtY(y, y0);
                            constexpr int kWM = SW::kWM, kWG = SW::kWG;
                            const double *wM = wt + (cX * 5 + cY) * kWM, *wX = wt + 25 * kWM + cX * kWG,
                                         *wY = wt + 25 * kWM + 5 * kWG + cY * kWG;
                            double eAcc[S];
#pragma unroll
                            for (int s = 0; s < S; s++) eAcc[s] = 0.0;
                            auto event = [&](int ti, double from, int to, double w) {
                                const double p = exp_1e7(from + v[to] + w - total);
                                tAcc[ti] += p;
                                eAcc[to] += p;
                            };
                            if (S == 5) {
                                event(0, fL[0], 1, wX[0]);
                                event(1, fL[1], 1, wX[1]);
                                event(2, fL[0], 3, wX[2]);
                                event(3, fL[3], 3, wX[3]);
                                event(4, fM[0], 0, wM[0]);
                                event(5, fM[1], 0, wM[1]);
                                event(6, fM[2], 0, wM[2]);
                                event(7, fM[3], 0, wM[3]);
                                event(8, fM[4], 0, wM[4]);
                                event(9, fU[0], 2, wY[0]);
                                event(10, fU[2], 2, wY[1]);
                                event(11, fU[0], 4, wY[2]);
                                event(12, fU[4], 4, wY[3]);
                            } else {
                                event(0, fL[0], 1, wX[0]);
                                event(1, fL[1], 1, wX[1]);
                                event(2, fL[2], 1, wX[2]);
                                event(3, fM[0], 0, wM[0]);
                                event(4, fM[1], 0, wM[1]);
                                event(5, fM[2], 0, wM[2]);
                                event(6, fU[0], 2, wY[0]);
                                event(7, fU[2], 2, wY[1]);
                                event(8, fU[1], 2, wY[2]);
                            }
                            if (cX < CPK_SYM_N && cY < CPK_SYM_N) {  // emissions are counted for ACGT x ACGT cells only (:429)
                                double *copy = eLds + (lane & (kExpectCopies - 1)) * 80;
#pragma unroll
                                for (int s = 0; s < S; s++) atomicAdd(&copy[s * 16 + cX * 4 + cY], eAcc[s]);
                            }
                        }
                        if (act) e2d--;
                    }
                }
            } else if (kIndel) {
                // ---------------- the three lists of the segment (:691-733): every emitted cell, diagonals ascending, x-y
                // descending inside a diagonal (the reference appends in x-y ascending order and the lists are reversed
                // on the way out, SURVEY 8b "list order") ----------------
                if (segOn && c == 0) {
#pragma unroll
                    for (int l = 0; l < NL; l++) a.segStarts[(size_t)l * a.nSegsTotal + rg.segOff + si] = countL[l];
                }
                int ed = segOn ? sg.tbPrev + 1 : 0;
                while (__ballot(segOn && ed <= sg.tbFrom)) {
                    const bool more = segOn && ed <= sg.tbFrom;
                    const int cnt = more ? (sg.tbFrom - ed + 1 < kPackChunk ? sg.tbFrom - ed + 1 : kPackChunk) : 0;
                    int x0, y0;
                    stage_chunk(more, ed, 1, cnt, 0, x0, y0);  // (the table entries; the symbol windows are not read)
                    for (int i = 0; i < kPackChunk; i++) {
                        if (!__ballot(i < cnt)) break;
                        const bool act = i < cnt;
                        const CpkDiag e = act ? unpack(ebuf[i]) : CpkDiag{0, 1, 0, 0};
                        const int W = e.width;
                        const bool on = act && c < W;
                        const int kc = on ? c : 0;  // every load is unconditional: cell 0 of a valid diagonal
                        const double total = ld_self(totals + (act ? (sg.tbFrom - ed) / CPK_REFRESH_PERIOD : 0));
                        const double *fr = ringAt(e);
                        const double *bo = bring + (size_t)(act ? e.cellOff - bBase + kc : 0) * S;
                        double fbv[NL];
#pragma unroll
                        for (int l = 0; l < NL; l++) fbv[l] = ld_self(fr + SW::ringIdx(W, l, kc)) + ld_self(bo + l);
                        const int x = ((ed + e.xmyL) >> 1) + kc, y = ed - x;
#pragma unroll
                        for (int l = 0; l < NL; l++) {
                            const bool cell = l == 0 ? (x > 0 && y > 0) : (l == 1 ? x > 0 : y > 0);  // :700, :711, :722
                            double p = (on && cell) ? exp(fbv[l] - total) : 0.0;
                            const bool keep = on && cell && p >= thr;
                            const unsigned long long mask = __ballot(keep);
                            if (keep) {
                                if (p > 1.0) p = 1.0;
                                const int pos = countL[l] + __popcll(mask & aboveMe);
                                if (pos < rg.outCap) {
                                    int32_t *o = a.triples + 3 * ((size_t)l * a.outTriplesPerList + rg.outOff + pos);
                                    o[0] = (int32_t)floor(p * (double)CPECAN_PROB_1);
                                    o[1] = x - 1;
                                    o[2] = y - 1;
                                }
                            }
                            countL[l] += __popcll(mask & groupBits);
                        }
                        if (act) ed++;
                    }
                }
            } else {
            // ---------------- thresholded posteriors from the candidates, walked backwards (:655-689) ----------------
                // (kModeTrace: the segment's own part of the region's slice -- the other segments are other items')
                int32_t *outSeg = MODE == kModeTrace ? out + 3 * (size_t)sg.outOff : out;
                const int outCap = MODE == kModeTrace ? sg.outCap : rg.outCap;
                if (segOn && c == 0) a.segStarts[rg.segOff + itemSeg + si] = MODE == kModeTrace ? sg.outOff : count;
                // kEmitU candidates per lane and pass: their loads (candidate, then the total of its diagonal) are in flight
                // together -- one candidate per lane was two dependent global round trips for every GW pairs
                constexpr int kEmitU = 4;
                for (int top = nCand; __ballot(segOn && top > 0); top -= kEmitU * GW) {
                    double fbv[kEmitU], tot[kEmitU];
                    long long xyv[kEmitU];
                    bool valid[kEmitU];
#pragma unroll
                    for (int u = 0; u < kEmitU; u++) {
                        const int i = top - 1 - (u * GW + c);
                        valid[u] = segOn && i >= 0;
                        const Candidate *cp = cand + (valid[u] ? i : 0);  // slot 0 exists in every group's list
                        fbv[u] = ld_self(&cp->fb);
                        xyv[u] = __hip_atomic_load(reinterpret_cast<const long long *>(&cp->x), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
#pragma unroll
                    for (int u = 0; u < kEmitU; u++) {
                        const int x = (int)(xyv[u] & 0xffffffffll), y = (int)(xyv[u] >> 32);
                        int jt = valid[u] ? (sg.tbFrom - (x + y)) / CPK_REFRESH_PERIOD : 0;
                        jt = jt < 0 ? 0 : jt;
                        tot[u] = ld_self(totals + jt);
                    }
#pragma unroll
                    for (int u = 0; u < kEmitU; u++) {
                        const int x = (int)(xyv[u] & 0xffffffffll), y = (int)(xyv[u] >> 32);
                        double p = valid[u] ? exp(fbv[u] - tot[u]) : 0.0;
                        const bool keep = valid[u] && p >= thr;
                        const unsigned long long mask = __ballot(keep);
                        if (keep) {
                            if (p > 1.0) p = 1.0;
                            const int pos = count + __popcll(mask & belowMe);
                            if (pos < outCap) {
                                outSeg[3 * (size_t)pos + 0] = (int32_t)floor(p * (double)CPECAN_PROB_1);
                                outSeg[3 * (size_t)pos + 1] = x - 1;
                                outSeg[3 * (size_t)pos + 2] = y - 1;
                            }
                        }
                        count += __popcll(mask & groupBits);
                    }
                }
                if (MODE == kModeTrace && segOn && c == 0) a.segCounts[rg.segOff + itemSeg + si] = count;
            }
            // ---------------- the traceback used the rolling buffers: restore F[dTop-1], F[dTop] ----------------
            if (MODE == kModeWhole && segOn && !sg.atEnd) {
#pragma unroll
                for (int back = 1; back >= 0; back--) {
                    const int dd = sg.dTop - back;
                    const CpkDiag e = back ? e2 : e1;
                    if (c < e.width) {
                        const double *src = ringAt(e);
                        double *cur = fbuf1(dd);
#pragma unroll
                        for (int s = 0; s < S; s++) cur[s + c * R] = ld_self(src + SW::ringIdx(e.width, s, c));
                    }
                }
            }
        }
        if (MODE == kModeWhole && have && c == 0) {
            if (kIndel) {
#pragma unroll
                for (int l = 0; l < NL; l++) a.outCounts[(size_t)l * a.geo.nRegions + r] = countL[l];
            } else {
                a.outCounts[r] = count;
            }
        }
    }
    if (kExpect) {
        // one partial result per wave, as in the sweep kernel: [0,25) transitions [from*S+to], [25,105) emissions, [105] likelihood
        __syncthreads();
        double *dst = a.expectOut + (size_t)blockIdx.x * 128;
        constexpr int kFrom5[13] = {0, 1, 0, 3, 0, 1, 2, 3, 4, 0, 2, 0, 4}, kTo5[13] = {1, 1, 3, 3, 0, 0, 0, 0, 0, 2, 2, 4, 4};
        constexpr int kFrom3[9] = {0, 1, 2, 0, 1, 2, 0, 2, 1}, kTo3[9] = {1, 1, 1, 0, 0, 0, 2, 2, 2};
        for (int i = lane; i < 25; i += CPK_WAVE) dst[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kNT; i++) {
            double v = tAcc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            const int idx = S == 5 ? kFrom5[i] * 5 + kTo5[i] : kFrom3[i] * 3 + kTo3[i];
            if (lane == 0) dst[idx] = v;
        }
        for (int i = lane; i < 80; i += CPK_WAVE) {
            double e = 0.0;
            for (int k = 0; k < kExpectCopies; k++) e += eLds[k * 80 + i];
            dst[25 + i] = e;
        }
        double lk = likelihood;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lk += __shfl_xor(lk, off);
        if (lane == 0) dst[105] = lk;
    }
}
