// cpk_team.inl -- the team sweep kernel: T waves of one workgroup share one wide DP region (match, indel and expectation emitters; LDS rows).
// Part of the single HIP translation unit cpecan_kernels.hip (included there, after cpk_sweep.inl); not compiled on its own.
//
// A band of a few hundred cells per diagonal needs tens of KB of LDS for its rolling rows: with one wave per region a CU
// holds three such regions, i.e. three waves, and every diagonal is a serial chain of up to a dozen 64-cell groups on one
// wave.  Here the T waves of a workgroup own ONE region: the tables, the symbol strings and the rolling rows are in LDS
// once, every wave takes a contiguous range of the diagonal's cells (whole groups of 64), and the waves meet at one
// barrier per diagonal.  The cell arithmetic is Sweep's (fwdCells / bwdCells: their contexts are row pointers and shifts);
// what differs from cpk_sweep.inl:
//   * three forward diagonals rotate through the rows (3S rows per position): a team cannot overwrite F[d-2] in place,
//     the single wave's direction trick does not carry over; the traceback uses rows 0..2S of the same buffer;
//   * candidates: every wave keeps the candidates of its cells of the current diagonal in registers; at the barrier the
//     waves exchange their counts through LDS, and each writes its run behind the runs of the waves before it -- the
//     segment's candidate list comes out in the single-wave order (diagonal descending, x-y ascending), so the totals
//     fold and the emission are Sweep's, run by wave 0 while the other waves restore the forward rows;
//   * the candidate bound (lastMax) is the maximum over the waves' maxima on refresh diagonals, exchanged the same way;
//   * F rows are read from the forward ring where they are used (the other waves of the team cover the latency).
constexpr int kTeamGroups = 3;  // 64-cell groups a wave handles per diagonal at most: bands up to 64 * T * kTeamGroups cells
constexpr int kTeamXchg = 48;   // doubles of LDS for the exchange area (counts per list, maxima, the region ticket)

// (expect: the workgroup's emission sums of the expectation emitter, kExpectCopies copies of 80, behind the exchange area)
__host__ __device__ constexpr int team_header_doubles(bool expect = false) {
    return kLdsCubics + kLdsEm + kLdsWeights + kTeamXchg + (expect ? kExpectCopies * 80 : 0);
}

// EMIT (round 4): CPECAN_EMIT_MATCH, or CPECAN_EMIT_INDEL -- the three lists of diagonalCalculationPosteriorProbs
// (pairwiseAligner.c:691-733: match, gapX, gapY; what getShiftedMEAAlignment needs): every state of every forward diagonal
// goes to the ring, a wave keeps the candidates of three lists per diagonal and the waves exchange three counts.
template <int S, int T, int EMIT = CPECAN_EMIT_MATCH>
__global__ void __launch_bounds__(CPK_WAVE *T) cpecan_pairhmm_team(const KArgs a) {
    // ... or CPECAN_EMIT_EXPECT: the traceback parks B of the emitted cells (`bring`, the layout Sweep::traceback writes),
    // wave 0 folds the totals, and the T waves share the second pass (Sweep::expectations: every T-th item of 64 cells each),
    // every wave with transition sums of its own, the emission sums of the workgroup in LDS.
    static_assert(EMIT == CPECAN_EMIT_MATCH || EMIT == CPECAN_EMIT_INDEL || EMIT == CPECAN_EMIT_EXPECT, "team kernel: match, match + indel lists, expectations");
    constexpr int NL = EMIT == CPECAN_EMIT_INDEL ? 3 : 1;
    constexpr bool kExpect = EMIT == CPECAN_EMIT_EXPECT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int R = 3 * S;
    using SweepT = Sweep<S, true, R>;
    const int tid = threadIdx.x, wave = tid / CPK_WAVE, lane = tid & (CPK_WAVE - 1);
    const CpkModel &m = *a.model;
    const int stride = a.geo.rollStride;

    // LDS (doubles): logAdd cubics | emissions | weights | exchange area | rolling rows | symbol strings
    logadd_fp_mode();  // (every wave: fill_cubics below sets it for the one wave that fills the table)
    if (wave == 0) {
        fill_cubics(lds);
        fill_weights<S>(lds + kLdsCubics + kLdsEm, m, a.kc, lane);
    }
    const Cubic *lg = reinterpret_cast<const Cubic *>(lds);
    double *em = lds + kLdsCubics;
    double *wt = lds + kLdsCubics + kLdsEm;
    int *xi = reinterpret_cast<int *>(lds + kLdsCubics + kLdsEm + kLdsWeights);  // [0]: ticket, [8 + parity * T + w]: counts
    float *xf = reinterpret_cast<float *>(xi + 8 + 2 * NL * T);              // [parity * T + w]: maxima (counts: [8 + (parity * NL + l) * T + w])
    double *eLds = lds + team_header_doubles();  // (expectation emitter only)
    double *roll = lds + team_header_doubles(kExpect);
    uint8_t *seqLds = reinterpret_cast<uint8_t *>(roll + (size_t)R * stride);
    for (int i = tid; i < R * stride; i += CPK_WAVE * T) roll[i] = NEG_INF;  // position 0 of every row stays the guard
    if (kExpect)
        for (int i = tid; i < kExpectCopies * 80; i += CPK_WAVE * T) eLds[i] = 0.0;
    constexpr int kNT = S == 5 ? 13 : 9;
    double tAcc[kNT];  // this wave's transition sums (expectation emitter), one per transition in list order
#pragma unroll
    for (int i = 0; i < kNT; i++) tAcc[i] = 0.0;
    double likelihood = 0.0;
    __syncthreads();

    // forward diagonal d (d >= -1), state s of cell k: frow(d)[s + k * R]
    auto frow = [&](int d) { return roll + R + ((d + 3) % 3) * S; };
    const size_t slot = blockIdx.x;
    const float logThr = (float)log(m.threshold);
    for (;;) {
        if (wave == 0) {  // wave-uniform test; inside, every lane takes part (see the note in cpecan_pairhmm_sweep)
            const unsigned int ticket = atomicAdd(a.queue, lane == 0 ? 1u : 0u);
            const int tk0 = __builtin_amdgcn_readfirstlane((int)ticket);
            if (lane == 0) xi[0] = tk0;
        }
        __syncthreads();
        const int tk = xi[0];
        __syncthreads();  // everyone has read the ticket before the next one overwrites it
        if (tk >= a.regionCount) break;
        const int r = a.regionBase + tk;
        const CpkRegion &rg = a.regions[r];
        const int lX = rg.lX, lY = rg.lY, N = lX + lY;
        const uint8_t *gx = a.symbols + rg.seqXOff, *gy = a.symbols + rg.seqYOff;
        {
            stage_symbols<CPK_WAVE * T>(seqLds, gx, lX + 2, tid);
            stage_symbols<CPK_WAVE * T>(seqLds + ((lX + 3) >> 1), gy, lY + 2, tid);
        }
        const CpkDiag *table = a.diags + rg.diagOff;
        SweepT sw{a,
                  a.kc,
                  DiagCache{table, N, 0, lane, 0, 0, 0, 0},
                  seqLds,
                  seqLds + ((lX + 3) >> 1),
                  roll,
                  em,
                  wt,
                  lg,
                  a.ring + slot * (size_t)a.geo.ringCells * S,
                  a.cand + slot * (size_t)a.geo.fbCells * NL,
                  nullptr,
                  a.cbuf + slot * (size_t)a.geo.refreshCells,
                  a.mbuf + slot * (size_t)a.geo.refreshCells,
                  a.totals + slot * (size_t)a.geo.maxRefresh,
                  stride,
                  lane,
                  lane * R,
                  N,
                  CpkDiag{},
                  CpkDiag{}};
        if (kExpect) {
            sw.bring = a.bring + slot * (size_t)a.geo.fbCells * S;  // B of a segment's emitted cells
            sw.expStride = T;  // the second pass: this wave takes every T-th item
            sw.expPhase = wave;
        }
        __syncthreads();  // symbols staged
        int count[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) count[l] = 0;
#ifdef CPK_DIAGNOSTICS
        if (a.geo.debug & 4) {  // diagnostic build (CPECAN_DEBUG_SKIP=4): regions are fetched and staged, nothing is computed
            if (tid == 0) a.outCounts[r] = 0;
            continue;
        }
#endif
        if (N > 0) {
            sw.dc.load(0);
            const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
            CpkDiag f1 = sw.dc.get(0, false), f2 = f1;  // table entries of the two previous forward diagonals
            if (tid < S) {
                frow(0)[tid] = startPrior[tid];
                sw.ringAt(f1)[tid] = startPrior[tid];
            }
            __syncthreads();
            // this wave's range of a diagonal of W cells: whole groups of 64, contiguous, in wave order
            auto range = [&](int W, int &lo, int &hi) {
                const int groups = (W + CPK_WAVE - 1) / CPK_WAVE, per = (groups + T - 1) / T;
                lo = wave * per * CPK_WAVE;
                hi = lo + per * CPK_WAVE;
                lo = lo < W ? lo : W;
                hi = hi < W ? hi : W;
            };
            int d = 1;
            int emitSeg = 0, emitFrom = a.segs[rg.segOff].tbFrom;
            for (int si = 0; si < rg.nSeg; si++) {
                const CpkSegment sg = a.segs[rg.segOff + si];
                // ---- forward sweep up to dTop (pairwiseAligner.c:609-629) ----
                while (d <= sg.dTop) {
                    sw.dc.load(d);
                    const int dEnd = d + CPK_WAVE - 1 < sg.dTop ? d + CPK_WAVE - 1 : sg.dTop;
                    for (; d <= dEnd; d++) {
                        while (d > emitFrom) emitFrom = a.segs[rg.segOff + ++emitSeg].tbFrom;
                        const bool all = EMIT != CPECAN_EMIT_MATCH || (emitFrom - d) % CPK_REFRESH_PERIOD == 0 || d >= sg.dTop - 1;
                        const CpkDiag g = sw.dc.at(d - sw.dc.base);
                        const int W = g.width;
                        typename SweepT::FwdCtx c;
                        c.d = d;
                        c.xlo = (d + g.xmyL) >> 1;
                        c.dlR = ((g.xmyL - 1 - f1.xmyL) >> 1) * R;
                        c.w1R = f1.width * R;
                        c.dmR = ((g.xmyL - f2.xmyL) >> 1) * R;
                        c.w2R = d >= 2 ? f2.width * R : 0;
                        c.p1 = frow(d - 1);
                        c.p2 = frow(d - 2);
                        double *cur = frow(d);
                        double *out = sw.ringAt(g);
                        int lo, hi;
                        range(W, lo, hi);
                        for (int kb = lo; kb < hi; kb += CPK_WAVE) {
                            const int k0 = kb + lane;
                            if (k0 < hi) {
                                const int kk[1] = {k0};
                                const int kkR[1] = {kb * R + lane * R};
                                double v[1][S];
                                sw.template fwdCells<1>(c, kk, kkR, v);
#pragma unroll
                                for (int s = 0; s < S; s++) cur[s + kkR[0]] = v[0][s];
                                out[SweepT::ringIdx(W, 0, k0)] = v[0][0];
                                if (all) {
#pragma unroll
                                    for (int s = 1; s < S; s++) out[SweepT::ringIdx(W, s, k0)] = v[0][s];
                                }
                            }
                        }
                        __syncthreads();
                        f2 = f1;
                        f1 = g;
                    }
                }
#ifdef CPK_DIAGNOSTICS
                if (a.geo.debug & 2) continue;  // diagnostic (CPECAN_DEBUG_SKIP=2): the forward sweep alone, no output
#endif
                // ---- traceback of the segment (pairwiseAligner.c:796-862) ----
                const double *endPrior = (sg.atEnd && rg.raggedRight) ? m.raggedEnd : m.end;
                double ep[S];
#pragma unroll
                for (int s = 0; s < S; s++) ep[s] = endPrior[s];
                const int J = sg.nRefresh;
                float lastMax = -__builtin_huge_valf();
                int nCand[NL];
#pragma unroll
                for (int l = 0; l < NL; l++) nCand[l] = 0;
                int untilRefresh = sg.dTop - sg.tbFrom, jr = 0;
                CpkDiag gb{}, ga{};
                CpkDiag g = sw.dc.get(sg.dTop, true);
                const int bBase = kExpect ? table[sg.tbPrev + 1].cellOff : 0;
                // F.match of this wave's cells, one diagonal ahead of its use (the ring is in HBM): group gi of the range
                auto loadF = [&](const CpkDiag &gd, double (&dst)[NL][kTeamGroups]) {
                    int flo, fhi;
                    range(gd.width, flo, fhi);
                    const double *src = sw.ringAt(gd);
#pragma unroll
                    for (int gi = 0; gi < kTeamGroups; gi++) {
                        int k = flo + gi * CPK_WAVE + lane;
                        k = k < gd.width ? k : gd.width - 1;
#pragma unroll
                        for (int l = 0; l < NL; l++) dst[l][gi] = ld_self(src + SweepT::ringIdx(gd.width, l, k > 0 ? k : 0));
                    }
                };
                double fCur[NL][kTeamGroups];
                loadF(g, fCur);
                for (int d2 = sg.dTop; d2 > sg.tbPrev;) {
                    sw.dc.load(d2 - 1 - (CPK_WAVE - 1));  // entries of the 64 diagonals ending at d2-1
                    for (int ci = CPK_WAVE - 1; ci >= 0 && d2 > sg.tbPrev; ci--, d2--) {
                        const bool seeded = d2 == sg.dTop;
                        const int W = g.width;
                        const bool emit = d2 <= sg.tbFrom;
                        const bool refresh = untilRefresh == 0;
                        const bool feeds = untilRefresh == 1 && d2 - 1 > sg.tbPrev;
                        const CpkDiag gnext = sw.dc.at(ci);  // entry of d2-1 (of diagonal 0 when d2 < 1: not used then)
                        double fNext[NL][kTeamGroups];
                        loadF(gnext, fNext);
                        double *curM = sw.bM1(d2), *curG = sw.bG1(d2);
                        const double *fsrc = sw.ringAt(g);
                        const int xlo = (d2 + g.xmyL) >> 1;
                        typename SweepT::BwdCtx c;
                        c.d2 = d2;
                        c.xlo = xlo;
                        c.dbR = ((g.xmyL - 1 - gb.xmyL) >> 1) * R;
                        c.wBR = seeded ? 0 : gb.width * R;
                        c.daR = ((g.xmyL - ga.xmyL) >> 1) * R;
                        c.wAR = (!seeded && d2 + 2 <= sg.dTop) ? ga.width * R : 0;
                        c.pb = sw.bG1(d2 + 1);
                        c.pa = sw.bM1(d2 + 2);
                        const float keepFrom = lastMax + logThr - kCandMargin;
                        int lo, hi;
                        range(W, lo, hi);
                        // this wave's candidates of the diagonal wait in registers for the other waves' counts
                        double pfb[NL][kTeamGroups];
                        int px[kTeamGroups];
                        unsigned long long pmask[NL][kTeamGroups];
                        int myCount[NL];
#pragma unroll
                        for (int l = 0; l < NL; l++) myCount[l] = 0;
                        float myMax = -__builtin_huge_valf();
                        // refresh diagonals need the other states of F as well (cell dot products): their loads go out
                        // here, ahead of the diagonal's arithmetic
                        double rf[kTeamGroups][S];
                        if (refresh) {
#pragma unroll
                            for (int gi = 0; gi < kTeamGroups; gi++) {
                                int k = lo + gi * CPK_WAVE + lane;
                                k = k < W ? k : W - 1;
#pragma unroll
                                for (int s2 = 1; s2 < S; s2++) rf[gi][s2] = ld_self(fsrc + SweepT::ringIdx(W, s2, k > 0 ? k : 0));
                            }
                        }
#pragma unroll
                        for (int gi = 0; gi < kTeamGroups; gi++) {
#pragma unroll
                            for (int l = 0; l < NL; l++) {
                                pmask[l][gi] = 0;
                                pfb[l][gi] = 0.0;
                            }
                            px[gi] = 0;
                            const int kb = lo + gi * CPK_WAVE;
                            if (kb < hi) {  // wave-uniform
                                const int k0 = kb + lane;
                                const bool on = k0 < hi;
                                const int kc = on ? k0 : hi - 1;
                                const int kR0 = kc * R;
                                double v[1][S];
                                if (seeded) {
#pragma unroll
                                    for (int s = 0; s < S; s++) v[0][s] = ep[s];
                                } else {
                                    const int kk[1] = {kc};
                                    const int kkR[1] = {kR0};
                                    sw.template bwdCells<1>(c, kk, kkR, v);
                                }
                                if (on) {
                                    curM[kR0] = v[0][0];
#pragma unroll
                                    for (int s = 1; s < S; s++) curG[s + kR0] = v[0][s];
                                }
                                const int x = xlo + kc, y = d2 - x;
                                if (kExpect && emit && on) {  // kept for the second pass, per group of gN cells, state-major
                                    const int gN = W - kb < CPK_WAVE ? W - kb : CPK_WAVE;
                                    double *bo = sw.bring + (size_t)(g.cellOff - bBase + kb) * S + lane;
#pragma unroll
                                    for (int s = 0; s < S; s++) bo[s * gN] = v[0][s];
                                }
                                if (emit || feeds) {
                                    const double f0 = fCur[0][gi];
                                    const double fb = f0 + v[0][0];
                                    if (feeds && on) sw.mbuf[(size_t)k0 * J + jr] = fb;  // series of the refresh point below (:647)
                                    if (emit) {
                                        px[gi] = x;
#pragma unroll
                                        for (int l = 0; l < NL; l++) {
                                            // match cells need x > 0 and y > 0, gapX cells x > 0, gapY cells y > 0 (:700, :711, :722)
                                            const bool cell = l == 0 ? (x > 0 && y > 0) : (l == 1 ? x > 0 : y > 0);
                                            const double fbl = l == 0 ? fb : fCur[l][gi] + v[0][l];
                                            const bool keep = !kExpect && on && cell && (float)fbl >= keepFrom;
                                            pmask[l][gi] = __ballot(keep);
                                            pfb[l][gi] = fbl;
                                            myCount[l] += __popcll(pmask[l][gi]);
                                        }
                                        if (refresh) {
                                            // cell dot product over the states (pairwiseAligner.c:402-408) and this wave's
                                            // share of the diagonal's largest F.m + B.m
                                            float fbf = -__builtin_huge_valf();
                                            if (on) {
                                                double t = fb;
#pragma unroll
                                                for (int s2 = 1; s2 < S; s2++)
                                                    t = logadd(lg, t, rf[gi][s2] + v[0][s2]);
                                                sw.cbuf[(size_t)k0 * J + jr] = t;
                                                if (x > 0 && y > 0) fbf = (float)fb;
                                            }
                                            myMax = fmaxf(myMax, wave_max_f32(fbf));
                                        }
                                    }
                                }
                            }
                        }
                        // ---- the waves meet: the diagonal is complete, counts and maxima are exchanged ----
                        const int par = d2 & 1;
                        if (lane == 0) {
#pragma unroll
                            for (int l = 0; l < NL; l++) xi[8 + (par * NL + l) * T + wave] = myCount[l];
                            xf[par * T + wave] = myMax;
                        }
                        __syncthreads();
                        float dmax = -__builtin_huge_valf();
#pragma unroll
                        for (int w = 0; w < T; w++) dmax = fmaxf(dmax, xf[par * T + w]);
#pragma unroll
                        for (int l = 0; l < NL; l++) {
                            int before = 0, total = 0;
#pragma unroll
                            for (int w = 0; w < T; w++) {
                                const int cw = xi[8 + (par * NL + l) * T + w];
                                before += w < wave ? cw : 0;
                                total += cw;
                            }
                            if (emit) {
                                int at = nCand[l] + before;
#pragma unroll
                                for (int gi = 0; gi < kTeamGroups; gi++) {
                                    const unsigned long long mk = pmask[l][gi];
                                    if (mk) {  // wave-uniform
                                        if ((mk >> lane) & 1ull) {
                                            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                                            Candidate cd;
                                            cd.fb = pfb[l][gi];
                                            cd.x = px[gi];
                                            cd.y = d2 - px[gi];
                                            sw.cand[(size_t)l * a.geo.fbCells + at + rank] = cd;
                                        }
                                        at += __popcll(mk);
                                    }
                                }
                                nCand[l] += total;
                            }
                        }
                        if (refresh) lastMax = fmaxf(dmax, lastMax - 1.0f);
                        ga = gb;
                        gb = g;
                        g = gnext;
#pragma unroll
                        for (int gi = 0; gi < kTeamGroups; gi++) {
#pragma unroll
                            for (int l = 0; l < NL; l++) {
                                asm volatile("" : "+v"(fNext[l][gi]));  // waited for here, a diagonal after the loads were issued
                                fCur[l][gi] = fNext[l][gi];
                            }
                        }
                        if (refresh) {
                            untilRefresh = CPK_REFRESH_PERIOD - 1;
                            jr++;
                        } else {
                            untilRefresh--;
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();  // every candidate and refresh series is written
                // wave 0 folds the totals and emits while the others already put the forward rows back
                if (wave == 0) {
                    sw.template foldTotals<true>(sg, table);
#pragma unroll
                    for (int l = 0; l < (kExpect ? 0 : NL); l++) {
                        if (lane == 0) a.segStarts[(size_t)l * a.nSegsTotal + rg.segOff + si] = count[l];
                        count[l] = sw.emitMatches(sg, sw.cand + (size_t)l * a.geo.fbCells, nCand[l],
                                                  a.triples + 3 * ((size_t)l * a.outTriplesPerList + rg.outOff), rg.outCap, count[l]);
                    }
                }
                if (!sg.atEnd) {
                    // the traceback reused the rows: put F[dTop-1] and F[dTop] back for the forward sweep
                    const CpkDiag gTopM1 = sw.dc.get(sg.dTop - 1, false);
                    const CpkDiag gTop = sw.dc.get(sg.dTop, false);
                    // (nobody reads backward rows any more: the barrier above came after the last diagonal; wave 0 is
                    // busy with the totals and the emission, the other waves do this)
                    for (int which = 0; which < 2 && wave > 0; which++) {
                        const CpkDiag &gd = which ? gTop : gTopM1;
                        const int dd = which ? sg.dTop : sg.dTop - 1;
                        double *cur = frow(dd);
                        const double *src = sw.ringAt(gd);
                        for (int k = tid - CPK_WAVE; k < gd.width; k += CPK_WAVE * (T - 1)) {
#pragma unroll
                            for (int s = 0; s < S; s++) cur[s + k * R] = ld_self(src + SweepT::ringIdx(gd.width, s, k));
                        }
                    }
                    f2 = gTopM1;
                    f1 = gTop;
                }
                if (kExpect) {
                    __threadfence_block();
                    __syncthreads();  // the totals are written
                    sw.expectations(sg, tAcc, eLds, likelihood);
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int l = 0; l < NL; l++)
            if (tid == 0) a.outCounts[(size_t)l * a.geo.nRegions + r] = count[l];
        // count lives in wave 0 only: tid 0 is lane 0 of wave 0
    }
    if (kExpect) {
        // one partial result per WAVE (cpecan_pairhmm_sweep writes one per workgroup of one wave): [0,25) transitions
        // [from*S+to], [25,105) emissions -- the workgroup's, by wave 0 -- [105] likelihood
        __syncthreads();
        double *dst = a.expectOut + ((size_t)blockIdx.x * T + wave) * 128;
        constexpr int kFrom5[13] = {0, 1, 0, 3, 0, 1, 2, 3, 4, 0, 2, 0, 4}, kTo5[13] = {1, 1, 3, 3, 0, 0, 0, 0, 0, 2, 2, 4, 4};
        constexpr int kFrom3[9] = {0, 1, 2, 0, 1, 2, 0, 2, 1}, kTo3[9] = {1, 1, 1, 0, 0, 0, 2, 2, 2};
        for (int i = lane; i < 106; i += CPK_WAVE) dst[i] = 0.0;
        __threadfence_block();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kNT; i++) {
            double v = tAcc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            const int idx = S == 5 ? kFrom5[i] * 5 + kTo5[i] : kFrom3[i] * 3 + kTo3[i];
            if (lane == 0) dst[idx] = v;
        }
        if (wave == 0) {
            for (int i = lane; i < 80; i += CPK_WAVE) {
                double e = 0.0;
                for (int k = 0; k < kExpectCopies; k++) e += eLds[k * 80 + i];
                dst[25 + i] = e;
            }
        }
        double v = likelihood;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) dst[105] = v;
    }
}
