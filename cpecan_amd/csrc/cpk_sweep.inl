// cpk_sweep.inl -- the sweep kernel: one wave per DP region (forward stream, traceback, expectation step, totals, emission).
// Part of the single HIP translation unit cpecan_kernels.hip (included there, in this order); not compiled on its own.

// ROWS: rows of the rolling buffers per position (2S+1 for one wave per region; the team kernel keeps three forward
// diagonals, 3S rows)
// COH: the forward ring is handed from wave to wave inside ONE launch (kModeFused): its stores and loads are device-scope
// (sc1, write-through / read-through) so that no L2 write-back or invalidate is needed around the hand-off
// RDBL: a table entry's ringOff counts doubles (the ring of a split region keeps only the states that are read back,
// cpk_table_gather.inl) instead of cells of S doubles
// ABS: the rolling rows are indexed by a cell's POSITION (its matrix diagonal, cpk_table_gather.inl "positions") instead of
// its rank on the anti-diagonal: neighbours sit at constant offsets, out-of-band ones read -inf without a range test.
// The rows are then TWO arrays of S rows (ROWS = S), one per parity of the diagonal, `setStride` doubles apart: S is odd
// for both models, so a lane stride of S * 8 bytes is as conflict-free as 2S + 1 rows were, without the padding row
// (round 4: 1.2 KB of LDS per wave at BASELINE config B, part of what the tenth wave per CU needs).
// XG: 64-lane groups per diagonal tracebackExpect() is unrolled for (1: no diagonal of the class is wider than a wave)
template <int S, bool FAST, int ROWS = 2 * S + 1, bool COH = false, bool RDBL = false, bool ABS = false, int XG = 2>
struct Sweep {
    const KArgs &a;
    const KConsts &m;  // kernarg-resident constants
    DiagCache dc;
    // padded symbol strings: symbol p of X is the base x-1 (p = 0 and p = lX+1 read as N).  FAST: two symbols per
    // byte in LDS (low nibble = even p); otherwise one byte per symbol in global memory.
    const uint8_t *sxp;
    const uint8_t *syp;
    // One 8-byte LDS read that the compiler may NOT pair with its neighbour into ds_read2_b64: on gfx950 a wave64
    // ds_read_b64 takes 2 LDS cycles (256 B/clk) and ds_read2_b64 8 (128 B/clk, MI355X_MICROARCH.md "LDS"), and the LDS
    // array, shared by the four SIMDs, is the busiest unit of this kernel.  A relaxed wave-scope atomic load is an
    // ordinary ds_read_b64 that the load/store optimiser leaves alone.
    __device__ __forceinline__ static double lds1(const double *p) {
        return FAST ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : *p;
    }
    __device__ __forceinline__ static double tab1(const double *p) {  // tables are in LDS in every variant
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __device__ __forceinline__ int symX(int p) const { return FAST ? (sxp[p >> 1] >> ((p & 1) * 4)) & 15 : sxp[p]; }
    __device__ __forceinline__ int symY(int p) const { return FAST ? (syp[p >> 1] >> ((p & 1) * 4)) & 15 : syp[p]; }
    double *roll;        // rolling buffers: `stride` positions of R = 2S+1 doubles; position 0 = -inf guard
    const double *em;    // LDS emissions: [0..24] match, [25..29] gapX, [30..34] gapY
    // LDS (emission + transition) sums, the second operand of every DP term `from + (eP + tP)` (pairwiseAligner.c:384):
    //   wt[(cX*5 + cY)*kWM + i]             match emission + {matchContinue, matchFromShortX, matchFromShortY, [matchFromLongX, matchFromLongY]}
    //   wt[25*kWM + cX*kWG + i]             gapX emission  + {open, extend, [longOpen, longExtend] | switchToX}
    //   wt[25*kWM + 5*kWG + cY*kWG + i]     gapY emission  + the same for Y
    // One table fetch replaces an emission fetch plus one fp64 add per term (13 adds per cell and direction).
    const double *wt;
    static constexpr int kWM = S == 5 ? 5 : 3, kWG = S == 5 ? 4 : 3;
    const Cubic *lg;     // LDS logAdd cubics
    double *ring;
    Candidate *cand;
    Candidate *stage;  // LDS: kStage candidates per output list, see traceback()
    double *cbuf, *mbuf, *totals;
    int stride;
    int lane;
    int laneR;  // lane * R
    int N;
    // forward sweep state: the two previous diagonals' table entries
    CpkDiag f1, f2;

    // Rolling buffers, position-major: element (row r, position i) is roll[i * R + r], R = 2S+1 rows, positions
    // 0..stride-1, position 0 of every row is the -inf guard.  With the row a compile-time offset the rows of one
    // position cost one address VGPR and immediate offsets (adjacent rows pair up into ds_read2/ds_write2_b64), and
    // a lane stride of R*8 bytes (R odd) is bank-conflict-free for 8-byte accesses.
    //  forward layout : two diagonals, F[d] = rows [(d&1)*S, (d&1)*S + S); F[d] overwrites F[d-2] in place
    //  backward layout: match row in a ring of three, B[d].match = row (d mod 3); the other states in two alternating
    //                   groups, B[d][s] = row 3 + (d&1)*(S-1) + (s-1) for s >= 1
    //  fbuf1/bM1/bG1 return the row pointer at position 1 (cell 0); bG1(d)[s + kR] is state s >= 1 of cell k
    static constexpr int R = ROWS;
    // Cell k of a diagonal lives at position k+1, i.e. at element offset k*R from a row pointer that already points at
    // position 1 (fbuf1/bM1/bG1 below).  Cell indices are kept premultiplied by R ("kR"): lane*R is computed once per
    // kernel and everything added to it per diagonal / per group is wave-uniform, so no per-access multiply is left.
    // sel(iR, wR): element offset of neighbour cell i of a diagonal with w cells (wR = w*R; w = 0: no such diagonal),
    // or the offset of the -inf guard (position 0) when the neighbour is outside the band.
    __device__ __forceinline__ static int sel(int iR, int wR) { return ABS ? iR : (((unsigned)iR < (unsigned)wR) ? iR : -R); }
    __device__ __forceinline__ double *fbuf1(int d) const { return roll + R + (d & 1) * S; }
    __device__ __forceinline__ double *bM1(int d) const { return roll + R + (d + 3) % 3; }
    __device__ __forceinline__ double *bG1(int d) const { return roll + R + 2 + (d & 1) * (S - 1); }
    // Forward ring in HBM, per diagonal of W cells: the match row [W], then the other states cell-major [W][S-1]
    // (the traceback reads the match row on its own; a cell's remaining states go out as one 32-byte run).
    // (Rings of a split region, RDBL: the match row is padded to an even number of doubles and every diagonal starts on
    // one -- cpecan_build_diag_table -- so that pairs of match cells and a cell's other states are 16-byte aligned.)
    __device__ __forceinline__ static size_t ringIdx(int W, int s, int k) {
        const int We = RDBL ? ((W + 1) & ~1) : W;
        return s == 0 ? (size_t)k : (size_t)We + (size_t)k * (S - 1) + (size_t)(s - 1);
    }
    __device__ __forceinline__ double *ringAt(const CpkDiag &g) const { return ring + (size_t)g.ringOff * (RDBL ? 1 : S); }
    // CPK_COH_ST / CPK_COH_LD = 0: timing-only builds (tools/ab_build.sh) that drop one side's device scope -- results
    // of the one-launch form are then undefined
    __device__ __forceinline__ static void ringSt(double *p, double v) {
        if (COH && CPK_COH_ST) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    }
    // One cell's forward values -> the ring (rs states: 1 = the match row only, S = all).  next / prev: the lane above /
    // below this one holds cell k + 1 / k - 1 of the same diagonal and stores it in this call.
    // One-launch form (COH): a device-scope store is one fabric write per LANE whatever its size -- 8-byte ones cost 2.7x
    // the time per byte of 16-byte ones (MI355X_MICROARCH.md; measured here: config B 92.8 ms, 87.6 with plain stores,
    // profiles/r02_ab_coherent_stores.log).  So the lane of an even cell takes its right neighbour's match value over
    // DPP and writes both as one aligned 16 bytes, the neighbour writes nothing, and a cell's other states leave as
    // 16-byte writes too (S - 1 is even, their run starts on an even double).
    __device__ __forceinline__ void ringPut(double *out, int W, int k, bool next, bool prev, const double (&v)[S], int rs) const {
#ifdef CPK_TIMING_NO_RING_STORES  // timing experiment (tools/ab_build.sh): what the forward sweep costs without its stores
        if (a.geo.maxWidth >= 0) return;
#endif
        if (!(COH && CPK_COH_ST && CPK_COH_PAIRS)) {
            if (rs > 0) {
                ringSt(out + ringIdx(W, 0, k), v[0]);
                if (rs > 1) {
#pragma unroll
                    for (int s = 1; s < S; s++) ringSt(out + ringIdx(W, s, k), v[s]);
                }
            }
            return;
        }
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const unsigned lo = (unsigned)__double_as_longlong(v[0]), hi = (unsigned)((unsigned long long)__double_as_longlong(v[0]) >> 32);
        // wave_shl:1: lane i <- lane i + 1 (0 where that lane is off or does not exist: bound_ctrl)
        const unsigned loN = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xf, 0xf, true);
        const unsigned hiN = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x130, 0xf, 0xf, true);
        if (rs <= 0) return;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ring, 0, 0x7fffffff, 0x00020000);  // the region's ring: < 2 GiB (cpk_device_upload)
        const bool even = (k & 1) == 0;
        const unsigned at = (unsigned)((out - ring) + k) * 8u;
        if (even && next) {
            __builtin_amdgcn_raw_buffer_store_b128(u4{lo, hi, loN, hiN}, rsrc, (int)at, 0, 16);  // aux 16 = sc1
        } else if (even || !prev) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u2{lo, hi}, rsrc, (int)at, 0, 16);
        }
        if (rs > 1) {
            const unsigned at2 = (unsigned)((out - ring) + ringIdx(W, 1, k)) * 8u;
#pragma unroll
            for (int s = 1; s + 1 < S; s += 2) {
                const unsigned long long x = (unsigned long long)__double_as_longlong(v[s]), y = (unsigned long long)__double_as_longlong(v[s + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(u4{(unsigned)x, (unsigned)(x >> 32), (unsigned)y, (unsigned)(y >> 32)}, rsrc,
                                                       (int)(at2 + 8u * (unsigned)(s - 1)), 0, 16);
            }
        }
    }
    __device__ __forceinline__ static double ringLd(const double *p) {
        return (COH && CPK_COH_LD) ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ld_self(p);
    }

    // ---- forward: impl/pairwiseAligner.c:609-629 with stateMachine{5,3}_cellCalculate as the per-cell body ----
    struct FwdCtx {
        int d, xlo, dlR, w1R, dmR, w2R;  // neighbour shifts and widths premultiplied by R
        const double *p1, *p2;           // F[d-1], F[d-2] rows at position 1
    };

    // NC cells (NC = 1 or 2, 64 lanes apart on the same diagonal) computed together.  Fold order per state is the
    // reference's transition-list order; independent folds advance in lock-step (logadd_n).
    template <int NC>
    __device__ __forceinline__ void fwdCells(const FwdCtx &c, const int (&k)[NC], const int (&kR)[NC],
                                             double (&v)[NC][S]) const {
        int cX[NC], cY[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int x = c.xlo + k[q], y = c.d - x;
#ifdef CPK_TIMING_NO_SYMBOLS  // timing experiment (tools/ab_build.sh): symbols from the lane number instead of LDS -- results invalid
            cX[q] = (x + lane) & 3;
            cY[q] = (y + 2 * lane) & 3;
#else
            cX[q] = symX(x);
            cY[q] = symY(y);
#endif
        }
        fwdCellsSym<NC>(c, cX, cY, kR, v);
    }
    // the same with the cells' symbols given (the packed kernel fetches them itself)
    template <int NC>
    __device__ __forceinline__ void fwdCellsSym(const FwdCtx &c, const int (&cX)[NC], const int (&cY)[NC],
                                                const int (&kR)[NC], double (&v)[NC][S]) const {
        const double *p1 = c.p1, *p2 = c.p2;
        if (S == 5) {
            // states: 0 match, 1 shortGapX, 2 shortGapY, 3 longGapX, 4 longGapY (stateMachine.c:261-263)
            double acc[NC * 5], t[NC * 5], m2[NC], m3[NC], m4[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX[q] * 5 + cY[q]) * kWM, *wX = wt + 25 * kWM + cX[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY[q] * kWG;
                const int iL = sel(kR[q] + c.dlR, c.w1R);
                const int iU = sel(kR[q] + c.dlR + R, c.w1R);
                const int iM = sel(kR[q] + c.dmR, c.w2R);
                const double lM = lds1(p1 + 0 + iL), lSX = lds1(p1 + 1 + iL), lLX = lds1(p1 + 3 + iL);
                const double uM = lds1(p1 + 0 + iU), uSY = lds1(p1 + 2 + iU), uLY = lds1(p1 + 4 + iU);
                const double mM = lds1(p2 + 0 + iM), mSX = lds1(p2 + 1 + iM), mSY = lds1(p2 + 2 + iM),
                             mLX = lds1(p2 + 3 + iM), mLY = lds1(p2 + 4 + iM);
                // first two terms of every state's fold: lower block :454-462, middle :463-470, upper :471-479
                acc[q * 5 + 0] = mM + tab1(wM + 0);
                t[q * 5 + 0] = mSX + tab1(wM + 1);
                acc[q * 5 + 1] = lM + tab1(wX + 0);
                t[q * 5 + 1] = lSX + tab1(wX + 1);
                acc[q * 5 + 2] = uM + tab1(wY + 0);
                t[q * 5 + 2] = uSY + tab1(wY + 1);
                acc[q * 5 + 3] = lM + tab1(wX + 2);
                t[q * 5 + 3] = lLX + tab1(wX + 3);
                acc[q * 5 + 4] = uM + tab1(wY + 2);
                t[q * 5 + 4] = uLY + tab1(wY + 3);
                m2[q] = mSY + tab1(wM + 2);
                m3[q] = mLX + tab1(wM + 3);
                m4[q] = mLY + tab1(wM + 4);
            }
            logadd_n<NC * 5>(lg, acc, t);
            // the match state folds three more terms, in order
            double am[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) am[q] = acc[q * 5 + 0];
            logadd_n<NC>(lg, am, m2);
            logadd_n<NC>(lg, am, m3);
            logadd_n<NC>(lg, am, m4);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                v[q][0] = am[q];
#pragma unroll
                for (int s2 = 1; s2 < 5; s2++) v[q][s2] = acc[q * 5 + s2];
            }
        } else {
            // states: 0 match, 1 gapX, 2 gapY; stateMachine.c:695-713
            double acc[NC * 3], t[NC * 3], u[NC * 3];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX[q] * 5 + cY[q]) * kWM, *wX = wt + 25 * kWM + cX[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY[q] * kWG;
                const int iL = sel(kR[q] + c.dlR, c.w1R);
                const int iU = sel(kR[q] + c.dlR + R, c.w1R);
                const int iM = sel(kR[q] + c.dmR, c.w2R);
                const double lM = lds1(p1 + 0 + iL), lGX = lds1(p1 + 1 + iL), lGY = lds1(p1 + 2 + iL);
                const double uM = lds1(p1 + 0 + iU), uGX = lds1(p1 + 1 + iU), uGY = lds1(p1 + 2 + iU);
                const double mM = lds1(p2 + 0 + iM), mGX = lds1(p2 + 1 + iM), mGY = lds1(p2 + 2 + iM);
                acc[q * 3 + 0] = mM + tab1(wM + 0);
                t[q * 3 + 0] = mGX + tab1(wM + 1);
                u[q * 3 + 0] = mGY + tab1(wM + 2);
                acc[q * 3 + 1] = lM + tab1(wX + 0);
                t[q * 3 + 1] = lGX + tab1(wX + 1);
                u[q * 3 + 1] = lGY + tab1(wX + 2);
                acc[q * 3 + 2] = uM + tab1(wY + 0);
                t[q * 3 + 2] = uGY + tab1(wY + 1);
                u[q * 3 + 2] = uGX + tab1(wY + 2);
            }
            logadd_n<NC * 3>(lg, acc, t);
            logadd_n<NC * 3>(lg, acc, u);
#pragma unroll
            for (int q = 0; q < NC; q++)
#pragma unroll
                for (int s2 = 0; s2 < 3; s2++) v[q][s2] = acc[q * 3 + s2];
        }
    }

    // ringStates: how many states of F[d] go to the forward ring (0, 1 = match row only, S = all)
    __device__ void forward(int d, const CpkDiag &g, int ringStates) {
        const int W = g.width;
        FwdCtx c;
        c.d = d;
        c.xlo = (d + g.xmyL) >> 1;
        const int dl = (g.xmyL - 1 - f1.xmyL) >> 1;  // lower neighbour (d-1, xmy-1) is cell k+dl, upper is k+dl+1
        const int dm = (g.xmyL - f2.xmyL) >> 1;      // middle neighbour (d-2, xmy) is cell k+dm
        c.dlR = dl * R;
        c.w1R = f1.width * R;
        c.dmR = dm * R;
        c.w2R = d >= 2 ? f2.width * R : 0;
        c.p1 = fbuf1(d - 1);
        c.p2 = fbuf1(d - 2);
        double *cur = fbuf1(d);  // same rows as F[d-2]: updated in place
        double *out = ringAt(g);
        // A group of 64 cells reads F[d-2] at k+dm and writes F[d] at k.  With dm >= 0 ascending groups never read a
        // position an earlier group has overwritten; with dm < 0 descending groups never do (DESIGN.md "LDS layout").
        const int nPass = (W + CPK_WAVE - 1) / CPK_WAVE;
        const bool ascending = dm >= 0;
        for (int i = 0; i < nPass; i++) {
            const int kb = (ascending ? i : nPass - 1 - i) * CPK_WAVE;
            const int k0 = kb + lane;
            if (k0 < W) {
                const int kk[1] = {k0};
                const int kkR[1] = {kb * R + laneR};
                double v[1][S];
                fwdCells<1>(c, kk, kkR, v);
#pragma unroll
                for (int s = 0; s < S; s++) cur[s + kkR[0]] = v[0][s];
                ringPut(out, W, k0, k0 + 1 < W && lane < CPK_WAVE - 1, lane > 0, v[0], ringStates);
            }
        }
        roll_fence<!FAST>();
        f2 = f1;
        f1 = g;
    }

    // ---- forward sweep as a STREAM of cells (LDS variant).  Diagonals of 101-155 cells fill groups of 64 lanes to 77 %:
    // the last group of a diagonal is mostly empty.  Here a diagonal's leftover cells (fewer than 64) wait and share
    // a group with the first cells of the next diagonal: lanes [0, r) finish diagonal A, lanes [r, 64) start
    // diagonal B = A+1.  Legal when (1) both diagonals run in the same direction (in-place rule above), and (2) B's
    // cells in the shared group only read cells of A that earlier groups have written:
    //   ascending : B cells [0, b) read F[A] up to index b + dl_B        -> need b + dl_B < first leftover cell of A
    //   descending: B cells [W_B - b, W_B) read F[A] down to W_B - b + dl_B -> need that >= end of A's leftover range
    // Within the shared group every load precedes every store (one instruction stream, LDS in order), so A's
    // reads of F[A-1] and B's in-place writes over F[A-1] do not collide.  Otherwise the leftover is flushed as a
    // partly filled group, as before.  Per-lane parameters of the shared group are selects between A's and B's
    // wave-uniform ones.
    struct FwdTail {
        bool has;
        bool asc;
        int lo, n;  // leftover cells [lo, lo + n)
        int W, ringStates;
        FwdCtx c;
        double *cur, *out;
    };
    FwdTail tail{};
    // expectation emitter: backward values of the emitted cells of the segment being traced back, [cell][S], written by
    // traceback() and read by expectations() (set by the kernel; null for the other emitters)
    double *bring = nullptr;

    // one group of cells of ONE diagonal: cells [kb, kb + 64) clipped to [lo, hi)
    __device__ __forceinline__ void fwdGroupUniform(const FwdCtx &c, double *cur, double *out, int W, int ringStates, int kb,
                                                    int lo, int hi) {
        const int k0 = kb + lane;
        if (k0 >= lo && k0 < hi) {
            const int kk[1] = {k0};
            const int kkR[1] = {kb * R + laneR};
            double v[1][S];
            fwdCells<1>(c, kk, kkR, v);
#pragma unroll
            for (int s = 0; s < S; s++) cur[s + kkR[0]] = v[0][s];
            ringPut(out, W, k0, k0 + 1 < hi && lane < CPK_WAVE - 1, k0 > lo && lane > 0, v[0], ringStates);
        }
    }

    __device__ void flushTail() {
        if (!tail.has) return;
        fwdGroupUniform(tail.c, tail.cur, tail.out, tail.W, tail.ringStates, tail.lo, tail.lo, tail.lo + tail.n);
        tail.has = false;
    }

    __device__ void forwardStream(int d, const CpkDiag &g, int ringStates) {
        const int W = g.width;
        FwdCtx c;
        c.d = d;
        c.xlo = (d + g.xmyL) >> 1;
        const int dl = (g.xmyL - 1 - f1.xmyL) >> 1;
        const int dm = (g.xmyL - f2.xmyL) >> 1;
        c.dlR = dl * R;
        c.w1R = f1.width * R;
        c.dmR = dm * R;
        c.w2R = d >= 2 ? f2.width * R : 0;
        c.p1 = fbuf1(d - 1);
        c.p2 = fbuf1(d - 2);
        double *cur = fbuf1(d);
        double *out = ringAt(g);
        const bool asc = dm >= 0;
        int lo = 0, hi = W;  // cells of this diagonal still to do
        if (tail.has) {
            const int r = tail.n;
            const int b = CPK_WAVE - r < W ? CPK_WAVE - r : W;
            const int kB0 = asc ? 0 : W - b;  // first cell of B's share
            const bool reads_done = asc ? (b + dl < tail.lo) : (kB0 + dl >= tail.lo + tail.n);
            if (tail.asc == asc && reads_done) {
                const bool inA = lane < r;
                const bool inB = !inA && lane - r < b;
                // idle lanes (a narrow B) recompute B's first cell and store nothing
                const int k = inA ? tail.lo + lane : (inB ? kB0 + lane - r : kB0);
                const int kR = inA ? tail.lo * R + laneR : (inB ? (kB0 - r) * R + laneR : kB0 * R);
                FwdCtx m;
                m.d = inA ? tail.c.d : c.d;
                m.xlo = inA ? tail.c.xlo : c.xlo;
                m.dlR = inA ? tail.c.dlR : c.dlR;
                m.w1R = inA ? tail.c.w1R : c.w1R;
                m.dmR = inA ? tail.c.dmR : c.dmR;
                m.w2R = inA ? tail.c.w2R : c.w2R;
                // rows by parity of the diagonal: A writes over F[A-2] in `tail.cur` and reads F[A-1] from the other set,
                // which is the set B = A+1 writes into: two selects cover p1, p2 and cur
                double *curL = inA ? tail.cur : cur;
                m.p1 = inA ? cur : tail.cur;
                m.p2 = curL;
                double *outL = inA ? tail.out : out;
                const int WL = inA ? tail.W : W;
                const int rsL = inA ? tail.ringStates : ringStates;
                const int kk[1] = {k};
                const int kkR[1] = {kR};
                double v[1][S];
                fwdCells<1>(m, kk, kkR, v);
                if (inA || inB) {
#pragma unroll
                    for (int s = 0; s < S; s++) curL[s + kR] = v[0][s];
                    // the two diagonals' cells never pair: A's are lanes [0, r), B's [r, r + b)
                    ringPut(outL, WL, k, inA ? lane + 1 < r : lane + 1 - r < b, inA ? lane > 0 : lane > r, v[0], rsL);
                }
                tail.has = false;
                if (asc) lo = b;
                else hi = W - b;
            } else {
                flushTail();
            }
        }
        // whole groups of this diagonal, in its direction; what is left over waits for the next diagonal -- except that
        // a diagonal of fewer than 64 cells can never share (its first cell would have to be behind the reads of the
        // next diagonal) and is done now.  ONE call site for all of these: every inlined copy of the group is ~1.4 KB
        // of code, and the kernel's hot loops compete for the instruction cache of two CUs.
        const bool lone = W < CPK_WAVE;
        while (hi - lo >= CPK_WAVE || (lone && hi > lo)) {
            const int kb = (asc || lone) ? lo : hi - CPK_WAVE;
            fwdGroupUniform(c, cur, out, W, ringStates, kb, lo, hi);
            if (lone) lo = hi;
            else if (asc) lo += CPK_WAVE;
            else hi -= CPK_WAVE;
        }
        if (hi > lo) {
            tail.has = true;
            tail.asc = asc;
            tail.lo = lo;
            tail.n = hi - lo;
            tail.W = W;
            tail.ringStates = ringStates;
            tail.c = c;
            tail.cur = cur;
            tail.out = out;
        }
        f2 = f1;
        f1 = g;
    }

    // =====================================================================================================================
    // Absolute-position sweeps (ABS).  A cell of diagonal d with x-y = xmy lives at position p = (xmy - B) >> 1 of the row
    // set of d's parity (rows (d & 1) * S + state), B an even base that changes rarely; the table builder lays the
    // positions out per diagonal and per sweep direction (cpk_table_gather.inl, "positions": bit 15 of a diagonal's value
    // = the base moves in front of it).  What that buys:
    //  * the neighbours of a cell are at CONSTANT offsets -- the middle one (d -+ 2, same x-y) at the cell's own position,
    //    in the same row set, so a diagonal's values replace those of d -+ 2 in place, position by position, whatever
    //    the order of the groups; the two of d -+ 1 at p - 1 and p (d even) or p and p + 1 (d odd) of the other set --
    //    no per-diagonal shifts, no range tests: a position outside the band holds -inf unless a band cell wrote it, and
    //    the re-base below wipes what a rectangle of the band leaves behind before the next one can read it;
    //  * a group shared by the last cells of one diagonal and the first cells of the next differs per lane in two LDS
    //    offsets, the diagonal number and the first x: the traceback can stream cells across diagonals as the forward
    //    sweep does (lane fill 80 % -> 95 %) for a dozen instructions per shared group instead of ~120
    //    (profiles/r02_ab_traceback_stream.log).
    // =====================================================================================================================
    struct AbsDiag {       // one diagonal as the groups of its cells see it (wave-uniform; per lane in a shared group)
        int d;             // the diagonal
        int xlo;           // x of its first cell
        int ownR;          // (position of its first cell - 1) * R: element offset of cell 0 from a row pointer at position 1
        int W;
        double *cur;       // row set of d's parity, position 1: the diagonal's values go there, over those of d -+ 2
        const double *lu;  // the other row set, shifted by d's parity: the neighbour at x-y - 1 is at the cell's own offset,
                           // the one at x-y + 1 R elements further
    };
    int setStride = 0;  // doubles between the row set of the even and the odd diagonals (S * stride; set by the kernel)
    __device__ __forceinline__ AbsDiag absDiag(int d, const CpkDiag &g, int pLo) const {
        AbsDiag c;
        c.d = d;
        c.xlo = (d + g.xmyL) >> 1;
        c.ownR = (pLo - 1) * R;
        c.W = g.width;
        c.cur = roll + R + (d & 1) * setStride;
        c.lu = roll + R + ((d + 1) & 1) * setStride + ((d & 1) - 1) * R;
        return c;
    }
    // every position of every row: -inf (a region's forward sweep and a segment's traceback start from empty rows)
    __device__ void absWipe() {
        for (int i = lane; i < 2 * setStride; i += CPK_WAVE) roll[i] = NEG_INF;
        roll_fence<false>();
    }
    // Moves the W cells of diagonal dd (first cell at position pOld of its row set) by `delta` positions and sets every
    // other position of that row set to -inf.  W = 0: the diagonal does not exist, the set is wiped.
    __device__ void absMoveRows(int dd, int pOld, int W, int delta) {
        double *set = roll + (dd & 1) * setStride;  // position 0
        const int nG = (W + CPK_WAVE - 1) / CPK_WAVE;
        if (delta != 0) {
            for (int i = 0; i < nG; i++) {  // memmove order: towards higher positions from the top group down
                const int k = (delta > 0 ? nG - 1 - i : i) * CPK_WAVE + lane;
                const int src = (pOld + k) * R;
                double v[S];
                if (k < W) {
#pragma unroll
                    for (int st = 0; st < S; st++) v[st] = lds1(set + src + st);
                }
                roll_fence<false>();
                if (k < W) {
#pragma unroll
                    for (int st = 0; st < S; st++) set[src + delta * R + st] = v[st];
                }
                roll_fence<false>();
            }
        }
        const int pNew = pOld + delta;
        for (int p = lane; p < stride; p += CPK_WAVE) {
            if (p < pNew || p >= pNew + W) {
#pragma unroll
                for (int st = 0; st < S; st++) set[p * R + st] = NEG_INF;
            }
        }
        roll_fence<false>();
    }
    // The base moves in front of diagonal d (first cell at position pLo under the new base): the two live diagonals --
    // e1 (d - dir, first cell at p1 under the old base) and, where `have2`, e2 (d - 2 dir, at p2) -- move along, everything
    // else becomes -inf.  dir = +1: forward sweep, -1: traceback.  Returns the shift in positions.
    __device__ int absRebase(int d, const CpkDiag &g, int pLo, int dir, const CpkDiag &e1, int p1, bool have1, const CpkDiag &e2, int p2, bool have2) {
        int delta = 0;
        if (have1) {
            const int bNew = g.xmyL - 2 * pLo - (d & 1), bOld = e1.xmyL - 2 * p1 - ((d - dir) & 1);
            delta = (bOld - bNew) >> 1;  // both bases are even
        }
        absMoveRows(d - dir, p1, have1 ? e1.width : 0, delta);
        absMoveRows(d - 2 * dir, p2, have2 ? e2.width : 0, delta);
        return delta;
    }

    // ---- forward sweep as a stream of cells over absolute positions (cf. forwardStream) ----
    struct AbsTail {
        bool has;
        int k0, n;   // leftover cells [k0, k0 + n) of diagonal c.d
        int rs;
        AbsDiag c;
        double *out;
    };
    AbsTail atail{};
    int apos1 = 0, apos2 = 0;  // first-cell positions of the diagonals d - 1 / d - 2 (f1 / f2) under the base in force

    // cells [kb, kb + 64) of one diagonal, clipped to [kb, hi)
    __device__ __forceinline__ void absFwdGroup(const AbsDiag &c, double *out, int rs, int kb, int hi) {
        const int k0 = kb + lane;
#ifdef CPK_TIMING_LANES  // timing experiment: only the first CPK_TIMING_LANES lanes of every group work (results invalid)
        if (k0 < hi && lane < CPK_TIMING_LANES) {
#else
        if (k0 < hi) {
#endif
            const int kk[1] = {k0};
            const int kkR[1] = {c.ownR + kb * R + laneR};
            FwdCtx f;
            f.d = c.d;
            f.xlo = c.xlo;
            f.dlR = 0;
            f.w1R = 0;
            f.dmR = 0;
            f.w2R = 0;
            f.p1 = c.lu;
            f.p2 = c.cur;
            double v[1][S];
            fwdCells<1>(f, kk, kkR, v);
#pragma unroll
            for (int st = 0; st < S; st++) c.cur[st + kkR[0]] = v[0][st];
            ringPut(out, c.W, k0, k0 + 1 < hi && lane < CPK_WAVE - 1, lane > 0, v[0], rs);
        }
    }
    // cells [kb, kb + 128) of one diagonal, clipped to [kb, hi): two cells per lane, 64 apart, in lock-step
    __device__ __forceinline__ void absFwdGroup2(const AbsDiag &c, double *out, int rs, int kb, int hi) {
        const int k0 = kb + lane, k1 = k0 + CPK_WAVE;
        const bool on1 = k1 < hi;
        // lanes without a second cell recompute their first one (k0 < hi holds for every lane: hi - kb > 64)
        const int kk[2] = {k0, on1 ? k1 : k0};
        const int kkR[2] = {c.ownR + kb * R + laneR, c.ownR + (on1 ? kb + CPK_WAVE : kb) * R + laneR};
        FwdCtx f;
        f.d = c.d;
        f.xlo = c.xlo;
        f.dlR = 0;
        f.w1R = 0;
        f.dmR = 0;
        f.w2R = 0;
        f.p1 = c.lu;
        f.p2 = c.cur;
        double v[2][S];
        fwdCells<2>(f, kk, kkR, v);
#pragma unroll
        for (int st = 0; st < S; st++) c.cur[st + kkR[0]] = v[0][st];
        if (on1) {
#pragma unroll
            for (int st = 0; st < S; st++) c.cur[st + kkR[1]] = v[1][st];
        }
        ringPut(out, c.W, k0, lane < CPK_WAVE - 1, lane > 0, v[0], rs);
        if (on1) ringPut(out, c.W, k1, k1 + 1 < hi && lane < CPK_WAVE - 1, lane > 0, v[1], rs);
    }
    __device__ void absFlushTail() {
        if (!atail.has) return;
        absFwdGroup(atail.c, atail.out, atail.rs, atail.k0, atail.k0 + atail.n);
        atail.has = false;
    }
    // pos: the diagonal's entry of KArgs::dpos, forward half (position of its first cell | base-moves flag << 15)
    __device__ void forwardStreamAbs(int d, const CpkDiag &g, int pos, int ringStates) {
        const int W = g.width;
        const int pLo = pos & 0x7fff;
        if (pos & 0x8000) {  // rare: once per rectangle of the band
            absFlushTail();
            const CpkDiag e1 = dc.table[d >= 1 ? d - 1 : 0], e2 = dc.table[d >= 2 ? d - 2 : 0];  // (not kept in registers)
            const int delta = absRebase(d, g, pLo, 1, e1, apos1, d >= 1, e2, apos2, d >= 2);
            apos1 += delta;
            apos2 += delta;
        }
        const AbsDiag c = absDiag(d, g, pLo);
        double *out = ringAt(g);
        int lo = 0;
#if defined(CPK_ABS_FWD_FORM) && CPK_ABS_FWD_FORM > 0
        // timing experiments (tools/ab_build.sh): no streaming; FORM 2: pairs of groups in lock-step (two cells per lane).
        // Measured (profiles/r03_forward_what_bounds_it.txt, r03_ab_forward_pairs_under_subscribed.txt): the pairs are 6 %
        // slower with every wave slot busy and within 1.5 % of the stream on launches with fewer regions than slots (1250 /
        // 2500 config-B pairs, config A) -- two cells per lane buy nothing on this kernel.
        for (; CPK_ABS_FWD_FORM == 2 && W - lo > CPK_WAVE; lo += 2 * CPK_WAVE) absFwdGroup2(c, out, ringStates, lo, W);
#ifdef CPK_TIMING_GROUP_REPEAT  // timing experiment: every group computed CPK_TIMING_GROUP_REPEAT times (0: not at all)
        for (; lo < W; lo += CPK_WAVE)
            for (int rep = 0; rep < CPK_TIMING_GROUP_REPEAT; rep++) absFwdGroup(c, out, ringStates, lo, W);
#else
        for (; lo < W; lo += CPK_WAVE) absFwdGroup(c, out, ringStates, lo, W);
#endif
        apos2 = apos1;
        apos1 = pLo;
        return;
#endif
        if (atail.has) {
            // Lanes [0, r) finish diagonal A = d - 1, lanes [r, r + b) start this one.  Its cells [0, b) read F[A] up to
            // position pLo + b - 1 + (d & 1), which earlier groups must have written: below the first leftover cell of A.
            // (They replace F[d - 2] = F[A - 1] at positions [pLo, pLo + b), below everything A's leftover cells read, and
            // within the group every load precedes every store.)
            const int r = atail.n;
            const int b = CPK_WAVE - r < W ? CPK_WAVE - r : W;
            if (c.ownR + (b + (d & 1)) * R < atail.c.ownR + atail.k0 * R) {
                const bool inA = lane < r;
                const bool inB = !inA && lane - r < b;
                // idle lanes (a narrow diagonal) recompute this diagonal's first cell and store nothing
                const int k = inA ? atail.k0 + lane : (inB ? lane - r : 0);
                const int kR = inA ? atail.c.ownR + atail.k0 * R + laneR : (inB ? c.ownR - r * R + laneR : c.ownR);
                FwdCtx m;
                m.d = inA ? atail.c.d : c.d;
                m.xlo = inA ? atail.c.xlo : c.xlo;
                m.dlR = 0;
                m.w1R = 0;
                m.dmR = 0;
                m.w2R = 0;
                m.p1 = inA ? atail.c.lu : c.lu;
                double *curL = inA ? atail.c.cur : c.cur;
                m.p2 = curL;
                double *outL = inA ? atail.out : out;
                const int WL = inA ? atail.c.W : W;
                const int rsL = inA ? atail.rs : ringStates;
                const int kk[1] = {k};
                const int kkR[1] = {kR};
                double v[1][S];
                fwdCells<1>(m, kk, kkR, v);
                if (inA || inB) {
#pragma unroll
                    for (int st = 0; st < S; st++) curL[st + kR] = v[0][st];
                    ringPut(outL, WL, k, inA ? lane + 1 < r : lane + 1 - r < b, inA ? lane > 0 : lane > r, v[0], rsL);
                }
                atail.has = false;
                lo = b;
            } else {
                absFlushTail();
            }
        }
        // whole groups now, the leftover waits for the next diagonal (ONE call site for the uniform group: code size)
        const int nWhole = (W - lo) >> 6;
        for (int i = 0; i < nWhole; i++, lo += CPK_WAVE) absFwdGroup(c, out, ringStates, lo, W);
        if (W > lo) {
            atail.has = true;
            atail.k0 = lo;
            atail.n = W - lo;
            atail.rs = ringStates;
            atail.c = c;
            atail.out = out;
        }
        apos2 = apos1;
        apos1 = pLo;
    }

    // Puts diagonal d of the forward ring back into its rolling buffer (after a traceback used the buffers).
    __device__ void reloadForward(const CpkDiag &g, int d) {
        const int W = g.width;
        double *cur = fbuf1(d);
        const double *src = ringAt(g);
        for (int kb = 0; kb < W; kb += CPK_WAVE) {
            const int k = kb + lane;
            if (k < W) {
#pragma unroll
                for (int s = 0; s < S; s++) cur[s + kb * R + laneR] = ld_self(src + ringIdx(W, s, k));
            }
        }
        roll_fence<!FAST>();
    }

    struct BwdCtx {
        int d2, xlo, dbR, wBR, daR, wAR;  // source shifts and widths premultiplied by R
        const double *pb, *pa;            // B[d2+1] gap rows, B[d2+2] match row, at position 1
    };
    // B[d2][k] gathered from B[d2+1], B[d2+2] in the reference's scatter order, NC cells at a time
    template <int NC>
    __device__ __forceinline__ void bwdCells(const BwdCtx &c, const int (&k)[NC], const int (&kR)[NC],
                                             double (&v)[NC][S]) const {
        int cX1[NC], cY1[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int x = c.xlo + k[q], y = c.d2 - x;
            cX1[q] = symX(x + 1);  // symbols of the source cells (x+1,.) and (.,y+1)
            cY1[q] = symY(y + 1);
        }
        bwdCellsSym<NC>(c, cX1, cY1, kR, v);
    }
    template <int NC>
    __device__ __forceinline__ void bwdCellsSym(const BwdCtx &c, const int (&cX1)[NC], const int (&cY1)[NC],
                                                const int (&kR)[NC], double (&v)[NC][S]) const {
        const double *pb = c.pb, *pa = c.pa;
        if (S == 5) {
            double acc[NC * 5], t[NC * 5], m2[NC], m3[NC], m4[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX1[q] * 5 + cY1[q]) * kWM, *wX = wt + 25 * kWM + cX1[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY1[q] * kWG;
                const int iU = sel(kR[q] + c.dbR, c.wBR);      // cell (x, y+1): its "upper" neighbour is the target
                const int iL = sel(kR[q] + c.dbR + R, c.wBR);  // cell (x+1, y): its "lower" neighbour is the target
                const int iA = sel(kR[q] + c.daR, c.wAR);      // cell (x+1, y+1): its "middle" neighbour is the target
                const double aM = lds1(pa + iA);
                const double uSY = lds1(pb + 2 + iU), uLY = lds1(pb + 4 + iU);
                const double lSX = lds1(pb + 1 + iL), lLX = lds1(pb + 3 + iL);
                // per target state: (1) middle term from d2+2, (2) upper-block terms, (3) lower-block terms
                acc[q * 5 + 0] = aM + tab1(wM + 0);
                t[q * 5 + 0] = uSY + tab1(wY + 0);
                m2[q] = uLY + tab1(wY + 2);
                m3[q] = lSX + tab1(wX + 0);
                m4[q] = lLX + tab1(wX + 2);
                acc[q * 5 + 1] = aM + tab1(wM + 1);
                t[q * 5 + 1] = lSX + tab1(wX + 1);
                acc[q * 5 + 2] = aM + tab1(wM + 2);
                t[q * 5 + 2] = uSY + tab1(wY + 1);
                acc[q * 5 + 3] = aM + tab1(wM + 3);
                t[q * 5 + 3] = lLX + tab1(wX + 3);
                acc[q * 5 + 4] = aM + tab1(wM + 4);
                t[q * 5 + 4] = uLY + tab1(wY + 3);
            }
            logadd_n<NC * 5>(lg, acc, t);
            double am[NC];
#pragma unroll
            for (int q = 0; q < NC; q++) am[q] = acc[q * 5 + 0];
            logadd_n<NC>(lg, am, m2);
            logadd_n<NC>(lg, am, m3);
            logadd_n<NC>(lg, am, m4);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                v[q][0] = am[q];
#pragma unroll
                for (int s2 = 1; s2 < 5; s2++) v[q][s2] = acc[q * 5 + s2];
            }
        } else {
            double acc[NC * 3], t[NC * 3], u[NC * 3];
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double *wM = wt + (cX1[q] * 5 + cY1[q]) * kWM, *wX = wt + 25 * kWM + cX1[q] * kWG,
                             *wY = wt + 25 * kWM + 5 * kWG + cY1[q] * kWG;
                const int iU = sel(kR[q] + c.dbR, c.wBR);
                const int iL = sel(kR[q] + c.dbR + R, c.wBR);
                const int iA = sel(kR[q] + c.daR, c.wAR);
                const double aM = lds1(pa + iA);
                const double uGY = lds1(pb + 2 + iU);
                const double lGX = lds1(pb + 1 + iL);
                acc[q * 3 + 0] = aM + tab1(wM + 0);
                t[q * 3 + 0] = uGY + tab1(wY + 0);
                u[q * 3 + 0] = lGX + tab1(wX + 0);
                acc[q * 3 + 1] = aM + tab1(wM + 1);
                t[q * 3 + 1] = uGY + tab1(wY + 2);
                u[q * 3 + 1] = lGX + tab1(wX + 1);
                acc[q * 3 + 2] = aM + tab1(wM + 2);
                t[q * 3 + 2] = uGY + tab1(wY + 1);
                u[q * 3 + 2] = lGX + tab1(wX + 2);
            }
            logadd_n<NC * 3>(lg, acc, t);
            logadd_n<NC * 3>(lg, acc, u);
#pragma unroll
            for (int q = 0; q < NC; q++)
#pragma unroll
                for (int s2 = 0; s2 < 3; s2++) v[q][s2] = acc[q * 3 + s2];
        }
    }

    // ---- traceback of one segment (pairwiseAligner.c:796-862).
    // The reference scatters from diagonal d2+1 / d2+2 into d2 (:392-395, :631-634); this gathers the same terms in
    // the same order.  Per emitted diagonal it forms fb = F.s + B.s for the NL emitted
    // states (match; plus gapX, gapY for the indel emitter, :691-733) and keeps the cells that can still reach the
    // threshold once the total is known; on refresh diagonals it also writes the two per-cell series whose sequential
    // logAdd folds give the total probability (:636-653).
    // nCand[l] receives the number of candidates appended to list l (visit order: diagonal descending, x-y ascending).
    template <int NL, bool CANDS>
    __device__ void traceback(const CpkSegment &sg, const double *endPrior, double *dbgFb, int (&nCand)[NL]) {
        const int J = sg.nRefresh;
        const float logThr = (float)log(m.threshold);  // -inf for threshold 0: every cell is a candidate
        // Candidates are staged in LDS (a ring of kStage slots per list) and go to HBM 64 at a time as one coalesced
        // store.  A store per group would sit between the F prefetch below and its use: loads and stores share vmcnt
        // on gfx9, the compiler then waits with vmcnt(0) at every diagonal, i.e. for the write acknowledgement too.
        int pend[NL], head[NL];  // staged entries and ring position of the oldest, per list (wave-uniform)
#pragma unroll
        for (int l = 0; l < NL; l++) nCand[l] = pend[l] = head[l] = 0;
        auto flush = [&](int l, int n) {  // the n <= 64 oldest staged candidates of list l -> cand[l][nCand[l]..]
            if (lane < n) {
                cand[(size_t)l * a.geo.fbCells + nCand[l] + lane] = stage[l * kStage + ((head[l] + lane) & (kStage - 1))];
            }
            head[l] = (head[l] + n) & (kStage - 1);
            pend[l] -= n;
            nCand[l] += n;
        };
        // expectation emitter: cells of the segment are numbered from the first cell of its lowest emitted diagonal
        const int bBase = (!CANDS && bring) ? dc.table[sg.tbPrev + 1].cellOff : 0;
        float lastMax = -__builtin_huge_valf();
        double ep[S];  // end prior: loaded AND waited for here (the empty asm consumes the registers); a value whose
                       // load may still be pending at the loop head costs a vmcnt(0) in front of every group
#pragma unroll
        for (int s = 0; s < S; s++) ep[s] = endPrior[s];
#pragma unroll
        for (int s = 0; s < S; s++) asm volatile("" : "+v"(ep[s]));
        CpkDiag gb{}, ga{};  // table entries of d2+1 and d2+2
        CpkDiag g = dc.get(sg.dTop, true);
        CpkDiag gnext = sg.dTop >= 1 ? dc.get(sg.dTop - 1, true) : CpkDiag{};  // entry of d2-1
        // F rows of the emitted states (list l emits state l), prefetched one diagonal ahead of their use.  wantF: the
        // emitted diagonals plus the one above the first refresh point (its F.m + B.m feeds the straddle term).
        // The loads are unconditional (lanes past the end of the diagonal re-read its last cell, diagonals that are not
        // emitted are read all the same): a predicate per load costs more instructions than the load.
        double fmCur[NL][kPrefetch];
        auto loadRows = [&](const CpkDiag &gd, double (&dst)[NL][kPrefetch]) {
            const double *src = ringAt(gd);
#pragma unroll
            for (int l = 0; l < NL; l++)
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
                    dst[l][q] = ringLd(src + ringIdx(gd.width, l, k < gd.width ? k : gd.width - 1));
                }
        };
        loadRows(g, fmCur);
#pragma unroll
        for (int l = 0; l < NL; l++)
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) asm volatile("" : "+v"(fmCur[l][q]));  // complete before the loop
        // Refresh points (every 10th emitted diagonal, counted from tbFrom) as a countdown: no division per diagonal.
        int untilRefresh = sg.dTop - sg.tbFrom;  // diagonals until the next refresh point
        int jr = 0;                              // ... and its index
        for (int d2 = sg.dTop; d2 > sg.tbPrev;) {
          // Table entries of the 64 diagonals ending at d2-2: each diagonal of the sweep needs one new entry, that of d2-2.
          dc.load(d2 - 2 - (CPK_WAVE - 1));
          for (int ci = CPK_WAVE - 1; ci >= 0 && d2 > sg.tbPrev; ci--, d2--) {
            const bool seeded = d2 == sg.dTop;
            const int W = g.width;
            const bool emit = d2 <= sg.tbFrom;
            const bool refresh = untilRefresh == 0;
            // "Matches straddling diagonal r" (pairwiseAligner.c:643-651) is a middle-block forward step from F[r-1] into the
            // cells of r+1, times B[r+1].  The match state is reached through the middle block only, so that step IS
            // F[r+1].match (same terms, same order: stateMachine.c:463-470 / :703-707), and the series to fold is
            // F[r+1].m + B[r+1].m -- the fb values this loop forms anyway, one diagonal before the refresh point.
            const bool feeds = untilRefresh == 1 && d2 - 1 > sg.tbPrev;
            const int jrNext = jr;
            // issue the loads for diagonal d2-1 now: one diagonal of arithmetic covers the HBM round trip
            double fmNext[NL][kPrefetch];
            loadRows(gnext, fmNext);
            const CpkDiag gnext2 = dc.at(ci);  // entry of d2-2 (of diagonal 0 when d2 < 2: not used then)
            double *curM = bM1(d2), *curG = bG1(d2);
            const double *fsrc = ringAt(g);
            const int xlo = (d2 + g.xmyL) >> 1;
            BwdCtx c;
            c.d2 = d2;
            c.xlo = xlo;
            c.dbR = ((g.xmyL - 1 - gb.xmyL) >> 1) * R;  // source (d2+1, xmy-1) is cell k+db, (d2+1, xmy+1) is k+db+1
            c.wBR = seeded ? 0 : gb.width * R;
            c.daR = ((g.xmyL - ga.xmyL) >> 1) * R;      // source (d2+2, xmy) is cell k+da
            c.wAR = (!seeded && d2 + 2 <= sg.dTop) ? ga.width * R : 0;
            c.pb = bG1(d2 + 1);
            c.pa = bM1(d2 + 2);
            const float keepFrom = lastMax + logThr - kCandMargin;  // wave-uniform
            // Refresh diagonals read the remaining states of F[d2] (cell dot products).  Those loads are issued HERE,
            // before the compute loop of the diagonal, and consumed after it, so their HBM latency hides behind a few
            // thousand cycles of arithmetic.
            double rfC[S][kPrefetch];  // F[d2][s][k], s >= NL   (rows < NL are in fmCur)
            if (refresh) {
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
#pragma unroll
                    for (int s2 = NL; s2 < S; s2++) rfC[s2][q] = k < W ? ringLd(fsrc + ringIdx(W, s2, k)) : 0.0;
                }
            }
            // One group of 64 cells.  Wave-uniform (the candidate counts must stay identical in every lane): lanes past
            // the end of the diagonal recompute its last cell and have their stores masked.  f0 = F[d2][l][k0].
            const int lastR = (W - 1) * R;
            auto group = [&](int kb, const double (&f0)[NL]) {
                const int k0 = kb + lane, kR0 = kb * R + laneR;
                const bool on = k0 < W;
                double v[1][S];
                if (seeded) {
                    // every cell of the top diagonal gets the end-state prior (pairwiseAligner.c:798-799)
#pragma unroll
                    for (int s = 0; s < S; s++) v[0][s] = ep[s];
                } else {
                    const int kk[1] = {on ? k0 : W - 1};
                    const int kkR[1] = {on ? kR0 : lastR};
                    bwdCells<1>(c, kk, kkR, v);
                }
                if (on) {
                    curM[kR0] = v[0][0];
#pragma unroll
                    for (int s = 1; s < S; s++) curG[s + kR0] = v[0][s];
                }
                if (!CANDS && bring && emit && on) {  // kept for the expectation step: it needs B again, not its neighbours
                    // per group of gN cells state-major, [s][cell]: every store (and every load of the expectation step)
                    // is one contiguous run of the group's lanes
                    const int gN = W - kb < CPK_WAVE ? W - kb : CPK_WAVE;
                    double *bo = bring + (size_t)(g.cellOff - bBase + kb) * S + lane;
#pragma unroll
                    for (int s = 0; s < S; s++) bo[s * gN] = v[0][s];
                }
                if (feeds && on) mbuf[(size_t)k0 * J + jrNext] = f0[0] + v[0][0];  // every cell of the diagonal (:647)
                if (emit) {
                    const int x = xlo + k0, y = d2 - x;
                    double fbv[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) fbv[l] = f0[l] + v[0][l];
                    if (on && dbgFb) dbgFb[g.cellOff + k0] = fbv[0];
                    // candidate filter: a cell survives when it is within log(threshold) - margin of the bound on the
                    // total probability (DESIGN.md "candidate filter").  Match cells need x > 0 and y > 0, gapX cells
                    // x > 0, gapY cells y > 0 (pairwiseAligner.c:680, :719, :725).
#pragma unroll
                    for (int l = 0; l < (CANDS ? NL : 0); l++) {
                        const bool cell = l == 0 ? (x > 0 && y > 0) : (l == 1 ? x > 0 : y > 0);
                        const bool keep = on && cell && (float)fbv[l] >= keepFrom;
                        const unsigned long long mask = __ballot(keep);
                        if (keep) {
                            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            Candidate cd;
                            cd.fb = fbv[l];
                            cd.x = x;
                            cd.y = y;
                            stage[l * kStage + ((head[l] + pend[l] + rank) & (kStage - 1))] = cd;
                        }
                        pend[l] += __popcll(mask);
                        if (pend[l] >= CPK_WAVE) flush(l, CPK_WAVE);
                    }
                }
            };
            // the first kPrefetch groups take F from the prefetched registers (compile-time group index) ...
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) {
                if (q * CPK_WAVE < W) {
                    double f0[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) f0[l] = fmCur[l][q];
                    group(q * CPK_WAVE, f0);
                }
            }
            // ... wider diagonals load it on the spot
            for (int kb = kPrefetch * CPK_WAVE; kb < W; kb += CPK_WAVE) {
                double f0[NL];
#pragma unroll
                for (int l = 0; l < NL; l++)
                    f0[l] = ((emit || feeds) && kb + lane < W) ? ringLd(fsrc + ringIdx(W, l, kb + lane)) : 0.0;
                group(kb, f0);
            }
            roll_fence<!FAST>();
            if (refresh) {
                // (a) cell dot products over states (cell_dotProduct, pairwiseAligner.c:402-408) and, for the candidate
                //     bound, this diagonal's largest F.m + B.m: renewed every 10th diagonal as
                //     max(this diagonal's maximum, old bound - 1); the reference itself asserts that consecutive totals
                //     differ by less than 1.0 (:834), so the decayed bound stays below the current total.
                float diagMax = -__builtin_huge_valf();
                auto dotCell = [&](int k, int kR, const double (&fRow)[S]) {
                    double t = fRow[0] + curM[kR];
                    const int x = xlo + k, y = d2 - x;
                    const float fbf = (x > 0 && y > 0) ? (float)t : -__builtin_huge_valf();
#pragma unroll
                    for (int s2 = 1; s2 < S; s2++) t = logadd(lg, t, fRow[s2] + curG[s2 + kR]);
                    cbuf[(size_t)k * J + jr] = t;
                    return fbf;
                };
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
                    float fbf = -__builtin_huge_valf();
                    if (q * CPK_WAVE < W) {  // wave-uniform
                        if (k < W) {
                            double fRow[S];
#pragma unroll
                            for (int s2 = 0; s2 < S; s2++) fRow[s2] = s2 < NL ? fmCur[s2][q] : rfC[s2][q];
                            fbf = dotCell(k, q * CPK_WAVE * R + laneR, fRow);
                        }
                        if (CANDS) diagMax = fmaxf(diagMax, wave_max_f32(fbf));
                    }
                }
                for (int kb = kPrefetch * CPK_WAVE; kb < W; kb += CPK_WAVE) {  // diagonals wider than the prefetch
                    const int k = kb + lane;
                    float fbf = -__builtin_huge_valf();
                    if (k < W) {
                        double fRow[S];
#pragma unroll
                        for (int s2 = 0; s2 < S; s2++) fRow[s2] = ringLd(fsrc + ringIdx(W, s2, k));
                        fbf = dotCell(k, kb * R + laneR, fRow);
                    }
                    if (CANDS) diagMax = fmaxf(diagMax, wave_max_f32(fbf));
                }
                if (CANDS) lastMax = fmaxf(diagMax, lastMax - 1.0f);
            }
            // slide the window of table entries and prefetched F rows down one diagonal
            ga = gb;
            gb = g;
            g = gnext;
            gnext = gnext2;
            // The empty asm consumes the prefetched registers HERE, one whole diagonal after their loads were issued and
            // before the next prefetch goes out: left to itself hipcc waits at the first use inside the next diagonal,
            // behind the next prefetch, with vmcnt(0) -- the full HBM round trip exposed on every diagonal.
#pragma unroll
            for (int l = 0; l < NL; l++)
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    asm volatile("" : "+v"(fmNext[l][q]));
                    fmCur[l][q] = fmNext[l][q];
                }
            if (refresh) {
                untilRefresh = CPK_REFRESH_PERIOD - 1;
                jr++;
            } else {
                untilRefresh--;
            }
          }
        }
#pragma unroll
        for (int l = 0; l < (CANDS ? NL : 0); l++) flush(l, pend[l]);
    }

    // ---- traceback of one segment over absolute positions, as a STREAM of cells (cf. traceback above: same arithmetic,
    // same candidate filter, same series for the totals).  The last cells of a diagonal (fewer than 64) may wait and share
    // a group with the first cells of the next lower one: lanes [0, off) finish diagonal A, lanes [off, 64) start B = A - 1.
    // Legal when B's cells in that group only read cells of A that earlier groups wrote (their neighbours at x-y -+ 1 lie
    // below A's waiting cells); B[B] replaces B[A + 1] in place at B's first positions, below everything A's waiting cells
    // read, and within a group every load precedes every store.  Visit order is kept (A's cells before B's, each by
    // ascending x-y), so the candidate list is unchanged.
    // These sweeps are bound by the NUMBER of instructions a wave issues, so the loop is laid out
    // for few of them per diagonal: the top diagonal (end prior, no neighbours) is peeled off; what only some diagonals do
    // -- the straddle series of a diagonal above a refresh point, the dot products of a refresh point -- runs as a pass
    // of its own behind the diagonal's groups, and such diagonals finish all their cells first; a waiting tail and the
    // diagonal it joins are of the same kind (both emitted or both not), so a group has no per-lane flags; the prefetch
    // of the next diagonal's F rows is three loads off one per-lane address (lanes outside the row read neighbouring ring
    // words, which the ring's padding makes legal, and ignore them).
    template <int NL, bool CANDS>
    __device__ void tracebackAbs(const CpkSegment &sg, const double *endPrior, double *dbgFb, int (&nCand)[NL]) {
        const int J = sg.nRefresh;
        const float logThr = (float)log(m.threshold);
        // Candidates wait in LDS -- kStageAbs slots per list, filled from slot 0 -- and go to HBM as ONE coalesced store of
        // everything that waits (round 4: the ring of 128 slots flushed 64 at a time took 2 KB of LDS per wave, this 0.5).
        int pend[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) nCand[l] = pend[l] = 0;
        auto flush = [&](int l) {
            if (lane < pend[l]) cand[(size_t)l * a.geo.fbCells + nCand[l] + lane] = stage[l * kStageAbs + lane];
            nCand[l] += pend[l];
            pend[l] = 0;
        };
        float lastMax = -__builtin_huge_valf();
        double keepFrom = lastMax;  // compared in double: one instruction per group instead of a conversion and a compare
        absWipe();  // nothing above the top diagonal exists: its neighbours read -inf
        // F rows of the emitted states, lane <-> cell k = q * 64 - off + lane of group q; unclamped (see above)
        double fmCur[NL][kPrefetch];
        auto loadRows = [&](const CpkDiag &gd, int offd, double (&dst)[NL][kPrefetch]) {
            static_assert(NL == 1, "the unclamped prefetch addresses the match row (cell k at element k)");
#ifdef CPK_TIMING_HOT_ROWS  // timing experiment: every prefetch reads the same (cache-hot) words -- results invalid
            const double *src = ring + lane + (offd & 1);
#else
            const double *src = ringAt(gd) + (lane - offd);  // lane - offd may be negative: the ring is padded in front
#endif
#ifdef CPK_TIMING_NO_PREFETCH  // timing experiment: no F rows at all
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) dst[0][q] = (double)(src != nullptr);
#else
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) dst[0][q] = ringLd(src + q * CPK_WAVE);
#endif
        };
        // posterior candidates of one group of an emitted diagonal: fb = F + B of the NL emitted states
        auto emitCells = [&](int d, int x, bool on, const double (&f0)[NL], const double (&v)[S], int dbgAt) {
            const int y = d - x;
            double fbv[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) fbv[l] = f0[l] + v[l];
            if (dbgFb && on) dbgFb[dbgAt] = fbv[0];
#pragma unroll
            for (int l = 0; l < (CANDS ? NL : 0); l++) {
                const bool cell = l == 0 ? (x > 0 && y > 0) : (l == 1 ? x > 0 : y > 0);
                const bool keep = on && cell && fbv[l] >= keepFrom;
                const unsigned long long mask = __ballot(keep);
                const int cnt = __popcll(mask);
                // what waits goes out at the END of a diagonal (behind the wait for the prefetch, see below); here only
                // when this group's candidates do not fit behind it (the top diagonal, where every cell is one; a threshold
                // of zero)
                const bool direct = pend[l] + cnt > kStageAbs;  // wave-uniform
                if (direct) flush(l);
                if (keep) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    Candidate cd;
                    cd.fb = fbv[l];
                    cd.x = x;
                    cd.y = y;
                    if (direct) cand[(size_t)l * a.geo.fbCells + nCand[l] + rank] = cd;  // ... and go straight to HBM then
                    else stage[l * kStageAbs + pend[l] + rank] = cd;
                }
                if (direct) nCand[l] += cnt;
                else pend[l] += cnt;
            }
        };
        // a refresh point: per-cell dot products over the states (cell_dotProduct, pairwiseAligner.c:402-408) into cbuf and,
        // for the candidate bound, the diagonal's largest F.m + B.m (renewed as max(this, old - 1): :834 bounds the drift)
        // rfC: the remaining states of F[d] (rows NL..S-1), prefetched before the diagonal's groups; fm: its emitted states
        // Global stores of the two series wait in registers (the first kPrefetch groups of a diagonal) and go out at the END
        // of the diagonal, behind the wait for the prefetched F rows and in front of the next prefetch: that wait counts
        // every store issued since (loads and stores share vmcnt on gfx9 and complete out of order, hipcc waits for all),
        // and a write acknowledgement is the slowest thing there is to wait for.  Issued there, they have a whole diagonal.
        double pendM[kPrefetch], pendC[kPrefetch];
        int pendMW = 0, pendMOff = 0, pendMj = 0, pendCW = 0, pendCOff = 0, pendCj = 0;  // widths (0: nothing waits), first lanes, series indices
        auto issueStores = [&]() {
            if (pendMW > 0) {
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE - pendMOff + lane;
                    if ((unsigned)k < (unsigned)pendMW) mbuf[(size_t)k * J + pendMj] = pendM[q];
                }
                pendMW = 0;
            }
            if (pendCW > 0) {
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE - pendCOff + lane;
                    if ((unsigned)k < (unsigned)pendCW) cbuf[(size_t)k * J + pendCj] = pendC[q];
                }
                pendCW = 0;
            }
#pragma unroll
            for (int l = 0; l < (CANDS ? NL : 0); l++)
                if (pend[l] >= kStageAbs / 2) flush(l);
        };
        auto dotCell = [&](const AbsDiag &cx, int k, const double (&fRow)[S], double &t) {
            const int kR = cx.ownR + k * R;
            t = fRow[0] + cx.cur[kR];
            const int x = cx.xlo + k, y = cx.d - x;
            const float fbf = (x > 0 && y > 0) ? (float)t : -__builtin_huge_valf();
#pragma unroll
            for (int s2 = 1; s2 < S; s2++) t = logadd(lg, t, fRow[s2] + cx.cur[s2 + kR]);
            return fbf;
        };
        auto refreshDots = [&](const AbsDiag &cx, const CpkDiag &g, int off, int jr, const double (&fm)[NL][kPrefetch],
                               const double (&rfC)[S][kPrefetch]) {
            const int W = g.width;
            const double *fsrc = ringAt(g);
            float diagMax = -__builtin_huge_valf();
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) {
                const int k = q * CPK_WAVE - off + lane;
                float fbf = -__builtin_huge_valf();
                if (q * CPK_WAVE - off < W) {  // wave-uniform
                    if ((unsigned)k < (unsigned)W) {
                        double fRow[S];
#pragma unroll
                        for (int s2 = 0; s2 < S; s2++) fRow[s2] = s2 < NL ? fm[s2][q] : rfC[s2][q];
                        fbf = dotCell(cx, k, fRow, pendC[q]);
                    }
                    if (CANDS) diagMax = fmaxf(diagMax, wave_max_f32(fbf));
                }
            }
            for (int kb = kPrefetch * CPK_WAVE - off; kb < W; kb += CPK_WAVE) {  // diagonals wider than the prefetch
                const int k = kb + lane;
                float fbf = -__builtin_huge_valf();
                if (k < W) {
                    double fRow[S];
#pragma unroll
                    for (int s2 = 0; s2 < S; s2++) fRow[s2] = ringLd(fsrc + ringIdx(W, s2, k));
                    double tk;
                    fbf = dotCell(cx, k, fRow, tk);
                    cbuf[(size_t)k * J + jr] = tk;
                }
                if (CANDS) diagMax = fmaxf(diagMax, wave_max_f32(fbf));
            }
            pendCW = W;
            pendCOff = off;
            pendCj = jr;
            if (CANDS) {
                lastMax = fmaxf(diagMax, lastMax - 1.0f);
                keepFrom = (double)(lastMax + logThr - kCandMargin);
            }
        };
        auto loadRefreshRows = [&](const CpkDiag &g, int off, double (&rfC)[S][kPrefetch]) {
            const double *fsrc = ringAt(g);
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) {
                const int k = q * CPK_WAVE - off + lane;
#pragma unroll
                for (int s2 = NL; s2 < S; s2++) rfC[s2][q] = (unsigned)k < (unsigned)g.width ? ringLd(fsrc + ringIdx(g.width, s2, k)) : 0.0;
            }
        };

        // ---- the top diagonal: every cell gets the end-state prior (pairwiseAligner.c:798-799); nothing is read
        CpkDiag g = dc.get(sg.dTop, true);
        int gpos = dc.posGet(sg.dTop, true) >> 16;  // backward half of the diagonal's positions
        int untilRefresh = sg.dTop - sg.tbFrom;  // diagonals until the next refresh point (every 10th emitted one, from tbFrom)
        int jr = 0;                              // ... and its index
        {
            const int W = g.width;
            const AbsDiag cx = absDiag(sg.dTop, g, gpos & 0x7fff);
            const bool emit = sg.dTop <= sg.tbFrom;  // the last segment of a region: the top diagonal is tbFrom (and a refresh point)
            const double *fsrc = ringAt(g);
            double ep[S];
#pragma unroll
            for (int st = 0; st < S; st++) ep[st] = endPrior[st];
            const bool feedsTop = untilRefresh == 1 && sg.dTop - 1 > sg.tbPrev;  // (traceBackDiagonals == 0 only)
            for (int kb = 0; kb < W; kb += CPK_WAVE) {
                const int k0 = kb + lane;
                const bool on = k0 < W;
                if (on) {
#pragma unroll
                    for (int st = 0; st < S; st++) cx.cur[st + cx.ownR + k0 * R] = ep[st];
                }
                if (emit || feedsTop) {
                    double f0[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) f0[l] = ringLd(fsrc + ringIdx(W, l, on ? k0 : W - 1));
                    if (feedsTop && on) mbuf[(size_t)k0 * J + jr] = f0[0] + ep[0];
                    if (emit) emitCells(sg.dTop, cx.xlo + k0, on, f0, ep, g.cellOff + k0);  // keepFrom is still -inf: every cell
                }
            }
            roll_fence<false>();
            if (untilRefresh == 0) {
                double fm[NL][kPrefetch], rfC[S][kPrefetch];
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    const int k = q * CPK_WAVE + lane;
#pragma unroll
                    for (int l = 0; l < NL; l++) fm[l][q] = ringLd(fsrc + ringIdx(W, l, k < W ? k : W - 1));
                }
                loadRefreshRows(g, 0, rfC);
                refreshDots(cx, g, 0, jr, fm, rfC);
                issueStores();
                untilRefresh = CPK_REFRESH_PERIOD - 1;
                jr++;
            } else {
                untilRefresh--;
            }
        }

        // ---- the diagonals below it
        int posb = gpos & 0x7fff, posa = 0;  // positions of the first cells of d2 + 1 and d2 + 2 under the base in force
        g = sg.dTop >= 1 ? dc.get(sg.dTop - 1, true) : CpkDiag{};
        gpos = sg.dTop >= 1 ? dc.posGet(sg.dTop - 1, true) >> 16 : 0;
        CpkDiag gnext = sg.dTop >= 2 ? dc.get(sg.dTop - 2, true) : CpkDiag{};
        int gnpos = sg.dTop >= 2 ? dc.posGet(sg.dTop - 2, true) >> 16 : 0;
        int off = 0;         // lane of the current diagonal's cell 0 in its first group
        bool carry = false;  // lanes [0, off) of that group hold the last cells of the diagonal above:
        AbsDiag tl{};        //   its context,
        int tlK0 = 0, tlCellOff = 0;
        double fTail[NL];    //   and its F values
#pragma unroll
        for (int l = 0; l < NL; l++) fTail[l] = 0.0;
        loadRows(g, 0, fmCur);
#pragma unroll
        for (int l = 0; l < NL; l++)
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) asm volatile("" : "+v"(fmCur[l][q]));
        for (int d2 = sg.dTop - 1; d2 > sg.tbPrev;) {
          dc.load(d2 - 2 - (CPK_WAVE - 1));  // the chunk of table entries that ends at d2 - 2
          for (int ci = CPK_WAVE - 1; ci >= 0 && d2 > sg.tbPrev; ci--, d2--) {
            const int W = g.width;
            const bool emit = d2 <= sg.tbFrom;
            const bool refresh = untilRefresh == 0;
            // "Matches straddling diagonal r" (pairwiseAligner.c:643-651) is F[r+1].m + B[r+1].m of every cell: the
            // diagonal above a refresh point writes that series (see traceback above)
            const bool feeds = untilRefresh == 1 && d2 - 1 > sg.tbPrev;
            const int pLo = gpos & 0x7fff;
            if (gpos & 0x8000) {  // rare; never with cells of the diagonal above waiting (carryNext below)
                const bool have2 = d2 + 2 <= sg.dTop;
                const CpkDiag gb = dc.table[d2 + 1], ga = dc.table[have2 ? d2 + 2 : d2 + 1];  // (not kept in registers: once per rectangle)
                const int delta = absRebase(d2, g, pLo, -1, gb, posb, true, ga, posa, have2);
                posb += delta;
                posa += delta;
            }
            const AbsDiag cx = absDiag(d2, g, pLo);
            // groups of this diagonal: qT whole ones (the first may be shared with the diagonal above), then r cells
            const int qT = (off + W) >> 6, r = (off + W) & (CPK_WAVE - 1);
            bool carryNext = false;
            if (!refresh && !feeds && r > 0 && qT >= 1 && qT < kPrefetch && d2 - 1 > sg.tbPrev && (d2 - 1 <= sg.tbFrom) == emit &&
                !(gnpos & 0x8000)) {
                const int bN = CPK_WAVE - r < gnext.width ? CPK_WAVE - r : gnext.width;
                // the next diagonal's cells [0, bN) read this one up to position pLoN + bN - 1 + ((d2 - 1) & 1): below the
                // first waiting cell, which sits at position pLo + qT * 64 - off
                carryNext = (gnpos & 0x7fff) + bN + ((d2 - 1) & 1) - 1 < pLo + qT * CPK_WAVE - off;
            }
            const int offNext = carryNext ? r : 0;
            double fmNext[NL][kPrefetch];
            loadRows(gnext, offNext, fmNext);  // one diagonal of arithmetic covers the round trip
            const CpkDiag gnext2 = dc.at(ci);
            const int gnpos2 = dc.posAt(ci) >> 16;
            double rfC[S][kPrefetch];  // a refresh point reads every state of F[d2]: requested here, used behind the groups
#ifndef CPK_TIMING_NO_PASSES
            if (refresh) loadRefreshRows(g, off, rfC);
#endif
            // One group of 64 cells: lane by lane cell k0 of diagonal t (wave-uniform except in the shared group); lanes that
            // are not `on` compute a cell of their diagonal all the same and store nothing.
            auto group = [&](const AbsDiag &t, int k0, int kR0, bool on, const double (&f0)[NL], int tCellOff) {
#ifdef CPK_TIMING_TRACE_REPEAT  // timing experiment: every traceback group computed this many times (0: not at all)
              for (int rep = 0; rep < CPK_TIMING_TRACE_REPEAT; rep++) {
#endif
                BwdCtx c;
                c.d2 = t.d;
                c.xlo = t.xlo;
                c.dbR = 0;
                c.wBR = 0;
                c.daR = 0;
                c.wAR = 0;
                c.pb = t.lu;
                c.pa = t.cur;
                const int kk[1] = {k0};
                const int kkR[1] = {kR0};
                double v[1][S];
                bwdCells<1>(c, kk, kkR, v);
                if (on) {
#pragma unroll
                    for (int st = 0; st < S; st++) t.cur[st + kR0] = v[0][st];
                }
#ifdef CPK_TIMING_TRACE_NO_EMIT
                if (emit && a.geo.maxWidth < 0) emitCells(t.d, t.xlo + k0, on, f0, v[0], tCellOff + k0);
#else
                if (emit) emitCells(t.d, t.xlo + k0, on, f0, v[0], tCellOff + k0);
#endif
#ifdef CPK_TIMING_TRACE_REPEAT
              }
#endif
            };
            int q0 = 0;
            if (carry) {  // the shared group: lanes [0, off) finish the diagonal above, the others start this one
                const bool inA = lane < off;
                const int b = CPK_WAVE - off < W ? CPK_WAVE - off : W;
                const bool on = lane < off + b;
                AbsDiag mx;
                mx.d = inA ? tl.d : cx.d;
                mx.xlo = inA ? tl.xlo : cx.xlo;
                mx.ownR = 0;
                mx.W = 0;
                mx.cur = inA ? tl.cur : cx.cur;
                mx.lu = inA ? tl.lu : cx.lu;
                // lanes past the end of a narrow diagonal compute its first cell
                const int k0 = inA ? tlK0 + lane : (on ? lane - off : 0);
                const int kR0 = inA ? tl.ownR + tlK0 * R + laneR : (on ? cx.ownR - off * R + laneR : cx.ownR);
                double f0[NL];
#pragma unroll
                for (int l = 0; l < NL; l++) f0[l] = inA ? fTail[l] : fmCur[l][0];
                group(mx, k0, kR0, on, f0, inA ? tlCellOff : g.cellOff);
                q0 = 1;
            }
            // this diagonal's own groups; its last cells wait for the next diagonal when they may (carryNext)
            const int nNow = carryNext ? qT : qT + (r > 0 ? 1 : 0);
#pragma unroll
            for (int q = 0; q < kPrefetch; q++) {
                if (q >= q0 && q < nNow) {
                    double f0[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) f0[l] = fmCur[l][q];
                    const int kb = q * CPK_WAVE - off;
                    const int k0 = kb + lane;
                    const bool on = k0 < W;
                    group(cx, on ? k0 : W - 1, on ? cx.ownR + kb * R + laneR : cx.ownR + (W - 1) * R, on, f0, g.cellOff);
                }
            }
            for (int q = kPrefetch; q < nNow; q++) {  // diagonals wider than the prefetch load F on the spot
                const int kb = q * CPK_WAVE - off;
                const int k0 = kb + lane;
                const bool on = k0 < W;
                double f0[NL];
#pragma unroll
                for (int l = 0; l < NL; l++) f0[l] = ringLd(ringAt(g) + ringIdx(W, l, on ? k0 : W - 1));
                group(cx, on ? k0 : W - 1, on ? cx.ownR + kb * R + laneR : cx.ownR + (W - 1) * R, on, f0, g.cellOff);
            }
#ifdef CPK_TIMING_NO_PASSES  // timing experiment: neither the straddle series nor the dot products
            if ((feeds || refresh) && a.geo.maxWidth < 0) {
#else
            if (feeds || refresh) {  // one diagonal in five: all its cells are done (no tail waits), a pass of its own follows
#endif
                roll_fence<false>();
                if (feeds) {  // F.m of the cells from the prefetched registers, B.m from the rows
#pragma unroll
                    for (int q = 0; q < kPrefetch; q++) {
                        const int k = q * CPK_WAVE - off + lane;
                        pendM[q] = (unsigned)k < (unsigned)W ? fmCur[0][q] + cx.cur[cx.ownR + k * R] : 0.0;
                    }
                    pendMW = W;
                    pendMOff = off;
                    pendMj = jr;
                    const double *fsrc = ringAt(g);
                    for (int kb = kPrefetch * CPK_WAVE - off; kb < W; kb += CPK_WAVE) {
                        const int k = kb + lane;
                        if (k < W) mbuf[(size_t)k * J + jr] = ringLd(fsrc + ringIdx(W, 0, k)) + cx.cur[cx.ownR + k * R];
                    }
                }
                if (refresh) refreshDots(cx, g, off, jr, fmCur, rfC);
            }
            // the cells that wait: their context and their F values (group qT of this diagonal's prefetched rows)
            if (carryNext) {
                tl = cx;
                tlK0 = qT * CPK_WAVE - off;
                tlCellOff = g.cellOff;
#pragma unroll
                for (int l = 0; l < NL; l++) {
                    double f = fmCur[l][kPrefetch - 1];
#pragma unroll
                    for (int q = kPrefetch - 2; q >= 1; q--) f = qT == q ? fmCur[l][q] : f;
                    fTail[l] = f;
                }
            }
            carry = carryNext;
            off = offNext;
            // slide the window of table entries, positions and prefetched F rows down one diagonal
            posa = posb;
            posb = pLo;
            g = gnext;
            gpos = gnpos;
            gnext = gnext2;
            gnpos = gnpos2;
            // The empty asm consumes the prefetched registers HERE, one whole diagonal after their loads were issued and
            // before the next prefetch goes out (left to itself hipcc waits at the first use, behind the next prefetch).
#pragma unroll
            for (int l = 0; l < NL; l++)
#pragma unroll
                for (int q = 0; q < kPrefetch; q++) {
                    asm volatile("" : "+v"(fmNext[l][q]));
                    fmCur[l][q] = fmNext[l][q];
                }
            issueStores();
            if (refresh) {
                untilRefresh = CPK_REFRESH_PERIOD - 1;
                jr++;
            } else {
                untilRefresh--;
            }
          }
        }
#pragma unroll
        for (int l = 0; l < (CANDS ? NL : 0); l++) flush(l);
    }

    // ---- expectation step (diagonalCalculationExpectations, pairwiseAligner.c:735-746; updateExpectations :418-432).
    // Second backward sweep of the segment, run once the totals are known.  For every emitted diagonal d2, every cell
    // of B[d2] and every transition into it: p = exp(F_nbr[from] + B[to] + (eP + tP) - total) with the neighbours
    // taken from F[d2-1] / F[d2-2]; T[from][to] += p, and E[to][cX][cY] += p when neither symbol is N.
    // The reference has already freed F[d2-2] at the lowest diagonal of a segment (:843-845), so the middle block
    // contributes nothing there; reproduced.  Sums are linear-space fp64: order-insensitive at the 1e-5 gate.
    // tAcc: per-lane sums, one per transition in list order; eLds: [state*16 + cX*4 + cY] in LDS (fp64 LDS atomics).
    static constexpr int kNT = S == 5 ? 13 : 9;

    // One emitted diagonal as the expectation step sees it (wave-uniform).
    struct ExpDiag {
        int d2, W, xlo, dl, dm, w1, w2, w2Load, cellOff;
        const double *f1, *f2;
        int jt;  // refresh point whose total normalises the diagonal
    };
    // One item of the step: up to 64 CELLS, the last nA cells [kA, kA + nA) of diagonal A and the first nB cells of the
    // next lower diagonal B.  (Round 3: the step has no dependency between cells, so it streams them across diagonals;
    // rounds 1-2 gave every diagonal groups of its own -- BASELINE config 5's 36-cell diagonals filled them to 56 %.)
    struct ExpItem {
        ExpDiag A, B;
        int kA, nA, nB;
        bool valid;
    };
    // What the step keeps per lane for its cell: B of the cell, F[d2-1] at the lower / upper neighbour, F[d2-2] at the
    // middle one, the total of its diagonal, and the cell itself (coordinates, which neighbours exist, first cell of its
    // diagonal or not).
    struct ExpLoads {
        double v[S], fL[S], fU[S], fM[S];
        double total;
        // the cell's coordinates (both below 2^30) with four flags in their top bits -- x: 30 = lower neighbour in the band,
        // 31 = upper; y: 30 = middle, 31 = cell 0 of its diagonal.  A lane without a cell has no flag set.
        unsigned x, y;
    };

    // expStride / expPhase (the team kernel): this wave takes the items whose number is expPhase modulo expStride
    int expStride = 1, expPhase = 0;
    __device__ void expectations(const CpkSegment &sg, double (&tAcc)[kNT], double *eLds, double &likelihood) {
        const int bBase = dc.table[sg.tbPrev + 1].cellOff;
        // kDepth items are in flight: the loads of an item (16 values per cell) are issued kDepth - 1 items before its
        // events are computed.  The pass stores nothing to global memory, and every load is unconditional (a lane without
        // a cell, an item behind the last one: valid words that are never used), so the wait in front of an item's events
        // can count the younger loads and leave them in flight.
        auto diag_of = [&](int d2) {
            ExpDiag e;
            const int dd = d2 > sg.tbPrev ? d2 : sg.tbPrev + 1;  // behind the last diagonal: the last one again, never used
            const CpkDiag g = dc.get(dd, true);
            const CpkDiag g1 = dc.get(dd - 1, true);      // F[d2-1]: always alive (d2-1 >= tbPrev)
            const bool haveM2 = dd - 2 >= sg.tbPrev;      // F[d2-2] is gone at d2 == tbPrev+1 (:843-845)
            const CpkDiag g2 = haveM2 ? dc.get(dd - 2, true) : g1;
            e.d2 = dd;
            e.W = g.width;
            e.xlo = (dd + g.xmyL) >> 1;
            e.dl = (g.xmyL - 1 - g1.xmyL) >> 1;  // lower neighbour (d2-1, xmy-1) is cell k+dl of F[d2-1]
            e.dm = (g.xmyL - g2.xmyL) >> 1;      // middle neighbour (d2-2, xmy) is cell k+dm of F[d2-2]
            e.w1 = g1.width;
            e.w2 = haveM2 ? g2.width : 0;
            e.w2Load = g2.width;                 // the row the middle loads read (g1's when F[d2-2] is gone: unused then)
            e.cellOff = g.cellOff;
            e.f1 = ringAt(g1);
            e.f2 = ringAt(g2);
            e.jt = (sg.tbFrom - dd) / CPK_REFRESH_PERIOD;
            return e;
        };
        // the stream's cursor: the next cell to hand out is cell curK of diagonal cur.d2 (curValid: there is one)
        ExpDiag cur = diag_of(sg.tbFrom);
        int curK = 0;
        bool curValid = sg.tbFrom > sg.tbPrev;
        auto next_item = [&]() {
            ExpItem it;
            it.valid = curValid;
            it.A = cur;
            it.kA = curK;
            it.nA = curValid ? (cur.W - curK < CPK_WAVE ? cur.W - curK : CPK_WAVE) : 0;
            it.B = cur;
            it.nB = 0;
            if (!curValid) return it;
            if (curK + it.nA < cur.W) {  // diagonal A goes on in the next item
                curK += it.nA;
                return it;
            }
            // A is finished: the first cells of the diagonal below share the item
            if (cur.d2 - 1 > sg.tbPrev) {
                cur = diag_of(cur.d2 - 1);
                const int room = CPK_WAVE - it.nA;
                it.B = cur;
                it.nB = cur.W < room ? cur.W : room;
                curK = it.nB;
                if (curK >= cur.W) {  // B fits whole: the cursor moves on to the diagonal below it (a third diagonal never shares)
                    if (cur.d2 - 1 > sg.tbPrev) {
                        cur = diag_of(cur.d2 - 1);
                        curK = 0;
                    } else {
                        curValid = false;
                    }
                }
            } else {
                curValid = false;
            }
            return it;
        };
        int itemNo = 0;
        auto next_mine = [&]() {  // the next item that is this wave's (every item with one wave per region)
            for (;;) {
                const ExpItem it = next_item();
                if (!it.valid || expStride == 1 || itemNo++ % expStride == expPhase) return it;
            }
        };
        auto issue = [&](const ExpItem &it, ExpLoads &L) {
            const bool inA = lane < it.nA;
            const bool has = lane < it.nA + it.nB;
            // per lane: the diagonal of its cell (lanes without a cell read B's first cell and use nothing)
            const int W = inA ? it.A.W : it.B.W, d2 = inA ? it.A.d2 : it.B.d2, xlo = inA ? it.A.xlo : it.B.xlo;
            const int dl = inA ? it.A.dl : it.B.dl, dm = inA ? it.A.dm : it.B.dm;
            const int w1 = inA ? it.A.w1 : it.B.w1, w2 = inA ? it.A.w2 : it.B.w2, w2Load = inA ? it.A.w2Load : it.B.w2Load;
            const int cellOff = inA ? it.A.cellOff : it.B.cellOff, jt = inA ? it.A.jt : it.B.jt;
            const double *f1 = inA ? it.A.f1 : it.B.f1, *f2 = inA ? it.A.f2 : it.B.f2;
            int k = inA ? it.kA + lane : (has ? lane - it.nA : 0);
            k = k < W ? k : W - 1;
            const int kL = k + dl, kU = k + dl + 1, kM = k + dm;
            const bool okL = (unsigned)kL < (unsigned)w1, okU = (unsigned)kU < (unsigned)w1, okM = (unsigned)kM < (unsigned)w2;
            const int qL = okL ? kL : 0, qU = okU ? kU : 0;
            const int qM = (unsigned)kM < (unsigned)w2Load ? kM : 0;
            // the traceback kept B per group of gN cells, state-major (traceback: `bring`)
            const int kb = k & ~(CPK_WAVE - 1);
            const int gN = W - kb < CPK_WAVE ? W - kb : CPK_WAVE;
            const double *bo = bring + (size_t)(cellOff - bBase + kb) * S + (k - kb);
            L.total = ld_self(totals + jt);
            const int x = xlo + k;
            L.x = (unsigned)x | (has && okL ? 1u << 30 : 0u) | (has && okU ? 1u << 31 : 0u);
            L.y = (unsigned)(d2 - x) | (has && okM ? 1u << 30 : 0u) | (has && k == 0 ? 1u << 31 : 0u);
#pragma unroll
            for (int s = 0; s < S; s++) {
                // 5 states: the lower block reads M, sX, lX, the upper block M, sY, lY; 3 states: all three
                const bool needL = S == 3 || s == 0 || s == 1 || s == 3, needU = S == 3 || s == 0 || s == 2 || s == 4;
                L.v[s] = ld_self(bo + s * gN);
                L.fL[s] = needL ? ld_self(f1 + ringIdx(w1, s, qL)) : 0.0;
                L.fU[s] = needU ? ld_self(f1 + ringIdx(w1, s, qU)) : 0.0;
                L.fM[s] = ld_self(f2 + ringIdx(w2Load, s, qM));
            }
        };
        auto events = [&](const ExpLoads &Lc) {
            if (Lc.y >> 31) likelihood += Lc.total;  // once per diagonal (:743), by the lane of its first cell: summed over lanes at the end
            const bool okL = (Lc.x >> 30) & 1u, okU = Lc.x >> 31, okM = (Lc.y >> 30) & 1u;
            if (okL || okU || okM) {  // (a cell none of whose neighbours is in the band has no event)
                double fL[S], fU[S], fM[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    fL[s] = okL ? Lc.fL[s] : NEG_INF;
                    fU[s] = okU ? Lc.fU[s] : NEG_INF;
                    fM[s] = okM ? Lc.fM[s] : NEG_INF;
                }
                const int cX = symX((int)(Lc.x & 0x3fffffffu)), cY = symY((int)(Lc.y & 0x3fffffffu));
                // (emission + transition) sums of the events, from the same LDS table as the sweeps (Sweep::wt)
                const double *wM = wt + (cX * 5 + cY) * kWM, *wX = wt + 25 * kWM + cX * kWG, *wY = wt + 25 * kWM + 5 * kWG + cY * kWG;
                const bool acgt = cX < CPK_SYM_N && cY < CPK_SYM_N;
                const int eIdx = cX * 4 + cY;
                const double total = Lc.total;
                double eAcc[S];  // this cell's events summed per target state: one LDS atomic per state, not per event
#pragma unroll
                for (int s = 0; s < S; s++) eAcc[s] = 0.0;
                // one (transition, emission) event: impl/pairwiseAligner.c:426-431
                auto event = [&](int ti, double from, int to, double w) {
                    const double p = exp_1e7(from + Lc.v[to] + w - total);
                    tAcc[ti] += p;
                    eAcc[to] += p;
                };
                if (S == 5) {
                    event(0, fL[0], 1, wX[0]);   // M -> shortX (open)
                    event(1, fL[1], 1, wX[1]);   // shortX -> shortX
                    event(2, fL[0], 3, wX[2]);   // M -> longX (open)
                    event(3, fL[3], 3, wX[3]);   // longX -> longX
                    event(4, fM[0], 0, wM[0]);   // M -> M
                    event(5, fM[1], 0, wM[1]);   // shortX -> M
                    event(6, fM[2], 0, wM[2]);   // shortY -> M
                    event(7, fM[3], 0, wM[3]);   // longX -> M
                    event(8, fM[4], 0, wM[4]);   // longY -> M
                    event(9, fU[0], 2, wY[0]);   // M -> shortY
                    event(10, fU[2], 2, wY[1]);  // shortY -> shortY
                    event(11, fU[0], 4, wY[2]);  // M -> longY
                    event(12, fU[4], 4, wY[3]);  // longY -> longY
                } else {
                    event(0, fL[0], 1, wX[0]);  // M -> gapX
                    event(1, fL[1], 1, wX[1]);  // gapX -> gapX
                    event(2, fL[2], 1, wX[2]);  // gapY -> gapX (switch)
                    event(3, fM[0], 0, wM[0]);
                    event(4, fM[1], 0, wM[1]);
                    event(5, fM[2], 0, wM[2]);
                    event(6, fU[0], 2, wY[0]);  // M -> gapY
                    event(7, fU[2], 2, wY[1]);  // gapY -> gapY
                    event(8, fU[1], 2, wY[2]);  // gapX -> gapY (switch)
                }
                if (acgt) {  // emissions are counted for ACGT x ACGT cells only (:429)
                    double *copy = eLds + (lane & (kExpectCopies - 1)) * 80;
#pragma unroll
                    for (int s = 0; s < S; s++) atomicAdd(&copy[s * 16 + eIdx], eAcc[s]);
                }
            }
        };
#ifndef CPK_EXP_DEPTH
#define CPK_EXP_DEPTH 4
#endif
        constexpr int kDepth = CPK_EXP_DEPTH;
        bool valid[kDepth];
        ExpLoads L[kDepth];
#pragma unroll
        for (int j = 0; j < kDepth; j++) {
            const ExpItem it = next_mine();
            valid[j] = it.valid;
            issue(it, L[j]);
        }
        for (bool more = valid[0]; more;) {
#pragma unroll
            for (int j = 0; j < kDepth; j++) {
                if (!valid[j]) {  // wave-uniform: the items behind it are invalid as well
                    more = false;
                    break;
                }
#pragma unroll
                for (int s = 0; s < S; s++) {  // the wait for this item's loads: kDepth - 1 younger items stay in flight
                    asm volatile("" : "+v"(L[j].v[s]), "+v"(L[j].fL[s]), "+v"(L[j].fU[s]), "+v"(L[j].fM[s]));
                }
                asm volatile("" : "+v"(L[j].total));
                events(L[j]);
                const ExpItem it = next_mine();  // the item behind the youngest one in flight
                valid[j] = it.valid;
                issue(it, L[j]);
            }
        }
    }

    // ---- expectation step inside the traceback (classes whose diagonals fit one 64-lane group).
    // expectations() above is a second pass: it reads B of every emitted cell back from `bring` (40 bytes a cell written,
    // 40 read) and gathers 11 forward values per cell from the ring (88 bytes).  Here the events of a diagonal are formed
    // right behind its backward cells: B is still in registers and the forward rows F[d2], F[d2-1], F[d2-2] sit in LDS,
    // each diagonal of the ring read ONCE (coalesced, a diagonal ahead).  What is not known yet is the normaliser: the total
    // of the diagonal's refresh window comes out of foldTotals() once the segment is done (its sequential logAdd folds
    // are the reference's order, :513-523 / :649, and the cubic logAdd makes the order matter at the 1e-4 level).  So the
    // events of window j are summed against a provisional reference ref_j (an fp32 log-sum-exp of the cell dot products of the
    // window's refresh diagonal, within ~1e-3 of the total), the window's sums go to global memory (96 doubles per ten
    // diagonals), and scaleWindows() multiplies them by exp(ref_j - total_j) afterwards: exp(x - total) =
    // exp(x - ref) * exp(ref - total).  The events are exp2f-accurate as in expectations(), their sums fp64.
    double *frow = nullptr;  // LDS: F of three diagonals, slot-major, position-major inside [position][S], position 0 = -inf guard
    double *eWin = nullptr;  // LDS: emission sums of the current window, kWinCopies copies of [state*16 + cX*4 + cY]
    static constexpr int kWinDoubles = 96;  // a window's record: kNT transition sums | [13] ref | [14] diagonals | [16, 96) emission sums

    static constexpr int kWinCopies = kExpectWinCopies;  // LDS copies of the window's emission sums (lane & 1)
    static constexpr int kExpGroups = XG;  // 64-lane groups per diagonal tracebackExpect() is unrolled for (class: <= 64 / <= 128 cells)
    __device__ void tracebackExpect(const CpkSegment &sg, const double *endPrior, double *dbgFb) {
        const int J = sg.nRefresh;
        const int FS = (a.geo.maxWidth + 1) * S;  // doubles per F slot
        double *wsum = bring;                     // the window records take the place of the B values
        double ep[S];
#pragma unroll
        for (int s = 0; s < S; s++) ep[s] = endPrior[s];
#pragma unroll
        for (int s = 0; s < S; s++) asm volatile("" : "+v"(ep[s]));
        // cell 0 of F[d2], F[d2-1], F[d2-2] (rotated every diagonal)
        double *pF0 = frow + S, *pF1 = frow + FS + S, *pF2 = frow + 2 * FS + S;
        // a diagonal of the ring -> registers (lanes past its end re-read its last cell) -> an F slot
        auto loadF = [&](const CpkDiag &gd, double (&f)[kExpGroups][S]) {
            const double *src = ringAt(gd);
#pragma unroll
            for (int q = 0; q < kExpGroups; q++) {
                if (q > 0 && q * CPK_WAVE >= gd.width) break;  // wave-uniform
                const int k = q * CPK_WAVE + lane < gd.width ? q * CPK_WAVE + lane : gd.width - 1;
#pragma unroll
                for (int s = 0; s < S; s++) f[q][s] = ringLd(src + ringIdx(gd.width, s, k));
            }
        };
        auto storeF = [&](double *cell0, int Wd, const double (&f)[kExpGroups][S]) {
#pragma unroll
            for (int q = 0; q < kExpGroups; q++) {
                const int k = q * CPK_WAVE + lane;
                if (k < Wd) {
#pragma unroll
                    for (int s = 0; s < S; s++) cell0[k * S + s] = f[q][s];
                }
            }
        };
        CpkDiag gb{}, ga{};  // table entries of d2+1 and d2+2
        CpkDiag g = dc.get(sg.dTop, true);
        CpkDiag g1 = dc.get(sg.dTop >= 1 ? sg.dTop - 1 : 0, true);
        CpkDiag g2 = dc.get(sg.dTop >= 2 ? sg.dTop - 2 : 0, true);
        double fN[kExpGroups][S];
#pragma unroll
        for (int q = 0; q < kExpGroups; q++)
#pragma unroll
            for (int s = 0; s < S; s++) fN[q][s] = 0.0;
        loadF(g, fN);
        storeF(pF0, g.width, fN);
        loadF(g1, fN);
        storeF(pF1, g1.width, fN);
        loadF(g2, fN);  // F[dTop-2]: stored at the top of the first diagonal
        // This lane's transition sums of the current window -- at most ten diagonals, one cell each -- in fp32, as the events
        // themselves are (exp_1e7f: v_exp_f32); the window's sums over the lanes are fp32 too (wave_sum_f32), the windows of a
        // segment and everything above them fp64.  (Round 4: fp64 sums were 13 conversions per cell and, per window, 13 x 6
        // rounds of two ds_bpermute each.)
        float tW[kNT];
#pragma unroll
        for (int i = 0; i < kNT; i++) tW[i] = 0.0f;
        double ref = 0.0;  // the window's provisional normaliser (set on its refresh diagonal, before its first event)
        int jw = 0, nWin = 0;  // the window's index (= its refresh point's) and its diagonals so far
        // Global stores wait a diagonal in registers and go out at the top of the next one, in front of its F request: the
        // wait for a request counts the stores issued after it as well (loads and stores share vmcnt on gfx9 and complete
        // out of order, so hipcc waits for all of them), and a write acknowledgement is the slowest thing there is to
        // wait for.  Issued first, they have the whole diagonal.
        double pendM[kExpGroups], pendC[kExpGroups], pendRec[3];
        int pendMW = 0, pendMj = 0, pendCW = 0, pendCj = 0, pendRj = -1;  // widths (0: nothing pending), series indices; window (-1: none)
        auto issueStores = [&]() {
            if (pendMW > 0) {
#pragma unroll
                for (int q = 0; q < kExpGroups; q++)
                    if (q * CPK_WAVE + lane < pendMW) mbuf[(size_t)(q * CPK_WAVE + lane) * J + pendMj] = pendM[q];
                pendMW = 0;
            }
            if (pendCW > 0) {
#pragma unroll
                for (int q = 0; q < kExpGroups; q++)
                    if (q * CPK_WAVE + lane < pendCW) cbuf[(size_t)(q * CPK_WAVE + lane) * J + pendCj] = pendC[q];
                pendCW = 0;
            }
            if (pendRj >= 0) {
                double *rec = wsum + (size_t)pendRj * kWinDoubles;
                if (lane < 16) rec[lane] = pendRec[0];
                rec[16 + lane] = pendRec[1];
                if (lane < 16) rec[80 + lane] = pendRec[2];
                pendRj = -1;
            }
        };
        int untilRefresh = sg.dTop - sg.tbFrom;
        int jr = 0;
        // B.match rows of d2, d2+1, d2+2 (bM1: a ring of three; rotated every diagonal instead of two divisions by 3)
        double *pM0 = bM1(sg.dTop), *pM1 = bM1(sg.dTop + 1), *pM2 = bM1(sg.dTop + 2);
        for (int d2 = sg.dTop; d2 > sg.tbPrev;) {
          dc.load(d2 - 3 - (CPK_WAVE - 1));  // table entries of the 64 diagonals ending at d2-3: one new entry per diagonal
          for (int ci = CPK_WAVE - 1; ci >= 0 && d2 > sg.tbPrev; ci--, d2--) {
            const bool seeded = d2 == sg.dTop;
            const int W = g.width;
            const bool emit = d2 <= sg.tbFrom;
            const bool refresh = untilRefresh == 0;
            const bool feeds = untilRefresh == 1 && d2 - 1 > sg.tbPrev;  // see traceback(): the straddle series of the refresh point below
            // F[d2-2] arrives (its loads went out a diagonal ago) and takes the slot F[d2+1] has left; F[d2-3] is requested
#pragma unroll
            for (int q = 0; q < kExpGroups; q++)
#pragma unroll
                for (int s = 0; s < S; s++) asm volatile("" : "+v"(fN[q][s]));
            storeF(pF2, g2.width, fN);
            issueStores();
            const CpkDiag g3 = dc.at(ci);  // entry of d2-3 (of diagonal 0 below it: never used then)
            loadF(g3, fN);
            double *curM = pM0, *curG = bG1(d2);
            const int xlo = (d2 + g.xmyL) >> 1;
            BwdCtx c;
            c.d2 = d2;
            c.xlo = xlo;
            c.dbR = ((g.xmyL - 1 - gb.xmyL) >> 1) * R;
            c.wBR = gb.width * R;
            c.daR = ((g.xmyL - ga.xmyL) >> 1) * R;
            c.wAR = d2 + 2 <= sg.dTop ? ga.width * R : 0;
            c.pb = bG1(d2 + 1);
            c.pa = pM2;
            // backward cells of the diagonal, a group of 64 at a time; v stays in registers for the events
            double v[kExpGroups][S];
#pragma unroll
            for (int q = 0; q < kExpGroups; q++) {
                if (q * CPK_WAVE < W) {  // wave-uniform
                    const int k0 = q * CPK_WAVE + lane;
                    const bool on = k0 < W;
                    if (seeded) {
#pragma unroll
                        for (int s = 0; s < S; s++) v[q][s] = ep[s];  // (pairwiseAligner.c:798-799)
                    } else {
                        const int kk[1] = {on ? k0 : W - 1};
                        const int kkR[1] = {on ? k0 * R : (W - 1) * R};
                        double vv[1][S];
                        bwdCells<1>(c, kk, kkR, vv);
#pragma unroll
                        for (int s = 0; s < S; s++) v[q][s] = vv[0][s];
                    }
                    if (on) {
                        curM[k0 * R] = v[q][0];
#pragma unroll
                        for (int s = 1; s < S; s++) curG[s + k0 * R] = v[q][s];
                    }
                }
            }
            roll_fence<false>();  // the B rows for the next diagonal, the F slot for this one
            if (feeds || refresh || (emit && dbgFb)) {
                float tMax = -__builtin_huge_valf();
#pragma unroll
                for (int q = 0; q < kExpGroups; q++) {
                    if (q * CPK_WAVE < W) {
                        const int k0 = q * CPK_WAVE + lane;
                        const bool on = k0 < W;
                        const double *myF = pF0 + (on ? k0 : W - 1) * S;
                        const double fb0 = lds1(myF) + v[q][0];
                        if (feeds) pendM[q] = fb0;
                        if (emit && on && dbgFb) dbgFb[g.cellOff + k0] = fb0;
                        if (refresh) {
                            // cell dot products over states (cell_dotProduct, pairwiseAligner.c:402-408): the series foldTotals() folds
                            double t = fb0;
#pragma unroll
                            for (int s = 1; s < S; s++) t = logadd(lg, t, lds1(myF + s) + v[q][s]);
                            pendC[q] = t;
                            if (on) tMax = fmaxf(tMax, (float)t);
                        }
                    }
                }
                if (feeds) {
                    pendMW = W;
                    pendMj = jr;
                }
                if (refresh) {
                    pendCW = W;
                    pendCj = jr;
                    // ref = log of the sum of the cells' exp(dot product) in fp32, within ~1e-3 of the total that the sequential fp64
                    // folds will give: the events that carry the sums then have an exp2f argument near 0, as in expectations()
                    const float tm = wave_max_f32(tMax);
                    float es = 0.0f;
#pragma unroll
                    for (int q = 0; q < kExpGroups; q++)
                        if (q * CPK_WAVE + lane < W) es += __builtin_amdgcn_exp2f((float)(pendC[q] - (double)tm) * 1.44269504f);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) es += __shfl_xor(es, off);
                    ref = (double)tm + (double)(__builtin_amdgcn_logf(es) * 0.693147181f);
                    if (!(ref > -1e300 && ref < 1e300)) ref = 0.0;  // a diagonal without any probability: every event is exp(-inf) whatever the reference
                    jw = jr;
                }
            }
            if (emit) {
                // the events into the cells of d2 (updateExpectations, :418-432), against ref instead of the total
                const int dl = (g.xmyL - 1 - g1.xmyL) >> 1, dm = (g.xmyL - g2.xmyL) >> 1;
                const int w1 = g1.width, w2 = d2 - 2 >= sg.tbPrev ? g2.width : 0;  // F[d2-2] is gone at d2 == tbPrev+1 (:843-845)
#pragma unroll
                for (int q = 0; q < kExpGroups; q++) {
                    if (q * CPK_WAVE >= W) break;  // wave-uniform
                    const int k0 = q * CPK_WAVE + lane;
                    const bool on = k0 < W;
                    const int kL = k0 + dl, kM = k0 + dm;
                    const int iL = (unsigned)kL < (unsigned)w1 ? kL * S : -S, iU = (unsigned)(kL + 1) < (unsigned)w1 ? (kL + 1) * S : -S;
                    const int iM = (unsigned)kM < (unsigned)w2 ? kM * S : -S;
                    double fL[S], fU[S], fM[S];
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const bool needL = S == 3 || s == 0 || s == 1 || s == 3, needU = S == 3 || s == 0 || s == 2 || s == 4;
                        fL[s] = needL ? lds1(pF1 + iL + s) : 0.0;
                        fU[s] = needU ? lds1(pF1 + iU + s) : 0.0;
                        fM[s] = lds1(pF2 + iM + s);
                    }
                    const int x = xlo + (on ? k0 : W - 1), y = d2 - x;
                    const int cX = symX(x), cY = symY(y);
                    const double *wM = wt + (cX * 5 + cY) * kWM, *wX = wt + 25 * kWM + cX * kWG, *wY = wt + 25 * kWM + 5 * kWG + cY * kWG;
                    double vr[S];  // B - ref; a lane without a cell has no event (its reference is +inf: every exponent -inf)
                    const double refL = on ? ref : __builtin_huge_val();
#pragma unroll
                    for (int s = 0; s < S; s++) vr[s] = v[q][s] - refL;
                    float eAcc[S];  // this cell's events summed per target state: one LDS atomic per state, not per event
#pragma unroll
                    for (int s = 0; s < S; s++) eAcc[s] = 0.0f;
                    auto event = [&](int ti, double from, int to, double w) {
                        const float p = exp_1e7f(from + w + vr[to]);
                        tW[ti] += p;
                        eAcc[to] += p;
                    };
                    // the (emission + transition) sums of the events, fetched together: one LDS round trip, not one per event
                    double w[kNT];
#pragma unroll
                    for (int i = 0; i < kWG; i++) w[i] = tab1(wX + i);
#pragma unroll
                    for (int i = 0; i < kWM; i++) w[kWG + i] = tab1(wM + i);
#pragma unroll
                    for (int i = 0; i < kWG; i++) w[kWG + kWM + i] = tab1(wY + i);
                    if (S == 5) {  // the list order of expectations()
                        event(0, fL[0], 1, w[0]);
                        event(1, fL[1], 1, w[1]);
                        event(2, fL[0], 3, w[2]);
                        event(3, fL[3], 3, w[3]);
                        event(4, fM[0], 0, w[4]);
                        event(5, fM[1], 0, w[5]);
                        event(6, fM[2], 0, w[6]);
                        event(7, fM[3], 0, w[7]);
                        event(8, fM[4], 0, w[8]);
                        event(9, fU[0], 2, w[9]);
                        event(10, fU[2], 2, w[10]);
                        event(11, fU[0], 4, w[11]);
                        event(12, fU[4], 4, w[12]);
                    } else {
                        event(0, fL[0], 1, w[0]);
                        event(1, fL[1], 1, w[1]);
                        event(2, fL[2], 1, w[2]);
                        event(3, fM[0], 0, w[3]);
                        event(4, fM[1], 0, w[4]);
                        event(5, fM[2], 0, w[5]);
                        event(6, fU[0], 2, w[6]);
                        event(7, fU[2], 2, w[7]);
                        event(8, fU[1], 2, w[8]);
                    }
                    if (on && cX < CPK_SYM_N && cY < CPK_SYM_N) {  // emissions are counted for ACGT x ACGT cells only (:429)
                        double *copy = eWin + (lane & (kWinCopies - 1)) * 80 + cX * 4 + cY;
#pragma unroll
                        for (int s = 0; s < S; s++) atomicAdd(&copy[s * 16], (double)eAcc[s]);
                    }
                }
                nWin++;
                if (untilRefresh == 1 || d2 == sg.tbPrev + 1 || (refresh && CPK_REFRESH_PERIOD == 1)) {
                    // the window ends here (the next diagonal is a refresh point, or the segment's last): its record
                    double mine = lane == kNT ? ref : (double)nWin;  // [13] = ref, [14] = diagonals, [15] unused
#pragma unroll
                    for (int i = 0; i < kNT; i++) {
                        const float t = wave_sum_f32(tW[i]);
                        if (lane == i) mine = (double)t;
                        tW[i] = 0.0f;
                    }
                    pendRec[0] = mine;
                    roll_fence<false>();
#pragma unroll
                    for (int h = 0; h < 2; h++) {  // emission sums [lane] and [64 + lane]
                        const int e = h * CPK_WAVE + lane;
                        double sum = 0.0;
                        if (e < 80) {
#pragma unroll
                            for (int k = 0; k < kWinCopies; k++) {
                                sum += eWin[k * 80 + e];
                                eWin[k * 80 + e] = 0.0;
                            }
                        }
                        pendRec[1 + h] = sum;
                    }
                    roll_fence<false>();
                    pendRj = jw;
                    nWin = 0;
                }
            }
            // slide the window of table entries and F slots down one diagonal
            ga = gb;
            gb = g;
            g = g1;
            g1 = g2;
            g2 = g3;
            double *t = pF0;
            pF0 = pF1;
            pF1 = pF2;
            pF2 = t;
            t = pM2;  // d2 - 1 takes the row of d2 + 2
            pM2 = pM1;
            pM1 = pM0;
            pM0 = t;
            if (refresh) {
                untilRefresh = CPK_REFRESH_PERIOD - 1;
                jr++;
            } else {
                untilRefresh--;
            }
          }
        }
        issueStores();
#pragma unroll
        for (int q = 0; q < kExpGroups; q++)
#pragma unroll
            for (int s = 0; s < S; s++) asm volatile("" : "+v"(fN[q][s]));  // the last request is never used: let it land
    }

    // The window records of tracebackExpect(), scaled by exp(ref - total) now that foldTotals() has the totals, into the
    // kernel's sums: one lane per window.  likelihood: the total in force, once per emitted diagonal (:743).
    __device__ void scaleWindows(const CpkSegment &sg, double (&tAcc)[kNT], double *eLds, double &likelihood) {
        const int J = sg.nRefresh;
        const double *wsum = bring;
        for (int j0 = 0; j0 < J; j0 += CPK_WAVE) {
            const int j = j0 + lane;
            const bool on = j < J;
            const double *rec = wsum + (size_t)(on ? j : 0) * kWinDoubles;
            const double total = ld_self(totals + (on ? j : 0));
            const double c = on ? exp(ld_self(rec + kNT) - total) : 0.0;
            if (on) likelihood += ld_self(rec + kNT + 1) * total;
#pragma unroll
            for (int i = 0; i < kNT; i++) tAcc[i] += c * ld_self(rec + i);
            double *copy = eLds;  // (one copy in this variant: lds_expect_copies)
            for (int e = 0; e < S * 16; e++) {
                const double p = c * ld_self(rec + 16 + e);
                if (on) atomicAdd(&copy[e], p);
            }
        }
    }

    // ---- total probability at every refresh point of the segment: one lane per refresh point, each doing the
    // reference's sequential folds (dpDiagonal_dotProduct :513-523, then the straddle term :649).
    // TEAM: called by one wave of a multi-wave workgroup (no workgroup barrier at the end, a wave-level fence instead)
    template <bool TEAM = false>
    __device__ void foldTotals(const CpkSegment &sg, const CpkDiag *table) {
        const int J = sg.nRefresh;
        for (int j0 = 0; j0 < J; j0 += CPK_WAVE) {
            const int j = j0 + lane;
            const bool on = j < J;
            const int r = sg.tbFrom - CPK_REFRESH_PERIOD * (on ? j : 0);
            const int Wc = on ? table[r].width : 0;
            const int Wm = (on && r + 1 <= sg.dTop) ? table[r + 1].width : 0;
            double total = NEG_INF, straddle = NEG_INF;
            const int WcMax = wave_max_i32(Wc), WmMax = wave_max_i32(Wm);
            // loads are issued eight at a time, then folded in order; padding with -inf leaves the fold unchanged
            // because logAdd(x, -inf) returns x exactly.  The two series are independent chains of sequential logAdds:
            // they advance side by side (one lane cannot hide a logAdd's latency behind anything else here).
            const int WMax = WcMax > WmMax ? WcMax : WmMax;
            double ts[2] = {total, straddle};
            for (int k = 0; k < WMax; k += 8) {
                double x[8], y[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    x[i] = k + i < Wc ? ld_self(cbuf + (size_t)(k + i) * J + j) : NEG_INF;
                    y[i] = k + i < Wm ? ld_self(mbuf + (size_t)(k + i) * J + j) : NEG_INF;
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const double xy[2] = {x[i], y[i]};
                    logadd_n<2>(lg, ts, xy);
                }
            }
            total = ts[0];
            straddle = ts[1];
            if (on) {
                if (r + 1 <= sg.dTop) total = logadd(lg, total, straddle);
                totals[j] = total;
            }
        }
        roll_fence<!TEAM>();
    }

    // ---- thresholded posteriors (pairwiseAligner.c:655-689) from the candidate list, walked backwards so that the
    // output is in the reference's list order (diagonal ascending, x-y descending).
    __device__ int emitMatches(const CpkSegment &sg, const Candidate *cand, int nCand, int32_t *out, int outCap,
                               int count) {
        const double thr = m.threshold;
        for (int top = nCand; top > 0; top -= CPK_WAVE) {
            const int i = top - 1 - lane;
            const bool valid = i >= 0;
            double p = 0.0;
            int x = 0, y = 0;
            if (valid) {
                const double fbv = ld_self(&cand[i].fb);
                const long long xy = __hip_atomic_load(reinterpret_cast<const long long *>(&cand[i].x), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
                x = (int)(xy & 0xffffffffll);
                y = (int)(xy >> 32);
                const double total = ld_self(totals + (sg.tbFrom - (x + y)) / CPK_REFRESH_PERIOD);
                p = exp(fbv - total);
            }
            const bool keep = valid && p >= thr;
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                if (p > 1.0) p = 1.0;
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                const int pos = count + rank;
                if (pos < outCap) {
                    out[3 * (size_t)pos + 0] = (int32_t)floor(p * (double)CPECAN_PROB_1);
                    out[3 * (size_t)pos + 1] = x - 1;
                    out[3 * (size_t)pos + 2] = y - 1;
                }
            }
            count += __popcll(mask);
        }
        return count;
    }
};

// (emission + transition) sums of every DP term, see Sweep::wt for the layout
template <int S>
__device__ __forceinline__ void fill_weights(double *wt, const CpkModel &m, const KConsts &kc, int lane) {
    constexpr int kWM = S == 5 ? 5 : 3, kWG = S == 5 ? 4 : 3;
    for (int i = lane; i < 25 * kWM + 10 * kWG; i += CPK_WAVE) {
        double e, t;
        if (i < 25 * kWM) {
            const int j = i % kWM;
            e = m.matchEm[i / kWM];
            t = j == 0 ? kc.matchContinue : j == 1 ? kc.matchFromShortX : j == 2 ? kc.matchFromShortY
              : j == 3 ? kc.matchFromLongX : kc.matchFromLongY;
        } else {
            const int r = i - 25 * kWM, y = r >= 5 * kWG, c = (r - y * 5 * kWG) / kWG, j = (r - y * 5 * kWG) % kWG;
            e = y ? m.gapYEm[c] : m.gapXEm[c];
            if (S == 5) {
                t = j == 0 ? (y ? kc.shortOpenY : kc.shortOpenX) : j == 1 ? (y ? kc.shortExtendY : kc.shortExtendX)
                  : j == 2 ? (y ? kc.longOpenY : kc.longOpenX) : (y ? kc.longExtendY : kc.longExtendX);
            } else {
                t = j == 0 ? (y ? kc.shortOpenY : kc.shortOpenX) : j == 1 ? (y ? kc.shortExtendY : kc.shortExtendX)
                  : (y ? kc.shortSwitchToY : kc.shortSwitchToX);
            }
        }
        wt[i] = e + t;
    }
}

// EMIT: CPECAN_EMIT_MATCH (0), CPECAN_EMIT_INDEL (1) or kEmitForward (3: forward sweep only, total probability out)
constexpr int kEmitForward = 3;

// MODE: how a launch uses the kernel.
//   kModeWhole    a queue of REGIONS; a wave takes a region through forward sweeps and tracebacks, segment by segment, with
//                 the forward values of ONE segment in its per-wave ring (the default)
//   kModeForward  a queue of regions; forward sweep of the whole region into the REGION's own ring, no traceback
//   kModeTrace    a queue of (region, segment) ITEMS; traceback, totals and emission of one segment from the region's ring
// The last two are the two launches of a SPLIT class (cpecan_kernels.hip): when a class has fewer regions than the
// chip has wave slots (BASELINE config A: 1000 regions on 2048 slots; a strong-scaled batch on 8 GPUs), the tracebacks
// of a region -- independent of each other once its forward values exist (the backward sweep of a segment starts from
// a constant end-state vector, pairwiseAligner.c:798) -- become queue items of their own and fill the idle slots.
//   kModeFused    ONE launch for a split class: the queue holds the class's regions (forward sweep into the region's ring,
//                 as kModeForward) and, behind them, its (region, segment) items (as kModeTrace), segments in ascending
//                 order.  The forward wave of a region publishes how many of its segments have their forward values
//                 complete (release at agent scope); a wave that draws an item waits for that count (acquire).  A region's
//                 ticket is always drawn before any item's, so whatever an item waits for is in progress on a running wave
//                 or done: no deadlock whatever the number of resident waves.  The tracebacks of a region's first
//                 segments then run beside the forward sweeps of the class instead of behind the slowest of them.
constexpr int kModeWhole = 0, kModeForward = 1, kModeTrace = 2, kModeFused = 3;
// kModeFused hands the forward ring from wave to wave inside one launch with sc1 (device-scope write-through / read-through)
// accesses and a flag behind an acknowledged-stores wait: the hand-off MI355X_MICROARCH.md lists for gfx942 / gfx950, outside
// what the HIP memory model promises in general.  This translation unit is gfx950 code; refuse anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "cpecan_kernels.hip: the one-launch hand-off (kModeFused) relies on gfx942 / gfx950 sc1 semantics"
#endif

// WPS: waves per SIMD the registers are allocated for.  Two everywhere (the five-state match kernel needs ~230 VGPRs)
// except the three-state match kernels of classes with more regions than two waves per SIMD hold: at 168 VGPRs (a
// handful of spills) three fit, and a queue that long runs 9-34 % faster with them (cpk_device_upload).
// ABS: the sweeps of a split class over absolute positions (Sweep::forwardStreamAbs / tracebackAbs): match emitter, LDS rows,
// fixed expansion (KArgs::dpos holds the positions)
#ifndef CPK_FUSED_PRIO
#define CPK_FUSED_PRIO 0
#endif
#ifndef CPK_INSWEEP_PLAIN_FWD
#define CPK_INSWEEP_PLAIN_FWD 1  // bands this narrow have no whole group in front of a tail: the plain per-diagonal forward (1.2 % faster, config 5)
#endif
// INSWEEP: expectation emitter, every diagonal of the class within one 64-lane group: the events are formed inside the
// traceback (Sweep::tracebackExpect / scaleWindows) instead of in a second pass (Sweep::expectations)
// (INSWEEP = 1: a class without a diagonal wider than 64 cells -- one group per diagonal, nothing kept for a second)
template <int S, bool FAST, int EMIT, int MODE = kModeWhole, int WPS = CPK_SWEEP_WAVES, bool ABS = false, int INSWEEP = 0>
__global__ void __launch_bounds__(CPK_WAVE) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
cpecan_pairhmm_sweep(const KArgs a) {
    static_assert(!ABS || (FAST && MODE != kModeWhole && EMIT == CPECAN_EMIT_MATCH), "absolute positions: split classes of the match emitter");
    static_assert(!INSWEEP || (FAST && MODE == kModeWhole && EMIT == CPECAN_EMIT_EXPECT), "in-sweep events: expectation emitter, LDS rows");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const CpkModel &m = *a.model;
    const int stride = a.geo.rollStride;
    constexpr int R = ABS ? S : 2 * S + 1;  // rows of the rolling buffers (Sweep::R; absolute positions: two arrays of S rows)
    const int rollDoubles = (ABS ? 2 * S : 2 * S + 1) * stride;

    // LDS (doubles): logAdd cubics | emission tables | expectation sums | rolling buffers (FAST) | symbol strings (FAST)
    fill_cubics(lds);
    const Cubic *lg = reinterpret_cast<const Cubic *>(lds);
    double *em = lds + kLdsCubics;  // (no plain emission table any more: kLdsEm == 0, every term reads `wt`)
    double *wt = lds + kLdsCubics + kLdsEm;
    fill_weights<S>(wt, m, a.kc, lane);
    double *eLds = lds + kLdsCubics + kLdsEm + kLdsWeights;  // emission-expectation sums of this wave (expectation emitter)
    constexpr int kECopies = lds_expect_copies(INSWEEP != 0);
    if (EMIT == CPECAN_EMIT_EXPECT)
        for (int i = lane; i < kECopies * 80; i += CPK_WAVE) eLds[i] = 0.0;
    constexpr int kNT = S == 5 ? 13 : 9;
    double tAcc[kNT];
#pragma unroll
    for (int i = 0; i < kNT; i++) tAcc[i] = 0.0;
    double likelihood = 0.0;
    constexpr int kHeader = lds_header_doubles(EMIT, INSWEEP != 0);
    double *roll = FAST ? (lds + kHeader) : (a.groll + (size_t)blockIdx.x * a.geo.rollDoubles);
    Candidate *stageLds = reinterpret_cast<Candidate *>(lds + kHeader + (FAST ? (size_t)rollDoubles : 0));
    constexpr int kStageDoubles = MODE == kModeForward ? 0 : lds_stage_doubles(EMIT, ABS);  // a forward launch stages no candidates
    uint8_t *seqLds = reinterpret_cast<uint8_t *>(lds + kHeader + (size_t)rollDoubles + kStageDoubles);
    // every rolling cell starts as -inf; position 0 of each row is never written again (the guard)
    for (int i = lane; i < rollDoubles; i += CPK_WAVE) roll[i] = NEG_INF;
    // expectation step inside the traceback (Sweep::tracebackExpect): three F slots and the window's emission sums behind the strings
    constexpr bool expInSweep = INSWEEP != 0;
    double *frowLds = reinterpret_cast<double *>(seqLds + (a.geo.seqLdsBytes + 15) / 16 * 16);
    double *eWinLds = reinterpret_cast<double *>(frowLds + 3 * (a.geo.maxWidth + 1) * S);
    if (expInSweep) {
        for (int i = lane; i < 3 * (a.geo.maxWidth + 1) * S; i += CPK_WAVE) frowLds[i] = NEG_INF;
        for (int i = lane; i < kExpectWinCopies * 80; i += CPK_WAVE) eWinLds[i] = 0;
    }
    __syncthreads();

    const size_t slot = blockIdx.x;
    for (;;) {
        // Every lane takes part in the ticket fetch (lane 0 adds 1, the others add 0; hipcc folds this into one
        // atomic per wave).  Do NOT write this as `if (lane == 0) ticket = atomicAdd(..)`: hipcc 7.2 jump-threads
        // the lane test across the loop back-edge and re-runs the readfirstlane with 63 lanes -> endless loop.
#if defined(CPK_DIAGNOSTICS) && defined(CPK_TICKET_LANE0)
        // the form that hangs, kept for the ISA comparison only (profiles/r02_ticket_fetch_isa.txt)
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(a.queue, 1u);
#else
        const unsigned int ticket = atomicAdd(a.queue, lane == 0 ? 1u : 0u);
#endif
        const int tk = __builtin_amdgcn_readfirstlane((int)ticket);
        if (tk >= a.regionCount + (MODE == kModeFused ? a.itemCount : 0)) break;
        // kModeTrace: the queue holds (region, segment) items, longest first; kModeFused: regions, then items; else regions
        const bool traceRole = MODE == kModeTrace || (MODE == kModeFused && tk >= a.regionCount);  // wave-uniform
        const bool forwardRole = MODE == kModeForward || (MODE == kModeFused && !traceRole);
        const int ti = MODE == kModeFused ? tk - a.regionCount : tk;
#if CPK_FUSED_PRIO
        // One launch, regions and items in one queue: the forward sweep of a region is the chain everything else of the
        // region waits for, so its wave goes ahead of the item waves that share its SIMD (s_setprio: the arbiter picks the
        // highest-priority wave that can issue).
        if (MODE == kModeFused) {
            if (forwardRole) __builtin_amdgcn_s_setprio(CPK_FUSED_PRIO);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        const int r = traceRole ? a.items[ti].region : a.regionBase + tk;
        const int itemSeg = traceRole ? a.items[ti].seg : 0;

        const CpkRegion &rg = a.regions[r];
        const int lX = rg.lX, lY = rg.lY, N = lX + lY;
        const uint8_t *gx = a.symbols + rg.seqXOff, *gy = a.symbols + rg.seqYOff;
        if (FAST && (!ABS || a.geo.reserved0)) {  // (reserved0: CPECAN_ABS_WINDOWS=0, whole strings for the absolute-position sweeps too)
            // stage N + bases + N of both strings into LDS, two symbols per byte (per-cell reads come from here)
            // (absolute positions: the symbols a SEGMENT's diagonals touch, staged per segment below)
            stage_symbols<CPK_WAVE>(seqLds, gx, lX + 2, lane);
            stage_symbols<CPK_WAVE>(seqLds + ((lX + 3) >> 1), gy, lY + 2, lane);
            roll_fence<false>();
        }
        const CpkDiag *table = a.diags + rg.diagOff;
        Sweep<S, FAST, R, MODE == kModeFused, MODE != kModeWhole, ABS, INSWEEP == 1 ? 1 : 2> sw{a,
                          a.kc,
                          DiagCache{table, N, 0, lane, 0, 0, 0, 0, ABS ? a.dpos + rg.diagOff : nullptr, 0},
                          FAST ? seqLds : gx,
                          FAST ? seqLds + ((lX + 3) >> 1) : gy,
                          roll,
                          em,
                          wt,
                          lg,
                          MODE == kModeWhole ? a.ring + slot * (size_t)a.geo.ringCells * S : a.ring + (size_t)rg.ringBase,
                          a.cand + slot * (size_t)a.geo.fbCells * (EMIT == CPECAN_EMIT_INDEL ? 3 : 1),
                          stageLds,
                          a.cbuf + slot * (size_t)a.geo.refreshCells,
                          a.mbuf + slot * (size_t)a.geo.refreshCells,
                          a.totals + slot * (size_t)a.geo.maxRefresh,
                          stride,
                          lane,
                          lane * R,
                          N,
                          CpkDiag{},
                          CpkDiag{}};
        sw.setStride = S * stride;
        if (EMIT == CPECAN_EMIT_EXPECT) {
            // B of a segment's emitted cells (expectations()) or its window records (tracebackExpect())
            sw.bring = a.bring + slot * (size_t)(expInSweep ? (int64_t)a.geo.maxRefresh * sw.kWinDoubles : a.geo.fbCells * S);
            if (expInSweep) {
                sw.frow = frowLds;
                sw.eWin = eWinLds;
            }
        }
        constexpr int NL = EMIT == CPECAN_EMIT_INDEL ? 3 : 1;
        int count[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) count[l] = 0;
        if (EMIT == kEmitForward) {
            // getForwardProbWithBanding (pairwiseAligner.c:879-931): forward sweep over the whole matrix, then the
            // total probability of the last diagonal against the end prior; no traceback.
            double total = 0.0;  // LOG_ONE for two empty sequences (:889-891)
            if (N > 0) {
                sw.dc.load(0);
                const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
                const CpkDiag g0 = sw.dc.get(0, false);
                double *cur0 = sw.fbuf1(0);
                if (lane < S) cur0[lane] = startPrior[lane];
                roll_fence<!FAST>();
                sw.f1 = g0;
                sw.f2 = g0;
                for (int d = 1; d <= N;) {
                    sw.dc.load(d);  // table entries of diagonals d .. d+63
                    const int dEnd = d + CPK_WAVE - 1 < N ? d + CPK_WAVE - 1 : N;
                    for (; d <= dEnd; d++) {
                        if (FAST) sw.forwardStream(d, sw.dc.at(d - sw.dc.base), 0);  // nothing reads F back
                        else sw.forward(d, sw.dc.at(d - sw.dc.base), 0);
                    }
                }
                if (FAST) sw.flushTail();
                const double *endPrior = rg.raggedRight ? m.raggedEnd : m.end;
                const double *last = sw.fbuf1(N);
                const int W = sw.f1.width;
                total = NEG_INF;  // dpDiagonal_dotProduct (:513-523) over the cells of diagonal N, every lane alike
                for (int k = 0; k < W; k++) {
                    double t = last[0 + k * R] + endPrior[0];
#pragma unroll
                    for (int s = 1; s < S; s++) t = logadd(lg, t, last[s + k * R] + endPrior[s]);
                    total = logadd(lg, total, t);
                }
            }
            if (lane == 0) a.forwardOut[r] = total;
            continue;
        }
        if (N > 0) {
            sw.dc.load(0);
            // diagonal 0: the single cell (0,0) holds the start prior (pairwiseAligner.c:776-777)
            const double *startPrior = rg.raggedLeft ? m.raggedStart : m.start;
            if (!traceRole) {
                const CpkDiag g0 = sw.dc.get(0, false);
                double *cur = sw.fbuf1(0);
                if (ABS) {  // empty rows, the single cell of diagonal 0 at its position
                    sw.absWipe();
                    sw.apos1 = sw.apos2 = sw.dc.posGet(0, false) & 0x7fff;
                    cur += (sw.apos1 - 1) * R;
                }
                double *o0 = sw.ringAt(g0);
                if (lane < S) {
                    cur[lane] = startPrior[lane];
                    sw.ringSt(o0 + sw.ringIdx(1, lane, 0), startPrior[lane]);
                }
                roll_fence<!FAST>();
                sw.f1 = g0;
                sw.f2 = g0;
            }
            int d = 1;
            int emitSeg = 0, emitFrom = a.segs[rg.segOff].tbFrom;  // the segment whose traceback emits diagonal d: the first with tbFrom >= d
            int toRefresh = (emitFrom - 1) % CPK_REFRESH_PERIOD;    // (emitFrom - d) mod 10 as a countdown: no division per diagonal
            const int siFirst = traceRole ? itemSeg : 0, siEnd = traceRole ? itemSeg + 1 : rg.nSeg;
            for (int si = siFirst; si < siEnd; si++) {
                const CpkSegment sg = a.segs[rg.segOff + si];
                if (ABS && !a.geo.reserved0) {
                    // Symbol windows (round 4): only the symbols the diagonals of THIS step touch are staged -- the forward
                    // sweep of diagonals d .. dTop, or the traceback of tbPrev + 1 .. dTop, whose cells also read the symbols
                    // one past their own (Sweep::bwdCells) -- ~0.7 KB instead of the 2 KB of both whole strings of a 2 kb pair.
                    // The edges of a band that may run under absolute positions move one x-y step per diagonal
                    // (CpkRegion::absOk), so the smallest and the largest x and y of a diagonal never decrease with d: the
                    // window runs from the first cell of the lowest diagonal to the last cell of the highest.  The host sized
                    // the LDS for the largest window of the class (cpecan_host.c, RegionPlan::winBytes).
                    const int dLo = traceRole ? sg.tbPrev + 1 : d;
                    const CpkDiag gl = table[dLo], gh = table[sg.dTop];
                    const int xLoL = (dLo + gl.xmyL) >> 1, xHiL = xLoL + gl.width - 1;
                    const int xLoH = (sg.dTop + gh.xmyL) >> 1, xHiH = xLoH + gh.width - 1;
                    const int x0 = xLoL & ~1, y0 = (dLo - xHiL) & ~1;                 // even: two symbols per byte
                    const int x1 = xHiH + 1 < lX + 1 ? xHiH + 1 : lX + 1;            // last symbol index read (padded strings:
                    const int y1 = sg.dTop - xLoH + 1 < lY + 1 ? sg.dTop - xLoH + 1 : lY + 1;  // lX + 2 and lY + 2 entries)
                    const int nX = x1 - x0 + 1, nY = y1 - y0 + 1;
                    uint8_t *winY = seqLds + ((nX + 1) >> 1);
                    stage_symbols<CPK_WAVE>(seqLds, gx + x0, nX, lane);
                    stage_symbols<CPK_WAVE>(winY, gy + y0, nY, lane);
                    sw.sxp = seqLds - (x0 >> 1);
                    sw.syp = winY - (y0 >> 1);
                    roll_fence<false>();
                }
                // Which states of F[d] the traceback will read back: the match row always (posteriors), every state on the
                // refresh points of the segment that emits d (cell dot products, pairwiseAligner.c:636-653; the schedule is
                // known up front) and on the two diagonals the forward sweep is resumed from; the indel and expectation
                // emitters read every state of every diagonal.  For the match emitter this cuts the ring stores from 8*S to
                // ~8 + 0.8*(S-1) bytes per cell.
                while (!traceRole && d <= sg.dTop) {
                    sw.dc.load(d);  // table entries of diagonals d .. d+63
                    const int dEnd = d + CPK_WAVE - 1 < sg.dTop ? d + CPK_WAVE - 1 : sg.dTop;
                    for (; d <= dEnd; d++) {
                        if (d > emitFrom) {
                            while (d > emitFrom) emitFrom = a.segs[rg.segOff + ++emitSeg].tbFrom;  // the last segment ends at N
                            toRefresh = (emitFrom - d) % CPK_REFRESH_PERIOD;
                        }
                        const bool all = EMIT != CPECAN_EMIT_MATCH || toRefresh == 0 || d >= sg.dTop - 1;
                        toRefresh = toRefresh == 0 ? CPK_REFRESH_PERIOD - 1 : toRefresh - 1;
                        if (ABS) sw.forwardStreamAbs(d, sw.dc.at(d - sw.dc.base), sw.dc.posAt(d - sw.dc.base), all ? S : 1);
                        else if (FAST && !(INSWEEP && CPK_INSWEEP_PLAIN_FWD)) sw.forwardStream(d, sw.dc.at(d - sw.dc.base), all ? S : 1);
                        else sw.forward(d, sw.dc.at(d - sw.dc.base), all ? S : 1);
                    }
                }
                if (ABS && !traceRole) sw.absFlushTail();
                else if (FAST && !(INSWEEP && CPK_INSWEEP_PLAIN_FWD) && !traceRole) sw.flushTail();  // the traceback needs every cell of dTop
                if (forwardRole) {  // the tracebacks of this region are items of their own (of the next launch, or of this one)
                    if (MODE == kModeFused) {
                        // the ring stores are device-scope write-through (Sweep::ringSt): once they are acknowledged -- this
                        // fence is the wait, nothing else at workgroup scope -- they are where every XCD reads them, and the
                        // count goes out.  (An agent-scope release here writes back the XCD's whole L2: measured, 10 % slower.)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ring store of this wave is acknowledged
                        if (lane == 0) __hip_atomic_store(a.progress + r, si + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    continue;
                }
                if (MODE == kModeFused) {
                    // wait until the region's forward wave has passed this segment's top diagonal (bounded: a count that
                    // never comes is reported, not waited for)
                    int seen = 0;
                    for (int spin = 0; spin < a.geo.fusedSpin; spin++) {
                        seen = __hip_atomic_load(a.progress + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen > si) break;
                        __builtin_amdgcn_s_sleep(16);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // ordering only: the ring loads are device-scope (ringLd)
                    if (__builtin_amdgcn_readfirstlane(seen) <= si) {
                        if (lane == 0) a.progress[a.geo.nRegions] = 1;  // the error word behind the counts
                        continue;
                    }
                }
#ifdef CPK_DIAGNOSTICS
                if (a.geo.debug & 2) continue;  // diagnostic build only: time the forward sweep alone (no traceback, no output)
#endif
                const double *endPrior = (sg.atEnd && rg.raggedRight) ? m.raggedEnd : m.end;
                int nCand[NL];
                if constexpr (ABS) sw.template tracebackAbs<NL, true>(sg, endPrior, (a.geo.debug & 1) ? a.dbgFb + rg.dbgCellOff : nullptr, nCand);
                else if constexpr (INSWEEP != 0) sw.tracebackExpect(sg, endPrior, (a.geo.debug & 1) ? a.dbgFb + rg.dbgCellOff : nullptr);
                else sw.template traceback<NL, EMIT != CPECAN_EMIT_EXPECT>(sg, endPrior, (a.geo.debug & 1) ? a.dbgFb + rg.dbgCellOff : nullptr, nCand);
                roll_fence<true>();  // candidate / cbuf / mbuf stores of all lanes are complete before they are re-read
                sw.foldTotals(sg, table);
                if (a.geo.debug & 1) {
                    for (int d2 = sg.tbPrev + 1 + lane; d2 <= sg.tbFrom; d2 += CPK_WAVE)
                        a.dbgTotals[rg.dbgDiagOff + d2] = ld_self(sw.totals + (sg.tbFrom - d2) / CPK_REFRESH_PERIOD);
                }
                if (EMIT == CPECAN_EMIT_EXPECT) {
                    if constexpr (INSWEEP != 0) sw.scaleWindows(sg, tAcc, eLds, likelihood);
                    else sw.expectations(sg, tAcc, eLds, likelihood);
                }
#pragma unroll
                for (int l = 0; l < (EMIT == CPECAN_EMIT_EXPECT ? 0 : NL); l++) {
                    if (traceRole) {
                        // the segment's own part of the region's output slice (the other segments are written by other waves)
                        const int n = sw.emitMatches(sg, sw.cand + (size_t)l * a.geo.fbCells, nCand[l],
                                                     a.triples + 3 * ((size_t)l * a.outTriplesPerList + rg.outOff + sg.outOff),
                                                     sg.outCap, 0);
                        if (lane == 0) {
                            a.segStarts[(size_t)l * a.nSegsTotal + rg.segOff + si] = sg.outOff;
                            a.segCounts[(size_t)l * a.nSegsTotal + rg.segOff + si] = n;
                        }
                    } else {
                        if (lane == 0) a.segStarts[(size_t)l * a.nSegsTotal + rg.segOff + si] = count[l];
                        count[l] = sw.emitMatches(sg, sw.cand + (size_t)l * a.geo.fbCells, nCand[l],
                                                  a.triples + 3 * ((size_t)l * a.outTriplesPerList + rg.outOff), rg.outCap,
                                                  count[l]);
                    }
                }
                if (MODE == kModeWhole && !sg.atEnd) {
                    // the traceback reused the rolling buffers: restore F[dTop-1], F[dTop] for the forward sweep
                    const CpkDiag gTopM1 = sw.dc.get(sg.dTop - 1, false);
                    const CpkDiag gTop = sw.dc.get(sg.dTop, false);
                    sw.reloadForward(gTopM1, sg.dTop - 1);
                    sw.reloadForward(gTop, sg.dTop);
                    sw.f2 = gTopM1;
                    sw.f1 = gTop;
                }
            }
        }
#pragma unroll
        for (int l = 0; l < NL; l++)
            if (MODE == kModeWhole && lane == 0) a.outCounts[(size_t)l * a.geo.nRegions + r] = count[l];
    }
    if (EMIT == CPECAN_EMIT_EXPECT) {
        // one partial result per resident wave: [0,25) transitions [from*S+to], [25,105) emissions, [105] likelihood
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
            for (int k = 0; k < kECopies; k++) e += eLds[k * 80 + i];
            dst[25 + i] = e;
        }
        {  // every lane holds the totals of the diagonals whose first cell it computed (Sweep::expectations)
            double v = likelihood;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) dst[105] = v;
        }
    }
}
