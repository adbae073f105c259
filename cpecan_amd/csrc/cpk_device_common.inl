// cpk_device_common.inl -- shared device-side definitions: kernel arguments, logAdd, lane helpers, the diagonal-table cache.
// Part of the single HIP translation unit cpecan_kernels.hip (included there, in this order); not compiled on its own.

// ------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------
struct Candidate {  // a cell that may pass the posterior threshold once the total probability is known
    double fb;      // F.match + B.match
    int32_t x, y;   // matrix coordinates
};

// Transition log-probabilities and the threshold travel BY VALUE in the kernel arguments: kernarg loads are scalar
// (s_load), so the hot loops never wait on vector memory for a model constant.
struct KConsts {
    double matchContinue;
    double matchFromShortX, matchFromShortY, matchFromLongX, matchFromLongY;
    double shortOpenX, shortOpenY, shortExtendX, shortExtendY, shortSwitchToX, shortSwitchToY;
    double longOpenX, longOpenY, longExtendX, longExtendY;
    double threshold;
};

struct KArgs {
    KConsts kc;
    const CpkRegion *regions;
    const CpkDiag *diags;
    const int32_t *dpos;  // per diagonal: positions of the absolute-position sweeps (cpk_table_gather.inl); null without them
    const CpkSegment *segs;
    const uint8_t *symbols;
    const CpkModel *model;
    CpkGeometry geo;
    double *ring;      // [slots][ringCells*S]   forward values of the live traceback segment
    Candidate *cand;   // [slots][candCells]     posterior candidates of the segment being traced back
    double *cbuf;      // [slots][refreshCells]  per-cell F.B dot products on refresh diagonals, [k][j]
    double *mbuf;      // [slots][refreshCells]  per-cell "match straddling the diagonal" terms, [k][j]
    double *totals;    // [slots][maxRefresh]
    double *groll;     // [slots][rollDoubles]   rolling buffers when they do not fit in LDS
    double *bring;     // [slots][fbCells*S]     expectation emitter: backward values of the emitted cells of the segment
    int32_t *outCounts;  // [nLists][nRegions]
    int32_t *segStarts;  // [nLists][nSegsTotal]
    int32_t *segCounts;  // [nLists][nSegsTotal]  triples of a segment (split classes: their segments are written apart)
    int *progress;       // [nRegions + 1] kModeFused: segments of a region whose forward values are complete; last word: error flag
    int32_t itemCount;   // kModeFused: items behind the regionCount regions of the queue
    const CpkItem *items;  // (region, segment) queue of a split class's traceback launch
    int32_t *triples;    // [nLists][outTriplesPerList*3]
    int64_t outTriplesPerList;
    int64_t nSegsTotal;
    unsigned int *queue;
    int32_t regionBase, regionCount;  // this launch works on regions [regionBase, regionBase + regionCount)
    double *forwardOut;  // [nRegions] total forward log-probability (forward mode)
    double *expectOut;   // [slots][128] per-wave expectation partial sums (expectation mode)
    double *dbgFb;
    double *dbgTotals;
};

// logAdd, impl/pairwiseAligner.c:287-307: with hi = max(x, y), lo = min(x, y), d = hi - lo the reference returns hi when
// lo == -inf or d >= 7.5, else lo + P(d), P one of four cubics chosen by d <= 1.0 / <= 2.5 / <= 4.5 / else, evaluated in
// Horner form with separately rounded multiplies and adds.  The coefficients are float literals in the reference, i.e.
// float32 values widened to double.
//
// Two builds of the same function:
//  * CPK_LOGADD_EXACT=1 (diagnostic build, `make EXACT=1`): the reference's operations one for one -- 11 fp64 + 9 integer
//    vector instructions.  Bit-identical to the oracle; used to tell a rounding difference from a logic error.
//  * default: 7 fp64 + 5 integer vector instructions, the kernel's largest single saving (16 logAdds per cell).
//      r = hi + Q(dc),  dc = min(|x - y|, 8),  Q(d) = P(d) - d  (the same cubic with c1 - 1), three FMAs,
//      and a FIFTH all-zero table row for dc >= 7.5, so that "return hi" needs no compare and no select:
//      lo + P(d) = hi - d + P(d) = hi + Q(d).  |x - y| is +inf when one operand is -inf and NaN when both are; the
//      IEEE minimum with 8 turns both into 8 (row 4, Q = 0), so -inf operands give exactly the reference's result.
//    Differences from the reference's arithmetic: the rounding of d is not cancelled exactly (<= 1 ulp of d), Horner is
//    fused, and c1 - 1 is rounded once: |delta| <= ~1e-15 per logAdd against values of 1e3..1e4 that both sides round
//    at 2e-13..2e-12 anyway.  Measured against the oracle: see tests/parity.py (log-space values agree to < 1e-9 over
//    4000-diagonal sweeps; posteriors to ~1e-10 relative; the gate is 1e-5).  A value of d equal to a threshold to
//    the last bit takes the upper cubic here and the lower one in the reference (the cubics differ by ~1e-4 there);
//    such a d has probability ~2^-50 per logAdd.
// The cubics live in an LDS table [row][c3, c2, c1, c0] of 32-byte rows read with two ds_read_b128: selecting four
// 64-bit coefficients with v_cndmask cost 24 VALU instructions per logAdd (47 % of the forward loop).  Rows 0..4 span
// 160 bytes, fewer than the 64 banks x 4 bytes: conflict-free for every combination of rows.
struct __attribute__((aligned(16))) Cubic {
    double c3, c2, c1, c0;
};

#ifndef CPK_LOGADD_EXACT
#define CPK_LOGADD_EXACT 0
#endif

// Shipped form: dc = min(|x - y|, 7.75) falls into one of 16 buckets of width 0.5 -- the thresholds 1.0, 2.5, 4.5, 7.5 are all
// multiples of 0.5 -- and the bucket is floor(2 dc): one multiply and one conversion, both exact (round 2 found the segment
// from the bit pattern of dc in four integer instructions).  The table holds, per bucket, its segment's cubic as TWO
// 16-byte halves in two arrays of 16 rows: {c3, c2} at element 2 b, {c1 - 1, c0} at element 32 + 2 b.  An array of 16 rows of
// 16 bytes covers the 64 banks exactly once, so each of the two ds_read_b128 of a logAdd is conflict-free whatever
// buckets the lanes hit (one row of 32 bytes per bucket is not: rows 8 apart share banks -- measured, 26 such rows: config
// B 88 -> 100 ms, profiles/r03_ab_cubic_table_forms.txt).
#if CPK_LOGADD_EXACT
constexpr int kCubicDoubles = 24;  // 5 rows of {c3, c2, c1, c0}, padded
#else
constexpr int kCubicDoubles = 64;  // 16 x {c3, c2}, then 16 x {c1 - 1, c0}
#endif
#ifndef CPK_LOGADD_OMOD
#define CPK_LOGADD_OMOD 0
#endif
// CPK_LOGADD_OMOD=1 (round 4; built, measured and NOT shipped -- tools/ab_build.sh omod -DCPK_LOGADD_OMOD=1): nine vector
// instructions per logAdd instead of ten, for 0.7 % at BASELINE config B (forward launch 35.0 -> 35.1 ms, traceback launch
// 48.2 -> 47.4), 0.9 % at config 5 and 1.5 % at config A (profiles/r04_ab_logadd_omod.txt) -- the forward group lost 12 of
// its 187 instructions and ran no faster: the sweeps are not bound by the number of vector instructions -- against a
// floating-point mode that differs from the reference's for every kernel.  How it works:
// The bucket of a logAdd is floor(2 dc) and the doubling is an instruction of its own (v_add_f64 dc, dc).  v_min_f64 can deliver 2 * min(|x - y|, 7.75) by itself through the VOP3
// output modifier (mul:2) -- which the hardware honours only with MODE.IEEE = 0 and the f64 denormals flushed
// (tools/omod_check.hip: ignored otherwise) -- and Horner runs on u = 2 dc with the coefficients scaled by exact powers of
// two (c3 / 8, c2 / 4, (c1 - 1) / 2, c0): every intermediate is the old one times 1/4, 1/2, 1, bit for bit.  Nine vector
// instructions per logAdd instead of ten.  What the mode change costs: signalling NaNs are not quieted (there are none),
// and an f64 result below 2.2e-308 becomes zero -- log-space values never are, and a posterior that small is far below
// any threshold (floor(p * 1e7) is 0 either way).  Every kernel that calls logadd sets the mode through fill_cubics().
__device__ __forceinline__ void logadd_fp_mode() {
#if CPK_LOGADD_OMOD
    __builtin_amdgcn_s_setreg((0 << 11) | (9 << 6) | 1, 0);  // hwreg(HW_REG_MODE, 9, 1): IEEE off
    __builtin_amdgcn_s_setreg((1 << 11) | (6 << 6) | 1, 0);  // hwreg(HW_REG_MODE, 6, 2): f64 / f16 denormals flushed
#endif
}
__device__ __forceinline__ void fill_cubics(double *t) {
    logadd_fp_mode();
    const float c[16] = {-0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                         -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                         -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                         -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f};
    for (int l = threadIdx.x; l < kCubicDoubles; l += CPK_WAVE) {
        int seg, coef;  // segment 0..3 (4: d >= 7.5, all zeros: the result is hi) and coefficient 0..3 = c3, c2, c1, c0
        if (CPK_LOGADD_EXACT) {
            seg = l >> 2;
            coef = l & 3;
        } else {
            const int bucket = (l & 31) >> 1;  // [0, 1) | [1, 2.5) | [2.5, 4.5) | [4.5, 7.5) | [7.5, 8) in steps of 0.5
            seg = bucket >= 15 ? 4 : bucket >= 9 ? 3 : bucket >= 5 ? 2 : bucket >= 2 ? 1 : 0;
            coef = (l >> 5) * 2 + (l & 1);
        }
        const int idx = seg * 4 + coef;
        float v = c[0];
#pragma unroll
        for (int i = 1; i < 16; i++) v = idx == i ? c[i] : v;
        double w = seg >= 4 ? 0.0 : (double)v;
        if (!CPK_LOGADD_EXACT && seg < 4 && coef == 2) w = w - 1.0;  // Q(d) = P(d) - d
        if (CPK_LOGADD_OMOD) w *= coef == 0 ? 0.125 : (coef == 1 ? 0.25 : (coef == 2 ? 0.5 : 1.0));  // Horner on u = 2 dc
        t[l] = w;
    }
}

#if CPK_LOGADD_EXACT
// The segment of d = hi - lo is found without fp64 compares.  For d >= 0 the IEEE bit pattern is monotone; the three
// thresholds (1.0, 2.5, 4.5) have a zero low dword and high dwords that are multiples of 2^17, so the segment is a
// function of the bucket  b = ((bits(d) + 2^49 - 1) >> 49) - (bits(1.0) >> 49), saturated at 0:
// b == 0 -> d <= 1;  1..10 -> (1, 2.5];  11..17 -> (2.5, 4.5];  18.. -> above, i.e. segment = number of set bits of
// {0, 10, 17} below position b.  Exact for every double, thresholds included.  NaN / +inf / d >= 8 may pick any
// segment: the result is `hi` then.
__device__ __forceinline__ int cubic_row(double d) {
    // bits(d) > bits(T)  <=>  bits(d) + 2^49 - 1 >= bits(T) + 2^49 for the three thresholds (multiples of 2^49), so the
    // bucket is the high part of one 64-bit add; the saturating subtract sends every d <= 1 (d == 0 included) to
    // bucket 0, and v_bfe_u32 reads only 5 bits of its width, which is harmless: buckets above 24 mean d >= 8.
    const unsigned long long bits = (unsigned long long)__double_as_longlong(d) + ((1ull << 49) - 1ull);
    const unsigned h = (unsigned)(bits >> 32) >> 17;
    const unsigned b = __builtin_elementwise_sub_sat(h, 0x3FF00000u >> 17);
    return __builtin_popcount(__builtin_amdgcn_ubfe((1u << 0) | (1u << 10) | (1u << 17), 0u, b));
}

__device__ __forceinline__ double logadd(const Cubic *tab, double x, double y) {
    const double hi = __builtin_fmax(x, y);
    const double lo = __builtin_fmin(x, y);
    const double d = hi - lo;  // +inf / NaN exactly when the reference's LOG_ZERO tests fire: both fail d < 7.5
    const Cubic q = tab[cubic_row(d)];
    double r = q.c3 * d;
    r = r + q.c2;
    r = r * d;
    r = r + q.c1;
    r = r * d;
    r = r + q.c0;
    r = r + lo;
    return (d < 7.5) ? r : hi;
}
#else
constexpr double kLogAddClamp = 7.75;  // any value in [7.5, 8): the last bucket, whose row is all zeros
struct __attribute__((aligned(16))) CubicHalf {
    double a, b;
};
#if CPK_LOGADD_OMOD
// u = 2 * min(|d|, 7.75) in one instruction (see logadd_fp_mode); NaN (-inf - -inf) and +inf give 15.5, the all-zero bucket
__device__ __forceinline__ double logadd_arg(double d) {
    double u;
    asm("v_min_f64 %0, |%1|, %2 mul:2" : "=v"(u) : "v"(d), "s"(kLogAddClamp));
    return u;
}
__device__ __forceinline__ Cubic cubic_fetch(const Cubic *tab, double u) {
    const unsigned bucket = (unsigned)u;  // v_cvt_u32_f64
#else
__device__ __forceinline__ double logadd_arg(double d) { return __builtin_fmin(__builtin_fabs(d), kLogAddClamp); }
__device__ __forceinline__ Cubic cubic_fetch(const Cubic *tab, double dc) {
    const unsigned bucket = (unsigned)(dc * 2.0);  // v_mul_f64, v_cvt_u32_f64: exact (truncation of an exact product)
#endif
    const CubicHalf *t = reinterpret_cast<const CubicHalf *>(tab);
    const CubicHalf h = t[bucket], l = t[16 + bucket];
    return Cubic{h.a, h.b, l.a, l.b};
}

__device__ __forceinline__ double logadd(const Cubic *tab, double x, double y) {
    const double hi = __builtin_fmax(x, y);
    const double dc = logadd_arg(x - y);
    const Cubic q = cubic_fetch(tab, dc);
    double r = __builtin_fma(q.c3, dc, q.c2);
    r = __builtin_fma(r, dc, q.c1);
    r = __builtin_fma(r, dc, q.c0);
    return hi + r;
}
#endif

// exp(x) for the expectation sums only (probabilities: x <= ~1), whose gate is 1e-5 relative (SURVEY 8a row a11:
// linear-space sums, order-insensitive at 1e-5); the posterior emitters keep the exact exp.  Round 3: 2^(x log2 e) with the
// product formed in double and handed to v_exp_f32 as a float -- four instructions (round 2 split off the integer part in
// double and scaled with v_ldexp_f64: eight, ~1e-7).  The float carries x log2 e to 2^-24 relative, i.e. the result to
// |x log2 e| * 4e-8 relative: 8e-7 for an event 1e-6 times less likely than its cell's total (|x log2 e| = 20), less for
// the likelier ones that make up the sums; events below 2^-126 count as zero.  -inf (an unreachable transition) gives 0,
// and so does NaN (-inf - -inf: a window reference or total of -inf, e.g. a zero-probability region under a loaded HMM
// with -inf transitions): fmax returns its non-NaN operand, one instruction -- a NaN here would poison the sums of the
// whole batch, whose per-wave partials are added up on the host.
__device__ __forceinline__ float exp_1e7f(double x) {
    return __builtin_amdgcn_exp2f((float)__builtin_fmax(x * 1.4426950408889634 /* log2(e) */, -200.0));
}
__device__ __forceinline__ double exp_1e7(double x) { return (double)exp_1e7f(x); }

// Packs the nSym one-byte symbols at src into LDS, two to a byte (low nibble = even index; a missing partner reads as
// N), with LANES threads.  The loads of a batch go out together and are clamped instead of predicated: as a loop of
// load-then-store every iteration was a global round trip of its own -- 16 in a row for a 1 kb pair, for every region and
// (in a split class) every traceback item.
template <int LANES>
__device__ __forceinline__ void stage_symbols(uint8_t *dst, const uint8_t *src, int nSym, int tid) {
    const int nOut = (nSym + 1) >> 1;
    constexpr int kBatch = 4;
    for (int i0 = 0; i0 < nOut; i0 += LANES * kBatch) {
        int lo[kBatch], hi[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const int i = i0 + j * LANES + tid;
            const int e = 2 * i < nSym ? 2 * i : nSym - 1, o = 2 * i + 1 < nSym ? 2 * i + 1 : nSym - 1;
            lo[j] = src[e];
            hi[j] = src[o];
        }
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const int i = i0 + j * LANES + tid;
            if (i < nOut) dst[i] = (uint8_t)(lo[j] | ((2 * i + 1 < nSym ? hi[j] : CPK_SYM_N) << 4));
        }
    }
}

// N independent logAdds advanced in lock-step stages (compare/select -> table fetch -> Horner) so that the N LDS
// table fetches are in flight together instead of one fetch + wait per logAdd.  acc[i] = logAdd(acc[i], t[i]).
template <int N>
__device__ __forceinline__ void logadd_n(const Cubic *tab, double (&acc)[N], const double (&t)[N]) {
#if CPK_LOGADD_EXACT
    double hi[N], lo[N], d[N];
    Cubic q[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        hi[i] = __builtin_fmax(acc[i], t[i]);
        lo[i] = __builtin_fmin(acc[i], t[i]);
        d[i] = hi[i] - lo[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) q[i] = tab[cubic_row(d[i])];
    double r[N];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = q[i].c3 * d[i];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + q[i].c2;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] * d[i];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + q[i].c1;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] * d[i];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + q[i].c0;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = r[i] + lo[i];
#pragma unroll
    for (int i = 0; i < N; i++) acc[i] = (d[i] < 7.5) ? r[i] : hi[i];
#else
    double hi[N], dc[N];
    Cubic q[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        hi[i] = __builtin_fmax(acc[i], t[i]);
        dc[i] = logadd_arg(acc[i] - t[i]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) q[i] = cubic_fetch(tab, dc[i]);
    double r[N];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(q[i].c3, dc[i], q[i].c2);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(r[i], dc[i], q[i].c1);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(r[i], dc[i], q[i].c0);
#pragma unroll
    for (int i = 0; i < N; i++) acc[i] = hi[i] + r[i];
#endif
}

// Row position of neighbour cell i of a diagonal that has `w` cells (w = 0: the diagonal does not exist):
// cells sit at positions 1..w, position 0 of every row is a permanent -inf guard.
__device__ __forceinline__ int guard_pos(int i, int w) { return ((unsigned)i < (unsigned)w) ? i + 1 : 0; }

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = v > o ? v : o;
    }
    return v;
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// Sum of v over the 64 lanes, in every lane: an inclusive scan inside the rows of 16 lanes (row_shr), the rows joined with
// row_bcast:15 / :31 -- six v_add_f32 with a DPP operand and one v_readlane, no LDS (six rounds of __shfl_xor are twelve
// ds_bpermute_b32 and their waits for a double).  The order of the additions is fixed (lane order within pairs, quads, ...).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_or_zero_f32<0x111, 0xf>(v);  // row_shr:1
    v += dpp_or_zero_f32<0x112, 0xf>(v);  // row_shr:2
    v += dpp_or_zero_f32<0x114, 0xf>(v);  // row_shr:4
    v += dpp_or_zero_f32<0x118, 0xf>(v);  // row_shr:8: lane 15 of every row holds the row's sum
    v += dpp_or_zero_f32<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero_f32<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <bool GLOBAL_ROLL>
__device__ __forceinline__ void roll_fence() {
    if (GLOBAL_ROLL) {
        __syncthreads();  // workgroup-scope release/acquire on global memory (single-wave workgroup)
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// data this wave wrote earlier in the same launch: always a vector load, never the scalar cache
__device__ __forceinline__ double ld_self(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// 64 consecutive entries of the region's diagonal table, one per lane, handed out with v_readlane: the sweeps
// touch the table once per diagonal and must not wait on a memory load for it.
struct DiagCache {
    const CpkDiag *table;
    int last;  // highest valid diagonal (N)
    int base;  // diagonal held by lane 0
    int lane;
    int eXmyL, eWidth, eRing, eCell;
    // absolute-position sweeps only: the region's slice of KArgs::dpos and the chunk's entries of it
    const int32_t *posTable = nullptr;
    int ePos = 0;

    __device__ __forceinline__ void load(int b) {
        base = b;
        int i = b + lane;
        i = i < 0 ? 0 : (i > last ? last : i);
        const int4 e = *reinterpret_cast<const int4 *>(table + i);
        eXmyL = e.x;
        eWidth = e.y;
        eRing = e.z;
        eCell = e.w;
        if (posTable) ePos = posTable[i];
        // consume the loaded registers here so the s_waitcnt for this load sits inside the (rare) refill branch;
        // otherwise hipcc puts a vmcnt(0) at the branch merge and every diagonal waits for its ring stores
        asm volatile("" ::"v"(eXmyL), "v"(eWidth), "v"(eRing), "v"(eCell), "v"(ePos));
    }
    __device__ __forceinline__ int posAt(int l) const { return __builtin_amdgcn_readlane(ePos, l); }
    __device__ __forceinline__ int posGet(int d, bool descending) {
        if (d < base || d >= base + CPK_WAVE) load(descending ? d - (CPK_WAVE - 1) : d);
        return __builtin_amdgcn_readlane(ePos, __builtin_amdgcn_readfirstlane(d - base));
    }
    // entry held by lane l of the current chunk (l wave-uniform).  The hot loops walk a chunk with load() outside the
    // loop over its 64 diagonals: a lazy refill inside the loop costs a range check, a branch and a round of VGPR
    // copies at its merge point on every diagonal.
    __device__ __forceinline__ CpkDiag at(int l) const {
        CpkDiag g;
        g.xmyL = __builtin_amdgcn_readlane(eXmyL, l);
        g.width = __builtin_amdgcn_readlane(eWidth, l);
        g.ringOff = __builtin_amdgcn_readlane(eRing, l);
        g.cellOff = __builtin_amdgcn_readlane(eCell, l);
        return g;
    }
    // descending = the caller walks towards lower diagonals (refill so that d is the LAST lane of the chunk)
    __device__ __forceinline__ CpkDiag get(int d, bool descending) {
        if (d < base || d >= base + CPK_WAVE) load(descending ? d - (CPK_WAVE - 1) : d);
        const int l = __builtin_amdgcn_readfirstlane(d - base);
        CpkDiag g;
        g.xmyL = __builtin_amdgcn_readlane(eXmyL, l);
        g.width = __builtin_amdgcn_readlane(eWidth, l);
        g.ringOff = __builtin_amdgcn_readlane(eRing, l);
        g.cellOff = __builtin_amdgcn_readlane(eCell, l);
        return g;
    }
};

#ifndef CPK_COH_ST
#define CPK_COH_ST 1
#endif
#ifndef CPK_COH_PAIRS
#define CPK_COH_PAIRS 1  // 0: the one-launch form stores every ring double on its own (A/B builds; Sweep::ringPut)
#endif
#ifndef CPK_COH_LD
#define CPK_COH_LD 1
#endif
#ifndef CPK_SWEEP_WAVES
#define CPK_SWEEP_WAVES 2  // waves per SIMD the sweep kernel's registers are allocated for
#endif
constexpr int kLdsCubics = kCubicDoubles;  // the logAdd table (fill_cubics)
constexpr int kLdsEm = 0;  // (rounds 1-2 kept 40 doubles of plain emissions here; every reader uses the (emission + transition) table)
// doubles of LDS in front of the rolling rows: cubics + emissions (+ expectation sums for that emitter only)
constexpr int kExpectCopies = 4;  // emission-expectation sums are kept in 4 LDS copies (lane & 3): fewer atomic collisions
// ... and so are the sums of one refresh window (Sweep::tracebackExpect): fp64 as well -- ds_add_f32 is the slow one on
// gfx950 (20 000 config-5 pairs: 95 ms with fp32 window sums in 2, 4 or 8 copies, 70 ms with fp64 ones in 2 or 4)
constexpr int kExpectWinCopies = 2;
// (a class whose events are formed inside the traceback adds to the kernel's emission sums once per window only: one copy)
__host__ __device__ constexpr int lds_expect_copies(bool inSweep) { return inSweep ? 1 : kExpectCopies; }
constexpr int kLdsWeights = 168;  // (emission + transition) sums, see Sweep::wt: 25*5 + 5*4 + 5*4 = 165, padded
__host__ __device__ constexpr int lds_header_doubles(int emit, bool inSweep = false) {
    return kLdsCubics + kLdsEm + kLdsWeights + (emit == CPECAN_EMIT_EXPECT ? lds_expect_copies(inSweep) * 80 : 0);
}
// doubles of LDS behind the rolling rows for the candidate staging rings (16-byte Candidates, 128 per list; none for
// the forward-only and expectation emitters)
constexpr int kStage = 128;  // LDS staging slots per candidate list (two waves' worth: flushed 64 at a time)
// ... of the absolute-position traceback: flushed whole at the end of a diagonal once half full -- a diagonal leaves two or
// three candidates --, and passed by when one group brings more than fit (Sweep::tracebackAbs)
constexpr int kStageAbs = 32;
__host__ __device__ constexpr int lds_stage_doubles(int emit, bool abs = false) {
    return emit == CPECAN_EMIT_MATCH ? 2 * (abs ? kStageAbs : kStage) : (emit == CPECAN_EMIT_INDEL ? 3 * 2 * kStage : 0);
}
constexpr int kPrefetch = 3;  // passes (of 64 cells) of F.match prefetched one diagonal ahead in the traceback
constexpr float kCandMargin = 3.0f;  // log-space slack of the candidate filter (see DESIGN.md "candidate filter")

// FAST: rolling diagonals and the two symbol strings live in LDS.  !FAST: both stay in global memory (bands wider
// than the LDS budget, or sequences too long for it); same arithmetic, workgroup-scope fences.
