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

// logAdd, impl/pairwiseAligner.c:287-307.  hi/lo form: with d = hi - lo the reference returns hi when
// lo == -inf or d >= 7.5, else lo + P(d).  d is +inf when only lo is -inf and NaN when both are, and
// both fail (d < 7.5), so one comparison covers the reference's two tests.  The cubic's coefficients are
// float literals in the reference, i.e. float32 values widened to double; Horner with separate mul/add.
// The four cubics live in a 128-byte LDS table [segment][c3,c2,c1,c0] read with two ds_read_b128: selecting
// four 64-bit coefficients with v_cndmask cost 24 VALU instructions per logAdd (47 % of the forward loop).
struct __attribute__((aligned(16))) Cubic {
    double c3, c2, c1, c0;
};

// The segment of d = hi - lo is found without fp64 compares.  For d >= 0 the IEEE bit pattern is monotone; the three
// thresholds (1.0, 2.5, 4.5) have a zero low dword and high dwords that are multiples of 2^17, so the segment is a
// function of the bucket  b = ((bits(d) + 2^49 - 1) >> 49) - (bits(1.0) >> 49), saturated at 0:
// b == 0 -> d <= 1;  1..10 -> (1, 2.5];  11..17 -> (2.5, 4.5];  18.. -> above, i.e. segment = number of set bits of
// {0, 10, 17} below position b.  Exact for every double, thresholds included.  NaN / +inf / d >= 8 may pick any
// segment: the result is `hi` then.  The table keeps 4 rows of 32 bytes (conflict-free for ds_read_b128); a 26-row
// table indexed by bucket measured 100x the LDS bank conflicts.
__device__ __forceinline__ void fill_cubics(double *t) {
    const float c[16] = {-0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                         -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                         -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                         -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f};
    const int l = threadIdx.x;
    if (l < 16) {
        float v = c[0];
#pragma unroll
        for (int i = 1; i < 16; i++) v = l == i ? c[i] : v;
        t[l] = (double)v;
    }
}

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
    const double d = hi - lo;
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

// exp(x) to a relative error of ~1e-7 for x <= ~1 (probabilities): 2^(x log2 e) with the integer part split off in
// double, the fraction through v_exp_f32, and the scaling by v_ldexp_f64 -- 8 instructions instead of the ~35 of the
// double-precision exp.  Only for the expectation sums, whose gate is 1e-5 relative (linear-space
// sums, order-insensitive at 1e-5); the posterior emitters keep the exact exp.
__device__ __forceinline__ double exp_1e7(double x) {
    // branch-free: -inf, NaN (an unreachable transition) and anything below 2^-1100 end as ldexp(.., -1100) == 0
    const double y = __builtin_fmax(x * 1.4426950408889634 /* log2(e) */, -1100.0);
    const double yi = __builtin_rint(y);
    const float yf = (float)(y - yi);         // in [-0.5, 0.5]
    return __builtin_ldexp((double)__builtin_amdgcn_exp2f(yf), (int)yi);
}

// N independent logAdds advanced in lock-step stages (compare/select -> table fetch -> Horner) so that the N LDS
// table fetches are in flight together instead of one fetch + wait per logAdd.  acc[i] = logAdd(acc[i], t[i]).
template <int N>
__device__ __forceinline__ void logadd_n(const Cubic *tab, double (&acc)[N], const double (&t)[N]) {
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

    __device__ __forceinline__ void load(int b) {
        base = b;
        int i = b + lane;
        i = i < 0 ? 0 : (i > last ? last : i);
        const int4 e = *reinterpret_cast<const int4 *>(table + i);
        eXmyL = e.x;
        eWidth = e.y;
        eRing = e.z;
        eCell = e.w;
        // consume the loaded registers here so the s_waitcnt for this load sits inside the (rare) refill branch;
        // otherwise hipcc puts a vmcnt(0) at the branch merge and every diagonal waits for its ring stores
        asm volatile("" ::"v"(eXmyL), "v"(eWidth), "v"(eRing), "v"(eCell));
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

#ifndef CPK_SWEEP_WAVES
#define CPK_SWEEP_WAVES 2  // waves per SIMD the sweep kernel's registers are allocated for
#endif
constexpr int kLdsCubics = 16;  // 4 rows x 4 coefficients
// doubles of LDS in front of the rolling rows: cubics + emissions (+ expectation sums for that emitter only)
constexpr int kExpectCopies = 4;  // emission-expectation sums are kept in 4 LDS copies (lane & 3): fewer atomic collisions
constexpr int kLdsWeights = 168;  // (emission + transition) sums, see Sweep::wt: 25*5 + 5*4 + 5*4 = 165, padded
__host__ __device__ constexpr int lds_header_doubles(int emit) {
    return kLdsCubics + 40 + kLdsWeights + (emit == CPECAN_EMIT_EXPECT ? kExpectCopies * 80 : 0);
}
// doubles of LDS behind the rolling rows for the candidate staging rings (16-byte Candidates, 128 per list; none for
// the forward-only and expectation emitters)
__host__ __device__ constexpr int lds_stage_doubles(int emit) {
    return emit == CPECAN_EMIT_MATCH ? 2 * 128 : (emit == CPECAN_EMIT_INDEL ? 3 * 2 * 128 : 0);
}
constexpr int kStage = 128;  // LDS staging slots per candidate list (two waves' worth: flushed 64 at a time)
constexpr int kPrefetch = 3;  // passes (of 64 cells) of F.match prefetched one diagonal ahead in the traceback
constexpr float kCandMargin = 3.0f;  // log-space slack of the candidate filter (see DESIGN.md "candidate filter")

// FAST: rolling diagonals and the two symbol strings live in LDS.  !FAST: both stay in global memory (bands wider
// than the LDS budget, or sequences too long for it); same arithmetic, workgroup-scope fences.
