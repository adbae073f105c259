// Diagnostic microbenchmark (not product): SIMD cycles per logAdd for the kernel's own logadd_n<5> code path
// (LDS cubic table, bucket index), registers only otherwise.  Reports ns and cycles@2.36GHz per logAdd per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CPK_WAVE 64
struct __attribute__((aligned(16))) Cubic { double c3, c2, c1, c0; };
__device__ __forceinline__ int cubic_row(double d) {
    const int lo = __double2loint(d), hi = __double2hiint(d);
    const int h = hi - (lo == 0 ? 1 : 0);
    int b = (h >> 17) - ((0x3FF00000 >> 17) - 1);
    b = b < 0 ? 0 : (b > 31 ? 31 : b);
    const unsigned below = (1u << b) - 1u;
    return __builtin_popcount(below & ((1u << 0) | (1u << 10) | (1u << 17)));
}
template <int N>
__device__ __forceinline__ void logadd_n(const Cubic *tab, double (&acc)[N], const double (&t)[N]) {
    double hi[N], lo[N], d[N]; Cubic q[N];
#pragma unroll
    for (int i = 0; i < N; i++) { hi[i] = __builtin_fmax(acc[i], t[i]); lo[i] = __builtin_fmin(acc[i], t[i]); d[i] = hi[i] - lo[i]; }
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
// the shipped form (cpk_device_common.inl, CPK_LOGADD_EXACT=0): r = hi + Q(min(|x-y|, 8)), three FMAs, 5-row table
__device__ __forceinline__ int cubic_row_fast(double dc) {
    const unsigned hs = (unsigned)((unsigned long long)__double_as_longlong(dc) >> 32) >> 17;
    const unsigned b = __builtin_elementwise_sub_sat(hs, (0x3FF00000u >> 17) - 1u);
    return __builtin_popcount(__builtin_amdgcn_ubfe((1u << 0) | (1u << 10) | (1u << 17) | (1u << 23), 0u, b));
}
template <int N>
__device__ __forceinline__ void logadd_fast_n(const Cubic *tab, double (&acc)[N], const double (&t)[N]) {
    double hi[N], dc[N]; Cubic q[N];
#pragma unroll
    for (int i = 0; i < N; i++) { hi[i] = __builtin_fmax(acc[i], t[i]); dc[i] = __builtin_fmin(__builtin_fabs(acc[i] - t[i]), 8.0); }
#pragma unroll
    for (int i = 0; i < N; i++) q[i] = tab[cubic_row_fast(dc[i])];
    double r[N];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(q[i].c3, dc[i], q[i].c2);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(r[i], dc[i], q[i].c1);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_fma(r[i], dc[i], q[i].c0);
#pragma unroll
    for (int i = 0; i < N; i++) acc[i] = hi[i] + r[i];
}
template <int N, bool FASTV>
__global__ void __launch_bounds__(64) k(double *out, int iters) {
    __shared__ Cubic tab[5];
    if (threadIdx.x < 5) tab[threadIdx.x] = Cubic{-0.01 * (threadIdx.x + 1), 0.1, 0.5, 0.69};
    __syncthreads();
    double acc[N], t[N];
    for (int i = 0; i < N; i++) { acc[i] = -1.0 - i * 0.37 - threadIdx.x * 0.01; t[i] = -2.0 - i * 0.21 - (threadIdx.x & 7) * 0.9; }
    for (int it = 0; it < iters; it++) {
        if (FASTV) logadd_fast_n<N>(tab, acc, t);
        else logadd_n<N>(tab, acc, t);
#pragma unroll
        for (int i = 0; i < N; i++) { t[i] = t[i] - 0.001; acc[i] = acc[i] - 0.7; }
    }
    double s = 0; for (int i = 0; i < N; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int N, bool FASTV> void run(int wavesPerSimd) {
    const int blocks = 256 * 4 * wavesPerSimd, iters = 20000;
    double *d; hipMalloc(&d, (size_t)blocks * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<N, FASTV><<<blocks, 64>>>(d, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<N, FASTV><<<blocks, 64>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double perSimd = ms * 1e6 / ((double)iters * N * wavesPerSimd);
    printf("%s logadd_n<%d> waves/SIMD=%d: %.2f ns = %.1f cycles@2.36GHz per logAdd per SIMD (incl. 2 adds/logAdd of loop overhead)\n", FASTV ? "fast " : "exact", N, wavesPerSimd, perSimd, perSimd * 2.36);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 3, 4, 6, 8}) { run<1, false>(w); run<5, false>(w); run<1, true>(w); run<5, true>(w); }
    return 0;
}
