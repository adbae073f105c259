// Diagnostic microbenchmark (not product): how a WIDE diagonal sweep scales when the waves of one workgroup share a
// region -- rolling rows in LDS, the 64-cell groups of a diagonal dealt out to the waves, one barrier per diagonal.
// The cell body has the forward sweep's shape: five states, neighbours from the two previous diagonals, eight logAdds
// (the kernel's cubic-table code).  Prints time per diagonal and cells/s for 1, 2, 4, 8 waves per workgroup with one
// workgroup per CU, and for as many workgroups per CU as the LDS allows.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/team_scale tools/team_scale.hip && tools/team_scale [W] [diagonals]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int S = 5;
struct __attribute__((aligned(16))) Cubic { double c3, c2, c1, c0; };
__device__ __forceinline__ int cubic_row(double d) {
    const int lo = __double2loint(d), hi = __double2hiint(d);
    const int h = hi - (lo == 0 ? 1 : 0);
    int b = (h >> 17) - ((0x3FF00000 >> 17) - 1);
    b = b < 0 ? 0 : (b > 31 ? 31 : b);
    const unsigned below = (1u << b) - 1u;
    return __builtin_popcount(below & ((1u << 0) | (1u << 10) | (1u << 17)));
}
__device__ __forceinline__ double logadd(const Cubic *tab, double a, double t) {
    const double hi = __builtin_fmax(a, t), lo = __builtin_fmin(a, t), d = hi - lo;
    const Cubic q = tab[cubic_row(d)];
    double r = q.c3 * d;
    r = r + q.c2;
    r = r * d;
    r = r + q.c1;
    r = r * d;
    r = r + q.c0;
    r = r + lo;
    return d < 7.5 ? r : hi;
}
// rows[3][S][W + 2]: diagonals d, d-1, d-2 rotate through three buffers (a team cannot overwrite in place)
__global__ void team_sweep(double *out, int W, int nDiag, int regionsPerGroup) {
    extern __shared__ double lds[];
    Cubic *tab = reinterpret_cast<Cubic *>(lds);
    double *rows = lds + 16;
    const int stride = W + 2, T = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    if (threadIdx.x < 4) tab[threadIdx.x] = Cubic{-0.01 * (threadIdx.x + 1), 0.1, 0.5, 0.69};
    double acc = 0;
    for (int rg = 0; rg < regionsPerGroup; rg++) {
        for (int i = threadIdx.x; i < 3 * S * stride; i += blockDim.x) rows[i] = -1.0 - 0.001 * (i % 97);
        __syncthreads();
        for (int d = 2; d < nDiag; d++) {
            const double *p1 = rows + ((d - 1) % 3) * S * stride, *p2 = rows + ((d - 2) % 3) * S * stride;
            double *cur = rows + (d % 3) * S * stride;
            for (int kb = wave * 64; kb < W; kb += 64 * T) {
                const int k = kb + lane < W ? kb + lane : W - 1;
                const double em = -0.3 - 0.01 * ((k + d) & 7);
                // middle block: the match state from all five states of (d-2, k)
                double m = p2[0 * stride + k + 1] + em;
                m = logadd(tab, m, p2[1 * stride + k + 1] + em - 0.1);
                m = logadd(tab, m, p2[2 * stride + k + 1] + em - 0.1);
                m = logadd(tab, m, p2[3 * stride + k + 1] + em - 0.2);
                m = logadd(tab, m, p2[4 * stride + k + 1] + em - 0.2);
                // lower / upper blocks: the gap states from (d-1, k) and (d-1, k+1)
                const double gx = logadd(tab, p1[0 * stride + k] - 2.3, p1[1 * stride + k] - 0.4);
                const double lx = logadd(tab, p1[0 * stride + k] - 4.6, p1[3 * stride + k] - 0.1);
                const double gy = logadd(tab, p1[0 * stride + k + 2] - 2.3, p1[2 * stride + k + 2] - 0.4);
                const double ly = logadd(tab, p1[0 * stride + k + 2] - 4.6, p1[4 * stride + k + 2] - 0.1);
                if (kb + lane < W) {
                    cur[0 * stride + k + 1] = m;
                    cur[1 * stride + k + 1] = gx;
                    cur[2 * stride + k + 1] = gy;
                    cur[3 * stride + k + 1] = lx;
                    cur[4 * stride + k + 1] = ly;
                }
                acc += m;
            }
            __syncthreads();
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1500, nDiag = argc > 2 ? atoi(argv[2]) : 3000;
    const size_t lds = sizeof(double) * (16 + 3 * S * (size_t)(W + 2));
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 512 * 64 * cus * 8);
    if (lds > 160 * 1024) {
        printf("a band of %d cells needs %zu KB of LDS with three rotating diagonals: more than a CU has\n", W, lds >> 10);
        return 1;
    }
    (void)hipFuncSetAttribute((const void *)team_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("band of %d cells, %d diagonals, %zu KB of LDS per region (three rotating diagonals x %d states)\n", W, nDiag, lds >> 10, S);
    for (int perCU = 1; perCU <= 2; perCU++) {
        const int groupsPerCU = perCU == 1 ? 1 : (int)((160 * 1024) / lds);
        if (perCU == 2 && groupsPerCU <= 1) break;
        for (int T = 1; T <= 16; T *= 2) {
            if (T * groupsPerCU > 32) break;
            const int blocks = cus * groupsPerCU;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(team_sweep, dim3(blocks), dim3(64 * T), lds, 0, out, W, 64, 1);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(team_sweep, dim3(blocks), dim3(64 * T), lds, 0, out, W, nDiag, 1);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double cells = (double)blocks * (double)W * (nDiag - 2);
            printf("%2d workgroup(s) per CU x %2d waves: %8.2f ms  %6.2f us per diagonal  %.3e cells/s (forward-like body only)\n",
                   groupsPerCU, T, ms, 1e3 * ms / (nDiag - 2), cells / (ms * 1e-3));
        }
    }
    return 0;
}
