// Diagnostic microbenchmark (not part of the product): cycles per wave64 VALU instruction on gfx950 for the
// instruction kinds the sweep kernel is made of.  One workgroup of 256 threads = one wave per SIMD of one CU
// (or 512 threads = two waves per SIMD), 8 independent chains per lane so dependencies never stall issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP>
__global__ void k(double *out, int iters, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
    double b = seed * 0.5 + 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a[i]), "v"(b) : "vcc");
                if (OP == 5) {
                    int lo = __double2loint(a[i]);
                    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(lo) : );
                    a[i] = __hiloint2double(__double2hiint(a[i]), lo);
                }
                if (OP == 6) {
                    int lo = __double2loint(a[i]);
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(lo));
                    a[i] = __hiloint2double(__double2hiint(a[i]), lo);
                }
                if (OP == 7) {
                    float f = (float)i;
                    asm volatile("v_add_f32 %0, %0, %0" : "+v"(f));
                    a[i] += f * 0.0;
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[8192 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = (double)(t1 - t0);
}

template <int OP>
void run(const char *name, int threads, int blocks) {
    double *d;
    hipMalloc(&d, (4096 + 4096 * 64) * sizeof(double));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 200, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, iters, 1.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> h(8192);
    hipMemcpy(h.data(), d, 8192 * sizeof(double), hipMemcpyDeviceToHost);
    double ticks = h[4096];
    double instr = (double)iters * REP;
    const int wavesPerSimd = threads / 256;
    // wall-clock: each SIMD executed wavesPerSimd * instr wave-instructions in ms
    double nsPerSimdInstr = ms * 1e6 / (instr * wavesPerSimd);
    printf("%-14s waves/SIMD=%d blocks=%3d  wall: %.3f ns per SIMD wave-instr (= %.2f cycles @2.4GHz)   memtime ticks/instr/wave=%.2f  tick=%.3f ns\n",
           name, wavesPerSimd, blocks, nsPerSimdInstr, nsPerSimdInstr * 2.4, ticks / instr, ms * 1e6 / ticks);
    hipFree(d);
}

int main() {
    for (int blocks : {1, 256}) {
        for (int threads : {256, 512, 1024}) {
            run<0>("v_add_f64", threads, blocks);
            run<1>("v_mul_f64", threads, blocks);
            run<2>("v_fma_f64", threads, blocks);
            run<3>("v_max_f64", threads, blocks);
            run<4>("v_cmp_lt_f64", threads, blocks);
            run<6>("v_add_u32", threads, blocks);
        }
    }
    return 0;
}
