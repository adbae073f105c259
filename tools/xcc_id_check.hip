// which XCD runs which workgroup: HW_REG_XCC_ID against blockIdx.x % 8 for a launch of 2016 single-wave workgroups
// (the one-launch form's size).  hipcc --offload-arch=gfx950 -o xcc_check tools/xcc_id_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
    const int id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;  // hwreg(HW_REG_XCC_ID, 0, 4)
    if (threadIdx.x == 0) o[blockIdx.x] = id;
}
int main() {
    const int n = 2016;
    int *d, h[n];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int hist[16] = {0}, agree = 0;
    for (int i = 0; i < n; i++) {
        hist[h[i] & 15]++;
        agree += h[i] == i % 8;
    }
    printf("workgroups per XCC id:");
    for (int i = 0; i < 16; i++) printf(" %d", hist[i]);
    printf("\nXCC id == blockIdx %% 8 for %d of %d; first 24 ids:", agree, n);
    for (int i = 0; i < 24; i++) printf(" %d", h[i]);
    printf("\n");
    return 0;
}
