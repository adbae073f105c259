// prints what v_mov_b32_dpp wave_shl:1 / wave_shr:1 deliver on this GPU (lane i <- lane i+1 / i-1?), including at row
// borders (lanes 15|16, 31|32) and with the source lane masked off by EXEC.  hipcc --offload-arch=gfx950 -o dpp_check tools/dpp_shift_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
    const int lane = threadIdx.x;
    int v = 100 + lane;
    o[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
    o[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
    if (lane % 5 != 0) {  // lanes 0, 5, 10, ... are disabled: what do their neighbours read?
        o[128 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
        o[192 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
    }
}
int main() {
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"wave_shl:1", "wave_shr:1", "wave_shl:1, lanes %5==0 off", "wave_shr:1, lanes %5==0 off"};
    for (int j = 0; j < 4; j++) {
        printf("%s:", names[j]);
        for (int i = 0; i < 64; i++) printf(" %d", h[64 * j + i]);
        printf("\n");
    }
    return 0;
}
