// Diagnostic (not product): the two forms of the work-queue ticket fetch of a persistent wave, compiled side by side so
// that their ISA can be compared (profiles/r02_ticket_fetch_isa.txt).  Form A is the natural one and is NOT used: with
// hipcc 7.2 -O3 the lane test is threaded across the loop back-edge (see the ISA); form B is what the kernels use
// (cpk_sweep.inl, cpk_team.inl, cpk_packed.inl).
#include <hip/hip_runtime.h>

__device__ __forceinline__ void work(double *out, int item, int lane) {
    double v = out[(size_t)item * 64 + lane];
    for (int i = 0; i < 8; i++) v = v * 1.0000001 + 0.5;
    out[(size_t)item * 64 + lane] = v;
}

// form A: one lane fetches, the ticket is broadcast
extern "C" __global__ void __launch_bounds__(64) ticket_lane0(unsigned int *queue, int count, double *out) {
    const int lane = threadIdx.x;
    for (;;) {
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(queue, 1u);
        const int tk = __builtin_amdgcn_readfirstlane((int)ticket);
        if (tk >= count) break;
        work(out, tk, lane);
    }
}

// form B: every lane takes part (lane 0 adds 1, the others 0)
extern "C" __global__ void __launch_bounds__(64) ticket_all_lanes(unsigned int *queue, int count, double *out) {
    const int lane = threadIdx.x;
    for (;;) {
        const unsigned int ticket = atomicAdd(queue, lane == 0 ? 1u : 0u);
        const int tk = __builtin_amdgcn_readfirstlane((int)ticket);
        if (tk >= count) break;
        work(out, tk, lane);
    }
}
