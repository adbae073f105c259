// diagnostic: does v_min_f64 ... mul:2 (the output modifier) deliver 2 * min(|d|, c) on this GPU once MODE.IEEE is cleared
// with s_setreg, and what do the special values give?  (LLVM: "omod is ignored by hardware if IEEE bit is enabled".)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/omod_check tools/omod_check.hip && /tmp/omod_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(const double *x, const double *y, double *o, int n, int clearIeee) {
    if (clearIeee >= 1) __builtin_amdgcn_s_setreg((0 << 11) | (9 << 6) | 1, 0);  // hwreg(HW_REG_MODE, 9, 1) = 0: IEEE off
    if (clearIeee >= 2) __builtin_amdgcn_s_setreg((1 << 11) | (6 << 6) | 1, 0);  // hwreg(HW_REG_MODE, 6, 2) = 0: f64 / f16 denormals flushed
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i] - y[i];
    const double c = 7.75;
    double u;
    asm volatile("v_min_f64 %0, |%1|, %2 mul:2" : "=v"(u) : "v"(d), "v"(c));
    o[i] = u;
    o[n + i] = fmax(x[i], y[i]);  // what max gives for the specials under this mode
}

int main() {
    std::vector<double> x, y;
    const double inf = INFINITY, den = 4.9e-324;
    const double sp[][2] = {{0, 0}, {1, 1}, {-inf, -inf}, {-inf, 3}, {3, -inf}, {den, 0}, {1e-310, 0}, {7.5, 0}, {0, 7.75}, {0, 100}, {1e300, -1e300}, {-0.0, 0.0}};
    for (auto &p : sp) { x.push_back(p[0]); y.push_back(p[1]); }
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < 100000; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double a = -(double)(s % 1000003) / 64.0;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double b = a + ((double)(s % 2000001) - 1000000.0) / 65536.0;
        x.push_back(a); y.push_back(b);
    }
    const int n = (int)x.size();
    double *dx, *dy, *dout;
    hipMalloc(&dx, 8 * n); hipMalloc(&dy, 8 * n); hipMalloc(&dout, 16 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), 8 * n, hipMemcpyHostToDevice);
    for (int clear = 0; clear < 3; clear++) {
        std::vector<double> o(2 * n);
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, dout, n, clear);
        hipMemcpy(o.data(), dout, 16 * n, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; i++) {
            const double d = x[i] - y[i];
            double want = 2.0 * fmin(fabs(d), 7.75);
            if (std::isnan(d)) want = 15.5;
            if (memcmp(&want, &o[i], 8) != 0 && !(want == 0.0 && o[i] == 0.0)) bad++;
        }
        printf("MODE.IEEE %s: %ld of %d values differ from 2 * min(|x - y|, 7.75)\n", clear == 2 ? "cleared, f64 denormals flushed" : clear ? "cleared" : "as launched", bad, n);
        for (int i = 0; i < 12; i++) printf("   x %-10g y %-10g -> u %-10g  max %g\n", x[i], y[i], o[i], o[n + i]);
    }
    return 0;
}
