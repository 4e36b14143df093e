// Accuracy of the rounding-faithful policy's exp / expm1 / tanh (fib_tf_amd/csrc/models.hpp) on the device itself —
// v_exp_f32 and v_rcp_f32 are the real instructions here, not stand-ins — against double precision on the host.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tools/ubench/acc_rf.hip -o tools/ubench/acc_rf
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../fib_tf_amd/csrc/models.hpp"
using namespace fib;

__global__ void eval(const float *x, float *e, float *m, float *t, float *h, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e[i] = exp_rf(x[i]);
    m[i] = expm1_rf(x[i]);
    t[i] = tanh_rf(x[i]);
    h[i] = __builtin_amdgcn_exp2f(x[i]);
}
static double ulps(float got, double want)
{
    const float w = (float)want;
    if (!std::isfinite(w) || w == 0) return got == w ? 0 : 1e9;
    int ex; std::frexp(w, &ex);
    if (ex < -125) ex = -125;
    return std::fabs((double)got - want) / std::ldexp(1.0, ex - 24);
}
int main()
{
    struct Range { float lo, hi; } ranges[] = {{1e-7f, 87.0f}};
    for (int sign = -1; sign <= 1; sign += 2) {
        std::vector<float> xs;
        uint32_t lo, hi; std::memcpy(&lo, &ranges[0].lo, 4); std::memcpy(&hi, &ranges[0].hi, 4);
        for (uint32_t u = lo; u < hi; u += 11) { float x; std::memcpy(&x, &u, 4); xs.push_back(sign * x); }
        const int n = (int)xs.size();
        float *d; hipMalloc(&d, 5 * (size_t)n * sizeof(float));
        hipMemcpy(d, xs.data(), n * sizeof(float), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(eval, dim3((n + 255) / 256), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n, n);
        std::vector<float> r(4 * (size_t)n);
        if (hipMemcpy(r.data(), d + n, 4 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
        double w[4] = {0, 0, 0, 0}; float at[4] = {0, 0, 0, 0}; long over[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const double x = xs[i];
            const double want[4] = {std::exp(x), std::expm1(x), std::tanh(x), std::exp2(x)};
            for (int f = 0; f < 4; ++f) {
                const double e = ulps(r[(size_t)f * n + i], want[f]);
                if (e > w[f]) { w[f] = e; at[f] = xs[i]; }
                if (e > 1.0) ++over[f];
            }
        }
        const char *names[4] = {"exp_rf", "expm1_rf", "tanh_rf", "v_exp_f32 (2^x)"};
        for (int f = 0; f < 4; ++f)
            printf("sign %+d  %-16s %9d samples: worst %.3f ulp at x = %.9g; above 1 ulp: %.4f %%\n", sign, names[f], n, w[f], at[f], 100.0 * over[f] / n);
        hipFree(d);
    }
    return 0;
}
