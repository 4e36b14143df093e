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

__global__ void eval(const float *x, float *e, float *m, float *t, float *h, float *s1, float *s2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e[i] = exp_rf(x[i]);
    m[i] = expm1_rf(x[i]);
    t[i] = tanh_rf(x[i]);
    h[i] = __builtin_amdgcn_exp2f(x[i]);
    s1[i] = one_plus_tanh_rf(x[i]);              // the sum as the kernels form it since round 4
    s2[i] = 1.0f + tanh_rf(x[i]);                // ... and through the stand-alone tanh (before)
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
        float *d; hipMalloc(&d, 7 * (size_t)n * sizeof(float));
        hipMemcpy(d, xs.data(), n * sizeof(float), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(eval, dim3((n + 255) / 256), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n,
                           d + 5 * (size_t)n, d + 6 * (size_t)n, n);
        std::vector<float> r(6 * (size_t)n);
        if (hipMemcpy(r.data(), d + n, 6 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
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
        // 1 + tanh(x): the sum the reference forms (fenton.py:83,90).  Error against the true sum in units of 2^-24 (half an ulp
        // of a sum in [1, 2), one ulp of one in [0.5, 1)): what reaches the state is the ABSOLUTE error of the sum.  Also: how often
        // each form returns the correctly rounded sum RN(1 + tanh x), and how often the two forms agree bit for bit.
        double wa[2] = {0, 0}; float ata[2] = {0, 0}; long exact_rn[2] = {0, 0}, same = 0; double sum_abs[2] = {0, 0};
        for (int i = 0; i < n; ++i) {
            const double want = 1.0 + std::tanh((double)xs[i]);
            const float rn = (float)want;
            for (int f = 0; f < 2; ++f) {
                const float got = r[(size_t)(4 + f) * n + i];
                const double ea = std::fabs((double)got - want) / std::ldexp(1.0, -24);
                sum_abs[f] += ea;
                if (ea > wa[f]) { wa[f] = ea; ata[f] = xs[i]; }
                if (got == rn) ++exact_rn[f];
            }
            if (r[(size_t)4 * n + i] == r[(size_t)5 * n + i]) ++same;
        }
        const char *sn[2] = {"2 - 2/(e^2x + 1)", "1 + tanh_rf(x)"};
        for (int f = 0; f < 2; ++f)
            printf("sign %+d  1 + tanh as %-18s: worst absolute error %.3f x 2^-24 at x = %.9g, mean %.4f; equal to RN(1 + tanh x): %.3f %%\n",
                   sign, sn[f], wa[f], ata[f], sum_abs[f] / n, 100.0 * exact_rn[f] / n);
        printf("sign %+d  the two forms return the same float for %.3f %% of the arguments\n", sign, 100.0 * same / n);
        hipFree(d);
    }
    return 0;
}
