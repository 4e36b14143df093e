// accuracy of the branch-free tanh (float32 arithmetic emulated with float ops + fmaf), vs double tanh
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static float exp2_hw(float f) { return (float)exp2((double)f); }        // stand-in for v_exp_f32 (<= 1 ulp): here 0.5 ulp
static float rcp_hw(float d) { return (float)(1.0 / (double)d); }        // stand-in for v_rcp_f32
static float tanh_new(float x)
{
    const float a = fabsf(x);
    // small arguments: odd polynomial (Cephes tanhf range |x| < 0.625)
    const float z = a * a;
    float p = -5.70498872745e-3f;
    p = fmaf(p, z, 2.06390887954e-2f);
    p = fmaf(p, z, -5.37397155531e-2f);
    p = fmaf(p, z, 1.33314422036e-1f);
    p = fmaf(p, z, -3.33332819422e-1f);
    const float small = fmaf(p * z, a, a);
    // large arguments: 1 - 2/(exp(2a) + 1), exp through 2^f with a two-term log2(e)
    const float y = a + a;
    const float L2E_hi = 1.44269504088896340736f, L2E_lo = 1.92596299112661746e-8f;   // log2(e) = hi + lo
    const float t = y * L2E_hi;
    const float n = rintf(t);
    const float f = (fmaf(y, L2E_hi, -t) + y * L2E_lo) + (t - n);
    const float e = ldexpf(exp2_hw(f), (int)n);
    const float d = e + 1.0f;
    float r = rcp_hw(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    const float big = fmaf(-2.0f, r, 1.0f);
    const float res = a < 0.625f ? small : big;
    return copysignf(res, x);
}
static double ulp_err(float got, double want)
{
    float w = (float)want;
    int e; frexpf(w == 0 ? 1e-30f : w, &e);
    double ulp = ldexp(1.0, e - 24);
    return fabs((double)got - want) / ulp;
}
int main()
{
    double worst = 0, worst_libm = 0; float wx = 0;
    uint32_t lo, hi; float fl = 1e-6f, fh = 100.0f; memcpy(&lo, &fl, 4); memcpy(&hi, &fh, 4);
    long cnt = 0, over1 = 0;
    for (uint32_t u = lo; u < hi; u += 7) {
        float x; memcpy(&x, &u, 4);
        double want = tanh((double)x);
        double e = ulp_err(tanh_new(x), want);
        double e2 = ulp_err(tanhf(x), want);
        if (e > worst) { worst = e; wx = x; }
        if (e2 > worst_libm) worst_libm = e2;
        if (e > 1.0) ++over1;
        ++cnt;
    }
    printf("samples %ld: worst error %.3f ulp at x = %.9g (glibc tanhf worst %.3f ulp); > 1 ulp: %ld (%.4f %%)\n", cnt, worst, wx, worst_libm, over1, 100.0 * over1 / cnt);
    return 0;
}
