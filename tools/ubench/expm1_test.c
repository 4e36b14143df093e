// accuracy of the branch-free expm1 / exp of the rounding-faithful policy (float32 arithmetic, fmaf), vs double
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static float exp2_hw(float f) { return (float)exp2((double)f); }        // stand-in for v_exp_f32
static float exp_new(float x)
{
    const float L2E_HI = 1.44269504088896340736f, L2E_LO = 1.92596299112661746e-8f;
    x = fminf(fmaxf(x, -104.0f), 89.0f);
    const float t = x * L2E_HI;
    const float n = rintf(t);
    const float f = (fmaf(x, L2E_HI, -t) + x * L2E_LO) + (t - n);
    return ldexpf(exp2_hw(f), (int)n);
}
static float expm1_new(float x)
{
    // |x| < 0.35: Taylor to x^7 (truncation 2e-8 relative); otherwise exp(x) - 1
    float p = 1.0f / 5040.0f;
    p = fmaf(p, x, 1.0f / 720.0f);
    p = fmaf(p, x, 1.0f / 120.0f);
    p = fmaf(p, x, 1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, 0.5f);
    const float small = fmaf(p * x, x, x);
    const float big = exp_new(x) - 1.0f;
    return !(fabsf(x) >= 0.35f) ? small : big;
}
static double ulp_err(float got, double want)
{
    float w = (float)want;
    if (!isfinite(w) || w == 0) return got == w ? 0 : 1e9;
    int e; frexpf(w, &e);
    return fabs((double)got - want) / ldexp(1.0, e - 24);
}
int main()
{
    for (int sign = -1; sign <= 1; sign += 2) {
        double w1 = 0, w2 = 0, wl1 = 0, wl2 = 0; float x1 = 0, x2 = 0;
        uint32_t lo, hi; float fl = 1e-7f, fh = 88.0f; memcpy(&lo, &fl, 4); memcpy(&hi, &fh, 4);
        long cnt = 0;
        for (uint32_t u = lo; u < hi; u += 5) {
            float x; memcpy(&x, &u, 4); x *= sign;
            double e1 = ulp_err(expm1_new(x), expm1((double)x)), e2 = ulp_err(exp_new(x), exp((double)x));
            double l1 = ulp_err(expm1f(x), expm1((double)x)), l2 = ulp_err(expf(x), exp((double)x));
            if (e1 > w1) { w1 = e1; x1 = x; } if (e2 > w2) { w2 = e2; x2 = x; }
            if (l1 > wl1) wl1 = l1; if (l2 > wl2) wl2 = l2;
            ++cnt;
        }
        printf("sign %+d, %ld samples: expm1 worst %.3f ulp at %.9g (glibc %.3f) | exp worst %.3f ulp at %.9g (glibc %.3f)\n", sign, cnt, w1, x1, wl1, w2, x2, wl2);
    }
    return 0;
}
