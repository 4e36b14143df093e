// models.hpp — per-cell kinetics of the three ionic models as gfx950 device code.
//
// Each model exposes
//     NVAR                       number of state arrays (variable 0 = potential)
//     struct Consts              per-launch scalars (kernarg -> SGPRs)
//     step<P, MODE>(s, Vc, lap, k, sub)
// which advances the NVAR values of ONE cell held in registers `s` by one sub-step, given
// Vc  = the boundary-enforced potential at the cell (ionic.py:107-113) and
// lap = laplace(V0) at the cell (ionic.py:44-60, incl. the phase-field term).
// MODE selects which variables are assigned (the dead arithmetic is eliminated at compile
// time): that is how Courtemanche's fast / slow assign groups (court.py:94-103) are built.
//
// P is the arithmetic policy:
//   Exact — one float32 rounding per reference op: IEEE division, tanh/exp/expm1 within ~1 ulp (tanh_rf, exp_rf,
//           expm1_rf below), ocml logf, no FMA contraction (the translation unit is built with -ffp-contract=off).  This
//           is the op order TensorFlow's CPU kernels execute for the reference graph.
//   Fast  — v_rcp_f32 / v_exp_f32 / v_log_f32 based forms (FIBHIP_FAST), a few ulp per op.
//
// The formulas follow fenton.py:46-108, br.py:125-332 and court.py:124-429 (cited per block).
#pragma once
#ifndef __HIPCC_RTC__            // hiprtc (in-process builds of traced models) brings its own runtime declarations
#include <hip/hip_runtime.h>
#endif

// Builds of this translation unit other than the stock library (a Beeler-Reuter table baked in, a traced model compiled
// in) put everything into an inline namespace named after the build: the same kernel then has a DIFFERENT symbol in every
// code object a process may hold at once (tools that key kernels by name — rocprofv3's kernel trace — met two
// `fib::copy_kernel` of two fat binaries in round 2 and crashed).
#ifdef FIB_BUILD_TAG
#define FIB_TAG_BEGIN inline namespace FIB_BUILD_TAG {
#define FIB_TAG_END }
#else
#define FIB_TAG_BEGIN
#define FIB_TAG_END
#endif

namespace fib {
FIB_TAG_BEGIN

#define FIB_DEV __device__ __forceinline__

// the two type traits the kernels need, spelled out: hiprtc has no <type_traits>
template <class A, class B> struct same_type { static constexpr bool value = false; };
template <class A> struct same_type<A, A> { static constexpr bool value = true; };
template <class...> using void_of = void;

// ---- R-wide value type --------------------------------------------------------------------------
// The SIMD issues a wave's next VALU instruction at full rate only if it does not depend on the one
// just issued (measured, tools/ubench/valu.hip: one dependent chain 2.1 ns per instruction per SIMD,
// two or more independent chains 1.2 ns, at 4 waves per SIMD).  A lane of the strip kernel owns R
// independent cells, so the kinetics are written once over `vf<R>`: every source line expands to R
// adjacent independent instructions (operation-major order) instead of R long dependent chains.
template <int R>
struct vf {
    float v[R];
};
template <int R>
struct vm {
    bool v[R];
};
#define FIB_VEC_BINOP(OP)                                                                    \
    template <int R> FIB_DEV vf<R> operator OP(const vf<R> &a, const vf<R> &b)               \
    { vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] OP b.v[r]; return o; }   \
    template <int R> FIB_DEV vf<R> operator OP(const vf<R> &a, float b)                       \
    { vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] OP b; return o; }        \
    template <int R> FIB_DEV vf<R> operator OP(float a, const vf<R> &b)                       \
    { vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a OP b.v[r]; return o; }
FIB_VEC_BINOP(+)
FIB_VEC_BINOP(-)
FIB_VEC_BINOP(*)
template <int R> FIB_DEV vf<R> operator-(const vf<R> &a)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = -a.v[r]; return o; }
template <int R> FIB_DEV vm<R> vgt(const vf<R> &a, float b)
{ vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] > b; return o; }
template <int R> FIB_DEV vf<R> vsel(const vm<R> &m, const vf<R> &a, const vf<R> &b)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = m.v[r] ? a.v[r] : b.v[r]; return o; }
template <int R> FIB_DEV vf<R> vfma(const vf<R> &a, float b, const vf<R> &c)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = __builtin_fmaf(a.v[r], b, c.v[r]); return o; }
template <int R> FIB_DEV vf<R> vfma(const vf<R> &a, const vf<R> &b, const vf<R> &c)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = __builtin_fmaf(a.v[r], b.v[r], c.v[r]); return o; }
template <int R> FIB_DEV vf<R> vfma(const vf<R> &a, float b, float c)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = __builtin_fmaf(a.v[r], b, c); return o; }
template <int R, class F> FIB_DEV vf<R> vmap(const vf<R> &a, F f)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = f(a.v[r]); return o; }
template <int R, class F> FIB_DEV vf<R> vzip(const vf<R> &a, const vf<R> &b, F f)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = f(a.v[r], b.v[r]); return o; }
template <int R, class F> FIB_DEV vf<R> vzip(const vf<R> &a, float b, F f)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = f(a.v[r], b); return o; }
template <int R, class F> FIB_DEV vf<R> vzip(float a, const vf<R> &b, F f)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = f(a, b.v[r]); return o; }
#define FIB_VEC_CMP(NAME, OP)                                                                \
    template <int R> FIB_DEV vm<R> NAME(const vf<R> &a, const vf<R> &b)                      \
    { vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] OP b.v[r]; return o; }   \
    template <int R> FIB_DEV vm<R> NAME(const vf<R> &a, float b)                              \
    { vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] OP b; return o; }        \
    template <int R> FIB_DEV vm<R> NAME(float a, const vf<R> &b)                              \
    { vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a OP b.v[r]; return o; }        \
    FIB_DEV bool NAME(float a, float b) { return a OP b; }
FIB_VEC_CMP(vcmp_gt, >)
FIB_VEC_CMP(vcmp_ge, >=)
FIB_VEC_CMP(vcmp_lt, <)
FIB_VEC_CMP(vcmp_le, <=)
FIB_VEC_CMP(vcmp_eq, ==)
FIB_VEC_CMP(vcmp_ne, !=)
template <int R> FIB_DEV vm<R> vm_and(const vm<R> &a, const vm<R> &b)
{ vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] && b.v[r]; return o; }
template <int R> FIB_DEV vm<R> vm_or(const vm<R> &a, const vm<R> &b)
{ vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = a.v[r] || b.v[r]; return o; }
template <int R> FIB_DEV vm<R> vm_not(const vm<R> &a)
{ vm<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = !a.v[r]; return o; }
FIB_DEV bool vm_and(bool a, bool b) { return a && b; }
FIB_DEV bool vm_or(bool a, bool b) { return a || b; }
FIB_DEV bool vm_not(bool a) { return !a; }
template <class T> struct bcast;
template <> struct bcast<float> { static FIB_DEV float of(float x) { return x; } };
template <int R> struct bcast<vf<R>> { static FIB_DEV vf<R> of(float x) { vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = x; return o; } };
template <class T> FIB_DEV T T_of(float x) { return bcast<T>::of(x); }
// clamp(a*b + c, 0, 1) in one instruction (v_fma_f32 ... clamp)
template <int R> FIB_DEV vf<R> vfma_sat(const vf<R> &a, float b, float c)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = __builtin_amdgcn_fmed3f(__builtin_fmaf(a.v[r], b, c), 0.0f, 1.0f); return o; }
FIB_DEV float vfma_sat(float a, float b, float c) { return __builtin_amdgcn_fmed3f(__builtin_fmaf(a, b, c), 0.0f, 1.0f); }
// scalar spellings of the same helpers, so one body serves float and vf<R>
FIB_DEV bool vgt(float a, float b) { return a > b; }
FIB_DEV float vsel(bool m, float a, float b) { return m ? a : b; }
FIB_DEV float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <class F> FIB_DEV float vmap(float a, F f) { return f(a); }
template <class F> FIB_DEV float vzip(float a, float b, F f) { return f(a, b); }

// Division by a value whose correctly rounded reciprocal rc = RN(1/c) is known (a compile-time
// constant, or the per-cell 1/(4ϕ) prepared once): q = RN(a*rc); r = a - q*c (exact, one FMA);
// RN(q + r*rc) is the correctly rounded quotient RN(a/c) (Markstein's theorem) — bit-identical to
// IEEE division at 3 instructions instead of the ~11 of the generic expansion.  Checked exhaustively
// over all 2^23 significands for every constant of the three models (tools/ubench/divtest.c);
// it can differ only when a/c is subnormal (by at most one subnormal ulp).
// exp and expm1 for the rounding-faithful policy, branch-free.
//   exp:   v_exp_f32 on x*log2(e) and ONE correction step for the rounding of that product (log2(e) in two terms):
//          7 instructions against ocml's 18; <= 1 ulp + the instruction's own error (measured on the device,
//          tools/ubench/acc_rf.hip).  Results below 2^-126 flush to zero.
//   expm1: no transcendental instruction at all: x = n ln2 + r, expm1(x) = 2^n expm1(r) + (2^n - 1) with the degree-7
//          Taylor form on |r| <= 0.35; 17 instructions against ocml's 31, <= 0.9 ulp for x < 0 — the only sign a positive
//          time constant gives rush_larsen — and <= 1.7 ulp for x > 0.
static FIB_DEV float exp_core(float x)                              // x <= 88.72 (e must stay finite)
{
    constexpr float L2E_HI = 1.44269504088896340736f, L2E_LO = 1.92596299112661746e-8f, LN2 = 0.693147180559945f;
    const float t = x * L2E_HI;
    const float lo = __builtin_fmaf(x, L2E_LO, __builtin_fmaf(x, L2E_HI, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, lo * LN2, e);
}
static FIB_DEV float exp_rf(float x) { return exp_core(__builtin_fminf(x, 88.72f)); }   // exp(88.72) is the last finite value
static FIB_DEV float expm1_rf(float x)
{
    constexpr float L2E = 1.44269504088896340736f, LN2_HI = 0.693147180559945f, LN2_LO = -1.90465429995776804525e-9f;
    x = __builtin_fminf(__builtin_fmaxf(x, -104.0f), 88.0f);         // 2^n stays finite (and -inf stays out of n*ln2)
    const float n = __builtin_rintf(x * L2E);
    const float r = __builtin_fmaf(n, -LN2_LO, __builtin_fmaf(n, -LN2_HI, x));
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    const float q = __builtin_fmaf(p * r, r, r);
    const float s = __builtin_ldexpf(1.0f, (int)n);
    return __builtin_fmaf(s, q, s - 1.0f);
}
// tanh for the rounding-faithful policy: branch-free, <= 1.5 ulp of the true tanh over the whole float32 range,
// measured on the device (tools/ubench/acc_rf.hip: 22 M arguments per sign; glibc's tanhf: 2.2 ulp, the float32 tanh
// TensorFlow's CPU kernels use is a rational approximation of a few ulp too) at ~23 instructions and two
// transcendental issues, where ocml's tanhf runs both of its divergent branches at ~4x that.
//   |x| < 0.625:  x + x*z*P(z), z = x^2              (Cephes tanhf's polynomial)
//   otherwise:    1 - 2/(e + 1), e = exp(2|x|) by exp_core above, the quotient by v_rcp_f32 + one Newton step
static FIB_DEV float tanh_rf(float x)
{
    const float a = __builtin_fabsf(x);
    const float z = a * a;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    const float small = __builtin_fmaf(p * z, a, a);
    const float y = 2.0f * __builtin_fminf(a, 10.0f);                 // tanh(10) rounds to 1; keeps e finite
    const float e = exp_core(y);
    const float d = e + 1.0f;
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    const float big = __builtin_fmaf(-2.0f, r, 1.0f);
    const float res = !(a >= 0.625f) ? small : big;                  // (a NaN takes the polynomial and stays a NaN)
    return __builtin_copysignf(res, x);
}

// 1 + tanh(x) for the rounding-faithful policy, where the reference forms exactly that sum (fenton.py:83 `1 + tanh(..)`, :90
// `0.5*(1 + tanh(..))`): 2 - 2/(e^{2x} + 1), e by exp_core, the quotient by v_rcp_f32 + one Newton step, the last step ONE fused
// multiply-add.  What the sum needs of tanh is ABSOLUTE accuracy — it lives in [0, 2], and RN(1 + t) throws away whatever
// relative accuracy a small t had — so tanh's polynomial branch for small arguments (there for the RELATIVE accuracy of tanh
// alone) and the select between the two branches are dead weight here: 13 instructions instead of 25.  Measured on the device
// against double precision (tools/ubench/acc_rf.hip, 22 M arguments per sign): see profiles/r04_accuracy_one_plus_tanh.txt —
// the error of the sum is within that of 1.0f + tanh_rf(x), whose tanh is itself within 1.5 ulp.
static FIB_DEV float one_plus_tanh_rf(float x)
{
    const float y = 2.0f * __builtin_amdgcn_fmed3f(x, -50.0f, 10.0f);   // 1 + tanh rounds to 0 / 2 beyond; keeps e finite and > 0 paths exact
    const float e = exp_core(y);
    const float d = e + 1.0f;
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_fmaf(-2.0f, r, 2.0f);
}

// Fenton, fewer instructions per cell (the kernels' sub-steps cost what their arithmetic costs).  Fast policy: bit 0 =
// U*G(b) as U - U*H(b) (exact: one multiply-add for a Heaviside and a product), bit 1 = the second sigmoid itself instead of one minus
// its complement (52 -> 50 per cell).  Rounding-faithful policy, bit 4: where one factor of a*b + c is 0, 0.5 or 1 (the Heavisides) or
// exactly 0.5, the product is EXACT and the reference's two roundings are the one rounding of a fused multiply-add — the same bits in
// one instruction: H*tau_a + x, H*(r_sp - r_sn) + r_sn, x*0.5 - S, and (U - u_0)*G as U - U*H (kernel 18.3 -> 17.3 us per tick).
// 0 = the forms of rounds 2-3, kept for same-box A/B (tools/r04_l.sh, r04_q.sh).
#ifndef FIB_FENTON_FEWER
#define FIB_FENTON_FEWER 19
#endif
// Beeler-Reuter, fast policy (br_step.inc): 1 = every a*b+c of the currents that the reference leaves as two operations as ONE
// multiply-add and log C as the bare v_log_f32 (C is never subnormal); 2 = constant factors folded into neighbouring multiply-adds.
// 222 -> 202 instructions per cell; the error along the golden trajectories is the one of the old forms (tools/r04_p.sh).  0 = the
// forms of rounds 1-3, kept for same-box A/B.  (Measured and NOT taken: the twelve sums by Horner's rule, x as one multiply-add —
// 13 instructions fewer, but the ascending sums are what the reference computes and its rounding errors are what parity is
// measured against: 4-10 x the distance from the golden trajectories)
#ifndef FIB_BR_FEWER
#define FIB_BR_FEWER 2
#endif
struct Exact {
    // a*b + c: the reference rounds the product and the sum separately
    template <class T, class B, class C>
    static FIB_DEV auto mad(const T &a, const B &b, const C &c) { return a * b + c; }
    template <class T> static FIB_DEV T mad3(const T &a, const T &b, const T &c) { return a * b + c; }
    template <class T>
    static FIB_DEV T divc(const T &a, float c, float rc)
    {
        const T q = a * rc;
        const T r = vfma(-q, c, a);
        return vfma(r, rc, q);
    }
    template <class T>
    static FIB_DEV T tanhv(const T &a) { return vmap(a, [](float x) { return tanh_rf(x); }); }
    // 1 + tanh(a) and 0.5*(1 + tanh(a)) - s, as the reference writes them (fenton.py:83,90)
#ifdef FIB_EXACT_TANH_SUM_VIA_TANH                 // (the form before round 4: the sum through the stand-alone tanh; A/B builds only)
    template <class T>
    static FIB_DEV T one_plus_tanh(const T &a) { return 1.0f + tanhv(a); }
#else
    template <class T>
    static FIB_DEV T one_plus_tanh(const T &a) { return vmap(a, [](float x) { return one_plus_tanh_rf(x); }); }
#endif
    template <class T>
    static FIB_DEV T half_one_plus_tanh_minus(const T &a, const T &s)
    {
#if FIB_FENTON_FEWER & 16      // (halving is exact: the two roundings of x * 0.5 + (-s) are the one rounding of a fused multiply-add)
        return vfma(one_plus_tanh(a), 0.5f, -s);
#else
        return one_plus_tanh(a) * 0.5f + (-s);
#endif
    }
    template <class A, class T>
    static FIB_DEV T div(const A &a, const T &b) { return vzip(a, b, [](float x, float y) { return x / y; }); }
    template <class T> static FIB_DEV T rcp(const T &a) { return vmap(a, [](float x) { return 1.0f / x; }); }
    template <class T> static FIB_DEV T exp(const T &a) { return vmap(a, [](float x) { return exp_rf(x); }); }
    template <class T> static FIB_DEV T expm1(const T &a) { return vmap(a, [](float x) { return expm1_rf(x); }); }
    template <class T> static FIB_DEV T log(const T &a) { return vmap(a, [](float x) { return logf(x); }); }
    // generated code (traced models): division by an arbitrary constant stays a true IEEE division here
    template <class T> static FIB_DEV T divk(const T &a, float c) { return vmap(a, [c](float x) { return x / c; }); }
    template <class T> static FIB_DEV T expm1g(const T &a) { return expm1(a); }
    static FIB_DEV float tanh(float a) { return tanh_rf(a); }   // == tanhv<float>
    static FIB_DEV float sqrt(float a) { return sqrtf(a); }
};

struct Fast {
    // a*b + c as ONE fused multiply-add, written explicitly (never left to -ffp-contract): every kernel
    // variant then rounds identically, so fusion depth / tile shape never change a bit of the result
    template <class T, class C>
    static FIB_DEV T mad(const T &a, float b, const C &c) { return vfma(a, b, c); }
    template <class T> static FIB_DEV T mad3(const T &a, const T &b, const T &c) { return vfma(a, b, c); }
    template <class T>
    static FIB_DEV T divc(const T &a, float, float rc) { return a * rc; }
    template <class T>
    static FIB_DEV T tanhv(const T &a)
    {   // 1 - 2/(exp(2a)+1), operation-major over the R cells
        const T t = a * (2.0f * 1.44269504088896340736f);
        const T e = vmap(t, [](float x) { return __builtin_amdgcn_exp2f(x); });
        const T d = e + 1.0f;
        const T q = vmap(d, [](float x) { return __builtin_amdgcn_rcpf(x); });
        return 1.0f - 2.0f * q;
    }
    // q = 1/(exp(2a)+1):  1 + tanh = 2 - 2q,  0.5*(1 + tanh) - s = (1 - q) - s   (one instruction fewer each)
    template <class T>
    static FIB_DEV T sigm_q(const T &a)
    {
        const T t = a * (2.0f * 1.44269504088896340736f);
        const T e = vmap(t, [](float x) { return __builtin_amdgcn_exp2f(x); });
        return vmap(e + 1.0f, [](float x) { return __builtin_amdgcn_rcpf(x); });
    }
    // q = 1/(2^x + 1) for an exponent already scaled to base 2 (the caller folds 2*log2(e) and any affine map of its
    // argument into ONE multiply-add)
    template <class T>
    static FIB_DEV T sigm_q2(const T &x)
    {
#ifdef FIB_DIAG_NOTRANS                     // (diagnostic builds only: what do the transcendental instructions cost?)
        return x * 0.001f + 0.5f;
#else
#if defined(FIB_DIAG_NOEXP)                 // (diagnostic builds: the two transcendentals priced separately)
        const T e = x * 0.001f;
#else
        const T e = vmap(x, [](float y) { return __builtin_amdgcn_exp2f(y); });
#endif
#if defined(FIB_DIAG_NORCP)
        return (e + 1.0f) * 0.37f;
#else
        return vmap(e + 1.0f, [](float y) { return __builtin_amdgcn_rcpf(y); });
#endif
#endif
    }
    template <class T>
    static FIB_DEV T one_plus_tanh(const T &a) { return vfma(sigm_q(a), -2.0f, 2.0f); }
    template <class T>
    static FIB_DEV T half_one_plus_tanh_minus(const T &a, const T &s) { return (1.0f - sigm_q(a)) - s; }
    template <class T> static FIB_DEV T rcp(const T &a) { return vmap(a, [](float x) { return __builtin_amdgcn_rcpf(x); }); }
    template <class A, class T>
    static FIB_DEV T div(const A &a, const T &b) { return a * rcp(b); }
    template <class T> static FIB_DEV T exp(const T &a) { return vmap(a, [](float x) { return __expf(x); }); }
    // expm1 for the Rush-Larsen factor: argument is -dt/tau.  exp(x)-1 keeps an absolute error of
    // ~1 ulp(1) = 6e-8, which is below the float32 resolution of the gate value it multiplies into.
    template <class T> static FIB_DEV T expm1(const T &a) { return exp(a) - 1.0f; }
    template <class T> static FIB_DEV T log(const T &a) { return vmap(a, [](float x) { return __logf(x); }); }
    template <class T> static FIB_DEV T divk(const T &a, float c) { return a * (1.0f / c); }
    // expm1 for arbitrary user expressions: exp(x)-1 loses everything for |x| << 1, so small arguments take
    // the cubic Taylor form (relative error < 1e-7 below 0.03)
    template <class T> static FIB_DEV T expm1g(const T &a)
    {
        return vmap(a, [](float x) {
            const float big = __expf(x) - 1.0f;
            const float small = x * __builtin_fmaf(x, __builtin_fmaf(x, 0.16666667f, 0.5f), 1.0f);
            return fabsf(x) < 0.03f ? small : big;
        });
    }
    static FIB_DEV float tanh(float a) { return tanhv<float>(a); }
    static FIB_DEV float sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
};

static FIB_DEV float sgnf(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
static FIB_DEV float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
template <int R> static FIB_DEV vf<R> clipf(const vf<R> &x, float lo, float hi)
{ vf<R> o; _Pragma("unroll") for (int r = 0; r < R; ++r) o.v[r] = fminf(fmaxf(x.v[r], lo), hi); return o; }

// pointwise ops of the traced-model code generator (fib_tf_amd/traced.py), over float and vf<R>
// (1 + sign(x)) * 0.5 and (1 - sign(x)) * 0.5 for ANY finite or infinite x, subnormals included: two scalings by
// 2^120 take every nonzero |x| past 1, the clamp does the rest (2 instructions instead of compare/select/add/mul)
template <class T> static FIB_DEV T g_heav(const T &x) { const T y = x * 0x1p120f; return vfma_sat(y, 0x1p120f, 0.5f); }
template <class T> static FIB_DEV T g_heav_not(const T &x) { const T y = x * 0x1p120f; return vfma_sat(y, -0x1p120f, 0.5f); }
template <class T> static FIB_DEV T g_sign(const T &a) { return vmap(a, [](float x) { return sgnf(x); }); }
template <class T> static FIB_DEV T g_abs(const T &a) { return vmap(a, [](float x) { return fabsf(x); }); }
template <class A, class B> static FIB_DEV auto g_max(const A &a, const B &b)
{ return vzip(a, b, [](float x, float y) { return fmaxf(x, y); }); }
template <class A, class B> static FIB_DEV auto g_min(const A &a, const B &b)
{ return vzip(a, b, [](float x, float y) { return fminf(x, y); }); }
template <class T> static FIB_DEV T g_pow(const T &a, float e) { return vmap(a, [e](float x) { return powf(x, e); }); }
template <class P, class T> static FIB_DEV T g_sqrt(const T &a) { return vmap(a, [](float x) { return P::sqrt(x); }); }

// (1 + sign(x)) * 0.5 and (1 - sign(x)) * 0.5 (fenton.py:73-79): the three values {0, 0.5, 1}, produced as
// clamp(0.5 +- x * 2^27, 0, 1) — one instruction, no compare (compares write SGPRs and issue at ~60 %
// of the plain-VALU rate, tools/ubench/valu2.hip).  Exact for every argument this model forms: x is
// U - 0.23 or U - 0.3, an exact float difference whose nonzero magnitude is at least one ulp of the
// threshold (2^-26), so |x| * 2^27 >= 2 whenever x != 0.
template <class T> static FIB_DEV T heav(const T &x) { return vfma_sat(x, 134217728.0f, 0.5f); }
template <class T> static FIB_DEV T heav_not(const T &x) { return vfma_sat(x, -134217728.0f, 0.5f); }

// rush_larsen, ionic.py:115-123.  mdt = float(-dt)
template <class P, class T>
static FIB_DEV T rush_larsen(const T &g, const T &ginf, const T &tau, float mdt)
{
    if constexpr (same_type<P, Fast>::value) {
        // g + (g - g_inf) (e - 1) = g_inf + (g - g_inf) e with e = exp(-dt/tau) = 2^(rcp(tau) * (-dt log2 e)): one multiply and
        // one add fewer per gate (exp(x) - 1 carried an absolute error of an ulp of 1 already)
        const T e = vmap(P::rcp(tau) * (mdt * 1.44269504088896340736f), [](float x) { return __builtin_amdgcn_exp2f(x); });
        return clipf(vfma(g - ginf, e, ginf), 0.00001f, 0.99999f);
    } else {
        return clipf(P::mad3(g - ginf, P::expm1(P::div(mdt, tau)), g), 0.00001f, 0.99999f);
    }
}
// tau is a Python constant: expm1(float(-dt/tau)) is formed once on the host
static FIB_DEV float rush_larsen_c(float g, float ginf, float em1)
{
    return clipf(g + (g - ginf) * em1, 0.00001f, 0.99999f);
}

#define FC(x) ((float)(x))
// x / (float constant c), through the constant's correctly rounded reciprocal
#define DC(x, c) P::divc((x), FC(c), 1.0f / FC(c))

// =====================================================================================
// Fenton 4v  (fenton.py:46-108)
// =====================================================================================
struct Fenton {
    static constexpr int NVAR = 4;
    static constexpr int DEFAULT_STEPS = 10;      // fenton.py:135-138
    enum { MODE_ALL = 0 };
    struct Consts {
        float dt;      // float(self.dt)
        float ddt;     // float(self.diff * self.dt)   (product formed in double, fenton.py:103)
        // Fast policy only: the explicit-Euler updates of the two gates with dt folded in on the host (double, rounded
        // once):  V1 = V*(1 - dt/tau_vp) | V*(1 - dt/tau_vn) + dt/tau_vn,  likewise W (fenton.py:87-88,105-106)
        float cvp, cvn, dvn, cwp, cwn, dwn;
    };
    static constexpr unsigned mask(int) { return 0xFu; }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float U0, float lap, const Consts &k, int)
    {
        float dU;
        pre<P, float>(s, dU, k);
        post<P, float>(s, dU, U0, lap, k);
    }
    // the R cells of a lane at once (strip kernel): same arithmetic per cell, operation-major order
    static constexpr bool HAS_VEC = true;
    // The update splits into a part that needs neither the Laplacian nor the enforced potential (the whole reaction
    // term: it reads the raw U, fenton.py:101) and the potential's own line (fenton.py:103).  (Running the first part
    // while the stencil window's LDS reads are in flight was tried and bought nothing: with four waves per SIMD that
    // latency is covered already)
    template <class P, int MODE, int R>
    static FIB_DEV void stepN_pre(float (&s)[R][NVAR], float (&dU)[R], const Consts &k)
    {
        vf<R> t[NVAR], d;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < NVAR; ++v) t[v].v[r] = s[r][v];
        pre<P, vf<R>>(t, d, k);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dU[r] = d.v[r];
#pragma unroll
            for (int v = 1; v < NVAR; ++v) s[r][v] = t[v].v[r];
        }
    }
    template <class P, int MODE, int R>
    static FIB_DEV void stepN_post(float (&s)[R][NVAR], const float (&dU)[R], const float (&U0)[R], const float (&lap)[R],
                                   const Consts &k)
    {
#pragma unroll
        for (int r = 0; r < R; ++r) post<P, float>(s[r], dU[r], U0[r], lap[r], k);
    }
    template <class P, int MODE, int R>
    static FIB_DEV void stepN(float (&s)[R][NVAR], const float (&U0)[R], const float (&lap)[R], const Consts &k, int)
    {
        float dU[R];
        stepN_pre<P, MODE, R>(s, dU, k);
        stepN_post<P, MODE, R>(s, dU, U0, lap, k);
    }
    // keep the per-step scalars in VGPRs: a VALU op with an SGPR source issues at ~60 % of the rate
    // of an all-VGPR one (tools/ubench/valu2.hip)
    static FIB_DEV Consts pinned(const Consts &k)
    {
        Consts c = k;
        asm volatile("" : "+v"(c.dt), "+v"(c.ddt), "+v"(c.cvp), "+v"(c.cvn), "+v"(c.dvn), "+v"(c.cwp), "+v"(c.cwn), "+v"(c.dwn));
        return c;
    }
    // the same with ONE scalar left where the compiler puts it (an SGPR: its one use per cell and sub-step issues a little
    // slower): for the kernel that would otherwise exceed its register budget by exactly one — the multi-tick kernel under
    // the rounding-faithful policy, which spilled a register to scratch memory (kernels.hpp strip_body)
    static FIB_DEV Consts pinned_spare(const Consts &k)
    {
        Consts c = k;
        asm volatile("" : "+v"(c.dt), "+v"(c.cvp), "+v"(c.cvn), "+v"(c.dvn), "+v"(c.cwp), "+v"(c.cwn), "+v"(c.dwn));
        return c;
    }
    // everything but the potential: s[1..3] advance, dU_out = the reaction term of the potential
    template <class P, class T>
    static FIB_DEV void pre(T (&s)[NVAR], T &dU_out, const Consts &k)
    {
#include "fenton_step.inc"
    }
    // U1 = U0 + dt*dU + (diff*dt)*lap(U0), fenton.py:103 (P::mad: two roundings under Exact, one FMA under Fast)
    template <class P, class T>
    static FIB_DEV void post(T (&s)[NVAR], const T &dU, const T &U0, const T &lap, const Consts &k)
    {
        s[0] = P::mad(lap, k.ddt, P::mad(dU, k.dt, U0));
    }
};

// the same kinetics behind fenton_simple.py's zero-padded convolution Laplacian (kernels.hpp ZeroPadOf)
struct FentonZP : Fenton {
    static constexpr bool ZEROPAD = true;
};

// =====================================================================================
// Beeler-Reuter  (br.py:125-332)
// =====================================================================================
#ifdef FIB_BR_TABLE_INC
#include FIB_BR_TABLE_INC       // static constexpr float FIB_BR_CHEB[108]: fib_tf_amd/br.py's table, baked
#endif
struct BeelerReuter {
    static constexpr int NVAR = 8;                // V C M H J D F XI, br.py:87-94
    static constexpr int DEFAULT_STEPS = 5;       // br.py:98-107
    template <class C> static FIB_DEV const C &pinned(const C &k) { return k; }
    enum { MODE_DIRECT = 0, MODE_CHEBY = 1 };
    struct Consts {
        float dt, ddt;
        float mdt;          // float(-dt)
        float mdt_skip;     // float(-(dt*5)): slow gates on sub-step 0 when skip (br.py:99,195)
        int skip;           // config['skip']
        float cheb[12 * 9]; // br.py:327 coefficients, row order as fibhip.h documents
    };
    static constexpr unsigned mask(int) { return 0xFFu; }

    // calc_alpha_bata_tf, br.py:255-264, with the row of ab_coef (br.py:49-62) as template constants.
    // The table is float32; the d/f rows are pre-multiplied by 2 in double first.
    template <class P, class T>
    static FIB_DEV T ab(const T &v, float c0, float c1, float c2, float c3, float c4, float c5, float c6)
    {
        const T e1 = P::exp(c1 * (v + c2));
        const T den = P::exp(c5 * (v + c2)) + c6;
        if (c3 == 0.0f) return P::div(c0 * e1, den);
        return P::div(c0 * e1 + c3 * (v + c4), den);
    }
    template <class P, class T>
    static FIB_DEV void inf_tau(const T &a, const T &b, T &inf, T &tau)
    {   // calc_inf_tau, br.py:266-273
        inf = P::div(a, a + b);
        tau = P::div(1.0f, a + b);
    }
    // expand_chebyshev device part, br.py:329-331:  r = d0; r += d_i * S_i  (i ascending).
    // P::mad: product and sum rounded separately under Exact (the reference's rounding points), one fused
    // multiply-add per term under Fast
    template <class P, class T>
    static FIB_DEV T cheb(const float *d, const T (&S)[9])
    {
        T r = P::mad(S[1], d[1], d[0]);
#pragma unroll
        for (int i = 2; i <= 8; ++i) r = P::mad(S[i], d[i], r);
        return r;
    }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V0, float lap, const Consts &k, int sub)
    {
        body<P, MODE, float>(s, V0, lap, k, sub);
    }
    // the R cells of a lane at once (strip kernel), operation-major like Fenton::stepN
    static constexpr bool HAS_VEC = true;
    template <class P, int MODE, int R>
    static FIB_DEV void stepN(float (&s)[R][NVAR], const float (&V0)[R], const float (&lap)[R], const Consts &k, int sub)
    {
        vf<R> t[NVAR], v0, l;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int v = 0; v < NVAR; ++v) t[v].v[r] = s[r][v];
            v0.v[r] = V0[r];
            l.v[r] = lap[r];
        }
        body<P, MODE, vf<R>>(t, v0, l, k, sub);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < NVAR; ++v) s[r][v] = t[v].v[r];
    }
    template <class P, int MODE, class T>
    static FIB_DEV void body(T (&s)[NVAR], const T &V0, const T &lap, const Consts &k, int sub)
    {
#include "br_step.inc"
    }
};

// =====================================================================================
// Courtemanche  (court.py:124-429)
// =====================================================================================
struct CourtConsts {
    float dtf, dts;        // float(δt) for the fast / slow sets (court.py:118-122)
    float mdt_f, mdt_s;    // float(-δt)
    float ddt;             // float(diff * dt)
    float em1_fCa, em1_u;  // expm1(float(-δt/tau)) for the two constant-tau gates (:189,:243)
    float chronic;         // 1.0 / 0.0
    // Python-side products that depend on `chronic` (court.py:193,194,218)
    float c_to, c_Kur, c_CaL;
};

// US = court_ultra.py's optional 22nd gate `_us_` (ultra-slow sodium inactivation,
// court_ultra.py:81-82,198-199,221-222,445-450); only meaningful with MODE_ALL
template <bool US>
struct CourtT {
    static constexpr int NVAR = 21 + (US ? 1 : 0);
    static constexpr int DEFAULT_STEPS = 1;       // court.py:92
    static constexpr bool HAS_VEC = false;
    template <class C> static FIB_DEV const C &pinned(const C &k) { return k; }
    enum { iV, iNa_i, i_m, i_h, i_j, iK_i, i_oa, i_oi, i_ua, i_ui, i_xr, i_xs, iCa_i, i_d, i_f, i_f_Ca,
           iCa_rel, i_u, i_v, i_w, iCa_up, i_us };
    // MODE_FAST: the 4 fast_states (court.py:42,94-102); MODE_SLOW: the other 17 (court.py:103);
    // MODE_ALL: all 21 in one evaluation (court_ultra.py:107-111)
    // MODE_FASTSLOW: a fast tick and the 'slow' op the driver fires right after it (court.py:612-617) in ONE launch:
    // the kernel runs MODE_FAST, publishes the new potential through LDS, then MODE_SLOW on the post-fast state
    enum { MODE_FAST = 0, MODE_SLOW = 1, MODE_ALL = 2, MODE_FASTSLOW = 3 };
    static constexpr unsigned FAST_MASK = 0xFu, ALL_MASK = (1u << NVAR) - 1u;
    static constexpr unsigned mask(int mode)
    {
        return mode == MODE_FAST ? FAST_MASK : (mode == MODE_SLOW ? (ALL_MASK & ~FAST_MASK) : ALL_MASK);
    }
    using Consts = CourtConsts;

    struct Inter {
        float d_inf, tau_d, f_inf, tau_f, tau_w, w_inf, m_inf, tau_m, h_inf, tau_h, j_inf, tau_j;
        float tau_oa, oa_inf, tau_oi, oi_inf, tau_ua, ua_inf, tau_ui, ui_inf, tau_xr, xr_inf, tau_xs, xs_inf;
        float g_Kur, f_NaK, i_NaCaa, i_NaCab, i_K1a, i_Kra;
        float us_inf, tau_us;
    };

    // calc_inter(V, tf), court.py:273-429.  Fully inlined; whatever a MODE does not use is dead code.
    template <class P>
    static FIB_DEV void calc_inter(float V, Inter &o)
    {
#include "court_inter.inc"
    }

    // tf.pow(x, 3): float32 pow.  x*x*x is within 1 ulp of the correctly rounded cube.
    static FIB_DEV float pow3(float x) { return (x * x) * x; }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V, float lap, const Consts &k, int)
    {
#include "court_step.inc"
    }
};
using Courtemanche = CourtT<false>;
using CourtemancheUS = CourtT<true>;

// -------------------------------------------------------------------------------------------------------------------
// Courtemanche under the fast policy: the fast tick with the slow variables folded.
// court.py:94-103 assigns V, Na_i, m, h on every tick and the other 17 variables only when the driver fires 'slow'
// (every 10th tick, court.py:612-617).  Between two 'slow' ops the fast tick needs those 17 only through
//     S1   = c_to oa^3 oi + Cm g_Ks xs^2          (times V - E_K: i_to + i_Ks,            court.py:193,197)
//     PKur = c_Kur ua^3 ui                         (times g_Kur(V) (V - E_K): i_Kur,       court.py:194)
//     EK   = RT/F log(K_o / K_i)                                                            (court.py:191)
//     PCaL = c_CaL d f f_Ca                        (times V - 65: i_Ca_L,                  court.py:218)
//     C0   = i_CaP - Cm g_B_Ca E_Ca                (Ca_i is a slow variable,               court.py:219-221)
// and j, xr, Ca_i themselves.  'slow' (and any write to the state from outside) refreshes the five arrays; the fast
// tick then reads 12 arrays instead of 16 and drops two logarithms, four divisions and the cubes.  The sums are
// re-associated and fused (FMA), equal exponentials are formed once, 1/tau is never inverted twice: a few ulp per
// current, like everything under the fast policy — the rounding-faithful policy never takes this path.
// Arrays 21..25 of the kernels' pointer table are the five aggregates (fibhip.hip: fibhip_ctx::agg).
struct CourtAgg : CourtT<false> {
    typedef CourtT<false> Base;
    static constexpr int NAGG = 5;
    static constexpr int NVAR = Base::NVAR + NAGG;
    enum { a_S1 = Base::NVAR, a_PKur, a_EK, a_PCaL, a_C0 };
    enum { MODE_AGG = 4 };                        // only the aggregates, from the state as it stands
    static constexpr unsigned AGG_MASK = ((1u << NAGG) - 1u) << Base::NVAR;
    static constexpr unsigned mask(int mode)
    {
        return mode == MODE_FAST ? FAST_MASK : (mode == MODE_AGG ? AGG_MASK : (Base::mask(mode) | AGG_MASK));
    }
    static FIB_DEV float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
    static FIB_DEV float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
    static FIB_DEV float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

    static FIB_DEV void aggregates(float (&s)[NVAR], const Consts &k)
    {
        constexpr double R = 8.3143, T = 310, Fd = 96.4867, Cm = 100, K_o = 5.4, g_Ks = 0.12941176, Ca_o = 1.8;
        constexpr double i_CaP_max = 0.275, g_B_Ca = 0.001131;
        constexpr float RTF = FC((R * T) / Fd), RT2F = FC((R * T) / (2.0 * Fd));
        s[a_S1] = fm(FC(Cm * g_Ks) * s[i_xs], s[i_xs], ((k.c_to * pow3(s[i_oa])) * s[i_oi]));
        s[a_PKur] = (k.c_Kur * pow3(s[i_ua])) * s[i_ui];
        s[a_EK] = RTF * Fast::log(Fast::div(FC(K_o), s[iK_i]));
        s[a_PCaL] = ((k.c_CaL * s[i_d]) * s[i_f]) * s[i_f_Ca];
        const float Cai = s[iCa_i];
        const float i_CaP = Fast::div(FC(Cm * i_CaP_max) * Cai, FC(0.0005) + Cai);
        const float E_Ca = RT2F * Fast::log(Fast::div(FC(Ca_o), Cai));
        s[a_C0] = fm(E_Ca, FC(-(Cm * g_B_Ca)), i_CaP);
    }

    // the fast set (V, Na_i, m, h) from the boundary-enforced potential, its Laplacian and the aggregates
    static FIB_DEV void fast(float (&s)[NVAR], float V, float lap, const Consts &k)
    {
        constexpr double L2E = 1.44269504088896340736, LN2 = 0.69314718055994530942;
        constexpr double R = 8.3143, T = 310, Fd = 96.4867, Cm = 100, g_Na = 7.8, Na_o = 140, K_o = 5.4;
        constexpr double Km_K_o = 1.5, i_NaK_max = 0.59933874, g_B_Na = 0.0006744375, g_K1 = 0.09;
        constexpr double g_Kr = 0.029411765, Ca_o = 1.8, I_NaCa_max = 1600, K_mNa = 87.5, K_mCa = 1.38, K_sat = 0.1;
        constexpr double gamma_ = 0.35, V_cell = 20100, V_i = V_cell * 0.68;
        constexpr double FRT = Fd / (R * T), RTF = (R * T) / Fd;
        const float Nai = s[iNa_i], m = s[i_m], h = s[i_h], j = s[i_j], xr = s[i_xr], Cai = s[iCa_i];
        const float eps = V * FC(1e-20);                                                           // court.py:298

        // --- m (court.py:320-329) and h (:331-344): 1/tau = alpha + beta is used as it is
        const float vq = V + FC(47.13);
        const float n_m = FC(0.32) * vq, d_m = 1.0f - ex2(vq * FC(-0.1 * L2E));
        const float al_m = (fabsf(vq) < FC(0.001)) ? eps + FC(3.2) : n_m * rcpf(d_m);
        const float rate_m = fm(ex2(V * FC(-L2E / 11.0)), FC(0.08), al_m);
        const float m_inf = al_m * rcpf(rate_m);
        const float m1 = clipf(fm(m - m_inf, ex2(rate_m * (k.mdt_f * FC(L2E))), m_inf), 0.00001f, 0.99999f);   // m_inf + (m - m_inf) e^(-dt/tau)

        const bool lo = V < -40.0f;
        // one exponential serves both branches: exp((V+80)/-6.8) below -40 mV, exp((V+10.66)/-11.1) above
        const float e_a = ex2(fm(V, lo ? FC(L2E / -6.8) : FC(L2E / -11.1), lo ? FC(80.0 * L2E / -6.8) : FC(10.66 * L2E / -11.1)));
        const float be_lo = fm(ex2(V * FC(0.35 * L2E)), 310000.0f, FC(3.56) * ex2(V * FC(0.079 * L2E)));
        const float be_hi = rcpf(fm(e_a, FC(0.13), FC(0.13)));
        const float al_h = lo ? FC(0.135) * e_a : eps;
        const float rate_h = al_h + (lo ? be_lo : be_hi);
        const float h_inf = al_h * rcpf(rate_h);
        const float h1 = clipf(fm(h - h_inf, ex2(rate_h * (k.mdt_f * FC(L2E))), h_inf), 0.00001f, 0.99999f);

        // --- sodium (court.py:206-215): E_Na and (K_m/Na_i)^1.5 from ONE logarithm
        const float l2 = __builtin_amdgcn_logf(Nai);                                              // log2(Na_i)
        const float vENa = V - fm(l2, FC(-RTF * LN2), FC(RTF * 4.9416424226093039));             // log(140) = 4.94164...
        const float gNa = fm(((FC(Cm * g_Na) * pow3(m)) * h), j, FC(Cm * g_B_Na));               // i_Na + i_B_Na = gNa (V - E_Na)
        const float i_Nas = gNa * vENa;
        const float x = V * FC(FRT * L2E);                                                        // F V / (R T), base-2 scaled
        const float e8 = ex2(x * -0.1f), e9 = ex2(-x);                                           // court.py:417
        const float pw = ex2(fm(l2, -1.5f, FC(1.5 * 3.3219280948873623)));                        // (Km_Na_i / Na_i)^1.5, Km_Na_i = 10
        const float i_NaK = FC(Cm * i_NaK_max * (K_o / (K_o + Km_K_o))) *
                            rcpf(fm(e9, FC(0.0365), fm(e8, FC(0.1245), 1.0f)) * (1.0f + pw));    // :202,417
        const float e10 = ex2(x * FC(gamma_ - 1.0)), e11 = ex2(x * FC(gamma_));                   // :419-423 (e10 twice there)
        constexpr double cd = (K_mNa * K_mNa * K_mNa + Na_o * Na_o * Na_o) * (K_mCa + Ca_o);
        const float i_NaCa = fm(FC(Cm * I_NaCa_max * Ca_o / cd) * e11, pow3(Nai), -(FC(Cm * I_NaCa_max * Na_o * Na_o * Na_o / cd) * e10) * Cai) *
                             rcpf(fm(e10, FC(K_sat), 1.0f));                                      // :211

        // --- potassium (court.py:191-204): everything times V - E_K
        const float vEK = V - s[a_EK];
        const float i_K1a = FC(Cm * g_K1) * rcpf(1.0f + ex2(fm(V, FC(0.07 * L2E), FC(0.07 * 80.0 * L2E))));          // :425
        const float i_Kra = FC(Cm * g_Kr) * rcpf(1.0f + ex2(fm(V, FC(L2E / 22.4), FC(15.0 * L2E / 22.4))));          // :427
        const float g_Kur = fm(rcpf(1.0f + ex2(fm(V, FC(L2E / -13.0), FC(-15.0 * L2E / -13.0)))), FC(0.05), FC(0.005));  // :415
        const float X = fm(g_Kur, s[a_PKur], fm(i_Kra, xr, i_K1a + s[a_S1]));

        // --- potential (court.py:217-229)
        const float i_Cas = fm(s[a_PCaL], V - 65.0f, fm(V, FC(Cm * 0.001131), s[a_C0]));          // i_Ca_L + i_B_Ca + i_CaP
        const float isum = fm(X, vEK, i_Nas) + ((i_NaK + i_NaCa) + i_Cas);
        s[iV] = fm(lap, k.ddt, fm(isum, k.dtf * FC(-1.0 / Cm), V));
        s[iNa_i] = fm(fm(3.0f, i_NaK + i_NaCa, i_Nas), k.dtf * FC(-1.0 / (V_i * Fd)), Nai);
        s[i_m] = m1;
        s[i_h] = h1;
    }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V, float lap, const Consts &k, int sub)
    {
        if constexpr (MODE == MODE_FAST) {
            fast(s, V, lap, k);
        } else if constexpr (MODE == MODE_AGG) {
            aggregates(s, k);
        } else {                                   // MODE_SLOW (also the second pass of MODE_FASTSLOW)
            float b[Base::NVAR];
#pragma unroll
            for (int v = 0; v < Base::NVAR; ++v) b[v] = s[v];
            Base::template step<P, MODE>(b, V, lap, k, sub);
#pragma unroll
            for (int v = 0; v < Base::NVAR; ++v) s[v] = b[v];
            aggregates(s, k);
        }
    }
};

// =====================================================================================
// A model traced from a user's reference-style Python file (fib_tf_amd/traced.py) is emitted as
// `struct Custom` into a generated header and compiled in here (FIBHIP_CUSTOM).
// =====================================================================================
#ifdef FIB_CUSTOM_MODEL_INC
#include FIB_CUSTOM_MODEL_INC
#endif

FIB_TAG_END
}  // namespace fib
