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
//   Exact — one float32 rounding per reference op: IEEE division, ocml expf/expm1f/logf/tanhf,
//           no FMA contraction (the translation unit is built with -ffp-contract=off).  This
//           is the op order TensorFlow's CPU kernels execute for the reference graph.
//   Fast  — v_rcp_f32 / v_exp_f32 / v_log_f32 based forms (FIBHIP_FAST), a few ulp per op.
//
// The formulas follow fenton.py:46-108, br.py:125-332 and court.py:124-429 (cited per block).
#pragma once
#include <hip/hip_runtime.h>

namespace fib {

#define FIB_DEV __device__ __forceinline__

// Division by a value whose correctly rounded reciprocal rc = RN(1/c) is known (a compile-time
// constant, or the per-cell 1/(4ϕ) prepared once): q = RN(a*rc); r = a - q*c (exact, one FMA);
// RN(q + r*rc) is the correctly rounded quotient RN(a/c) (Markstein's theorem) — bit-identical to
// IEEE division at 3 instructions instead of the ~11 of the generic expansion.  Checked exhaustively
// over all 2^23 significands for every constant of the three models (tools/ubench/divtest.c);
// it can differ only when a/c is subnormal (by at most one subnormal ulp).
struct Exact {
    static constexpr bool CONTRACT = false;   // never fuse a*b+c: the reference rounds every op
    static FIB_DEV float divc(float a, float c, float rc)
    {
        const float q = a * rc;
        const float r = __builtin_fmaf(-q, c, a);
        return __builtin_fmaf(r, rc, q);
    }
    static FIB_DEV float div(float a, float b) { return a / b; }
    static FIB_DEV float rcp(float a) { return 1.0f / a; }
    static FIB_DEV float exp(float a) { return expf(a); }
    static FIB_DEV float expm1(float a) { return expm1f(a); }
    static FIB_DEV float log(float a) { return logf(a); }
    static FIB_DEV float tanh(float a) { return tanhf(a); }
    static FIB_DEV float sqrt(float a) { return sqrtf(a); }
};

struct Fast {
    static constexpr bool CONTRACT = true;    // the kinetics may use FMA (fewer roundings, ~25 % fewer
                                              // instructions); stencil + phase term never do
    static FIB_DEV float divc(float a, float, float rc) { return a * rc; }
    static FIB_DEV float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static FIB_DEV float rcp(float a) { return __builtin_amdgcn_rcpf(a); }
    static FIB_DEV float exp(float a) { return __expf(a); }
    // expm1 for the Rush-Larsen factor: argument is -dt/tau.  exp(x)-1 keeps an absolute error of
    // ~1 ulp(1) = 6e-8, which is below the float32 resolution of the gate value it multiplies into.
    static FIB_DEV float expm1(float a) { return __expf(a) - 1.0f; }
    static FIB_DEV float log(float a) { return __logf(a); }
    static FIB_DEV float tanh(float a)
    {   // 1 - 2/(exp(2a)+1); saturates correctly for |a| large (exp -> inf or 0)
        return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * a) + 1.0f);
    }
    static FIB_DEV float sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
};

static FIB_DEV float sgnf(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
// (1 + sign(x)) * 0.5 and (1 - sign(x)) * 0.5 (fenton.py:73-79) as selects: the same three values
// {0, 0.5, 1} bit for bit, 4 instructions instead of 7
static FIB_DEV float heav(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? 0.0f : 0.5f); }
static FIB_DEV float heav_not(float x) { return x > 0.0f ? 0.0f : (x < 0.0f ? 1.0f : 0.5f); }
static FIB_DEV float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// rush_larsen, ionic.py:115-123.  mdt = float(-dt)
template <class P>
static FIB_DEV float rush_larsen(float g, float ginf, float tau, float mdt)
{
    return clipf(g + (g - ginf) * P::expm1(P::div(mdt, tau)), 0.00001f, 0.99999f);
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
    };
    static constexpr unsigned mask(int) { return 0xFu; }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float U0, float lap, const Consts &k, int)
    {
        if constexpr (P::CONTRACT) step_fused<P>(s, U0, lap, k); else step_plain<P>(s, U0, lap, k);
    }
    template <class P>
    static FIB_DEV void step_plain(float (&s)[NVAR], float U0, float lap, const Consts &k)
    {
#include "fenton_step.inc"
    }
    template <class P>
    static FIB_DEV void step_fused(float (&s)[NVAR], float U0, float lap, const Consts &k)
    {
#pragma clang fp contract(fast)
#include "fenton_step.inc"
    }
};

// =====================================================================================
// Beeler-Reuter  (br.py:125-332)
// =====================================================================================
struct BeelerReuter {
    static constexpr int NVAR = 8;                // V C M H J D F XI, br.py:87-94
    static constexpr int DEFAULT_STEPS = 5;       // br.py:98-107
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
    template <class P>
    static FIB_DEV float ab(float v, float c0, float c1, float c2, float c3, float c4, float c5, float c6)
    {
        const float e1 = P::exp(c1 * (v + c2));
        const float den = P::exp(c5 * (v + c2)) + c6;
        if (c3 == 0.0f) return P::div(c0 * e1, den);
        return P::div(c0 * e1 + c3 * (v + c4), den);
    }
    template <class P>
    static FIB_DEV void inf_tau(float a, float b, float &inf, float &tau)
    {   // calc_inf_tau, br.py:266-273
        inf = P::div(a, a + b);
        tau = P::div(1.0f, a + b);
    }
    // expand_chebyshev device part, br.py:329-331:  r = d0; r += d_i * S_i  (i ascending)
    template <class P>
    static FIB_DEV float cheb(const float *d, const float (&S)[9])
    {
        if constexpr (false && P::CONTRACT) {
#pragma clang fp contract(fast)
            float r = d[0];
#pragma unroll
            for (int i = 1; i <= 8; ++i) r = r + d[i] * S[i];   // 8 FMAs: same order, fewer roundings
            return r;
        } else {
            float r = d[0];
#pragma unroll
            for (int i = 1; i <= 8; ++i) r = r + d[i] * S[i];
            return r;
        }
    }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V0, float lap, const Consts &k, int sub)
    {
        // measured: Beeler-Reuter at one sub-step per launch is launch/latency-bound, FMA contraction buys
        // nothing there, so both policies keep the reference's rounding points
        step_plain<P, MODE>(s, V0, lap, k, sub);
    }
    template <class P, int MODE>
    static FIB_DEV void step_plain(float (&s)[NVAR], float V0, float lap, const Consts &k, int sub)
    {
#include "br_step.inc"
    }
    template <class P, int MODE>
    static FIB_DEV void step_fused(float (&s)[NVAR], float V0, float lap, const Consts &k, int sub)
    {
#pragma clang fp contract(fast)
#include "br_step.inc"
    }
};

// =====================================================================================
// Courtemanche  (court.py:124-429)
// =====================================================================================
struct Courtemanche {
    static constexpr int NVAR = 21;
    static constexpr int DEFAULT_STEPS = 1;       // court.py:92
    enum { iV, iNa_i, i_m, i_h, i_j, iK_i, i_oa, i_oi, i_ua, i_ui, i_xr, i_xs, iCa_i, i_d, i_f, i_f_Ca,
           iCa_rel, i_u, i_v, i_w, iCa_up };
    // MODE_FAST: the 4 fast_states (court.py:42,94-102); MODE_SLOW: the other 17 (court.py:103);
    // MODE_ALL: all 21 in one evaluation (court_ultra.py:107-111)
    enum { MODE_FAST = 0, MODE_SLOW = 1, MODE_ALL = 2 };
    static constexpr unsigned FAST_MASK = 0xFu, ALL_MASK = (1u << 21) - 1u;
    static constexpr unsigned mask(int mode)
    {
        return mode == MODE_FAST ? FAST_MASK : (mode == MODE_SLOW ? (ALL_MASK & ~FAST_MASK) : ALL_MASK);
    }
    struct Consts {
        float dtf, dts;        // float(δt) for the fast / slow sets (court.py:118-122)
        float mdt_f, mdt_s;    // float(-δt)
        float ddt;             // float(diff * dt)
        float em1_fCa, em1_u;  // expm1(float(-δt/tau)) for the two constant-tau gates (:189,:243)
        float chronic;         // 1.0 / 0.0
        // Python-side products that depend on `chronic` (court.py:193,194,218)
        float c_to, c_Kur, c_CaL;
    };

    struct Inter {
        float d_inf, tau_d, f_inf, tau_f, tau_w, w_inf, m_inf, tau_m, h_inf, tau_h, j_inf, tau_j;
        float tau_oa, oa_inf, tau_oi, oi_inf, tau_ua, ua_inf, tau_ui, ui_inf, tau_xr, xr_inf, tau_xs, xs_inf;
        float g_Kur, f_NaK, i_NaCaa, i_NaCab, i_K1a, i_Kra;
    };

    // calc_inter(V, tf), court.py:273-429.  Fully inlined; whatever a MODE does not use is dead code.
    template <class P>
    static FIB_DEV void calc_inter(float V, Inter &o)
    {
        calc_inter_plain<P>(V, o);      // see BeelerReuter::step: no contraction where it buys nothing
    }
    template <class P>
    static FIB_DEV void calc_inter_plain(float V, Inter &o)
    {
#include "court_inter.inc"
    }
    template <class P>
    static FIB_DEV void calc_inter_fused(float V, Inter &o)
    {
#pragma clang fp contract(fast)
#include "court_inter.inc"
    }

    // tf.pow(x, 3): float32 pow.  x*x*x is within 1 ulp of the correctly rounded cube.
    static FIB_DEV float pow3(float x) { return (x * x) * x; }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V, float lap, const Consts &k, int)
    {
        step_plain<P, MODE>(s, V, lap, k);
    }
    template <class P, int MODE>
    static FIB_DEV void step_plain(float (&s)[NVAR], float V, float lap, const Consts &k)
    {
#include "court_step.inc"
    }
    template <class P, int MODE>
    static FIB_DEV void step_fused(float (&s)[NVAR], float V, float lap, const Consts &k)
    {
#pragma clang fp contract(fast)
#include "court_step.inc"
    }
};

}  // namespace fib
