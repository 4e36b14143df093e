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
        // constants fenton.py:49-71
        // divisors appear literally below: tau_d 0.065, tau_si = tau_so 31.8364, c_so 0.02, tau_vp 3.33,
        // tau_vn 19.2, tau_wp 160, tau_wn1 = tau_wn2 75
        constexpr float tau_a = FC(0.009), u_c = FC(0.23), u_w = FC(0.146), u_0 = FC(0.0), u_m = FC(1.0),
                        u_csi = FC(0.8), u_so = FC(0.3), r_sn = FC(1.2), k_ = FC(3.0), b_so = FC(0.84);
        constexpr float half_aso = FC(0.5 * (0.115 - 0.009));   // 0.5*(a_so - tau_a) in double
        constexpr float rsp_m_rsn = FC(0.02 - 1.2);             // (r_sp - r_sn) in double
        const float U = s[0], V = s[1], W = s[2], S = s[3];     // raw U: fenton.py:101

        const float Huc = (1.0f + sgnf(U - u_c)) * 0.5f;        // H(), :73-75
        const float Huso = (1.0f + sgnf(U - u_so)) * 0.5f;
        const float Guso = (1.0f - sgnf(U - u_so)) * 0.5f;      // G(), :77-79

        const float I_fi = DC((((-V) * Huc) * (U - u_c)) * (u_m - U), 0.065);                // :81
        const float I_si = DC((-W) * S, 31.8364);                                             // :82
        const float I_so = (half_aso * (1.0f + P::tanh(DC(U - b_so, 0.02))) +
                            DC((U - u_0) * Guso, 31.8364)) + Huso * tau_a;                    // :83-84
        const float dU = -((I_fi + I_si) + I_so);                                                // :86
        const float dV = (U > u_c) ? DC(-V, 3.33) : DC(1.0f - V, 19.2);              // :87
        const float dW = (U > u_c) ? DC(-W, 160.0)
                                   : ((U > u_w) ? DC(1.0f - W, 75.0) : DC(1.0f - W, 75.0)); // :88
        const float r_s = rsp_m_rsn * Huc + r_sn;                                                // :89
        const float dS = r_s * (0.5f * (1.0f + P::tanh((U - u_csi) * k_)) - S);                  // :90

        s[0] = (U0 + k.dt * dU) + k.ddt * lap;                                                   // :103
        s[1] = V + k.dt * dV;
        s[2] = W + k.dt * dW;
        s[3] = S + k.dt * dS;
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
    static FIB_DEV float cheb(const float *d, const float (&S)[9])
    {
        float r = d[0];
#pragma unroll
        for (int i = 1; i <= 8; ++i) r = r + d[i] * S[i];
        return r;
    }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V0, float lap, const Consts &k, int sub)
    {
        const float C = s[1], M = s[2], H = s[3], J = s[4], D = s[5], F = s[6], XI = s[7];
        // n = number of dt the slow gates advance (br.py:98-107): 1, or with skip 5 / 0
        const bool slow = k.skip ? (sub == 0) : true;
        const float mdtn = k.skip ? k.mdt_skip : k.mdt;
        float M1, H1, J1 = J, D1 = D, F1 = F, XI1 = XI;
        if (MODE == MODE_CHEBY) {                                   // br.py:207-252
            constexpr float xmid = FC(0.5 * (30.0 + -90.0)), xhalf = FC(0.5 * (30.0 - -90.0));
            float S[9];
            // always the correctly rounded quotient: the degree-8 sums amplify an ulp of x by ~1e2
            const float x = Exact::divc(V0 - xmid, xhalf, 1.0f / xhalf);           // :215
            S[0] = 1.0f; S[1] = x;                                  // calc_chebyshev_leading :289-301
#pragma unroll
            for (int i = 2; i <= 8; ++i) S[i] = (2.0f * x) * S[i - 1];
            M1 = rush_larsen<P>(M, cheb(k.cheb + 0 * 9, S), cheb(k.cheb + 2 * 9, S), k.mdt);
            H1 = rush_larsen<P>(H, cheb(k.cheb + 1 * 9, S), cheb(k.cheb + 3 * 9, S), k.mdt);
            if (slow) {
                XI1 = rush_larsen<P>(XI, cheb(k.cheb + 4 * 9, S), cheb(k.cheb + 8 * 9, S), mdtn);
                J1 = rush_larsen<P>(J, cheb(k.cheb + 5 * 9, S), cheb(k.cheb + 9 * 9, S), mdtn);
                D1 = rush_larsen<P>(D, cheb(k.cheb + 6 * 9, S), cheb(k.cheb + 10 * 9, S), mdtn);
                F1 = rush_larsen<P>(F, cheb(k.cheb + 7 * 9, S), cheb(k.cheb + 11 * 9, S), mdtn);
            }
        } else {                                                    // br.py:175-205
            float inf, tau;
            inf_tau<P>(ab<P>(V0, FC(0.0000f), FC(0.0f), FC(47.f), FC(-1.0f), FC(47.f), FC(-0.1f), FC(-1.0f)),
                       ab<P>(V0, FC(40.f), FC(-0.056f), FC(72.f), 0.0f, 0.0f, 0.0f, 0.0f), inf, tau);
            M1 = rush_larsen<P>(M, inf, tau, k.mdt);
            inf_tau<P>(ab<P>(V0, FC(0.126f), FC(-.25f), FC(77.f), 0.0f, 0.0f, 0.0f, 0.0f),
                       ab<P>(V0, FC(1.7f), FC(0.0f), FC(22.5f), 0.0f, 0.0f, FC(-0.082f), FC(1.0f)), inf, tau);
            H1 = rush_larsen<P>(H, inf, tau, k.mdt);
            if (slow) {
                inf_tau<P>(ab<P>(V0, FC(0.0005f), FC(0.083f), FC(50.f), 0.0f, 0.0f, FC(0.057f), 1.0f),
                           ab<P>(V0, FC(0.0013f), FC(-0.06f), FC(20.f), 0.0f, 0.0f, FC(-0.04f), 1.0f), inf, tau);
                XI1 = rush_larsen<P>(XI, inf, tau, mdtn);
                inf_tau<P>(ab<P>(V0, FC(0.055f), FC(-.25f), FC(78.0f), 0.0f, 0.0f, FC(-0.2f), 1.0f),
                           ab<P>(V0, FC(0.3f), FC(0.0f), FC(32.f), 0.0f, 0.0f, FC(-0.1f), 1.0f), inf, tau);
                J1 = rush_larsen<P>(J, inf, tau, mdtn);
                inf_tau<P>(ab<P>(V0, FC(2 * 0.095), FC(-0.01f), FC(-5.f), 0.0f, 0.0f, FC(-0.072f), 1.0f),
                           ab<P>(V0, FC(2 * 0.07), FC(-0.017f), FC(44.f), 0.0f, 0.0f, FC(0.05f), 1.0f), inf, tau);
                D1 = rush_larsen<P>(D, inf, tau, mdtn);
                inf_tau<P>(ab<P>(V0, FC(2 * 0.012), FC(-0.008f), FC(28.f), 0.0f, 0.0f, FC(0.15f), 1.0f),
                           ab<P>(V0, FC(2 * 0.0065), FC(-0.02f), FC(30.f), 0.0f, 0.0f, FC(-0.2f), 1.0f), inf, tau);
                F1 = rush_larsen<P>(F, inf, tau, mdtn);
            }
        }
        // currents from the OLD gates, br.py:150-165
        const float iK1 =
            0.35f * (P::div(4.0f * (P::exp(0.04f * (V0 + 85.0f)) - 1.0f),
                            P::exp(0.08f * (V0 + 53.0f)) + P::exp(0.04f * (V0 + 53.0f))) +
                     0.2f * P::div(V0 + 23.0f, 1.0f - P::exp(-0.04f * (V0 + 23.0f))));
        const float ix1 = P::div((XI * 0.8f) * (P::exp(0.04f * (V0 + 77.0f)) - 1.0f), P::exp(0.04f * (V0 + 35.0f)));
        const float iNa = (1.0f * (((((4.0f * M) * M) * M) * H) * J + 0.005f)) * (V0 - 50.0f);
        const float ECa = FC(0.0 - 82.3) - 13.0278f * P::log(C);
        const float iCa = ((FC(1.0 * 0.09) * D) * F) * (V0 - ECa);
        const float I_sum = ((iK1 + ix1) + iNa) + iCa;
        const float V1 = clipf((V0 + k.ddt * lap) - (k.dt * I_sum), -85.0f, 25.0f);   // :167-168
        const float dC = -1.0e-7f * iCa + 0.07f * (1.0e-7f - C);                                   // :170
        s[0] = V1;
        s[1] = C + k.dt * dC;
        s[2] = M1; s[3] = H1; s[4] = J1; s[5] = D1; s[6] = F1; s[7] = XI1;
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
        constexpr double R = 8.3143, T = 310, Fd = 96.4867, Cm = 100, Na_o = 140, g_K1 = 0.09, g_Kr = 0.029411765;
        constexpr double Ca_o = 1.8, I_NaCa_max = 1600, K_mNa = 87.5, K_mCa = 1.38, K_sat = 0.1, gamma_ = 0.35,
                         sigma = 1.0;
        constexpr float RT = FC(R * T);
        const float eps = V * FC(1e-20);                                                          // :298
        o.d_inf = P::rcp(1.0f + P::exp(P::divc(V + 10.0f, -8.0f, 1.0f / (-8.0f))));                                // :300
        {                                                                                          // :303-307
            const float a = P::div(4.579f, 1.0f + P::exp(P::div(V + 10.0f, FC(-6.24))));
            const float vp = V + FC(10.0001);
            const float e = P::exp(P::divc(vp, FC(-6.24), 1.0f / (FC(-6.24))));
            const float b = P::div(1.0f - e, (FC(0.0350000) * vp) * (1.0f + e));
            o.tau_d = (fabsf(vp) < FC(1.0e-10)) ? a : b;
        }
        {                                                                                          // :309
            const float e = P::exp(P::divc(-(V + 28.0f), FC(6.9), 1.0f / (FC(6.9))));
            o.f_inf = P::div(e, 1.0f + e);
        }
        {                                                                                          // :310
            constexpr float sq = FC(0.0337) * FC(0.0337);
            const float vv = (V + 10.0f) * (V + 10.0f);
            o.tau_f = 9.0f * P::rcp(FC(0.0197000) * P::exp((-sq) * vv) + FC(0.02));
        }
        {                                                                                          // :312-316
            const float vm = V - FC(7.9);
            const float e = P::exp(P::divc(-vm, 5.0f, 1.0f / (5.0f)));
            const float b = P::div(6.0f * (1.0f - e), ((1.0f + FC(0.3) * e) * 1.0f) * vm);
            o.tau_w = (fabsf(vm) < FC(1.0e-10)) ? eps + FC((6.0 * 0.2) / 1.3) : b;
        }
        o.w_inf = 1.0f - P::rcp(1.0f + P::exp(P::divc(-(V - 40.0f), 17.0f, 1.0f / (17.0f))));                      // :318
        float al, be;
        {                                                                                          // :320-329
            const float vp = V - FC(-47.13), vq = V + FC(47.13);
            al = (fabsf(vp) < FC(0.001)) ? eps + FC(3.2) : P::div(FC(0.32) * vq, 1.0f - P::exp(FC(-0.1) * vq));
            be = FC(0.08) * P::exp(P::divc(-V, 11.0f, 1.0f / (11.0f)));
            o.m_inf = P::div(al, al + be);
            o.tau_m = P::rcp(al + be);
        }
        const bool lo = V < -40.0f;
        {                                                                                          // :331-344
            al = lo ? FC(0.135) * P::exp(P::divc(V + 80.0f, FC(-6.8), 1.0f / (FC(-6.8)))) : eps;
            be = lo ? FC(3.56) * P::exp(FC(0.079) * V) + 310000.0f * P::exp(FC(0.35) * V)
                    : P::rcp(FC(0.13) * (1.0f + P::exp(P::divc(V + FC(10.66), FC(-11.1), 1.0f / (FC(-11.1))))));
            o.h_inf = P::div(al, al + be);
            o.tau_h = P::rcp(al + be);
        }
        {                                                                                          // :346-359
            al = lo ? P::div(((-127140.0f * P::exp(FC(0.2444) * V)) - FC(3.474e-05) * P::exp(FC(-0.04391) * V)) *
                                 (V + FC(37.78)),
                             1.0f + P::exp(FC(0.311) * (V + FC(79.23))))
                    : eps;
            be = lo ? P::div(FC(0.1212) * P::exp(FC(-0.01052) * V), 1.0f + P::exp(FC(-0.1378) * (V + FC(40.14))))
                    : P::div(FC(0.3) * P::exp(FC(-2.535e-07) * V), 1.0f + P::exp(FC(-0.1) * (V + 32.0f)));
            o.j_inf = P::div(al, al + be);
            o.tau_j = P::rcp(al + be);
        }
        const float v10 = V - -10.0f;
        {                                                                                          // :361-365, :373-377
            al = FC(0.65) * P::rcp(P::exp(P::divc(v10, -8.5f, 1.0f / (-8.5f))) + P::exp(P::divc(v10 - 40.0f, -59.0f, 1.0f / (-59.0f))));
            be = FC(0.65) * P::rcp(2.5f + P::exp(P::divc(v10 + 72.0f, 17.0f, 1.0f / (17.0f))));
            o.tau_oa = P::divc(P::rcp(al + be), 3.0f, 1.0f / (3.0f));
            o.tau_ua = o.tau_oa;            // alpha_ua/beta_ua are the same expressions (:373-376)
            o.oa_inf = P::rcp(1.0f + P::exp(P::divc(v10 + FC(10.47), FC(-17.54), 1.0f / (FC(-17.54)))));
            o.ua_inf = P::rcp(1.0f + P::exp(P::divc(v10 + FC(20.3), FC(-9.6), 1.0f / (FC(-9.6)))));
        }
        {                                                                                          // :367-371
            al = P::rcp(FC(18.53) + 1.0f * P::exp(P::divc(v10 + FC(103.7), FC(10.95), 1.0f / (FC(10.95)))));
            be = P::rcp(FC(35.56) + 1.0f * P::exp(P::divc(v10 - FC(8.74), FC(-7.44), 1.0f / (FC(-7.44)))));
            o.tau_oi = P::divc(P::rcp(al + be), 3.0f, 1.0f / (3.0f));
            o.oi_inf = P::rcp(1.0f + P::exp(P::divc(v10 + FC(33.1), FC(5.3), 1.0f / (FC(5.3)))));
        }
        {                                                                                          // :379-383
            al = P::rcp(21.0f + 1.0f * P::exp(P::divc(v10 - 195.0f, -28.0f, 1.0f / (-28.0f))));
            be = P::rcp(P::exp(P::divc(v10 - 168.0f, -16.0f, 1.0f / (-16.0f))));
            o.tau_ui = P::divc(P::rcp(al + be), 3.0f, 1.0f / (3.0f));
            o.ui_inf = P::rcp(1.0f + P::exp(P::divc(v10 - FC(109.45), FC(27.48), 1.0f / (FC(27.48)))));
        }
        {                                                                                          // :385-398
            const float va = V + FC(14.1), vb = V - FC(3.3328);
            al = (fabsf(va) < FC(1.0e-10)) ? eps + FC(0.0015)
                                           : P::div(FC(0.0003) * va, 1.0f - P::exp(P::div(va, -5.0f)));
            be = (fabsf(vb) < FC(1.0e-10)) ? eps + FC(0.000378361)
                                           : P::div(FC(7.3898e-05) * vb, P::exp(P::div(vb, FC(5.1237))) - 1.0f);
            o.tau_xr = P::rcp(al + be);
            o.xr_inf = P::rcp(1.0f + P::exp(P::divc(va, -6.5f, 1.0f / (-6.5f))));
        }
        {                                                                                          // :400-413
            const float vs = V - FC(19.9);
            const bool sing = fabsf(vs) < FC(1.0e-10);
            al = sing ? eps + FC(0.00068) : P::div(FC(4.0e-05) * vs, 1.0f - P::exp(P::div(vs, -17.0f)));
            be = sing ? eps + FC(0.000315) : P::div(FC(3.5e-05) * vs, P::exp(P::div(vs, 9.0f)) - 1.0f);
            o.tau_xs = 0.5f * P::rcp(al + be);
            o.xs_inf = P::sqrt(P::rcp(1.0f + P::exp(P::divc(vs, FC(-12.7), 1.0f / (FC(-12.7))))));
        }
        o.g_Kur = FC(0.005) + P::div(FC(0.05), 1.0f + P::exp(P::div(V - 15.0f, -13.0f)));         // :415
        o.f_NaK = P::rcp((1.0f + FC(0.1245) * P::exp(P::divc(FC(-0.1 * Fd) * V, RT, 1.0f / (RT)))) +
                         FC(0.0365 * sigma) * P::exp(P::divc(FC(-Fd) * V, RT, 1.0f / (RT))));                    // :417
        const float i_NaCad = FC((K_mNa * K_mNa * K_mNa + Na_o * Na_o * Na_o) * (K_mCa + Ca_o)) *
                              (1.0f + FC(K_sat) * P::exp(P::divc((FC(gamma_ - 1.0) * V) * FC(Fd), RT, 1.0f / (RT))));  // :419
        o.i_NaCaa = P::div(FC(Cm * I_NaCa_max) * (P::exp(P::div(FC(gamma_ * Fd) * V, RT)) * FC(Ca_o)), i_NaCad);  // :421
        o.i_NaCab = P::div(FC(Cm * I_NaCa_max) *
                               (P::exp(P::div(FC((gamma_ - 1.0) * Fd) * V, RT)) * FC(Na_o * Na_o * Na_o)),
                           i_NaCad);                                                               // :423
        o.i_K1a = P::div(FC(Cm * g_K1), 1.0f + P::exp(FC(0.07) * (V + 80.0f)));                    // :425
        o.i_Kra = P::div(FC(Cm * g_Kr), 1.0f + P::exp(P::div(V + 15.0f, FC(22.4))));              // :427
    }

    // tf.pow(x, 3): float32 pow.  x*x*x is within 1 ulp of the correctly rounded cube.
    static FIB_DEV float pow3(float x) { return (x * x) * x; }

    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float V, float lap, const Consts &k, int)
    {
        constexpr bool FASTV = (MODE != MODE_SLOW), SLOWV = (MODE != MODE_FAST);
        // MODE_ALL = court_ultra.py single rate: every variable uses dt (court_ultra.py:127-128)
        const float dts = (MODE == MODE_ALL) ? k.dtf : k.dts;
        const float mdt_s = (MODE == MODE_ALL) ? k.mdt_f : k.mdt_s;
        // constants court.py:129-163
        constexpr double R = 8.3143, T = 310, Fd = 96.4867, Cm = 100, g_Na = 7.8, Na_o = 140, K_o = 5.4;
        constexpr double g_Ks = 0.12941176, Km_Na_i = 10, Km_K_o = 1.5, i_NaK_max = 0.59933874, i_CaP_max = 0.275;
        constexpr double g_B_Na = 0.0006744375, g_B_Ca = 0.001131, g_B_K = 0, Ca_o = 1.8, K_rel = 30, tau_tr = 180;
        constexpr double I_up_max = 0.005, K_up = 0.00092, Ca_up_max = 15, CMDN_max = 0.05, TRPN_max = 0.07;
        constexpr double CSQN_max = 10, Km_CMDN = 0.00238, Km_TRPN = 0.0005, Km_CSQN = 0.8, V_cell = 20100;
        constexpr double V_i = V_cell * 0.68, V_rel = 0.0048 * V_cell, V_up = 0.0552 * V_cell;
        constexpr float RTF = FC((R * T) / Fd), RT2F = FC((R * T) / (2.0 * Fd)), ViF = FC(V_i * Fd);

        Inter q;
        calc_inter<P>(V, q);
        float o[NVAR];
        // gates, court.py:175-189
        o[i_d] = rush_larsen<P>(s[i_d], q.d_inf, q.tau_d, mdt_s);
        o[i_f] = rush_larsen<P>(s[i_f], q.f_inf, q.tau_f, mdt_s);
        o[i_w] = rush_larsen<P>(s[i_w], q.w_inf, q.tau_w, mdt_s);
        o[i_m] = rush_larsen<P>(s[i_m], q.m_inf, q.tau_m, k.mdt_f);
        o[i_h] = rush_larsen<P>(s[i_h], q.h_inf, q.tau_h, k.mdt_f);
        o[i_j] = rush_larsen<P>(s[i_j], q.j_inf, q.tau_j, mdt_s);
        o[i_oa] = rush_larsen<P>(s[i_oa], q.oa_inf, q.tau_oa, mdt_s);
        o[i_oi] = rush_larsen<P>(s[i_oi], q.oi_inf, q.tau_oi, mdt_s);
        o[i_ua] = rush_larsen<P>(s[i_ua], q.ua_inf, q.tau_ua, mdt_s);
        o[i_ui] = rush_larsen<P>(s[i_ui], q.ui_inf, q.tau_ui, mdt_s);
        o[i_xr] = rush_larsen<P>(s[i_xr], q.xr_inf, q.tau_xr, mdt_s);
        o[i_xs] = rush_larsen<P>(s[i_xs], q.xs_inf, q.tau_xs, mdt_s);
        const float f_Ca_inf = P::rcp(1.0f + P::divc(s[iCa_i], FC(0.00035), 1.0f / (FC(0.00035))));
        o[i_f_Ca] = rush_larsen_c(s[i_f_Ca], f_Ca_inf, k.em1_fCa);
        // potassium, court.py:191-204
        const float E_K = RTF * P::log(P::div(FC(K_o), s[iK_i]));
        const float vEK = V - E_K;
        const float i_K1 = q.i_K1a * vEK;
        const float i_to = ((k.c_to * pow3(s[i_oa])) * s[i_oi]) * vEK;
        const float i_Kur = (((k.c_Kur * q.g_Kur) * pow3(s[i_ua])) * s[i_ui]) * vEK;
        const float i_Kr = (q.i_Kra * s[i_xr]) * vEK;
        const float i_Ks = (FC(Cm * g_Ks) * (s[i_xs] * s[i_xs])) * vEK;
        const float nr = P::div(FC(Km_Na_i), s[iNa_i]);
        const float i_NaK = P::div(FC(Cm * i_NaK_max) * q.f_NaK, 1.0f + P::sqrt(pow3(nr))) * FC(K_o / (K_o + Km_K_o));
        const float i_B_K = FC(Cm * g_B_K) * vEK;
        o[iK_i] = s[iK_i] +
                  P::divc(2.0f * i_NaK - (((((i_K1 + i_to) + i_Kur) + i_Kr) + i_Ks) + i_B_K), ViF, 1.0f / (ViF)) * dts;
        // sodium, court.py:206-215
        const float E_Na = RTF * P::log(P::div(FC(Na_o), s[iNa_i]));
        const float i_Na = (((FC(Cm * g_Na) * pow3(s[i_m])) * s[i_h]) * s[i_j]) * (V - E_Na);
        const float i_NaCa = q.i_NaCaa * pow3(s[iNa_i]) - q.i_NaCab * s[iCa_i];
        const float i_B_Na = FC(Cm * g_B_Na) * (V - E_Na);
        o[iNa_i] = s[iNa_i] + P::divc(-3.0f * i_NaK - ((3.0f * i_NaCa + i_B_Na) + i_Na), ViF, 1.0f / (ViF)) * k.dtf;
        // calcium currents + potential, court.py:217-229
        const float i_st = 0.0f;
        const float i_Ca_L = (((k.c_CaL * s[i_d]) * s[i_f]) * s[i_f_Ca]) * (V - 65.0f);
        const float i_CaP = P::div(FC(Cm * i_CaP_max) * s[iCa_i], FC(0.0005) + s[iCa_i]);
        const float E_Ca = RT2F * P::log(P::div(FC(Ca_o), s[iCa_i]));
        const float i_B_Ca = FC(Cm * g_B_Ca) * (V - E_Ca);
        const float isum = (((((((((((i_Na + i_K1) + i_to) + i_Kur) + i_Kr) + i_Ks) + i_B_Na) + i_B_Ca) + i_NaK) +
                              i_CaP) + i_NaCa) + i_Ca_L) + i_st;
        const float DV = V + P::divc(-isum, FC(Cm), 1.0f / (FC(Cm))) * k.dtf;
        o[iV] = DV + k.ddt * lap;
        // SR release / uptake, court.py:232-256
        const float i_rel = (((FC(K_rel) * (s[i_u] * s[i_u])) * s[i_v]) * s[i_w]) * (s[iCa_rel] - s[iCa_i]);
        const float i_tr = P::divc(s[iCa_up] - s[iCa_rel], FC(tau_tr), 1.0f / (FC(tau_tr)));
        {
            const float t = s[iCa_rel] + FC(Km_CSQN);
            o[iCa_rel] = s[iCa_rel] + ((i_tr - i_rel) * P::rcp(1.0f + P::div(FC(CSQN_max * Km_CSQN), t * t))) * dts;
        }
        const float Fn = 1000.0f * (FC(1.0e-15 * V_rel) * i_rel -
                                    FC(1.0e-15 / (2.0 * Fd)) * (0.5f * i_Ca_L - FC(0.2) * i_NaCa));
        const float u_inf = P::rcp(1.0f + P::exp(P::divc(-(Fn - FC(3.4175e-13)), FC(1.367e-15), 1.0f / (FC(1.367e-15)))));
        // MODE_ALL (court_ultra) integrates u with dt: expm1(float(-dt/tau_u)) is k.em1_u there too
        o[i_u] = rush_larsen_c(s[i_u], u_inf, k.em1_u);
        const float tau_v = FC(1.91) + FC(2.09) * u_inf;
        const float v_inf = 1.0f - P::rcp(1.0f + P::exp(P::divc(-(Fn - FC(6.835e-14)), FC(1.367e-15), 1.0f / (FC(1.367e-15)))));
        o[i_v] = rush_larsen<P>(s[i_v], v_inf, tau_v, mdt_s);
        const float i_up = P::div(FC(I_up_max), 1.0f + P::div(FC(K_up), s[iCa_i]));
        const float i_up_leak = P::divc(FC(I_up_max) * s[iCa_up], FC(Ca_up_max), 1.0f / (FC(Ca_up_max)));
        o[iCa_up] = s[iCa_up] + (i_up - (i_up_leak + P::divc(i_tr * FC(V_rel), FC(V_up), 1.0f / (FC(V_up))))) * dts;
        // intracellular calcium, court.py:258-265
        const float B1 = P::divc(2.0f * i_NaCa - ((i_CaP + i_Ca_L) + i_B_Ca), FC(2.0 * V_i * Fd), 1.0f / (FC(2.0 * V_i * Fd))) +
                         P::divc(FC(V_up) * (i_up_leak - i_up) + i_rel * FC(V_rel), FC(V_i), 1.0f / (FC(V_i)));
        const float t1 = s[iCa_i] + FC(Km_TRPN), t2 = s[iCa_i] + FC(Km_CMDN);
        const float B2 = (1.0f + P::div(FC(TRPN_max * Km_TRPN), t1 * t1)) + P::div(FC(CMDN_max * Km_CMDN), t2 * t2);
        o[iCa_i] = s[iCa_i] + P::div(B1, B2) * dts;

#pragma unroll
        for (int v = 0; v < NVAR; ++v) {
            const bool fastvar = (FAST_MASK >> v) & 1u;
            if ((fastvar && FASTV) || (!fastvar && SLOWV)) s[v] = o[v];
        }
    }
};

}  // namespace fib
