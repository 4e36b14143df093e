/*
 * fib_oracle.c — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity ORACLE and the `cpu_baseline` leg of bench.py.  It is never
 * linked into, loaded by or called from the product (fib_tf_amd/): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * It restates, op for op in float32 (one rounding per reference op, no FMA contraction:
 * build with -ffp-contract=off, no -ffast-math), what one `solve()` of each reference
 * model computes.  Python-float sub-expressions of the reference are evaluated in double
 * and rounded to float once, exactly where TensorFlow converts them.  Citations are
 * file:line into the reference tree (siravan/fib_tf).
 *
 * Pinning: checked against the tests/golden/ .npz fixtures, which were produced by the reference's own
 * functions running under a float32 NumPy stand-in for TensorFlow (tests/golden/
 * make_golden.py).  Arithmetic-only paths (boundary, Laplacian, phase field) must match
 * those fixtures bit for bit; paths through tanh/exp/expm1/log/pow match to a few ulp
 * (libm here vs NumPy there vs Eigen in real TensorFlow).  TensorFlow itself is absent
 * from the build container, so the last-ulp behaviour of TF's kernels is unpinned.
 *
 * State layout everywhere: SoA slab  [nvar][H][W] float32, row-major.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define F(x) ((float)(x))

/* Test knob (default 0 = plain libm): every expf result moved by `orc_exp_ulps` float32 neighbours.  It models "any
 * exp() within that many ulp of libm's" — TensorFlow's Eigen kernels, NumPy's SIMD loops and the GPU's v_exp_f32 all
 * are such implementations.  The parity tests use it to bound what the reference's own ill-conditioned quotients
 * (the removable singularities of court.py:303-410) do to the last bit of exp. */
static int orc_exp_ulps = 0;
void orc_set_exp_ulps(int k) { orc_exp_ulps = k; }
static inline float orc_expf(float x)
{
    float r = expf(x);
    for (int k = orc_exp_ulps; k > 0; --k) r = nextafterf(r, INFINITY);
    for (int k = orc_exp_ulps; k < 0; ++k) r = nextafterf(r, -INFINITY);
    return r;
}
#define expf(x) orc_expf(x)

/* ------------------------------------------------------------------------- */
/* geometry helpers                                                           */
/* ------------------------------------------------------------------------- */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflecti(int v, int n) { return v < 0 ? -v : (v >= n ? 2 * n - 2 - v : v); }

/* enforce_boundary, ionic.py:107-113: interior re-padded SYMMETRIC by 1.
 * out[r][c] = in[clamp(r,1,H-2)][clamp(c,1,W-2)]                                     */
void orc_enforce_boundary(int H, int W, const float *in, float *out)
{
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c)
            out[r * W + c] = in[clampi(r, 1, H - 2) * W + clampi(c, 1, W - 2)];
}

/* laplace(X0) (+ phase_field when phi != NULL), ionic.py:44-60 and :70-81.
 * X = REFLECT-pad(X0); evaluation order as written in ionic.py:51-53 / :78-80.       */
void orc_laplace(int H, int W, const float *X0, const float *phi, float *out)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < H; ++r) {
        const int rn = reflecti(r - 1, H), rs = reflecti(r + 1, H);
        for (int c = 0; c < W; ++c) {
            const int cw = reflecti(c - 1, W), ce = reflecti(c + 1, W);
            const float N = X0[rn * W + c], S = X0[rs * W + c];
            const float Wv = X0[r * W + cw], E = X0[r * W + ce];
            const float NW = X0[rn * W + cw], SW = X0[rs * W + cw];
            const float NE = X0[rn * W + ce], SE = X0[rs * W + ce];
            const float C = X0[r * W + c];
            float l = (((N + S) + Wv) + E) + 0.5f * (((NW + SW) + NE) + SE);
            l = l - 6.0f * C;
            if (phi) {
                const float f = ((S - N) * (phi[rs * W + c] - phi[rn * W + c]) +
                                 (E - Wv) * (phi[r * W + ce] - phi[r * W + cw])) /
                                (4.0f * phi[r * W + c]);
                l = l + f;
            }
            out[r * W + c] = l;
        }
    }
}

/* phase_field alone, ionic.py:70-81 (X0 is the un-padded array; pad is REFLECT)      */
void orc_phase_field(int H, int W, const float *X0, const float *phi, float *out)
{
    for (int r = 0; r < H; ++r) {
        const int rn = reflecti(r - 1, H), rs = reflecti(r + 1, H);
        for (int c = 0; c < W; ++c) {
            const int cw = reflecti(c - 1, W), ce = reflecti(c + 1, W);
            out[r * W + c] = ((X0[rs * W + c] - X0[rn * W + c]) * (phi[rs * W + c] - phi[rn * W + c]) +
                              (X0[r * W + ce] - X0[r * W + cw]) * (phi[r * W + ce] - phi[r * W + cw])) /
                             (4.0f * phi[r * W + c]);
        }
    }
}

/* rush_larsen, ionic.py:115-123:  clip(g + (g - g_inf) * expm1(-dt/tau), 1e-5, 0.99999)
 * `mdt` is float(-dt) (the Python float -dt converted when it meets the tensor).     */
static inline float rush_larsen(float g, float ginf, float tau, float mdt)
{
    float r = g + (g - ginf) * expm1f(mdt / tau);
    r = fmaxf(r, 0.00001f);
    return fminf(r, 0.99999f);
}
/* same with tau a Python constant: expm1 argument pre-rounded on the host            */
static inline float rush_larsen_c(float g, float ginf, float em1)
{
    float r = g + (g - ginf) * em1;
    r = fmaxf(r, 0.00001f);
    return fminf(r, 0.99999f);
}

void orc_rush_larsen(long n, const float *g, const float *ginf, const float *tau, double dt, float *out)
{
    const float mdt = F(-dt);
    for (long i = 0; i < n; ++i) out[i] = rush_larsen(g[i], ginf[i], tau[i], mdt);
}

/* pace op, ionic.py:144-163: pot = max(pot, s), s = min_v outside the rectangle       */
void orc_pace(int H, int W, float *pot, int r0, int r1, int c0, int c1, float v, float min_v)
{
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const float s = (r >= r0 && r < r1 && c >= c0 && c < c1) ? v : min_v;
            pot[r * W + c] = fmaxf(pot[r * W + c], s);
        }
}

static inline float sgnf(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

/* ------------------------------------------------------------------------- */
/* Fenton 4v, fenton.py:46-108                                                 */
/* ------------------------------------------------------------------------- */

static inline void fenton_diff(float U, float V, float W, float S, float *dU, float *dV, float *dW, float *dS)
{
    /* constants fenton.py:49-71 */
    const float tau_vp = F(3.33), tau_vn = F(19.2), tau_wp = F(160.0), tau_wn1 = F(75.0), tau_wn2 = F(75.0);
    const float tau_d = F(0.065), tau_si = F(31.8364), tau_so = F(31.8364), tau_a = F(0.009);
    const float u_c = F(0.23), u_w = F(0.146), u_0 = F(0.0), u_m = F(1.0), u_csi = F(0.8), u_so = F(0.3);
    const float r_sn = F(1.2), k_ = F(3.0), b_so = F(0.84), c_so = F(0.02);
    const float half_aso = F(0.5 * (0.115 - 0.009));      /* 0.5 * (a_so - tau_a), Python double */
    const float rsp_m_rsn = F(0.02 - 1.2);                /* (r_sp - r_sn), Python double         */

    const float Huc = (1.0f + sgnf(U - u_c)) * 0.5f;      /* H(), fenton.py:73-75 */
    const float Huso = (1.0f + sgnf(U - u_so)) * 0.5f;
    const float Guso = (1.0f - sgnf(U - u_so)) * 0.5f;    /* G(), fenton.py:77-79 */

    const float I_fi = ((((-V) * Huc) * (U - u_c)) * (u_m - U)) / tau_d;                 /* :81 */
    const float I_si = ((-W) * S) / tau_si;                                              /* :82 */
    const float I_so = (half_aso * (1.0f + tanhf((U - b_so) / c_so)) +
                        ((U - u_0) * Guso) / tau_so) + Huso * tau_a;                     /* :83-84 */
    *dU = -((I_fi + I_si) + I_so);                                                       /* :86 */
    *dV = (U > u_c) ? (-V) / tau_vp : (1.0f - V) / tau_vn;                               /* :87 */
    *dW = (U > u_c) ? (-W) / tau_wp : ((U > u_w) ? (1.0f - W) / tau_wn2 : (1.0f - W) / tau_wn1); /* :88 */
    const float r_s = rsp_m_rsn * Huc + r_sn;                                            /* :89 */
    *dS = r_s * (0.5f * (1.0f + tanhf((U - u_csi) * k_)) - S);                           /* :90 */
}

void orc_fenton_diff(long n, const float *U, const float *V, const float *W, const float *S,
                     float *dU, float *dV, float *dW, float *dS)
{
    for (long i = 0; i < n; ++i) fenton_diff(U[i], V[i], W[i], S[i], dU + i, dV + i, dW + i, dS + i);
}

/* one solve(), fenton.py:95-108.  in/out: slabs [4][H][W] (U,V,W,S); scratch: 2*H*W    */
void orc_fenton_step(int H, int W, double dt, double diff, const float *phi,
                     const float *in, float *out, float *scratch)
{
    const long n = (long)H * W;
    float *U0 = scratch, *lap = scratch + n;
    const float dtf = F(dt), ddt = F(diff * dt);          /* self.diff * self.dt in double   */
    orc_enforce_boundary(H, W, in, U0);
    orc_laplace(H, W, U0, phi, lap);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        float dU, dV, dW, dS;
        const float U = in[i], V = in[n + i], Wg = in[2 * n + i], S = in[3 * n + i];
        fenton_diff(U, V, Wg, S, &dU, &dV, &dW, &dS);     /* raw U, fenton.py:101 */
        out[i] = (U0[i] + dtf * dU) + ddt * lap[i];       /* :103 */
        out[n + i] = V + dtf * dV;
        out[2 * n + i] = Wg + dtf * dW;
        out[3 * n + i] = S + dtf * dS;
    }
}

/* ------------------------------------------------------------------------- */
/* Beeler-Reuter, br.py:125-332                                                */
/* ------------------------------------------------------------------------- */

/* ab_coef, br.py:49-62 (float32 table; d/f rows pre-multiplied by 2 in double first)  */
static void br_ab_coef(float c[12][7])
{
    const double t[12][7] = {
        {0.0005, 0.083, 50., 0.0, 0.0, 0.057, 1.0},   {0.0013, -0.06, 20., 0.0, 0.0, -0.04, 1.0},
        {0.0000, 0.0, 47., -1.0, 47., -0.1, -1.0},    {40., -0.056, 72., 0.0, 0.0, 0.0, 0.0},
        {0.126, -.25, 77., 0.0, 0.0, 0.0, 0.0},       {1.7, 0.0, 22.5, 0.0, 0.0, -0.082, 1.0},
        {0.055, -.25, 78.0, 0.0, 0.0, -0.2, 1.0},     {0.3, 0.0, 32., 0.0, 0.0, -0.1, 1.0},
        {2 * 0.095, -0.01, -5., 0.0, 0.0, -0.072, 1.0}, {2 * 0.07, -0.017, 44., 0.0, 0.0, 0.05, 1.0},
        {2 * 0.012, -0.008, 28., 0.0, 0.0, 0.15, 1.0},  {2 * 0.0065, -0.02, 30., 0.0, 0.0, -0.2, 1.0}};
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 7; ++j) c[i][j] = F(t[i][j]);
}

/* calc_alpha_bata_tf, br.py:255-264 */
static inline float br_ab(float v, const float *c)
{
    if (c[3] == 0.0f)
        return (c[0] * expf(c[1] * (v + c[2]))) / (expf(c[5] * (v + c[2])) + c[6]);
    return (c[0] * expf(c[1] * (v + c[2])) + c[3] * (v + c[4])) / (expf(c[5] * (v + c[2])) + c[6]);
}

/* calc_inf_tau, br.py:266-273 */
static inline void br_inf_tau(float v, const float *ca, const float *cb, float *inf, float *tau)
{
    const float a = br_ab(v, ca), b = br_ab(v, cb);
    *inf = a / (a + b);
    *tau = 1.0f / (a + b);
}

/* expand_chebyshev's device part, br.py:329-331 with Ts from calc_chebyshev_leading
 * (br.py:289-301): r = d0; r += d_i * S_i, i ascending; S_1 = x, S_i = (2*x)*S_{i-1}.
 * d: 9 float32 coefficients (each float64 coefficient rounded separately).           */
static inline float br_cheb(const float *d, const float *S)
{
    float r = d[0];
    for (int i = 1; i <= 8; ++i) r = r + d[i] * S[i];
    return r;
}

/* one solve(state, n), br.py:125-173.  in/out: [8][H][W] = V,C,M,H,J,D,F,XI.
 * cheb: NULL (direct gates, br.py:175-205) or float32[12][9] in the row order
 * m_inf,h_inf,m_tau,h_tau,xi_inf,j_inf,d_inf,f_inf,xi_tau,j_tau,d_tau,f_tau
 * (br.py:223-240).  scratch: 2*H*W.                                                   */
void orc_br_step(int H, int W, double dt, double diff, const float *phi, const float *cheb, int nslow,
                 const float *in, float *out, float *scratch)
{
    const long n = (long)H * W;
    float *V0a = scratch, *lap = scratch + n;
    float ab[12][7];
    br_ab_coef(ab);
    const float dtf = F(dt), ddt = F(diff * dt);
    const float mdt = F(-dt), mdtn = F(-(dt * nslow));    /* dt*n is a Python float, br.py:195 */
    const float xmid = F(0.5 * (30.0 + -90.0)), xhalf = F(0.5 * (30.0 - -90.0)); /* br.py:215 */
    orc_enforce_boundary(H, W, in, V0a);
    orc_laplace(H, W, V0a, phi, lap);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        const float V0 = V0a[i];
        const float C = in[n + i], M = in[2 * n + i], Hg = in[3 * n + i], J = in[4 * n + i];
        const float D = in[5 * n + i], Fg = in[6 * n + i], XI = in[7 * n + i];
        float M1, H1, J1 = J, D1 = D, F1 = Fg, XI1 = XI;
        if (cheb) {                                        /* br.py:207-252 */
            float S[9];
            const float x = (V0 - xmid) / xhalf;
            S[0] = 1.0f; S[1] = x;
            for (int k = 2; k <= 8; ++k) S[k] = (2.0f * x) * S[k - 1];
            M1 = rush_larsen(M, br_cheb(cheb + 0 * 9, S), br_cheb(cheb + 2 * 9, S), mdt);
            H1 = rush_larsen(Hg, br_cheb(cheb + 1 * 9, S), br_cheb(cheb + 3 * 9, S), mdt);
            if (nslow > 0) {
                XI1 = rush_larsen(XI, br_cheb(cheb + 4 * 9, S), br_cheb(cheb + 8 * 9, S), mdtn);
                J1 = rush_larsen(J, br_cheb(cheb + 5 * 9, S), br_cheb(cheb + 9 * 9, S), mdtn);
                D1 = rush_larsen(D, br_cheb(cheb + 6 * 9, S), br_cheb(cheb + 10 * 9, S), mdtn);
                F1 = rush_larsen(Fg, br_cheb(cheb + 7 * 9, S), br_cheb(cheb + 11 * 9, S), mdtn);
            }
        } else {                                           /* br.py:175-205 */
            float inf, tau;
            br_inf_tau(V0, ab[2], ab[3], &inf, &tau); M1 = rush_larsen(M, inf, tau, mdt);
            br_inf_tau(V0, ab[4], ab[5], &inf, &tau); H1 = rush_larsen(Hg, inf, tau, mdt);
            if (nslow > 0) {
                br_inf_tau(V0, ab[0], ab[1], &inf, &tau); XI1 = rush_larsen(XI, inf, tau, mdtn);
                br_inf_tau(V0, ab[6], ab[7], &inf, &tau); J1 = rush_larsen(J, inf, tau, mdtn);
                br_inf_tau(V0, ab[8], ab[9], &inf, &tau); D1 = rush_larsen(D, inf, tau, mdtn);
                br_inf_tau(V0, ab[10], ab[11], &inf, &tau); F1 = rush_larsen(Fg, inf, tau, mdtn);
            }
        }
        /* currents, br.py:150-165 (old gate values) */
        const float iK1 = 0.35f * ((4.0f * (expf(0.04f * (V0 + 85.0f)) - 1.0f)) /
                                       (expf(0.08f * (V0 + 53.0f)) + expf(0.04f * (V0 + 53.0f))) +
                                   0.2f * ((V0 + 23.0f) / (1.0f - expf(-0.04f * (V0 + 23.0f)))));
        const float ix1 = ((XI * 0.8f) * (expf(0.04f * (V0 + 77.0f)) - 1.0f)) / expf(0.04f * (V0 + 35.0f));
        const float iNa = (1.0f * (((((4.0f * M) * M) * M) * Hg) * J + 0.005f)) * (V0 - 50.0f);
        const float ECa = F(0.0 - 82.3) - 13.0278f * logf(C);
        const float iCa = ((F(1.0 * 0.09) * D) * Fg) * (V0 - ECa);
        const float I_sum = ((iK1 + ix1) + iNa) + iCa;
        float V1 = (V0 + ddt * lap[i]) - (dtf * I_sum) / 1.0f;             /* br.py:167-168 */
        V1 = fminf(fmaxf(V1, -85.0f), 25.0f);
        const float dC = -1.0e-7f * iCa + 0.07f * (1.0e-7f - C);           /* br.py:170 */
        out[i] = V1;
        out[n + i] = C + dtf * dC;
        out[2 * n + i] = M1; out[3 * n + i] = H1; out[4 * n + i] = J1;
        out[5 * n + i] = D1; out[6 * n + i] = F1; out[7 * n + i] = XI1;
    }
}

/* ------------------------------------------------------------------------- */
/* Courtemanche, court.py:124-429                                              */
/* ------------------------------------------------------------------------- */
enum { cV, cNa_i, c_m, c_h, c_j, cK_i, c_oa, c_oi, c_ua, c_ui, c_xr, c_xs, cCa_i, c_d, c_f, c_f_Ca,
       cCa_rel, c_u, c_v, c_w, cCa_up, COURT_NVAR };

typedef struct {
    float d_inf, tau_d, f_inf, tau_f, tau_w, w_inf, m_inf, tau_m, h_inf, tau_h, j_inf, tau_j;
    float tau_oa, oa_inf, tau_oi, oi_inf, tau_ua, ua_inf, tau_ui, ui_inf, tau_xr, xr_inf, tau_xs, xs_inf;
    float g_Kur, f_NaK, i_NaCaa, i_NaCab, i_K1a, i_Kra;
} court_inter;

#define RCP(x) (1.0f / (x))

/* calc_inter(V, tf), court.py:273-429 */
static inline void court_calc_inter(float V, court_inter *o)
{
    const double R = 8.3143, T = 310, Fd = 96.4867, Cm = 100, Na_o = 140, g_K1 = 0.09, g_Kr = 0.029411765;
    const double Ca_o = 1.8, I_NaCa_max = 1600, K_mNa = 87.5, K_mCa = 1.38, K_sat = 0.1, gamma_ = 0.35, sigma = 1.0;
    const float RT = F(R * T);
    const float eps = V * F(1e-20);                                                   /* :298 */

    o->d_inf = RCP(1.0f + expf((V + 10.0f) / -8.0f));                                  /* :300 */
    {                                                                                  /* :303-307 */
        const float a = 4.579f / (1.0f + expf((V + 10.0f) / F(-6.24)));
        const float vp = V + F(10.0001);
        const float b = (1.0f - expf(vp / F(-6.24))) / ((F(0.0350000) * vp) * (1.0f + expf(vp / F(-6.24))));
        o->tau_d = (fabsf(vp) < F(1.0e-10)) ? a : b;
    }
    {                                                                                  /* :309 */
        const float e = expf((-(V + 28.0f)) / F(6.9));
        o->f_inf = e / (1.0f + e);
    }
    {                                                                                  /* :310 */
        const float s = F(0.0337) * F(0.0337);
        const float vv = (V + 10.0f) * (V + 10.0f);
        o->tau_f = 9.0f * RCP(F(0.0197000) * expf((-s) * vv) + F(0.02));
    }
    {                                                                                  /* :312-316 */
        const float vm = V - F(7.9);
        const float e = expf((-vm) / 5.0f);
        const float b = (6.0f * (1.0f - e)) / (((1.0f + F(0.3) * e) * 1.0f) * vm);
        o->tau_w = (fabsf(vm) < F(1.0e-10)) ? eps + F((6.0 * 0.2) / 1.3) : b;
    }
    o->w_inf = 1.0f - RCP(1.0f + expf((-(V - 40.0f)) / 17.0f));                        /* :318 */
    float alpha, beta;
    {                                                                                  /* :320-329 */
        const float vp = V - F(-47.13);
        const float vq = V + F(47.13);
        alpha = (fabsf(vp) < F(0.001)) ? eps + F(3.2) : (F(0.32) * vq) / (1.0f - expf(F(-0.1) * vq));
        beta = F(0.08) * expf((-V) / 11.0f);
        o->m_inf = alpha / (alpha + beta);
        o->tau_m = RCP(alpha + beta);
    }
    {                                                                                  /* :331-344 */
        const int lo = V < -40.0f;
        alpha = lo ? F(0.135) * expf((V + 80.0f) / F(-6.8)) : eps;
        beta = lo ? F(3.56) * expf(F(0.079) * V) + 310000.0f * expf(F(0.35) * V)
                  : RCP(F(0.13) * (1.0f + expf((V + F(10.66)) / F(-11.1))));
        o->h_inf = alpha / (alpha + beta);
        o->tau_h = RCP(alpha + beta);
    }
    {                                                                                  /* :346-359 */
        const int lo = V < -40.0f;
        alpha = lo ? (((-127140.0f * expf(F(0.2444) * V)) - F(3.474e-05) * expf(F(-0.04391) * V)) * (V + F(37.78))) /
                         (1.0f + expf(F(0.311) * (V + F(79.23))))
                   : eps;
        beta = lo ? (F(0.1212) * expf(F(-0.01052) * V)) / (1.0f + expf(F(-0.1378) * (V + F(40.14))))
                  : (F(0.3) * expf(F(-2.535e-07) * V)) / (1.0f + expf(F(-0.1) * (V + 32.0f)));
        o->j_inf = alpha / (alpha + beta);
        o->tau_j = RCP(alpha + beta);
    }
    const float v10 = V - -10.0f;
    {                                                                                  /* :361-365 */
        alpha = F(0.65) * RCP(expf(v10 / -8.5f) + expf((v10 - 40.0f) / -59.0f));
        beta = F(0.65) * RCP(2.5f + expf((v10 + 72.0f) / 17.0f));
        o->tau_oa = RCP(alpha + beta) / 3.0f;
        o->oa_inf = RCP(1.0f + expf((v10 + F(10.47)) / F(-17.54)));
    }
    {                                                                                  /* :367-371 */
        alpha = RCP(F(18.53) + 1.0f * expf((v10 + F(103.7)) / F(10.95)));
        beta = RCP(F(35.56) + 1.0f * expf((v10 - F(8.74)) / F(-7.44)));
        o->tau_oi = RCP(alpha + beta) / 3.0f;
        o->oi_inf = RCP(1.0f + expf((v10 + F(33.1)) / F(5.3)));
    }
    {                                                                                  /* :373-377 */
        alpha = F(0.65) * RCP(expf(v10 / -8.5f) + expf((v10 - 40.0f) / -59.0f));
        beta = F(0.65) * RCP(2.5f + expf((v10 + 72.0f) / 17.0f));
        o->tau_ua = RCP(alpha + beta) / 3.0f;
        o->ua_inf = RCP(1.0f + expf((v10 + F(20.3)) / F(-9.6)));
    }
    {                                                                                  /* :379-383 */
        alpha = RCP(21.0f + 1.0f * expf((v10 - 195.0f) / -28.0f));
        beta = RCP(expf((v10 - 168.0f) / -16.0f));
        o->tau_ui = RCP(alpha + beta) / 3.0f;
        o->ui_inf = RCP(1.0f + expf((v10 - F(109.45)) / F(27.48)));
    }
    {                                                                                  /* :385-398 */
        const float va = V + F(14.1), vb = V - F(3.3328);
        alpha = (fabsf(va) < F(1.0e-10)) ? eps + F(0.0015) : (F(0.0003) * va) / (1.0f - expf(va / -5.0f));
        beta = (fabsf(vb) < F(1.0e-10)) ? eps + F(0.000378361) : (F(7.3898e-05) * vb) / (expf(vb / F(5.1237)) - 1.0f);
        o->tau_xr = RCP(alpha + beta);
        o->xr_inf = RCP(1.0f + expf(va / -6.5f));
    }
    {                                                                                  /* :400-413 */
        const float vs = V - F(19.9);
        const int sing = fabsf(vs) < F(1.0e-10);
        alpha = sing ? eps + F(0.00068) : (F(4.0e-05) * vs) / (1.0f - expf(vs / -17.0f));
        beta = sing ? eps + F(0.000315) : (F(3.5e-05) * vs) / (expf(vs / 9.0f) - 1.0f);
        o->tau_xs = 0.5f * RCP(alpha + beta);
        o->xs_inf = sqrtf(RCP(1.0f + expf(vs / F(-12.7))));
    }
    o->g_Kur = F(0.005) + F(0.05) / (1.0f + expf((V - 15.0f) / -13.0f));              /* :415 */
    o->f_NaK = RCP((1.0f + F(0.1245) * expf((F(-0.1 * Fd) * V) / RT)) +
                   F(0.0365 * sigma) * expf((F(-Fd) * V) / RT));                       /* :417 */
    const float i_NaCad = F((K_mNa * K_mNa * K_mNa + Na_o * Na_o * Na_o) * (K_mCa + Ca_o)) *
                          (1.0f + F(K_sat) * expf(((F(gamma_ - 1.0) * V) * F(Fd)) / RT)); /* :419 */
    o->i_NaCaa = (F(Cm * I_NaCa_max) * (expf((F(gamma_ * Fd) * V) / RT) * F(Ca_o))) / i_NaCad;      /* :421 */
    o->i_NaCab = (F(Cm * I_NaCa_max) * (expf((F((gamma_ - 1.0) * Fd) * V) / RT) * F(Na_o * Na_o * Na_o))) /
                 i_NaCad;                                                              /* :423 */
    o->i_K1a = F(Cm * g_K1) / (1.0f + expf(F(0.07) * (V + 80.0f)));                    /* :425 */
    o->i_Kra = F(Cm * g_Kr) / (1.0f + expf((V + 15.0f) / F(22.4)));                    /* :427 */
}

/* exports the 30 intermediates in courtemanche.h:105-134 order (for the cross-check
 * against oracle/_ref/generate_table, generate_table.cpp:14-22)                       */
void orc_court_calc_inter(float V, float *out30)
{
    court_inter q;
    court_calc_inter(V, &q);
    const float v[30] = {q.d_inf, q.f_inf, q.tau_w, q.tau_d, q.tau_f, q.w_inf, q.m_inf, q.h_inf, q.j_inf,
                         q.tau_oa, q.tau_oi, q.tau_ua, q.tau_ui, q.tau_xr, q.tau_xs, q.tau_m, q.tau_h, q.tau_j,
                         q.oa_inf, q.oi_inf, q.ua_inf, q.ui_inf, q.xr_inf, q.xs_inf, q.g_Kur, q.f_NaK,
                         q.i_NaCaa, q.i_NaCab, q.i_K1a, q.i_Kra};
    memcpy(out30, v, sizeof v);
}

/* one solve(State), court.py:124-271: ALL 21 new values are produced (State1); the
 * caller picks the fast subset {V,Na_i,m,h} or the slow subset (court.py:94-103).
 * in/out: [21][H][W] in the insertion order of court.py:57-78.  scratch: 2*H*W.        */
void orc_court_step_rate(int H, int W, double dt, double diff, const float *phi, int chronic_flag,
                         double slow_mult, const float *in, float *out, float *scratch);

void orc_court_step(int H, int W, double dt, double diff, const float *phi, int chronic_flag,
                    const float *in, float *out, float *scratch)
{
    orc_court_step_rate(H, W, dt, diff, phi, chronic_flag, 10.0, in, out, scratch);   /* court.py:122 */
}

/* slow_mult = 10: court.py (δt = 10·dt for the slow set, court.py:118-122);
 * slow_mult = 1 : court_ultra.py (δt = dt for every variable, court_ultra.py:127-128)              */
static void court_step_impl(int H, int W, double dt, double diff, const float *phi, int chronic_flag,
                            double slow_mult, const float *in, float *out, float *scratch,
                            const float *us_in, float *us_out);

void orc_court_step_rate(int H, int W, double dt, double diff, const float *phi, int chronic_flag,
                         double slow_mult, const float *in, float *out, float *scratch)
{
    court_step_impl(H, W, dt, diff, phi, chronic_flag, slow_mult, in, out, scratch, NULL, NULL);
}

/* us_in/us_out: court_ultra.py's optional 22nd array `_us_` (court_ultra.py:198-199,221-222,445-450) */
static void court_step_impl(int H, int W, double dt, double diff, const float *phi, int chronic_flag,
                            double slow_mult, const float *in, float *out, float *scratch,
                            const float *us_in, float *us_out)
{
    const long n = (long)H * W;
    float *Va = scratch, *lap = scratch + n;
    /* constants court.py:129-163 (Python numbers, double) */
    const double R = 8.3143, T = 310, Fd = 96.4867, Cm = 100, g_Na = 7.8, Na_o = 140, K_o = 5.4, g_to = 0.1652;
    const double g_Ks = 0.12941176, g_Ca_L = 0.12375, Km_Na_i = 10, Km_K_o = 1.5, i_NaK_max = 0.59933874;
    const double i_CaP_max = 0.275, g_B_Na = 0.0006744375, g_B_Ca = 0.001131, g_B_K = 0, Ca_o = 1.8, K_rel = 30;
    const double tau_tr = 180, I_up_max = 0.005, K_up = 0.00092, Ca_up_max = 15, CMDN_max = 0.05, TRPN_max = 0.07;
    const double CSQN_max = 10, Km_CMDN = 0.00238, Km_TRPN = 0.0005, Km_CSQN = 0.8, V_cell = 20100;
    const double V_i = V_cell * 0.68, tau_f_Ca = 2.0, tau_u = 8.0, V_rel = 0.0048 * V_cell, V_up = 0.0552 * V_cell;
    const double chronic = chronic_flag ? 1.0 : 0.0;
    const double dt_fast = dt, dt_slow = dt * slow_mult;   /* δt(), court.py:118-122 */
    const float mdt_f = F(-dt_fast), mdt_s = F(-dt_slow), dtf = F(dt_fast), dts = F(dt_slow);
    const float ddt = F(diff * dt_fast);
    const float em1_fCa = expm1f(F(-dt_slow / tau_f_Ca)), em1_u = expm1f(F(-dt_slow / tau_u));
    const float RTF = F((R * T) / Fd), RT2F = F((R * T) / (2.0 * Fd)), ViF = F(V_i * Fd);

    orc_enforce_boundary(H, W, in, Va);
    orc_laplace(H, W, Va, phi, lap);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        float s[COURT_NVAR];
        for (int k = 0; k < COURT_NVAR; ++k) s[k] = in[k * n + i];
        const float V = Va[i];
        court_inter q;
        court_calc_inter(V, &q);
        float *o = out + i;
        /* gates, court.py:175-189 */
        o[c_d * n] = rush_larsen(s[c_d], q.d_inf, q.tau_d, mdt_s);
        o[c_f * n] = rush_larsen(s[c_f], q.f_inf, q.tau_f, mdt_s);
        o[c_w * n] = rush_larsen(s[c_w], q.w_inf, q.tau_w, mdt_s);
        o[c_m * n] = rush_larsen(s[c_m], q.m_inf, q.tau_m, mdt_f);
        o[c_h * n] = rush_larsen(s[c_h], q.h_inf, q.tau_h, mdt_f);
        o[c_j * n] = rush_larsen(s[c_j], q.j_inf, q.tau_j, mdt_s);
        o[c_oa * n] = rush_larsen(s[c_oa], q.oa_inf, q.tau_oa, mdt_s);
        o[c_oi * n] = rush_larsen(s[c_oi], q.oi_inf, q.tau_oi, mdt_s);
        o[c_ua * n] = rush_larsen(s[c_ua], q.ua_inf, q.tau_ua, mdt_s);
        o[c_ui * n] = rush_larsen(s[c_ui], q.ui_inf, q.tau_ui, mdt_s);
        o[c_xr * n] = rush_larsen(s[c_xr], q.xr_inf, q.tau_xr, mdt_s);
        o[c_xs * n] = rush_larsen(s[c_xs], q.xs_inf, q.tau_xs, mdt_s);
        const float f_Ca_inf = RCP(1.0f + s[cCa_i] / F(0.00035));
        o[c_f_Ca * n] = rush_larsen_c(s[c_f_Ca], f_Ca_inf, em1_fCa);
        /* potassium currents, court.py:191-204 */
        const float E_K = RTF * logf(F(K_o) / s[cK_i]);
        const float vEK = V - E_K;
        const float i_K1 = q.i_K1a * vEK;
        const float i_to = ((F((1.0 - 0.5 * chronic) * Cm * g_to) * powf(s[c_oa], 3.0f)) * s[c_oi]) * vEK;
        const float i_Kur = (((F((1.0 - 0.5 * chronic) * Cm) * q.g_Kur) * powf(s[c_ua], 3.0f)) * s[c_ui]) * vEK;
        const float i_Kr = (q.i_Kra * s[c_xr]) * vEK;
        const float i_Ks = (F(Cm * g_Ks) * (s[c_xs] * s[c_xs])) * vEK;
        const float i_NaK = ((F(Cm * i_NaK_max) * q.f_NaK) / (1.0f + sqrtf(powf(F(Km_Na_i) / s[cNa_i], 3.0f)))) *
                            F(K_o / (K_o + Km_K_o));
        const float i_B_K = F(Cm * g_B_K) * vEK;
        o[cK_i * n] = s[cK_i] +
                      ((2.0f * i_NaK - (((((i_K1 + i_to) + i_Kur) + i_Kr) + i_Ks) + i_B_K)) / ViF) * dts;
        /* sodium, court.py:206-215 */
        const float E_Na = RTF * logf(F(Na_o) / s[cNa_i]);
        float i_Na = (((F(Cm * g_Na) * powf(s[c_m], 3.0f)) * s[c_h]) * s[c_j]) * (V - E_Na);
        if (us_in) {
            /* alpha_us, beta_us, us_infinity, tau_us: court_ultra.py:445-450 */
            const float a_us = F(3e-5) * (0.5f * (1.0f - tanhf((V - F(-83.0)) / F(23.0))));
            const float b_us = F(1e-5) * (0.5f * (1.0f + tanhf((V - F(-83.0 + 30)) / F(23.0))));
            us_out[i] = rush_larsen(us_in[i], a_us / (a_us + b_us), RCP(a_us + b_us), mdt_s);   /* :198-199 */
            i_Na = i_Na * us_in[i];                                                              /* :221-222 */
        }
        const float i_NaCa = q.i_NaCaa * powf(s[cNa_i], 3.0f) - q.i_NaCab * s[cCa_i];
        const float i_B_Na = F(Cm * g_B_Na) * (V - E_Na);
        o[cNa_i * n] = s[cNa_i] + ((-3.0f * i_NaK - ((3.0f * i_NaCa + i_B_Na) + i_Na)) / ViF) * dtf;
        /* calcium currents and the voltage, court.py:217-229 */
        const float i_st = 0.0f;
        const float i_Ca_L = (((F((1.0 - 0.7 * chronic) * Cm * g_Ca_L) * s[c_d]) * s[c_f]) * s[c_f_Ca]) * (V - 65.0f);
        const float i_CaP = (F(Cm * i_CaP_max) * s[cCa_i]) / (F(0.0005) + s[cCa_i]);
        const float E_Ca = RT2F * logf(F(Ca_o) / s[cCa_i]);
        const float i_B_Ca = F(Cm * g_B_Ca) * (V - E_Ca);
        const float isum = (((((((((((i_Na + i_K1) + i_to) + i_Kur) + i_Kr) + i_Ks) + i_B_Na) + i_B_Ca) + i_NaK) +
                              i_CaP) + i_NaCa) + i_Ca_L) + i_st;
        const float DV = V + ((-isum) / F(Cm)) * dtf;
        o[cV * n] = DV + ddt * lap[i];
        /* SR release / uptake, court.py:232-256 */
        const float i_rel = (((F(K_rel) * (s[c_u] * s[c_u])) * s[c_v]) * s[c_w]) * (s[cCa_rel] - s[cCa_i]);
        const float i_tr = (s[cCa_up] - s[cCa_rel]) / F(tau_tr);
        {
            const float t = s[cCa_rel] + F(Km_CSQN);
            o[cCa_rel * n] = s[cCa_rel] + ((i_tr - i_rel) * RCP(1.0f + F(CSQN_max * Km_CSQN) / (t * t))) * dts;
        }
        const float Fn = 1000.0f * (F(1.0e-15 * V_rel) * i_rel -
                                    F(1.0e-15 / (2.0 * Fd)) * (0.5f * i_Ca_L - F(0.2) * i_NaCa));
        const float u_inf = RCP(1.0f + expf((-(Fn - F(3.4175e-13))) / F(1.367e-15)));
        o[c_u * n] = rush_larsen_c(s[c_u], u_inf, em1_u);
        const float tau_v = F(1.91) + F(2.09) * u_inf;
        const float v_inf = 1.0f - RCP(1.0f + expf((-(Fn - F(6.835e-14))) / F(1.367e-15)));
        o[c_v * n] = rush_larsen(s[c_v], v_inf, tau_v, mdt_s);
        const float i_up = F(I_up_max) / (1.0f + F(K_up) / s[cCa_i]);
        const float i_up_leak = (F(I_up_max) * s[cCa_up]) / F(Ca_up_max);
        o[cCa_up * n] = s[cCa_up] + (i_up - (i_up_leak + (i_tr * F(V_rel)) / F(V_up))) * dts;
        /* intracellular calcium, court.py:258-265 */
        const float B1 = (2.0f * i_NaCa - ((i_CaP + i_Ca_L) + i_B_Ca)) / F(2.0 * V_i * Fd) +
                         (F(V_up) * (i_up_leak - i_up) + i_rel * F(V_rel)) / F(V_i);
        const float t1 = s[cCa_i] + F(Km_TRPN), t2 = s[cCa_i] + F(Km_CMDN);
        const float B2 = (1.0f + F(TRPN_max * Km_TRPN) / (t1 * t1)) + F(CMDN_max * Km_CMDN) / (t2 * t2);
        o[cCa_i * n] = s[cCa_i] + (B1 / B2) * dts;
    }
}

/* ------------------------------------------------------------------------- */
/* tick drivers (the schedule of define()/run()), used by tests and cpu_baseline */
/* ------------------------------------------------------------------------- */

/* Fenton: one tick = 10 x solve (fenton.py:133-138).  slab: [4][H][W], updated in place;
 * tmp: [4][H][W] + 2*H*W scratch                                                      */
void orc_fenton_run(int H, int W, double dt, double diff, const float *phi, float *slab, float *tmp,
                    int nsteps)
{
    const long n = (long)H * W;
    float *a = slab, *b = tmp, *scr = tmp + 4 * n;
    for (int s = 0; s < nsteps; ++s) {
        orc_fenton_step(H, W, dt, diff, phi, a, b, scr);
        float *t = a; a = b; b = t;
    }
    if (a != slab) memcpy(slab, a, 4 * n * sizeof(float));
}

/* Beeler-Reuter: one tick = 5 x solve(.,1), or with skip: solve(.,5) then 4 x solve(.,0)
 * (br.py:98-107).  slab [8][H][W]; tmp [8][H][W] + 2*H*W                               */
void orc_br_run(int H, int W, double dt, double diff, const float *phi, const float *cheb, int skip,
                float *slab, float *tmp, int nticks)
{
    const long n = (long)H * W;
    float *a = slab, *b = tmp, *scr = tmp + 8 * n;
    for (int t = 0; t < nticks; ++t)
        for (int s = 0; s < 5; ++s) {
            const int nslow = skip ? (s == 0 ? 5 : 0) : 1;
            orc_br_step(H, W, dt, diff, phi, cheb, nslow, a, b, scr);
            float *x = a; a = b; b = x;
        }
    if (a != slab) memcpy(slab, a, 8 * n * sizeof(float));
}

static const int court_is_fast[COURT_NVAR] = {1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

/* Courtemanche: every tick assigns the fast set (court.py:42,94-102 via ionic.py:203);
 * when tick % 10 == 0 the driver fires 'slow' (court.py:615-617): a SECOND evaluation of
 * solve on the post-fast state, assigning the other 17.  `tick0` = index of the first
 * tick of this call.  slab [21][H][W]; tmp [21][H][W] + 2*H*W                          */
void orc_court_run(int H, int W, double dt, double diff, const float *phi, int chronic, float *slab,
                   float *tmp, int tick0, int nticks, int slow_every)
{
    const long n = (long)H * W;
    float *scr = tmp + COURT_NVAR * n;
    for (int t = tick0; t < tick0 + nticks; ++t) {
        orc_court_step(H, W, dt, diff, phi, chronic, slab, tmp, scr);
        for (int k = 0; k < COURT_NVAR; ++k)
            if (court_is_fast[k]) memcpy(slab + k * n, tmp + k * n, n * sizeof(float));
        if (slow_every > 0 && t % slow_every == 0) {
            orc_court_step(H, W, dt, diff, phi, chronic, slab, tmp, scr);
            for (int k = 0; k < COURT_NVAR; ++k)
                if (!court_is_fast[k]) memcpy(slab + k * n, tmp + k * n, n * sizeof(float));
        }
    }
}

/* fenton_simple.py / fenton_jit.py: the Laplacian is tf.nn.depthwise_conv2d with the 3x3 kernel
 * [[.5,1,.5],[1,-6,1],[.5,1,.5]] and padding='SAME' (zeros outside the grid), fenton_simple.py:38-49.  TensorFlow does
 * not specify its accumulation order: row-major over the kernel here, as tests/golden/_standin does.        */
void orc_laplace_zeropad(int H, int W, const float *X, float *out)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
#define ZP(dr, dc) (((r + (dr)) < 0 || (r + (dr)) >= H || (c + (dc)) < 0 || (c + (dc)) >= W) ? 0.0f : X[(long)(r + (dr)) * W + (c + (dc))])
            float a = 0.5f * ZP(-1, -1);
            a = a + ZP(-1, 0);
            a = a + 0.5f * ZP(-1, 1);
            a = a + ZP(0, -1);
            a = a + (-6.0f * ZP(0, 0));
            a = a + ZP(0, 1);
            a = a + 0.5f * ZP(1, -1);
            a = a + ZP(1, 0);
            a = a + 0.5f * ZP(1, 1);
#undef ZP
            out[(long)r * W + c] = a;
        }
}

/* Fenton4vSimple.solve (fenton_simple.py:122-134) x nsteps: enforce_boundary, differentiate on the raw state, Euler
 * update with the zero-padded Laplacian of the boundary-enforced potential.  slab [4][H][W]; tmp [4][H][W] + 2*H*W */
void orc_fenton_simple_run(int H, int W, double dt, double diff, float *slab, float *tmp, int nsteps)
{
    const long n = (long)H * W;
    float *a = slab, *b = tmp, *U0 = tmp + 4 * n, *lap = tmp + 5 * n;
    const float dtf = F(dt), ddt = F(diff * dt);
    for (int s = 0; s < nsteps; ++s) {
        orc_enforce_boundary(H, W, a, U0);
        orc_laplace_zeropad(H, W, U0, lap);
        orc_fenton_diff(n, a, a + n, a + 2 * n, a + 3 * n, b, b + n, b + 2 * n, b + 3 * n);   /* dU dV dW dS */
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) {
            const float dU = b[i], dV = b[n + i], dW = b[2 * n + i], dS = b[3 * n + i];
            b[i] = (U0[i] + dtf * dU) + ddt * lap[i];                    /* fenton_simple.py:129 */
            b[n + i] = a[n + i] + dtf * dV;
            b[2 * n + i] = a[2 * n + i] + dtf * dW;
            b[3 * n + i] = a[3 * n + i] + dtf * dS;
        }
        float *t = a; a = b; b = t;
    }
    if (a != slab) memcpy(slab, a, 4 * n * sizeof(float));
}

void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* court_ultra.py:107-111: one solve per tick, all 21 variables assigned, single rate */
void orc_court_ultra_run(int H, int W, double dt, double diff, const float *phi, int chronic, float *slab,
                         float *tmp, int nticks)
{
    const long n = (long)H * W;
    float *scr = tmp + COURT_NVAR * n;
    for (int t = 0; t < nticks; ++t) {
        orc_court_step_rate(H, W, dt, diff, phi, chronic, 1.0, slab, tmp, scr);
        memcpy(slab, tmp, COURT_NVAR * n * sizeof(float));
    }
}

/* court_ultra.py with config['ultra_slow']: 22 arrays, `_us_` last.  slab [22][H][W]; tmp [22][H][W] + 2*H*W */
void orc_court_ultra_us_run(int H, int W, double dt, double diff, const float *phi, int chronic, float *slab,
                            float *tmp, int nticks)
{
    const long n = (long)H * W;
    float *scr = tmp + (COURT_NVAR + 1) * n;
    for (int t = 0; t < nticks; ++t) {
        court_step_impl(H, W, dt, diff, phi, chronic, 1.0, slab, tmp, scr, slab + COURT_NVAR * n,
                        tmp + COURT_NVAR * n);
        memcpy(slab, tmp, (COURT_NVAR + 1) * n * sizeof(float));
    }
}

/* the two extra intermediates of court_ultra.py:445-450 for one voltage */
void orc_court_us_inter(float V, float *us_inf, float *tau_us)
{
    const float a_us = F(3e-5) * (0.5f * (1.0f - tanhf((V - F(-83.0)) / F(23.0))));
    const float b_us = F(1e-5) * (0.5f * (1.0f + tanhf((V - F(-83.0 + 30)) / F(23.0))));
    *us_inf = a_us / (a_us + b_us);
    *tau_us = RCP(a_us + b_us);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
