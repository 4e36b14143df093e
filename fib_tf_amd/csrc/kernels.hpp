// kernels.hpp — the fused stencil + reaction kernels (gfx950).
//
// tick_kernel<M,P,MODE,K,TX,TY,NT,PHASE>
//   One workgroup advances one TX x TY tile of the grid by K sub-steps in a single launch
//   (temporal blocking).  K = 1 is the classic LDS-tiled fused step; K > 1 keeps the tile resident:
//   the potential lives in a double-buffered LDS tile with a halo, every other state variable and
//   the phase-field coefficients stay in registers of the thread that owns the cell, and HBM/L2 is
//   touched once per K steps.  The compute box is the tile grown by K-1 cells per side; its outer
//   ring goes stale by one cell per sub-step, so after K steps exactly the tile itself is still
//   exact — redundant rim compute instead of a grid-wide barrier per step (a launch boundary or a
//   grid barrier costs more than a whole 512x512 step).
//
//   Boundary conditions.  enforce_boundary (ionic.py:107-113) followed by the REFLECT pad of
//   laplace (ionic.py:49-50) means: every stencil tap at (r+dr, c+dc) reads the raw potential at
//   (clamp(r+dr,1,H-2), clamp(c+dc,1,W-2)).  The LDS tile is therefore filled through that clamp,
//   and after each sub-step the cells on domain row/col 1 and H-2/W-2 also refresh the border and
//   ghost copies next to them.  Border cells keep their own raw value in a register: Fenton's
//   reaction term reads it (fenton.py:101), nobody else does.
//
//   Cell -> thread map: the compute box is flattened row-major and dealt round-robin to the NT
//   threads, so consecutive lanes touch consecutive LDS words for all nine taps (conflict-free for
//   any tile shape) and every lane of every wave has work.
#pragma once
#include "models.hpp"

namespace fib {

struct Geo {
    int H, W;        // rows / cols of this slab (pitch == W)
    int Hg;          // rows of the whole grid
    int row_off;     // global row of local row 0
    int r0, r1;      // local rows [r0, r1) this launch computes and stores
    int tiles_x, ntiles;
};

template <int NVAR>
struct PtrTab {
    const float *in[NVAR];
    float *out[NVAR];
};

struct PhaseTab {    // derived from ϕ once at set_phase (ionic.py:78-80)
    const float *dpy;   // ϕ[r+1,c] - ϕ[r-1,c]   (REFLECT-padded)
    const float *dpx;   // ϕ[r,c+1] - ϕ[r,c-1]
    const float *q4;    // 4 * ϕ[r,c]
};

enum : unsigned {
    F_ACTIVE = 1u, F_WLDS = 2u, F_STORE = 4u,
    F_TOP = 8u, F_BOT = 16u, F_LEFT = 32u, F_RIGHT = 64u,
    F_TOP2 = 128u, F_BOT2 = 256u, F_LEFT2 = 512u, F_RIGHT2 = 1024u,
    F_EDGE_V = F_TOP | F_BOT, F_EDGE_H = F_LEFT | F_RIGHT
};

static FIB_DEV int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// 9-point Laplacian in the reference's evaluation order, ionic.py:51-53
static FIB_DEV float stencil9(float N, float S, float Wv, float E, float NW, float SW, float NE, float SE, float C)
{
    const float l = (((N + S) + Wv) + E) + 0.5f * (((NW + SW) + NE) + SE);
    return l - 6.0f * C;
}
// phase-field correction, ionic.py:78-80, from the pre-differenced ϕ terms
template <class P>
static FIB_DEV float phase_term(float N, float S, float Wv, float E, float dpy, float dpx, float q4)
{
    return P::div((S - N) * dpy + (E - Wv) * dpx, q4);
}

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous run of tiles so
// that the halos neighbouring tiles share are served by the same L2.  Speed only, never correctness.
static FIB_DEV int xcd_tile(int b, int ntiles)
{
    const int per = (ntiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

template <class M, class P, int MODE, int K, int TX, int TY, int NT, bool PHASE>
__global__ void __launch_bounds__(NT)
tick_kernel(Geo g, PtrTab<M::NVAR> pt, PhaseTab ph, typename M::Consts k, int sub0)
{
    constexpr int NV = M::NVAR;
    constexpr int CX = TX + 2 * (K - 1), CY = TY + 2 * (K - 1);   // compute box
    constexpr int LP = CX + 2, LQ = CY + 2;                        // LDS tile (box + ring)
    constexpr int NC = CX * CY, CPT = (NC + NT - 1) / NT, NL = LP * LQ;
    constexpr unsigned WMASK = M::mask(MODE);
    __shared__ float lds[(K > 1) ? 2 : 1][NL];

    const int tile = xcd_tile(blockIdx.x, g.ntiles);
    if (tile >= g.ntiles) return;                                  // whole workgroup, before any barrier
    const int tid = threadIdx.x;
    const int by = tile / g.tiles_x, bx = tile - by * g.tiles_x;
    const int x0 = bx * TX, y0 = g.r0 + by * TY;                   // tile origin (local rows)
    const int cx0 = x0 - (K - 1), cy0 = y0 - (K - 1);              // compute-box origin

    // ---- potential tile, through the boundary clamp -------------------------------------------
    const float *vin = pt.in[0];
    for (int i = tid; i < NL; i += NT) {
        const int ly = i / LP, lx = i - ly * LP;
        int yy = clampi(cy0 - 1 + ly + g.row_off, 1, g.Hg - 2) - g.row_off;
        yy = clampi(yy, 0, g.H - 1);                               // stay inside this slab
        const int xx = clampi(cx0 - 1 + lx, 1, g.W - 2);
        const float v = vin[(size_t)yy * g.W + xx];
        lds[0][i] = v;
        if (K > 1) lds[K > 1 ? 1 : 0][i] = v;
    }

    // ---- per-cell registers -------------------------------------------------------------------
    float s[CPT][NV];
    float pdy[CPT], pdx[CPT], pq4[CPT];
    int li[CPT], off[CPT];
    unsigned fl[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int e = tid + j * NT;
        const bool valid = e < NC;
        const int ee = valid ? e : 0;
        const int cyy = ee / CX, cxx = ee - cyy * CX;
        li[j] = (cyy + 1) * LP + cxx + 1;
        const int gy = cy0 + cyy, gx = cx0 + cxx, gyg = gy + g.row_off;
        const bool indom = valid && gx >= 0 && gx < g.W && gyg >= 0 && gyg < g.Hg && gy >= 0 && gy < g.H;
        off[j] = clampi(gy, 0, g.H - 1) * g.W + clampi(gx, 0, g.W - 1);
#pragma unroll
        for (int v = 0; v < NV; ++v) s[j][v] = pt.in[v][off[j]];
        if (PHASE) {
            pdy[j] = ph.dpy[off[j]];
            pdx[j] = ph.dpx[off[j]];
            pq4[j] = ph.q4[off[j]];
        }
        const bool border = gyg == 0 || gyg == g.Hg - 1 || gx == 0 || gx == g.W - 1;
        unsigned f = 0;
        if (indom) {
            f |= F_ACTIVE;
            if (!border) {
                f |= F_WLDS;
                if (gyg == 1) f |= F_TOP | (cyy >= 1 ? F_TOP2 : 0u);
                if (gyg == g.Hg - 2) f |= F_BOT | (cyy <= CY - 2 ? F_BOT2 : 0u);
                if (gx == 1) f |= F_LEFT | (cxx >= 1 ? F_LEFT2 : 0u);
                if (gx == g.W - 2) f |= F_RIGHT | (cxx <= CX - 2 ? F_RIGHT2 : 0u);
            }
            if (gy >= y0 && gy < min(y0 + TY, g.r1) && gx >= x0 && gx < x0 + TX) f |= F_STORE;
        }
        fl[j] = f;
    }
    __syncthreads();

    // ---- K fused sub-steps --------------------------------------------------------------------
#pragma unroll 1
    for (int st = 0; st < K; ++st) {
        const float *A = lds[(K > 1) ? (st & 1) : 0];
        float *B = lds[(K > 1) ? ((st & 1) ^ 1) : 0];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            if (fl[j] & F_ACTIVE) {
                const int i = li[j];
                const float N = A[i - LP], S = A[i + LP], Wv = A[i - 1], E = A[i + 1];
                const float NW = A[i - LP - 1], SW = A[i + LP - 1], NE = A[i - LP + 1], SE = A[i + LP + 1];
                const float C = A[i];
                float l = stencil9(N, S, Wv, E, NW, SW, NE, SE, C);
                if (PHASE) l = l + phase_term<P>(N, S, Wv, E, pdy[j], pdx[j], pq4[j]);   // ionic.py:58
                M::template step<P, MODE>(s[j], C, l, k, sub0 + st);
            }
        }
        if (K > 1 && st + 1 < K) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const unsigned f = fl[j];
                const int i = li[j];
                const float u = s[j][0];
                if (f & F_WLDS) B[i] = u;
                // refresh border + ghost copies (enforce_boundary + REFLECT), only in waves that own
                // such cells
                if (__builtin_amdgcn_ballot_w64((f & (F_EDGE_V | F_EDGE_H)) != 0)) {
                    if (f & F_TOP) B[i - LP] = u;
                    if (f & F_TOP2) B[i - 2 * LP] = u;
                    if (f & F_BOT) B[i + LP] = u;
                    if (f & F_BOT2) B[i + 2 * LP] = u;
                    if (f & F_LEFT) B[i - 1] = u;
                    if (f & F_LEFT2) B[i - 2] = u;
                    if (f & F_RIGHT) B[i + 1] = u;
                    if (f & F_RIGHT2) B[i + 2] = u;
                    if ((f & F_EDGE_V) && (f & F_EDGE_H)) {                // the four domain corners
                        const unsigned vf[4] = {F_TOP, F_TOP2, F_BOT, F_BOT2};
                        const int vo[4] = {-LP, -2 * LP, LP, 2 * LP};
                        const unsigned hf[4] = {F_LEFT, F_LEFT2, F_RIGHT, F_RIGHT2};
                        const int ho[4] = {-1, -2, 1, 2};
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                if ((f & vf[a]) && (f & hf[b])) B[i + vo[a] + ho[b]] = u;
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- write back the tile ------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        if (fl[j] & F_STORE) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if ((WMASK >> v) & 1u) pt.out[v][off[j]] = s[j][v];
        }
    }
}

// Pointwise re-evaluation without the stencil: Courtemanche's 'slow' op (court.py:103,615-617).
// Vc is the boundary-enforced potential of the cell, read straight through the clamp.
template <class M, class P, int MODE>
__global__ void __launch_bounds__(256)
pointwise_kernel(Geo g, PtrTab<M::NVAR> pt, typename M::Consts k)
{
    constexpr int NV = M::NVAR;
    constexpr unsigned WMASK = M::mask(MODE);
    const int n = (g.r1 - g.r0) * g.W;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int gy = g.r0 + e / g.W, gx = e % g.W;
        int yy = clampi(gy + g.row_off, 1, g.Hg - 2) - g.row_off;
        yy = clampi(yy, 0, g.H - 1);
        const int xx = clampi(gx, 1, g.W - 2);
        const float Vc = pt.in[0][(size_t)yy * g.W + xx];
        const int o = gy * g.W + gx;
        float s[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) s[v] = pt.in[v][o];
        M::template step<P, MODE>(s, Vc, 0.0f, k, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if ((WMASK >> v) & 1u) pt.out[v][o] = s[v];
    }
}

// The building blocks of IonicModel as stand-alone array ops (IonicModel.enforce_boundary / laplace /
// phase_field / rush_larsen are public methods of the reference, ionic.py:44-123).  Same device
// functions as the fused kernel; used for unit-level parity tests.
enum { OP_BOUNDARY = 0, OP_LAPLACE = 1, OP_PHASE = 2, OP_RUSH_LARSEN = 3 };
template <class P>
__global__ void unit_op_kernel(int op, int H, int W, const float *a, const float *b, const float *c,
                               const float *ph3, float mdt, float *out)
{
    const int n = H * W;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int y = e / W, x = e % W;
        if (op == OP_BOUNDARY) {
            out[e] = a[clampi(y, 1, H - 2) * W + clampi(x, 1, W - 2)];
        } else if (op == OP_RUSH_LARSEN) {
            out[e] = rush_larsen<P>(a[e], b[e], c[e], mdt);
        } else {   // REFLECT pad: ghost index -1 -> 1, H -> H-2
            const int yn = y == 0 ? 1 : y - 1, ys = y == H - 1 ? H - 2 : y + 1;
            const int xw = x == 0 ? 1 : x - 1, xe = x == W - 1 ? W - 2 : x + 1;
            const float N = a[yn * W + x], S = a[ys * W + x], Wv = a[y * W + xw], E = a[y * W + xe];
            float r = 0.0f;
            if (op == OP_LAPLACE)
                r = stencil9(N, S, Wv, E, a[yn * W + xw], a[ys * W + xw], a[yn * W + xe], a[ys * W + xe], a[e]);
            if (ph3) {
                const float f = phase_term<P>(N, S, Wv, E, ph3[e], ph3[n + e], ph3[2 * n + e]);
                r = (op == OP_LAPLACE) ? r + f : f;
            }
            out[e] = r;
        }
    }
}

// ϕ -> (dpy, dpx, q4), REFLECT-padded in GLOBAL coordinates (ionic.py:75-80)
__global__ void phase_prep_kernel(Geo g, const float *phi, float *dpy, float *dpx, float *q4)
{
    const int n = g.H * g.W;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int y = e / g.W, x = e % g.W, yg = y + g.row_off;
        int yn = yg - 1, ys = yg + 1, xw = x - 1, xe = x + 1;
        if (yn < 0) yn = 1;
        if (ys > g.Hg - 1) ys = g.Hg - 2;
        if (xw < 0) xw = 1;
        if (xe > g.W - 1) xe = g.W - 2;
        yn = clampi(yn - g.row_off, 0, g.H - 1);
        ys = clampi(ys - g.row_off, 0, g.H - 1);
        dpy[e] = phi[ys * g.W + x] - phi[yn * g.W + x];
        dpx[e] = phi[y * g.W + xe] - phi[y * g.W + xw];
        q4[e] = 4.0f * phi[e];
    }
}

// pace op, ionic.py:144-163:  pot = max(pot, s), s = v inside the global rectangle, min_v outside
__global__ void pace_kernel(Geo g, float *pot, int r0, int r1, int c0, int c1, float v, float min_v)
{
    const int n = g.H * g.W;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int yg = e / g.W + g.row_off, x = e % g.W;
        const float sv = (yg >= r0 && yg < r1 && x >= c0 && x < c1) ? v : min_v;
        pot[e] = fmaxf(pot[e], sv);
    }
}

}  // namespace fib
