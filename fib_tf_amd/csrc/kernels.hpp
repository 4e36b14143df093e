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
FIB_TAG_BEGIN

struct Geo {
    int H, W;        // rows / cols of this slab
    int pitch;       // floats between consecutive rows of ONE state array: W for the planar slab
                     // [nvar][H][W]; nvar*W for the row-interleaved slab [H][nvar][W] that row-block shards
                     // use (there the g halo rows of all arrays are one contiguous block = one message)
    int Hg;          // rows of the whole grid
    int row_off;     // global row of local row 0
    int r0, r1;      // local rows [r0, r1) this launch computes and stores ...
    int rb0, rb1;    // ... and, when ty_a < tile rows, a second band [rb0, rb1) served by the same launch
    int ty_a;        //     (the two edge strips of a row block); tile rows >= ty_a belong to the second band
    int tiles_x, ntiles;
};

// tile row `by` -> first local row of the tile and the end of the band it belongs to
static FIB_DEV void tile_rows(const Geo &g, int by, int TY, int &y0, int &rend)
{
    if (by < g.ty_a) {
        y0 = g.r0 + by * TY;
        rend = g.r1;
    } else {
        y0 = g.rb0 + (by - g.ty_a) * TY;
        rend = g.rb1;
    }
}

template <int NVAR>
struct PtrTab {
    const float *in[NVAR];
    float *out[NVAR];
};

struct PhaseTab {    // derived from ϕ once at set_phase (ionic.py:78-80)
    const float *dpy;   // ϕ[r+1,c] - ϕ[r-1,c]   (REFLECT-padded)
    const float *dpx;   // ϕ[r,c+1] - ϕ[r,c-1]
    const float *q4;    // 4 * ϕ[r,c]
    const float *r4;    // RN(1 / q4): lets the division by 4ϕ run as a 3-instruction exact form
    const float *pyr;   // RN(dpy * r4), RN(dpx * r4): all the fast policy needs of ϕ (its phase term is two FMAs on these
    const float *pxr;   //   products; formed per launch until round 3, now once at set_phase: 8 B per cell instead of 16)
    const float *phi;   // ϕ itself: one-sub-step launches stage a ϕ tile in LDS and difference it on the fly
                        // (4 B per cell of traffic instead of 16; the K-fused kernels read the prepared arrays
                        // once per K sub-steps and keep them in registers)
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
// The 3x3 convolution of fenton_simple.py:38-49 ([[.5,1,.5],[1,-6,1],[.5,1,.5]], padding SAME).  TensorFlow does
// not specify its accumulation order; this is the kernel's row-major order, as in tests/golden/_standin.
static FIB_DEV float stencil9_conv(float N, float S, float Wv, float E, float NW, float SW, float NE, float SE, float C)
{
    float a = 0.5f * NW;
    a = a + N;
    a = a + 0.5f * NE;
    a = a + Wv;
    a = a + (-6.0f * C);
    a = a + E;
    a = a + 0.5f * SW;
    a = a + S;
    return a + 0.5f * SE;
}
// phase-field correction, ionic.py:78-80, from the pre-differenced ϕ terms
// (the division is always the correctly rounded form: the Laplacian incl. its phase term is pure
// arithmetic and stays bit-identical to the reference under both arithmetic policies)
template <class P>
static FIB_DEV float phase_term(float N, float S, float Wv, float E, float dpy, float dpx, float q4, float r4)
{
    return Exact::divc((S - N) * dpy + (E - Wv) * dpx, q4, r4);
}

// The Laplacian of the FUSED kernels under the two arithmetic policies.  Exact: the reference's operations, one
// rounding each (stencil9 / phase_term).  Fast: re-associated row by row with the scalings (0.5*, -6*) and the phase
// quotient contracted into FMAs; every fused kernel uses these two functions, so fusion depth and tile shape still
// never change a bit of the result within a policy.  (The stand-alone array ops
// IonicModel.laplace / phase_field keep the exact form under both policies.)
template <class P>
static FIB_DEV float lap9(float N, float S, float Wv, float E, float NW, float SW, float NE, float SE, float C)
{
    if constexpr (same_type<P, Fast>::value) {
        // row by row: a(row) = centre + 0.5 (west + east) for the rows above and below, b = (west + east) - 6 centre for
        // the cell's own row.  The R cells of a lane share these row terms (a of the row below cell r is a of the row
        // above cell r+2, west + east of a row serves both forms): 19 operations for three cells instead of 24.
        const float an = __builtin_fmaf(0.5f, NW + NE, N), as = __builtin_fmaf(0.5f, SW + SE, S);
        return (an + as) + __builtin_fmaf(-6.0f, C, Wv + E);
    } else {
        return stencil9(N, S, Wv, E, NW, SW, NE, SE, C);
    }
}
template <class P>
static FIB_DEV float add_phase(float lap, float N, float S, float Wv, float E, float dpy, float dpx, float q4, float r4)
{
    if constexpr (same_type<P, Fast>::value)   // (dpx*r4 and dpy*r4 do not change during a launch: formed once, before the step loop)
        return __builtin_fmaf(E - Wv, dpx * r4, __builtin_fmaf(S - N, dpy * r4, lap));
    else
        return lap + phase_term<P>(N, S, Wv, E, dpy, dpx, q4, r4);
}

// What a thread of a K-fused kernel keeps of ϕ per cell, by arithmetic policy.  Exact: the four prepared arrays (the
// quotient by 4ϕ is the correctly rounded one).  Fast: the two products dpy*r4, dpx*r4 — prepared by phase_prep_kernel
// with the same single rounding the kernels used to apply per launch, so results are bit-identical to the four-array form.
template <class P>
struct PhaseCoef {
    float dpy, dpx, q4, r4;
    FIB_DEV void load(const PhaseTab &ph, int op)
    {
        dpy = ph.dpy[op];
        dpx = ph.dpx[op];
        q4 = ph.q4[op];
        r4 = ph.r4[op];
    }
    FIB_DEV float add(float lap, float N, float S, float Wv, float E) const
    {
        return add_phase<P>(lap, N, S, Wv, E, dpy, dpx, q4, r4);
    }
};
template <>
struct PhaseCoef<Fast> {
    float ay, ax;
    FIB_DEV void load(const PhaseTab &ph, int op)
    {
        ay = ph.pyr[op];
        ax = ph.pxr[op];
    }
    FIB_DEV float add(float lap, float N, float S, float Wv, float E) const
    {
        return __builtin_fmaf(E - Wv, ax, __builtin_fmaf(S - N, ay, lap));
    }
};

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous run of tiles so
// that the halos neighbouring tiles share are served by the same L2.  Speed only, never correctness.
static FIB_DEV int xcd_tile(int b, int ntiles)
{
    const int per = (ntiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

// FIBHIP_ZEROPAD — the Laplacian of fenton_simple.py: taps outside the grid read 0 and the nine products are
// accumulated in the kernel's row-major order.  A compile-time property of the model type (FentonZP), so that the
// other models' kernels carry none of it; tick_kernel only.
template <class M, class = void>
struct ZeroPadOf {
    static constexpr bool value = false;
};
template <class M>
struct ZeroPadOf<M, void_of<decltype(M::ZEROPAD)>> {
    static constexpr bool value = M::ZEROPAD;
};

// does MODE ask for two evaluations in one launch (Courtemanche::MODE_FASTSLOW)?
template <class M, class = void>
struct TwoPass {
    static constexpr bool of(int) { return false; }
    static constexpr int first(int mode) { return mode; }
    static constexpr int second(int mode) { return mode; }
};
template <class M>
struct TwoPass<M, void_of<decltype(M::MODE_FASTSLOW)>> {
    static constexpr bool of(int mode) { return mode == M::MODE_FASTSLOW; }
    static constexpr int first(int mode) { return mode == M::MODE_FASTSLOW ? M::MODE_FAST : mode; }
    static constexpr int second(int mode) { return mode == M::MODE_FASTSLOW ? M::MODE_SLOW : mode; }
};

template <class M, class P, int MODE, int K, int TX, int TY, int NT, bool PHASE>
__global__ void __launch_bounds__(NT)
tick_kernel(Geo g, PtrTab<M::NVAR> pt, PhaseTab ph, typename M::Consts k, int sub0)
{
    constexpr int NV = M::NVAR;
    constexpr int CX = TX + 2 * (K - 1), CY = TY + 2 * (K - 1);   // compute box
    constexpr int LP = CX + 2, LQ = CY + 2;                        // LDS tile (box + ring)
    constexpr int NC = CX * CY, CPT = (NC + NT - 1) / NT, NL = LP * LQ;
    constexpr unsigned WMASK = M::mask(MODE);
    constexpr bool PHI_TILE = PHASE && K == 1;
    constexpr bool ZP = ZeroPadOf<M>::value;
    __shared__ float lds[(K > 1) ? 2 : 1][NL];
    __shared__ float lphi[PHI_TILE ? NL : 1];

    const int tile = xcd_tile(blockIdx.x, g.ntiles);
    if (tile >= g.ntiles) return;                                  // whole workgroup, before any barrier
    auto &&kk = M::pinned(k);
    const int tid = threadIdx.x;
    const int by = tile / g.tiles_x, bx = tile - by * g.tiles_x;
    int y0, rend;
    tile_rows(g, by, TY, y0, rend);
    const int x0 = bx * TX;                                        // tile origin (local rows: y0)
    const int cx0 = x0 - (K - 1), cy0 = y0 - (K - 1);              // compute-box origin

    // ---- potential tile, through the boundary clamp -------------------------------------------
    const float *vin = pt.in[0];
    for (int i = tid; i < NL; i += NT) {
        const int ly = i / LP, lx = i - ly * LP;
        int yy = clampi(cy0 - 1 + ly + g.row_off, 1, g.Hg - 2) - g.row_off;
        yy = clampi(yy, 0, g.H - 1);                               // stay inside this slab
        const int xx = clampi(cx0 - 1 + lx, 1, g.W - 2);
        float v = vin[(size_t)yy * g.pitch + xx];
        if (ZP) {                                                  // outside the grid: 0 (conv2d padding='SAME')
            const int gyy = cy0 - 1 + ly + g.row_off, gxx = cx0 - 1 + lx;
            if (gyy < 0 || gyy > g.Hg - 1 || gxx < 0 || gxx > g.W - 1) v = 0.0f;
        }
        lds[0][i] = v;
        if (K > 1) lds[K > 1 ? 1 : 0][i] = v;
        if (PHI_TILE) {                                            // ϕ is REFLECT-padded, not clamped (ionic.py:75-76)
            int py = cy0 - 1 + ly + g.row_off, px = cx0 - 1 + lx;
            py = py < 0 ? -py : (py > g.Hg - 1 ? 2 * (g.Hg - 1) - py : py);
            px = px < 0 ? -px : (px > g.W - 1 ? 2 * (g.W - 1) - px : px);
            py = clampi(py - g.row_off, 0, g.H - 1);
            px = clampi(px, 0, g.W - 1);
            lphi[i] = ph.phi[(size_t)py * g.W + px];
        }
    }

    // ---- per-cell registers -------------------------------------------------------------------
    float s[CPT][NV];
    PhaseCoef<P> pc[CPT];
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
        const int oy = clampi(gy, 0, g.H - 1), ox = clampi(gx, 0, g.W - 1);
        off[j] = oy * g.pitch + ox;
#pragma unroll
        for (int v = 0; v < NV; ++v) s[j][v] = pt.in[v][off[j]];
        if (PHASE && !PHI_TILE) pc[j].load(ph, oy * g.W + ox);    // (the phase arrays are always planar)
        const bool border = gyg == 0 || gyg == g.Hg - 1 || gx == 0 || gx == g.W - 1;
        unsigned f = 0;
        if (indom) {
            f |= F_ACTIVE;
            if (!border) {
                f |= F_WLDS;
                constexpr bool ghost = !ZP;                        // the cells beyond the border stay 0 there
                if (gyg == 1) f |= F_TOP | (cyy >= 1 && ghost ? F_TOP2 : 0u);
                if (gyg == g.Hg - 2) f |= F_BOT | (cyy <= CY - 2 && ghost ? F_BOT2 : 0u);
                if (gx == 1) f |= F_LEFT | (cxx >= 1 && ghost ? F_LEFT2 : 0u);
                if (gx == g.W - 2) f |= F_RIGHT | (cxx <= CX - 2 && ghost ? F_RIGHT2 : 0u);
            }
            if (gy >= y0 && gy < min(y0 + TY, rend) && gx >= x0 && gx < x0 + TX) f |= F_STORE;
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
                float l = ZP ? stencil9_conv(N, S, Wv, E, NW, SW, NE, SE, C) : lap9<P>(N, S, Wv, E, NW, SW, NE, SE, C);
                if (PHI_TILE) {     // same arithmetic as phase_prep_kernel + add_phase (IEEE division = Exact::divc)
                    const float dy = lphi[i + LP] - lphi[i - LP], dx = lphi[i + 1] - lphi[i - 1];
                    if constexpr (same_type<P, Fast>::value) {
                        const float q4 = 4.0f * lphi[i];
                        l = add_phase<P>(l, N, S, Wv, E, dy, dx, q4, 1.0f / q4);           // r4 exactly as phase_prep_kernel forms it
                    } else {
                        l = l + ((S - N) * dy + (E - Wv) * dx) / (4.0f * lphi[i]);
                    }
                } else if (PHASE) {
                    l = pc[j].add(l, N, S, Wv, E);                                         // ionic.py:58
                }
                M::template step<P, TwoPass<M>::first(MODE)>(s[j], C, l, kk, sub0 + st);
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

    // ---- second evaluation on the post-update state (Courtemanche's 'slow' op, court.py:612-617) ----------
    // It sees the boundary-enforced NEW potential: a border cell reads its inward neighbour's new value, which
    // the host guarantees to be a cell of this same tile (fibhip.hip: lazy_fusable).
    if constexpr (TwoPass<M>::of(MODE)) {
        static_assert(K == 1, "two-pass modes are one sub-step per launch");
        __syncthreads();                                           // all taps of the old tile have been read
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (fl[j] & F_ACTIVE) lds[0][li[j]] = s[j][0];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            if (fl[j] & F_ACTIVE) {
                const int cyy = li[j] / LP - 1, cxx = li[j] - (cyy + 1) * LP - 1;
                const int ty = clampi(cy0 + cyy + g.row_off, 1, g.Hg - 2) - g.row_off;
                const int tx = clampi(cx0 + cxx, 1, g.W - 2);
                const float Vc = lds[0][(ty - cy0 + 1) * LP + (tx - cx0 + 1)];
                M::template step<P, TwoPass<M>::second(MODE)>(s[j], Vc, 0.0f, kk, 0);
            }
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


#ifdef FIB_STAMPS   // diagnostic build only (tools/ubench/stamp_strip.hip): per-wave s_memtime stamps
__device__ unsigned long long fib_stamps[4096 * 16];
#define FIB_STAMP(slot)                                                                          \
    do {                                                                                         \
        if ((threadIdx.x & 63) == 0 && (slot) < 16)                                              \
            fib_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) % 4096 * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// (tools/ubench/stamp_mt.hip: the phases of a tick boundary inside a multi-tick launch; the last boundary's values stay)
__device__ unsigned long long fib_bstamps[4096 * 16];
#define FIB_BSTAMP(slot)                                                                         \
    do {                                                                                         \
        if ((threadIdx.x & 63) == 0)                                                             \
            fib_bstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) % 4096 * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define FIB_BSTAMP_WAIT() __builtin_amdgcn_s_waitcnt(0x0070)      /* vmcnt(0) lgkmcnt(0): the phase's loads have landed */
// (round 4: when a wave ARRIVES at the barrier of a sub-step — arithmetic done, new potential written — against FIB_STAMP's
// "barrier passed and next window read": what a sub-step spends computing and what it spends waiting)
__device__ unsigned long long fib_wstamps[4096 * 16];
#define FIB_WSTAMP(slot)                                                                         \
    do {                                                                                         \
        if ((threadIdx.x & 63) == 0 && (slot) < 16)                                              \
            fib_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) % 4096 * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FIB_WSTAMP(slot) do { } while (0)
#define FIB_STAMP(slot) do { } while (0)
#define FIB_BSTAMP(slot) do { } while (0)
#define FIB_BSTAMP_WAIT() do { } while (0)
#endif

// strip_kernel<M,P,MODE,K,TX,TY,R,PHASE> — the K > 1 workhorse.
//   Same temporal blocking as tick_kernel, different work layout: the LDS tile is exactly 64 words
//   wide (compute box CX = TX + 2(K-1) <= 62 plus the two ring columns), lane l of every wave owns
//   column l, and wave w owns the R consecutive rows [wR, wR+R) of the compute box.  Consequences:
//     * every LDS access of a wave is 64 consecutive words: conflict-free, and a thread reads the
//       3 x (R+2) window of its R cells once per sub-step (3(R+2)/R instead of 9 reads per cell);
//     * rows are wave-uniform, so the rows that have gone stale (one more ring per sub-step) are
//       skipped with scalar branches — the box shrinks in y as the sub-steps proceed;
//     * the vertical border/ghost refresh is wave-uniform too; only the two edge columns need a
//       per-lane predicate.
// ---- several ticks in ONE launch: what a tile needs from its neighbours between two ticks ---------------------------
// A launch of `nticks` ticks keeps every workgroup resident on its tile: the tile's own cells stay in registers from
// tick to tick, and only the K-deep rim of the compute box (which went stale during the tick) is re-read — from what
// the up to eight neighbouring tiles published at the end of their tick.  No grid-wide barrier: a tile waits for its
// neighbours only.
//   * Payload: an exchange buffer of 16-byte cells [2 parities][NVAR/4][H*W] (the state arrays themselves are read at
//     the first tick and written at the last one only).  Stores and loads are write-through / L1-bypassing (sc1): the
//     vector L1 of a CU is never refreshed by another CU's stores and the XCDs' L2s are not coherent with each other
//     (MI355X_MICROARCH.md, "inter-workgroup visibility"); every wave drains its stores (s_waitcnt vmcnt(0)) before the
//     workgroup's barrier, after which ONE lane raises the tile's epoch word.
//   * Epoch words: one per tile, 256 bytes apart (words sharing a line serialise the pollers of a whole tile row on one
//     memory channel: measured 5.5 us per tick boundary against 3.1, tools/ubench/handoff.hip); they count ticks over
//     the life of the handle (epoch0 = their common value when the launch starts), so nothing is reset between launches.
//   * Two parities: a tile overwrites parity p two ticks after it published there, and by then every neighbour has
//     published the tick in between, for which it had to read p first.
//   * Every wait is BOUNDED (s_memrealtime, MT_WAIT_TICKS of 10 ns): a tile that gives up raises err[0], which every
//     waiting tile also polls, so the launch drains instead of hanging; the host reports the failure at its next
//     synchronisation point.  The host only uses this kernel when all tiles can be resident at once (tiles <= CUs)
//     and never runs two such launches of one process at the same time.
struct MtArgs {
    float *xb;            // exchange buffer
    unsigned *epoch;      // one word per tile, MT_EPOCH_STRIDE words apart
    unsigned *err;        // [0]: a tile gave up waiting; [MT_EPOCH_STRIDE]: the host's word as tile 0 passed it on; [2 * MT_EPOCH_STRIDE]:
                          // tiles that stopped where it said (counted)
    unsigned epoch0;      // value of every epoch word when the launch starts
    unsigned ticks_id;    // low half: ticks this launch advances; high half: the launch's id, 1 .. 65535 (the host's word names
                          // the launch it is meant for).  One word, and the host's word behind the tiles' words of `snap_flag`
                          // instead of a pointer of its own: Beeler-Reuter's kernel spills scalar registers as it is, and three
                          // more kernel arguments cost it 2.5 % (same-box A/B)
    // read-back inside the launch (fibhip.hip `run-ahead`): every tile also writes array `snap_var` of the state the launch
    // STARTS from into page-locked host memory during its first ticks and then raises its word in `snap_flag` (host memory
    // too, 64 bytes apart) to `snap_seq` — the host has the frame while the launch is still computing
    float *snap;
    unsigned *snap_flag;  // page-locked HOST memory (device address): MT_MAX_TILES words MT_SNAP_STRIDE apart, then the host's
                          // word {launch id << 16 | n}, written by the host while a launch runs — n = MT_CANCEL: not wanted any
                          // more, else: stop after n ticks (always allocated, with or without a frame to deliver)
    unsigned snap_seq;
    int snap_var;         // low byte: which array the frame is; the other three: the bound on a tile's wait for its neighbours in
                          // milliseconds (0 = MT_WAIT_TICKS; packed, not an argument of its own: see ticks_id)
};
#define FIB_STR2(x) #x
#define FIB_STR(x) FIB_STR2(x)
#ifndef FIB_POLL_SLEEP
#define FIB_POLL_SLEEP 1
#endif
#ifndef FIB_B_LASTWAVE
#define FIB_B_LASTWAVE 0
#endif
#ifndef FIB_B_INBOX
#define FIB_B_INBOX 0
#endif
constexpr int MT_SNAP_STRIDE = 16;                    // words between two tiles' words in snap_flag
constexpr int MT_MAX_TILES = 1024;                    // epoch / snap words allocated per handle (only grids of <= ncu tiles use them)
constexpr int MT_HOST_WORD_AT = MT_MAX_TILES * MT_SNAP_STRIDE;   // the host's word, in words from snap_flag
constexpr int MT_GIVEUP_WORD = 8;                     // ... and, this many words behind it, the id of the launch whose tile gave up first
constexpr int MT_EPOCH_STRIDE = 64;                   // words (256 bytes)
constexpr unsigned MT_CANCEL = 0xFFFFu;               // the host's word, low half: this launch is not wanted any more
constexpr unsigned long long MT_WAIT_TICKS = 200000000ull;   // 2 s of the 100 MHz s_memrealtime clock
constexpr int MT_WAIT_SHIFT = 17;                            // MtArgs::snap_var's upper three bytes count 2^17 such ticks (1.31 ms)
// The multi-tick kernels sit at their register limits (Fenton: 128 vector registers, and scalar registers spilled into vector
// lanes): an edit ANYWHERE in them — one `& 0xFF` in the prologue — re-draws the register allocation and moves the kernel by 2-3 %
// (round 4, tools/r04_h.sh -> profiles/r04_ab_kernel_variants.txt: round 3's text 12.28 us per tick, the same with the give-up
// word set by compare-and-swap 12.60, by a plain store 12.29, the wait bound as a shift 12.60 or 12.24 depending on what else is
// in, ...).  These switches keep the equivalent forms that were measured; the defaults are the combination that lost nothing
// against round 3's kernel for Fenton (12.32 / 12.28) and is the fastest measured for the other two (rounding-faithful Fenton 18.8
// against 21.5, Beeler-Reuter 15.36 against 15.75).
#ifndef FIB_WAIT_FORM
#define FIB_WAIT_FORM 0     // the wait bound: 0 = milliseconds in MtArgs::snap_var's upper bytes (0: MT_WAIT_TICKS), 1 = units of 2^17 ticks, 2 = MT_WAIT_TICKS
#endif
#ifndef FIB_GIVEUP_CAS
#define FIB_GIVEUP_CAS 0    // the give-up word: a plain store of the launch's id (every tile of a launch writes the same id, and a launch
                            // queued behind one that gave up finds the word before its own wait can run out) / compare-and-swap
#endif

typedef unsigned fib_v4u __attribute__((ext_vector_type(4)));
typedef float fib_v4f __attribute__((ext_vector_type(4)));

// M::pinned(k), or M::pinned_spare(k) where a model offers it and the kernel asks for it (one register less)
template <class M, class = void>
struct HasPinnedSpare {
    static constexpr bool value = false;
};
template <class M>
struct HasPinnedSpare<M, void_of<decltype(&M::pinned_spare)>> {
    static constexpr bool value = true;
};
template <class A, class B>
struct SameType {
    static constexpr bool value = false;
};
template <class A>
struct SameType<A, A> {
    static constexpr bool value = true;
};
template <class M, bool SPARE, class C>
static FIB_DEV decltype(auto) pinned_for(const C &k)
{
    if constexpr (SPARE && HasPinnedSpare<M>::value)
        return M::pinned_spare(k);
    else
        return M::pinned(k);
}

// strip_kernel<M,P,MODE,K,TX,TY,R,PHASE> / strip_mt_kernel<...> share this body (MT = several ticks per launch)
template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE, bool MT>
static FIB_DEV void strip_body(const Geo &g, const PtrTab<M::NVAR> &pt, const PhaseTab &ph, const typename M::Consts &k, int sub0,
                               const MtArgs &mt)
{
    constexpr int NV = M::NVAR;
    constexpr int CX = TX + 2 * (K - 1), CY = TY + 2 * (K - 1);
    static_assert(CX <= 62 && K > 1, "strip_kernel: compute box must fit 62 lanes");
    static_assert(!MT || (CX == 62 && TX >= K && TY >= K), "multi-tick launches: 62-column box, the rim inside the eight neighbours");
    constexpr int NW = (CY + R - 1) / R;
    constexpr int LP = 64, LQ = NW * R + 2, NL = LP * LQ;
    constexpr unsigned WMASK = M::mask(MODE);
    // The LDS image of the potential.  Strips of an ODD number of rows: row-major, one dword per cell, the 3 x (R+2) window read
    // as ds_read_b32.  Strips of an EVEN number of rows (round 4): rows 2k and 2k+1 of a column form one aligned 8-byte word —
    // element (row, col) at dword (row >> 1) * 128 + 2 col + (row & 1) — and the window, which then starts on an even row and has an
    // even number of rows, is read as 3 x (R+2)/2 ds_read_b64.  The LDS array serves a wave's ds_read_b64 in the two cycles it takes
    // for a ds_read_b32 (MI355X_MICROARCH.md, LDS: 256 against 128 B/clk), and the window reload of ALL waves at once, right behind
    // a sub-step's barrier, is LDS-bandwidth time on everybody's critical path (stamped build, profiles/r04_stamps_substeps.txt:
    // the LAST wave to reach the barrier still waits 450-560 cycles for its window — 15 waves x 15 dwords x 2 cycles).
    // Beeler-Reuter's two-row strips: 15.4 -> 15.1 us per tick.  Three-row strips would need two copies of the tile program (a
    // window starts on an even row in every other wave only): built and measured — the registers it costs the Fenton kernel, which
    // sits at its 128, outweigh the LDS cycles (12.3 -> 13.0 us per tick; four-row strips with the paired image: 13.8).
    // (only where ONE workgroup has the compute unit to itself: with several resident, as on grids beyond 704^2, another workgroup
    // computes while this one reloads its windows, and the opaque addresses of the separate ds_read_b64 only cost — Beeler-Reuter
    // 2048^2 with the paired image in strip_kernel: 180.9 -> 184.5 us per tick)
    constexpr bool PAIR = MT && (R % 2 == 0);
    constexpr int SPARE = PAIR ? R + 6 : R + 4;
    __shared__ __attribute__((aligned(16))) float lds[2][NL + SPARE * 64];   // (+ spare rows: see `wi`)
    __shared__ int mt_abort;
    __shared__ unsigned mt_arrive[1];                               // (FIB_B_LASTWAVE, measured and not taken: waves whose stores have been acknowledged)
    __shared__ float snapl[MT ? NW * R * 64 : 1];                    // multi-tick launches: the frame's values, parked for one tick

    const int tile = xcd_tile(blockIdx.x, g.ntiles);
    if (tile >= g.ntiles) return;
    FIB_STAMP(0);
    if (MT && threadIdx.x == 0) {
        mt_abort = 0;                                               // (read after the first tick's barriers)
        if (FIB_B_LASTWAVE) mt_arrive[0] = 0u;                         // (first added to after the first tick's barriers)
    }
    auto &&kk = pinned_for<M, (MT && SameType<P, Exact>::value)>(k);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int by = tile / g.tiles_x, bx = tile - by * g.tiles_x;
    int y0, rend;
    tile_rows(g, by, TY, y0, rend);
    const int x0 = bx * TX;
    const int cx0 = x0 - (K - 1), cy0 = y0 - (K - 1);
    const int gx = cx0 - 1 + lane;                                  // this lane's global column
    const int c0 = wave * R;                                        // first box row of this wave's strip

    // ---- the three tap columns of this lane, through the boundary clamp ------------------------------------
    // enforce_boundary + REFLECT: a tap at column c reads the raw potential of column clamp(c, 1, W-2).  The clamp
    // is a property of the LANE, so it lives in the tap ADDRESSES (computed once here) and the tile only ever holds
    // raw values at their own positions: border and ghost columns need no copies after a sub-step, and a tile at
    // the left or right edge of the domain costs what an interior tile costs.
    const int bW = clampi(gx - 1, 1, g.W - 2), bC = clampi(gx, 1, g.W - 2), bE = clampi(gx + 1, 1, g.W - 2);
    const int jW = clampi(bW - (cx0 - 1), 0, 63), jC = clampi(bC - (cx0 - 1), 0, 63), jE = clampi(bE - (cx0 - 1), 0, 63);
    auto brow = [&](int grow) {                                     // global row -> local row through the boundary clamp
        return clampi(clampi(grow, 1, g.Hg - 2) - g.row_off, 0, g.H - 1);
    };

    // ---- prologue: all global loads are issued before anything waits; the first sub-step's 3 x (R+2) window comes
    // straight from global memory (no tile fill, no barrier before the step loop)
    const float *vin = pt.in[0];
    float win[R + 2][3];
#pragma unroll
    for (int q = 0; q < R + 2; ++q) {
        const float *row = vin + (size_t)brow(cy0 + c0 - 1 + q + g.row_off) * g.pitch;
        win[q][0] = row[bW];
        win[q][1] = row[bC];
        win[q][2] = row[bE];
    }
    const bool lane_in = lane >= 1 && lane <= CX && gx >= 0 && gx < g.W;
    const bool col_border = gx == 0 || gx == g.W - 1;
    const bool store_col = lane_in && gx >= x0 && gx < x0 + TX;
    const bool wr = lane_in && !col_border;                         // this lane's cells are somebody's taps
    float s[R][NV];
    PhaseCoef<P> pc[R];
    int off[R];
    bool own[R];                                                    // the cells this thread stores: the tile proper
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = cy0 + c0 + r;
        const int oy = clampi(gy, 0, g.H - 1), ox = clampi(gx, 0, g.W - 1);
        off[r] = oy * g.pitch + ox;
        own[r] = store_col && gy >= y0 && gy < min(y0 + TY, rend) && gy < g.H;
#pragma unroll
        for (int v = 0; v < NV; ++v) s[r][v] = pt.in[v][off[r]];
        if (PHASE) pc[r].load(ph, oy * g.W + ox);                   // (the phase arrays are always planar)
    }
    // read-back inside the launch: this tile's cells of one array of the state the launch STARTS from go straight into
    // page-locked host memory, by system-scope (write-through) stores: plain stores stay in the L2 — frames came back with
    // cells of the previous read-back — and the L2 write-back of a system-scope release fence in every tile at the same
    // moment cost 15 us per launch.  The values are taken here; the stores are issued at the start of the SECOND tick and the
    // tile's word is raised at the boundary after it: issued here they had to drain at the first boundary together with the
    // tile's exchange stores, and a 1 MiB frame of PCIe writes kept every tile waiting ~10 us there.  (Launches of two ticks
    // have one boundary: stores here, word there.  Tried: the frame in three parts over three ticks, values re-read from the
    // slab — no faster, and the extra registers cost 2.7 %.)
    // (parked in LDS meanwhile: in registers they pushed the kernel to its 128-register budget and into scratch)
    const int snap_at = (MT && mt.snap) ? ((int)(mt.ticks_id & 0xFFFFu) >= 3 ? 1 : 0) : -1;
    if constexpr (MT) {
        if (mt.snap) {                                              // (wave-uniform)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float x = s[r][0];
#pragma unroll
#if FIB_WAIT_FORM == 2
                for (int v = 1; v < NV; ++v) x = mt.snap_var == v ? s[r][v] : x;
#else
                for (int v = 1; v < NV; ++v) x = (mt.snap_var & 0xFF) == v ? s[r][v] : x;
#endif
                snapl[(c0 + r) * 64 + lane] = x;                    // (read back by the same thread: no barrier needed)
            }
        }
    }
    // rows of the compute box that can still be correct at sub-step st: [lo0+st.., hi0-st..) unless
    // the box reaches the domain edge on that side (no staleness enters through a real boundary)
    const bool top_open = cy0 + g.row_off > 0, bot_open = cy0 + CY + g.row_off < g.Hg;
    // window addresses of this strip (paired image: of the 8-byte word that holds its first two rows)
    const int aW = PAIR ? (c0 >> 1) * 128 + 2 * jW : c0 * LP + jW, aC = PAIR ? (c0 >> 1) * 128 + 2 * jC : c0 * LP + jC,
              aE = PAIR ? (c0 >> 1) * 128 + 2 * jE : c0 * LP + jE;
    // dword offset of the row k rows below a strip's first row (tile row c0 + 1: always odd in the paired image), from that row's
    // address as `cell_at` gives it (paired image: the address of the row's 8-byte word)
    auto ro = [](int k) constexpr { return PAIR ? ((1 + k) >> 1) * 128 + ((1 + k) & 1) : k * LP; };
    auto cell_at = [](int row, int col) { return PAIR ? (row >> 1) * 128 + 2 * col : row * LP + col; };
    constexpr int SPARE_ROW = PAIR ? LQ + 3 : LQ + 2;                    // a strip's worth of rows nobody reads, behind the tile
    auto window = [&](const float *Bq, float (&w)[R + 2][3]) {          // the strip's 3 x (R+2) window out of the image Bq
        if constexpr (PAIR) {
            typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int m = 0; m < (R + 2) / 2; ++m) {
                // (the second and later words of a column from addresses the compiler cannot relate to the first: it would fuse
                // two reads into one ds_read2st64_b64, which the LDS serves in 8 cycles where two ds_read_b64 take 4)
                int oW = aW + m * 128, oC = aC + m * 128, oE = aE + m * 128;
                if (m > 0) asm volatile("" : "+v"(oW));
                const v2f a = *reinterpret_cast<const v2f *>(Bq + oW);
                if (m > 0) asm volatile("" : "+v"(oC));
                const v2f b = *reinterpret_cast<const v2f *>(Bq + oC);
                if (m > 0) asm volatile("" : "+v"(oE));
                const v2f c = *reinterpret_cast<const v2f *>(Bq + oE);
                w[2 * m][0] = a.x; w[2 * m + 1][0] = a.y;
                w[2 * m][1] = b.x; w[2 * m + 1][1] = b.y;
                w[2 * m][2] = c.x; w[2 * m + 1][2] = c.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < R + 2; ++q) {
                w[q][0] = Bq[aW + q * LP];
                w[q][1] = Bq[aC + q * LP];
                w[q][2] = Bq[aE + q * LP];
            }
        }
    };
    // ---- everything about the strip's rows that does not change from sub-step to sub-step, as wave-uniform scalars
    // (the step loop then spends its scalar instructions on two min/max and a few bit tests)
    const int g0 = cy0 + c0 + g.row_off;                            // global row of the strip's first row
    // rows that may ever be computed: inside the grid and inside this slab
    const int ra_fix = max(max(0, -g0), -(cy0 + c0));
    const int rb_fix = min(min(R, CY - c0), min(g.Hg - g.row_off, g.H) - (cy0 + c0));
    unsigned pub = 0;                                               // rows whose value other cells tap (not a border row)
    int top_r = -1, bot_r = -1;                                     // the strip row that is the grid's row 1 / H-2, if any
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (g0 + r != 0 && g0 + r != g.Hg - 1) pub |= 1u << r;
        if (g0 + r == 1) top_r = r;
        if (g0 + r == g.Hg - 2) bot_r = r;
    }
    // lanes whose cells nobody taps write to spare rows behind the tile instead of being masked out (row offsets
    // -2 .. R+1 are applied to this address)
    const int wi = wr ? cell_at(c0 + 1, lane) : cell_at(SPARE_ROW, lane);
    FIB_STAMP(1);
    // all prologue loads are consumed by the first sub-step anyway: drain them once here, so that the
    // compiler does not carry per-use `s_waitcnt vmcnt(n)` into every iteration of the step loop
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0) only
    FIB_STAMP(2);

    // two sub-steps per loop body: the loop-carried state then needs no register moves at the back edge (measured:
    // -5 % under Exact, -7 % for R = 4 strips, nothing for Fast R = 3; tools/ubench/diag_strip.hip)
#ifndef FIB_STEP_UNROLL
#define FIB_STEP_UNROLL 2
#endif
    constexpr int STEP_UNROLL = FIB_STEP_UNROLL;
    // A strip that stays whole for all K sub-steps — no row of it goes stale inside the tick, none is a border or a
    // ghost-source row of the grid (more than half of a tile's strips, and the ones that carry its own cells) — runs the
    // step loop without any of the row bookkeeping: ~35 scalar instructions and a dozen branches fewer per sub-step.
    // Only in the multi-tick kernel: the second copy of the step loop costs ~30 registers, which a grid with several
    // workgroups per compute unit pays with its occupancy (measured with the specialisation in every strip kernel: 512x512
    // multi-tick 12.53 -> 11.93 us per tick, but 4096x4096 388 -> 627 us and 1024x1024 35.0 -> 38.3); multi-tick grids have
    // at most one workgroup per compute unit by construction.
    // (and only where the registers are there and the bookkeeping is a visible share of the sub-step: four-row strips spilled
    // with the second loop — at 16 waves per workgroup the budget is 128 registers — and Beeler-Reuter's eight arrays with ~270
    // instructions per cell ran 1.5-3 % slower with it; both keep one loop)
#ifndef FIB_PRE_FIRST
#define FIB_PRE_FIRST 0
#endif
#ifndef FIB_WHOLE_NVR
#define FIB_WHOLE_NVR 12
#endif
    constexpr bool WHOLE_LOOP = MT && NV * R <= FIB_WHOLE_NVR;
    const bool whole = WHOLE_LOOP && ra_fix == 0 && rb_fix == R && (!top_open || c0 >= K - 1) && (!bot_open || c0 + R <= CY - (K - 1)) &&
                       pub == (1u << R) - 1u && top_r < 0 && bot_r < 0;
#ifdef FIB_LOOP_ALIGN           // (experiment: does the placement of the tick loop in the instruction stream matter?)
    asm volatile(".p2align " FIB_STR(FIB_LOOP_ALIGN));
#endif
#pragma unroll 1
    for (int tick = 0;; ++tick) {
    // the host's word is read over PCIe by ONE thread of the grid at the START of a tick and looked at at the tick's end: the
    // round trip hides behind the sub-steps, at the price of seeing the word a tick late (see the tick boundary below).
    // Issued HERE — nothing is outstanding at this point (the counter of outstanding loads is in order: in front of the rim
    // loads of a boundary the PCIe round trip would hold their wait back) — and defined and used inside one pass of the tick
    // loop: carried around the loop's back edge, the compiler waited for the load right where it was issued.
    unsigned hw = 0u;
    if constexpr (MT) {
        if (tile == 0 && threadIdx.x == 0) {
            typedef const __attribute__((address_space(1))) unsigned *gptr;     // (global, not flat: a flat load also counts
            gptr p = (gptr)(mt.snap_flag + MT_HOST_WORD_AT);                                        // as an LDS operation, which every barrier waits for)
            asm volatile("" : "+s"(p));                             // (a new address for the compiler in every tick: it had moved
            hw = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the load in front of the tick loop)
        }
    }
    if constexpr (MT) {
        if (tick == snap_at) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (own[r]) __hip_atomic_store(mt.snap + off[r], snapl[(c0 + r) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through
        }
    }
    if (WHOLE_LOOP && whole) {
#pragma unroll STEP_UNROLL
        for (int st = 0; st < K; ++st) {
            float *B = lds[(st & 1) ^ 1];
            float lp[R], cc[R];
#if FIB_PRE_FIRST
            // (experiment) the part of the update that needs no neighbour — the whole reaction term — is ISSUED before anything
            // waits for the window the barrier's other side has just requested from the LDS
            float dU[R];
            M::template stepN_pre<P, MODE, R>(s, dU, kk);
#if FIB_PRE_FIRST >= 2
            // two statements the compiler keeps in order: the first consumes everything the reaction term produced, the second
            // "produces" the window — so the wait for the LDS sits between them
#pragma unroll
            for (int r = 0; r < R; ++r) {
                asm volatile("" ::"v"(dU[r]));
#pragma unroll
                for (int v = 1; v < NV; ++v) asm volatile("" ::"v"(s[r][v]));
            }
#pragma unroll
            for (int r = 0; r < R + 2; ++r) asm volatile("" : "+v"(win[r][0]), "+v"(win[r][1]), "+v"(win[r][2]));
#else
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float l = lap9<P>(win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2], win[r][0], win[r + 2][0],
                                  win[r][2], win[r + 2][2], win[r + 1][1]);
                if (PHASE) l = pc[r].add(l, win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2]);
                lp[r] = l;
                cc[r] = win[r + 1][1];
            }
            M::template stepN_post<P, MODE, R>(s, dU, cc, lp, kk);
#else
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float l = lap9<P>(win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2], win[r][0], win[r + 2][0],
                                  win[r][2], win[r + 2][2], win[r + 1][1]);
                if (PHASE) l = pc[r].add(l, win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2]);
                lp[r] = l;
                cc[r] = win[r + 1][1];
            }
            if constexpr (M::HAS_VEC) {
                M::template stepN<P, MODE, R>(s, cc, lp, kk, sub0 + st);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) M::template step<P, MODE>(s[r], cc[r], lp[r], kk, sub0 + st);
            }
#endif
            if (st + 1 < K) {
#pragma unroll
                for (int r = 0; r < R; ++r) B[wi + ro(r)] = s[r][0];
                FIB_WSTAMP(st);
#ifndef FIB_DIAG_NO_BARRIER
                __syncthreads();
#endif
#ifndef FIB_DIAG_NO_RELOAD
                window(B, win);
#endif
#if FIB_PRE_FIRST == 3
                // (the state as the reaction term sees it exists only on this side of the barrier and of the window's reads)
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("" : "+v"(s[r][0]), "+v"(s[r][1]), "+v"(s[r][2]), "+v"(s[r][3])::"memory");
#endif
            }
            FIB_STAMP(3 + st);
        }
    } else {
#pragma unroll STEP_UNROLL
    for (int st = 0; st < K; ++st) {
        float *B = lds[(st & 1) ^ 1];
        // rows [ra, rb) of this wave's strip are live at this sub-step (wave-uniform): the box loses one ring per
        // sub-step on every side that is not the domain's edge
        const int ra = top_open ? max(ra_fix, st - c0) : ra_fix;
        const int rb = bot_open ? min(rb_fix, CY - st - c0) : rb_fix;
        if (ra == 0 && rb == R) {
            // ---- whole strip live: one straight-line block.  The R cells of a lane are independent,
            // so the scheduler can interleave their dependency chains; the 3 x (R+2) window is read once.
            float lp[R], cc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float l = lap9<P>(win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2], win[r][0], win[r + 2][0],
                                  win[r][2], win[r + 2][2], win[r + 1][1]);
                if (PHASE) l = pc[r].add(l, win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2]);
                lp[r] = l;
                cc[r] = win[r + 1][1];
            }
            if constexpr (M::HAS_VEC) {
                M::template stepN<P, MODE, R>(s, cc, lp, kk, sub0 + st);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) M::template step<P, MODE>(s[r], cc[r], lp[r], kk, sub0 + st);
            }
        } else {
            // (a strip of which only some rows are still live: at most two strips of a tile at any sub-step.  Running
            // the block above on all R rows instead was measured: 1 % slower under Fast, 4 % under Exact.)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r >= ra && r < rb) {                            // scalar branch
                    float l = lap9<P>(win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2], win[r][0], win[r + 2][0],
                                      win[r][2], win[r + 2][2], win[r + 1][1]);
                    if (PHASE) l = pc[r].add(l, win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2]);
                    M::template step<P, MODE>(s[r], win[r + 1][1], l, kk, sub0 + st);
                }
            }
        }
        // ---- publish the new potential, then fetch the next sub-step's window ---------------------------
        if (st + 1 < K) {
            const unsigned live = ra < rb ? ((1u << rb) - 1u) & ~((1u << ra) - 1u) : 0u;
            const unsigned m = live & pub;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((m >> r) & 1u) B[wi + ro(r)] = s[r][0];         // wave-uniform branch, no lane mask
            if (top_r >= 0 || bot_r >= 0) {                         // a strip that holds the grid's row 1 or H-2
                // enforce_boundary + REFLECT: the border and ghost rows above row 1 / below row H-2 take its new value
                // (the columns need nothing: their clamp is in the tap addresses)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r == top_r && ((live >> r) & 1u)) {
                        B[wi + ro(r - 1)] = s[r][0];
                        if (c0 + r >= 1) B[wi + ro(r - 2)] = s[r][0];
                    }
                    if (r == bot_r && ((live >> r) & 1u)) {
                        B[wi + ro(r + 1)] = s[r][0];
                        if (c0 + r + 1 < LQ - 2) B[wi + ro(r + 2)] = s[r][0];
                    }
                }
            }
            FIB_WSTAMP(st);
#ifndef FIB_DIAG_NO_BARRIER                 // (diagnostic builds of tools/ubench/diag_strip.hip only)
            __syncthreads();
#endif
#ifndef FIB_DIAG_NO_RELOAD
            window(B, win);
#endif
        }
        FIB_STAMP(3 + st);
    }
    }
    if (!MT || tick + 1 >= (int)(mt.ticks_id & 0xFFFFu)) break;

    // ================= between two ticks of one launch =================
    if constexpr (MT) {
        constexpr int NC4 = (NV + 3) / 4;                           // 16-byte cells per grid cell (the last one padded)
        const unsigned plane16 = (unsigned)(g.H * g.W) * 16u;       // bytes of one [H*W] array of 16-byte cells
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(mt.xb, 0, (int)(2u * NC4 * plane16), 0x00020000);
        const unsigned pbase = (unsigned)(tick & 1) * NC4 * plane16;
        FIB_BSTAMP(0);
        // ---- publish the tile: 16-byte cells, write-through -------------------------------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (own[r]) {
#pragma unroll
                for (int c = 0; c < NC4; ++c) {
                    // (a model whose arrays are not a multiple of four pads its last cell: the index is folded after unrolling)
                    const int i1 = 4 * c + 1 < NV ? 4 * c + 1 : NV - 1, i2 = 4 * c + 2 < NV ? 4 * c + 2 : NV - 1,
                              i3 = 4 * c + 3 < NV ? 4 * c + 3 : NV - 1;
                    const fib_v4f v = {s[r][4 * c], s[r][i1], s[r][i2], s[r][i3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fib_v4u, v), rs, (int)(pbase + c * plane16 + (unsigned)off[r] * 16u), 0, 16);
                }
            }
        }
        FIB_BSTAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // EVERY storing wave, before the barrier / before it is counted
        FIB_BSTAMP(2);
        const unsigned want = mt.epoch0 + (unsigned)tick + 1u;
#if FIB_B_LASTWAVE
        // No workgroup barrier between the stores and the epoch word: every wave counts itself in LDS once its stores have
        // been acknowledged, and the wave that comes LAST raises the word.  The strips that went stale early (the rim strips:
        // nothing to store) pass at once — wave 0 among them, so its poll of the neighbours' words is already running when the
        // tile's own word goes up — and nobody waits for the slowest strip twice.
        {
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(&mt_arrive[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old + 1u == (unsigned)NW * ((unsigned)tick + 1u) && lane == 0) {
                __hip_atomic_store(mt.epoch + (size_t)tile * MT_EPOCH_STRIDE, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (every wave's write-through stores to the host frame have been acknowledged before it was counted)
                if (tick == snap_at)
                    __hip_atomic_store(mt.snap_flag + (size_t)tile * MT_SNAP_STRIDE, mt.snap_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        FIB_BSTAMP(3);
#else
        __syncthreads();
        FIB_BSTAMP(3);
#if FIB_B_INBOX
        // PUSH instead of pull: the tile writes its tick count into a word of each neighbour's OWN line (slot = the direction it
        // is seen from), so that a tile polls ONE 64-byte line — eight neighbours in one memory request — instead of eight lines
        // on eight channels
        if (wave == 0 && lane < 8) {
            const int tiles_y = g.ntiles / g.tiles_x;
            const int d = lane < 4 ? lane : lane + 1;
            const int ny = by + d / 3 - 1, nx = bx + d % 3 - 1;
            const int od = 8 - d, slot = od < 4 ? od : od - 1;     // the direction this tile lies in, seen from that neighbour
            if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < g.tiles_x)
                __hip_atomic_store(mt.epoch + (size_t)(ny * g.tiles_x + nx) * MT_EPOCH_STRIDE + slot, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        if (threadIdx.x == 0) {
#if !FIB_B_INBOX
            __hip_atomic_store(mt.epoch + (size_t)tile * MT_EPOCH_STRIDE, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            // every wave's write-through stores to the host frame have been acknowledged (vmcnt(0) before the barrier above):
            // the word follows them
            if (tick == snap_at) {
                __hip_atomic_store(mt.snap_flag + (size_t)tile * MT_SNAP_STRIDE, mt.snap_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
#endif
        // ---- wait for the eight neighbours (bounded) ---------------------------------------------------------
        if (wave == 0) {
            const int tiles_y = g.ntiles / g.tiles_x;
            const int d = lane < 4 ? lane : lane + 1;               // 0..8 without the centre
            const int ny = by + d / 3 - 1, nx = bx + d % 3 - 1;
            const bool need = lane < 8 && ny >= 0 && ny < tiles_y && nx >= 0 && nx < g.tiles_x;
            // lane 8 watches the give-up word instead, lane 9 the host's word {launch id, n} (one line further; a launch that
            // ran ahead of the caller, fibhip.hip `run-ahead`): not this launch's id (or 0) = go on; n = MT_CANCEL: the results
            // are not wanted at all; else the caller wants the state after n ticks of this launch — this boundary if n ticks are
            // done now (leave through the write-back), not this tile's business yet if n is still ahead, too late if it is
            // behind.  (The id: earlier launches of the handle may still be queued or running when the word is written.)
            // The word gets here through tile 0, which reads the host's page-locked copy over PCIe — ONE read per tick for the
            // whole grid, issued at the start of a tick and looked at at its end, so the round trip hides behind the sub-steps —
            // and passes it on.  (Measured: a copy through a second stream does not land before the launch has ended, 238-387 us
            // at 512x512; every tile reading host memory itself costs 28 us per tick.)
            if (tile == 0) {
                const unsigned hws = __builtin_amdgcn_readfirstlane(hw);
                if ((hws >> 16) == (mt.ticks_id >> 16) && lane == 0)
                    __hip_atomic_store(mt.err + MT_EPOCH_STRIDE, hws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#if FIB_B_INBOX
            const unsigned *f = lane == 8 ? mt.err : (lane == 9 ? mt.err + MT_EPOCH_STRIDE
                                                                : mt.epoch + (size_t)tile * MT_EPOCH_STRIDE + (lane & 7));
#else
            const unsigned *f = lane == 8 ? mt.err : (lane == 9 ? mt.err + MT_EPOCH_STRIDE
                                                                : mt.epoch + (size_t)(need ? ny * g.tiles_x + nx : tile) * MT_EPOCH_STRIDE);
#endif
            const unsigned done = (unsigned)tick + 1u;
#if FIB_WAIT_FORM == 2
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + MT_WAIT_TICKS;
#elif FIB_WAIT_FORM == 1        // the bound in units of 2^17 ticks of 10 ns (1.31 ms), set by the host: a shift, no multiply, no default
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + ((unsigned long long)((unsigned)mt.snap_var >> 8) << MT_WAIT_SHIFT);
#else
            const unsigned wait_ms = (unsigned)mt.snap_var >> 8;
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (wait_ms ? (unsigned long long)wait_ms * 100000ull : MT_WAIT_TICKS);
#endif
            for (;;) {
                const unsigned e = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // one ballot for everything that is not the ordinary case: a tile gave up (lane 8), or the host's word concerns
                // this boundary (lane 9)
                const unsigned n_host = e & 0xFFFFu;
                const bool special = lane == 8 ? e != 0u
                                               : (lane == 9 && (e >> 16) == (mt.ticks_id >> 16) && (n_host == MT_CANCEL || n_host <= done));
                // (epochs are compared as differences: they may wrap)
                const bool wait_more = need && (int)(e - want) < 0;
                if (__builtin_amdgcn_ballot_w64(special || wait_more) == 0ull) break;      // the ordinary way out: ONE test
                const unsigned long long sp = __builtin_amdgcn_ballot_w64(special);
                if (sp != 0ull) {
                    const bool gave_up = (sp >> 8) & 1ull;
                    const bool stop_here = !gave_up && (__builtin_amdgcn_readlane(e, 9) & 0xFFFFu) == done;
                    if (lane == 0) {
                        // (the give-up word stands already — it names the launch whose tile gave up first — and stays as it is)
                        mt_abort = stop_here ? 2 : 1;               // 2: leave through the write-back (no neighbour is waited for: it
                    }                                               // may have left already); 1: the results are not wanted / void
                    break;
                }
                if (__builtin_amdgcn_s_memrealtime() > t_end) {
                    if (lane == 0) {
                        // the give-up word names the launch (its id is never 0): the host replays from the state THAT launch
                        // started from (fibhip.hip, `recover`); launches queued behind it find the word and leave at their first
                        // boundary without writing anything
#if FIB_GIVEUP_CAS
                        unsigned expected = 0u;
                        __hip_atomic_compare_exchange_strong(mt.err, &expected, mt.ticks_id >> 16, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
#else
                        __hip_atomic_store(mt.err, mt.ticks_id >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                        mt_abort = 1;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(FIB_POLL_SLEEP);
            }
        }
        FIB_BSTAMP(4);
        __syncthreads();
        FIB_BSTAMP(5);
        if (mt_abort == 1) {                                        // whole workgroup: the results of this launch are void
            // A tile that leaves because some tile gave up tells the host WHICH launch that was, in the host's own memory (page-
            // locked, behind the host's word): a synchronising call then reads a word of host memory instead of copying one from
            // the device behind every launch.  (Here, on the way out, and not where the wait runs out: the 64-bit address of a
            // system-scope store inside the poll loop cost the Fenton kernel, which sits at its 128 registers, three spills and
            // 2 % of its speed — same-box A/B against round 3's kernel, profiles/r04_ab_pair_lds.txt.)
            if (threadIdx.x == 0) {
                const unsigned who = __hip_atomic_load(mt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (who != 0u)
                    __hip_atomic_store(mt.snap_flag + MT_HOST_WORD_AT + MT_GIVEUP_WORD, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if (mt_abort == 2) {                                        // the caller wants exactly the ticks done so far: write them back
            if (threadIdx.x == 0) __hip_atomic_fetch_add(mt.err + 2 * MT_EPOCH_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // ---- the rim of the compute box, from what the neighbours published ------------------------------
        // (every load of handed-over bytes is an sc1 load; a thread outside the box or the grid reads a clamped
        // address like the prologue does: its values are never used)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!own[r]) {
#pragma unroll
                for (int c = 0; c < NC4; ++c) {
                    // (whole-vector bit cast: __builtin_bit_cast of ONE element of a vector reads element 0 for every index
                    // with this compiler)
                    const fib_v4f v = __builtin_bit_cast(
                        fib_v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pbase + c * plane16 + (unsigned)off[r] * 16u), 0, 16));
                    s[r][4 * c] = v.x;
                    if (4 * c + 1 < NV) s[r][4 * c + 1 < NV ? 4 * c + 1 : 0] = v.y;
                    if (4 * c + 2 < NV) s[r][4 * c + 2 < NV ? 4 * c + 2 : 0] = v.z;
                    if (4 * c + 3 < NV) s[r][4 * c + 3 < NV ? 4 * c + 3 : 0] = v.w;
                }
            }
        }
        // the potential of the two ring rows around the box (tapped by the first sub-step only), where they are
        // interior rows of the grid: loaded by the first / last wave
        const int gtop = cy0 - 1 + g.row_off, gbot = cy0 + CY + g.row_off;
        const int cxx = clampi(gx, 0, g.W - 1);
        float ring = 0.0f;
        const bool ring_top = wave == 0 && gtop >= 1 && gtop <= g.Hg - 2;
        const bool ring_bot = wave == NW - 1 && gbot >= 1 && gbot <= g.Hg - 2;
        if (ring_top)
            ring = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(pbase + (unsigned)((cy0 - 1) * g.W + cxx) * 16u), 0, 16));
        if (ring_bot)
            ring = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(pbase + (unsigned)((cy0 + CY) * g.W + cxx) * 16u), 0, 16));
        FIB_BSTAMP_WAIT();
        FIB_BSTAMP(6);
        // ---- the whole box's potential into the tile, as after a sub-step — plus the ring (columns 0 and 63 of the
        // tile, rows 0 and CY+1), which the sub-steps never write
        float *B0 = lds[0];
        const int wib = (gx >= 1 && gx <= g.W - 2) ? cell_at(c0 + 1, lane) : cell_at(SPARE_ROW, lane);
        {
            const unsigned live = ra_fix < rb_fix ? ((1u << rb_fix) - 1u) & ~((1u << ra_fix) - 1u) : 0u;
            const unsigned m = live & pub;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((m >> r) & 1u) B0[wib + ro(r)] = s[r][0];
            if (top_r >= 0 || bot_r >= 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r == top_r && ((live >> r) & 1u)) {
                        B0[wib + ro(r - 1)] = s[r][0];
                        if (c0 + r >= 1) B0[wib + ro(r - 2)] = s[r][0];
                    }
                    if (r == bot_r && ((live >> r) & 1u)) {
                        B0[wib + ro(r + 1)] = s[r][0];
                        if (c0 + r + 1 < LQ - 2) B0[wib + ro(r + 2)] = s[r][0];
                    }
                }
            }
            // (tile rows 0 and CY + 1; in the paired image an even row sits at its word's first dword, an odd one at the second)
            constexpr int RING_BOT = PAIR ? ((CY + 1) >> 1) * 128 + ((CY + 1) & 1) : (CY + 1) * LP;
            const int col = PAIR ? 2 * lane : lane, nobody = cell_at(SPARE_ROW, lane);
            if (ring_top) B0[(gx >= 1 && gx <= g.W - 2) ? col : nobody] = ring;
            if (ring_bot) B0[(gx >= 1 && gx <= g.W - 2) ? RING_BOT + col : nobody] = ring;
        }
        FIB_BSTAMP(7);
        __syncthreads();
        FIB_BSTAMP(8);
        window(B0, win);
        FIB_BSTAMP_WAIT();
        FIB_BSTAMP(9);
    }
    }

    // ---- write back ---------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (own[r]) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if ((WMASK >> v) & 1u) pt.out[v][off[r]] = s[r][v];
        }
    }
    FIB_STAMP(14);
}

template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE>
__global__ void __launch_bounds__(64 * ((TY + 2 * (K - 1) + R - 1) / R))
strip_kernel(Geo g, PtrTab<M::NVAR> pt, PhaseTab ph, typename M::Consts k, int sub0)
{
    strip_body<M, P, MODE, K, TX, TY, R, PHASE, false>(g, pt, ph, k, sub0, MtArgs{});
}

// the same tile program advancing `mt.nticks` ticks of K sub-steps each (K = the tick's sub-steps) in one launch
template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE>
__global__ void __launch_bounds__(64 * ((TY + 2 * (K - 1) + R - 1) / R))
strip_mt_kernel(Geo g, PtrTab<M::NVAR> pt, PhaseTab ph, typename M::Consts k, int sub0, MtArgs mt)
{
#ifndef FIB_MT_FULL_GEO
    // The host launches this kernel on whole single-device grids only (fibhip.hip mt_eligible: planar slab, no ghost rows, one band
    // of rows): say so, and seven of Geo's twelve scalars are constants or copies instead of live scalar registers — the kernel
    // spills scalar registers into vector lanes as it is, and sits at its 128 vector registers.
    g.pitch = g.W;
    g.Hg = g.H;
    g.row_off = 0;
    g.r0 = 0;
    g.r1 = g.H;
    g.rb0 = g.rb1 = 0;
    g.ty_a = 0x7fffffff;
#endif
    strip_body<M, P, MODE, K, TX, TY, R, PHASE, true>(g, pt, ph, k, sub0, mt);
}

// ---- wavefront-level neighbour access (gfx9 DPP wavefront shifts) -----------------------------------
// lane i reads the value lane i-1 / i+1 holds: the W / E taps of a row whose columns are the lanes of a
// wave.  The compiler folds the move into the consuming v_add/v_sub (`v_add_f32_dpp ... wave_shr:1`), so a
// horizontal tap costs no instruction of its own and no LDS access.  Lane 0 / lane 63 receive 0: they are
// the ring columns of the compute box, whose results are never used.
static FIB_DEV float lane_west(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
static FIB_DEV float lane_east(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
static FIB_DEV float lane_get(float v, int lane)     // lane: wave-uniform
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// 9-point Laplacian (+ phase term) of the cell whose row neighbours N, S and own value C sit in THIS lane's
// registers and whose column neighbours sit in the adjacent lanes.  Same operations in the same order as
// stencil9 / phase_term: NW + SW is the neighbouring lane's own N + S (the same float32 addition of the same
// two numbers), so `lane_west(N + S)` is bit-identical to forming it here.
template <class P, bool PHASE>
static FIB_DEV float stencil9_lanes(float N, float S, float C, const PhaseCoef<P> &pc)
{
    const float ns = N + S;
    const float Wv = lane_west(C), E = lane_east(C);
    const float l1 = (ns + Wv) + E, d = (lane_west(ns) + lane_east(N)) + lane_east(S);
    float r;
    if constexpr (same_type<P, Fast>::value)                      // lap9<Fast>'s row-by-row form, the taps by lane shifts
        r = (__builtin_fmaf(0.5f, lane_west(N) + lane_east(N), N) + __builtin_fmaf(0.5f, lane_west(S) + lane_east(S), S)) +
            __builtin_fmaf(-6.0f, C, Wv + E);
    else
        r = (l1 + 0.5f * d) - 6.0f * C;
    if (PHASE) r = pc.add(r, N, S, Wv, E);
    return r;
}

// rows_kernel<M,P,MODE,K,TX,TY,R,PHASE> — temporal blocking with the potential in REGISTERS.
//   Work layout as strip_kernel (lane = column of a 64-wide box, wave = R consecutive rows, K sub-steps per
//   launch on a box that shrinks by one ring per sub-step), but the potential never lives in an LDS tile:
//     * a lane keeps the R values of its column strip in registers; the N/S taps of the strip's inner rows are
//       those registers, the W/E/diagonal taps are DPP wavefront shifts of them (lane_west / lane_east);
//     * only the strip's first and last row travel between waves: 2 ds_write + 2 ds_read per wave and sub-step
//       (strip_kernel: 3(R+2) reads + R writes) through a double-buffered [wave][top|bottom][lane] exchange
//       array, one s_barrier per sub-step;
//     * sub-step 0 takes its halo rows straight from global memory: no LDS fill, no barrier in the prologue.
//   Boundary rule (enforce_boundary + REFLECT): a tap at (r, c) reads the raw potential at
//   (clamp(r,1,H-2), clamp(c,1,W-2)).  Each lane therefore carries, next to the raw value of its cell (Fenton's
//   reaction reads it on border cells), the ENFORCED value `e` its neighbours see; after a sub-step border and
//   ghost rows/columns take the new value of the adjacent interior row/column — register copies inside a wave
//   (readlane across columns), the exchanged edge row between waves.  Only tiles that touch the domain edge
//   run that code (block-uniform branch; EDGE = false compiles it away).
template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE, bool EDGE>
static FIB_DEV void rows_body(const Geo &g, const PtrTab<M::NVAR> &pt, const PhaseTab &ph, const typename M::Consts &k, int sub0,
                              int tile, float (*ex)[64])
{
    constexpr int NV = M::NVAR;
    constexpr int CX = TX + 2 * (K - 1), CY = TY + 2 * (K - 1);
    constexpr int NW = (CY + R - 1) / R;
    constexpr unsigned WMASK = M::mask(MODE);
    auto &&kk = M::pinned(k);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int by = tile / g.tiles_x, bx = tile - by * g.tiles_x;
    int y0, rend;
    tile_rows(g, by, TY, y0, rend);
    const int x0 = bx * TX;
    const int cx0 = x0 - (K - 1), cy0 = y0 - (K - 1);
    const int gx = cx0 - 1 + lane;                                  // this lane's global column
    const int c0 = wave * R;                                        // first box row of this wave
    const int g0 = cy0 + c0 + g.row_off, glast = g0 + R - 1;        // global rows of the strip's first / last row
    const bool lane_in = lane >= 1 && lane <= CX && gx >= 0 && gx < g.W;
    const bool store_col = lane_in && gx >= x0 && gx < x0 + TX;
    const int ox = clampi(gx, 0, g.W - 1);
    const int bxx = clampi(gx, 1, g.W - 2);                         // column through the boundary clamp
    FIB_STAMP(0);

    // ---- prologue: every global load is issued before anything waits -------------------------------------
    float s[R][NV], e[R];
    PhaseCoef<P> pc[R];
    int off[R];
    auto brow = [&](int grow) {                                     // global row -> local row through the boundary clamp
        return clampi(clampi(grow, 1, g.Hg - 2) - g.row_off, 0, g.H - 1);
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = cy0 + c0 + r;
        const int oy = clampi(gy, 0, g.H - 1);
        off[r] = oy * g.pitch + ox;
#pragma unroll
        for (int v = 0; v < NV; ++v) s[r][v] = pt.in[v][off[r]];
        if (EDGE) e[r] = pt.in[0][(size_t)brow(g0 + r) * g.pitch + bxx];
        if (PHASE) pc[r].load(ph, oy * g.W + ox);                   // (the phase arrays are always planar)
    }
    float eN = pt.in[0][(size_t)brow(g0 - 1) * g.pitch + bxx];      // the rows above / below the strip
    float eS = pt.in[0][(size_t)brow(glast + 1) * g.pitch + bxx];
    if (!EDGE) {
#pragma unroll
        for (int r = 0; r < R; ++r) e[r] = s[r][0];
    }
    const bool top_open = cy0 + g.row_off > 0, bot_open = cy0 + CY + g.row_off < g.Hg;
    // lanes that hold column 1 / W-2 (sources of the border and ghost columns), block-uniform
    const int l_c1 = clampi(2 - cx0, 0, 63), l_cw = clampi(g.W - 1 - cx0, 0, 63);
    const bool edge_h = (cx0 <= 1) || (cx0 + CX >= g.W - 1);
    const bool west_copy = gx <= 0, east_copy = gx >= g.W - 1;
    FIB_STAMP(1);
#ifdef FIB_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // diagnostic build: the load latency gets its own stamp
#endif
    FIB_STAMP(2);

#pragma unroll 1
    for (int st = 0; st < K; ++st) {
        const int need0 = top_open ? st : 0, need1 = bot_open ? CY - st : CY;
        int ra = max(0, need0 - c0), rb = min(R, need1 - c0);       // live rows of this strip (wave-uniform)
        ra = max(ra, -(cy0 + c0 + g.row_off));                      // global row >= 0
        rb = min(rb, min(g.Hg - g.row_off, g.H) - (cy0 + c0));      // global row < Hg, local row < H
        ra = max(ra, -(cy0 + c0));                                  // local row >= 0
        if (ra == 0 && rb == R) {
            float lp[R];
#pragma unroll
            for (int r = 0; r < R; ++r)
                lp[r] = stencil9_lanes<P, PHASE>(r == 0 ? eN : e[r - 1], r == R - 1 ? eS : e[r + 1], e[r], pc[r]);
            if constexpr (M::HAS_VEC) {
                M::template stepN<P, MODE, R>(s, e, lp, kk, sub0 + st);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) M::template step<P, MODE>(s[r], e[r], lp[r], kk, sub0 + st);
            }
        } else if (ra < rb) {
            // (all lanes stay active: the DPP taps of a live row need every lane's registers)
            float lp[R];
#pragma unroll
            for (int r = 0; r < R; ++r)
                lp[r] = stencil9_lanes<P, PHASE>(r == 0 ? eN : e[r - 1], r == R - 1 ? eS : e[r + 1], e[r], pc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r >= ra && r < rb) M::template step<P, MODE>(s[r], e[r], lp[r], kk, sub0 + st);   // scalar branch
        }
        if (st + 1 < K) {
            // ---- the enforced values the next sub-step's taps read --------------------------------------
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (!EDGE || (r >= ra && r < rb)) e[r] = s[r][0];
            if (EDGE) {
                if (edge_h) {                                       // border + ghost columns <- column 1 / W-2
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float a = lane_get(e[r], l_c1), b = lane_get(e[r], l_cw);
                        e[r] = west_copy ? a : (east_copy ? b : e[r]);
                    }
                }
                // border + ghost rows whose source row lives in this strip
#pragma unroll
                for (int r = R - 2; r >= 0; --r)
                    if (g0 + r == 0 || g0 + r == -1) e[r] = e[r + 1];
#pragma unroll
                for (int r = 1; r < R; ++r)
                    if (g0 + r == g.Hg - 1 || g0 + r == g.Hg) e[r] = e[r - 1];
            }
            // ---- the strip's edge rows, for the waves above and below --------------------------------------
            float(*slot)[64] = ex + ((st & 1) * NW) * 2;
            slot[wave * 2 + 0][lane] = e[0];
            slot[wave * 2 + 1][lane] = e[R - 1];
            __syncthreads();
            eN = slot[max(wave - 1, 0) * 2 + 1][lane];
            eS = slot[min(wave + 1, NW - 1) * 2 + 0][lane];
            if (EDGE) {
                if (g0 == 1 || g0 == 0) eN = e[0];                  // the row above is border row 0 / ghost row -1
                if (glast == g.Hg - 2 || glast == g.Hg - 1) eS = e[R - 1];
                if (glast == 0) {                                   // my last row is border row 0: row 1 is the next strip's
                    e[R - 1] = eS;
                    if (R >= 2) e[R - 2] = eS;
                }
                if (g0 == g.Hg - 1) {
                    e[0] = eN;
                    if (R >= 2) e[1] = eN;
                }
            }
        }
        FIB_STAMP(3 + st);
    }

    // ---- write back -------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = cy0 + c0 + r;
        if (store_col && gy >= y0 && gy < min(y0 + TY, rend) && gy < g.H) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if ((WMASK >> v) & 1u) pt.out[v][off[r]] = s[r][v];
        }
    }
    FIB_STAMP(14);
}

template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE>
__global__ void __launch_bounds__(64 * ((TY + 2 * (K - 1) + R - 1) / R))
rows_kernel(Geo g, PtrTab<M::NVAR> pt, PhaseTab ph, typename M::Consts k, int sub0)
{
    constexpr int CX = TX + 2 * (K - 1), CY = TY + 2 * (K - 1);
    static_assert(CX <= 62 && K > 1 && R >= 2, "rows_kernel: compute box must fit 62 lanes, strips of at least 2 rows");
    constexpr int NW = (CY + R - 1) / R;
    static_assert(NW <= 16, "rows_kernel: a workgroup has at most 16 waves");
    __shared__ float ex[2 * NW * 2][64];                            // [parity][wave][top|bottom][lane]

    const int tile = xcd_tile(blockIdx.x, g.ntiles);
    if (tile >= g.ntiles) return;
    const int by = tile / g.tiles_x, bx = tile - by * g.tiles_x;
    int y0, rend;
    tile_rows(g, by, TY, y0, rend);
    const int cx0 = bx * TX - (K - 1), cy0 = y0 - (K - 1) + g.row_off;
    // does the compute box (with its ring) reach the domain's border rows / columns?  block-uniform
    const bool edge = cx0 <= 1 || cx0 + CX >= g.W - 1 || cy0 <= 1 || cy0 + NW * R >= g.Hg - 1;
    if (edge)
        rows_body<M, P, MODE, K, TX, TY, R, PHASE, true>(g, pt, ph, k, sub0, tile, ex);
    else
        rows_body<M, P, MODE, K, TX, TY, R, PHASE, false>(g, pt, ph, k, sub0, tile, ex);
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
        const float Vc = pt.in[0][(size_t)yy * g.pitch + xx];
        const int o = gy * g.pitch + gx;
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
                const float f = phase_term<P>(N, S, Wv, E, ph3[e], ph3[n + e], ph3[2 * n + e], ph3[3 * n + e]);
                r = (op == OP_LAPLACE) ? r + f : f;
            }
            out[e] = r;
        }
    }
}

// ϕ -> (dpy, dpx, q4, r4) and the fast policy's two products, REFLECT-padded in GLOBAL coordinates (ionic.py:75-80)
__global__ void phase_prep_kernel(Geo g, const float *phi, float *dpy, float *dpx, float *q4, float *r4, float *pyr, float *pxr)
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
        r4[e] = 1.0f / q4[e];                       // IEEE division: correctly rounded reciprocal
        if (pyr) {                                  // one rounding each (-ffp-contract=off), as add_phase<Fast> forms them
            pyr[e] = dpy[e] * r4[e];
            pxr[e] = dpx[e] * r4[e];
        }
    }
}

// pace op, ionic.py:144-163:  pot = max(pot, s), s = v inside the global rectangle, min_v outside
__global__ void pace_kernel(Geo g, float *pot, int r0, int r1, int c0, int c1, float v, float min_v)
{
    const int n = g.H * g.W;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int y = e / g.W, yg = y + g.row_off, x = e % g.W;
        const float sv = (yg >= r0 && yg < r1 && x >= c0 && x < c1) ? v : min_v;
        const size_t o = (size_t)y * g.pitch + x;
        pot[o] = fmaxf(pot[o], sv);
    }
}

// calc_inter(V, mod) as a stand-alone op (court.py:273-429, court_ultra.py:445-450): the 32 voltage-only
// intermediates of n voltages, row k of `out` = k-th key in the reference dict's insertion order.
constexpr int COURT_NINTER = 32;
template <class P>
__global__ void court_inter_kernel(int n, const float *__restrict__ V, float *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        CourtemancheUS::Inter q;
        CourtemancheUS::calc_inter<P>(V[i], q);
        const float v[COURT_NINTER] = {q.d_inf, q.tau_d, q.f_inf, q.tau_f, q.tau_w, q.w_inf, q.m_inf, q.tau_m,
                                       q.h_inf, q.tau_h, q.j_inf, q.tau_j, q.tau_oa, q.oa_inf, q.tau_oi, q.oi_inf,
                                       q.tau_ua, q.ua_inf, q.tau_ui, q.ui_inf, q.tau_xr, q.xr_inf, q.tau_xs, q.xs_inf,
                                       q.g_Kur, q.f_NaK, q.i_NaCaa, q.i_NaCab, q.i_K1a, q.i_Kra, q.us_inf, q.tau_us};
#pragma unroll
        for (int k = 0; k < COURT_NINTER; ++k) out[(size_t)k * n + i] = v[k];
    }
}

// plain streaming copy, one 16-byte element per thread and as many workgroups as that takes: the bandwidth yardstick
// bench.py prints next to the roofline peak.  (tools/ubench/copybw.hip -> profiles/r02_copy_bandwidth_shapes.txt: this
// shape reaches the 6.3 TB/s the microarch guide quotes; grid-stride loops with non-temporal accesses stay at 4.6-5.7,
// reads alone run at 7.0, writes alone at 4.4 TB/s)
__global__ void __launch_bounds__(256) copy_kernel(const fib_v4f *__restrict__ src, fib_v4f *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

FIB_TAG_END
}  // namespace fib
