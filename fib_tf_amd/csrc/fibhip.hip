// fibhip.hip — C ABI (include/fibhip.h) over the gfx950 kernels of kernels.hpp.
//
// Host-side responsibilities: device memory for the SoA slabs (ping/pong), the per-tick launch
// plan (how many sub-steps each launch fuses), the per-variable buffer bookkeeping, the
// edge/interior split used by the row-block multi-GPU driver, and HIP-event timing.
// There is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/fibhip.h"
#include "kernels.hpp"

using namespace fib;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(...)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (__VA_ARGS__);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(FIBHIP_EHIP, "%s failed: %s (%s:%d)", #__VA_ARGS__, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                  \
    } while (0)

extern "C" const char *fibhip_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------
// kernel variants
// ------------------------------------------------------------------------------------------
constexpr int MT_MAX_TICKS = 32;          // default bound on the ticks of one launch (0.4 ms of Fenton 512x512)
constexpr int AT_MT_TICKS = 8;           // autotune times a multi-tick candidate as one launch of this many ticks
static const char *const MT_DEAD_MSG =
    "a multi-tick launch gave up (a tile waited its full bound for a neighbouring tile: were all workgroups resident? is another "
    "process holding the GPU?) and the state it started from could not be restored; the state of this handle is void — "
    "FIBHIP_MT=0 runs one launch per tick";

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

constexpr int FIB_MAXVAR = 26;   // CourtAgg: 21 state arrays + 5 aggregates (CourtemancheUS: 22)

struct LaunchCtx {
    Geo g;
    const float *in[FIB_MAXVAR];
    float *out[FIB_MAXVAR];
    PhaseTab ph;
    const void *consts;
    int sub0;
    // a kernel of a code object loaded at run time (fibhip_module_load) instead of one linked into this library
    hipFunction_t kern;
    int kind, K, TX, TY, NT, nvar, consts_bytes;
    MtArgs mt;                      // launch_strip_mt only: the ticks of this launch and what its tiles exchange through
};

typedef hipError_t (*launch_fn)(hipStream_t, const LaunchCtx &);

template <class M, class P, int MODE, int K, int TX, int TY, int NT, bool PHASE>
static hipError_t launch_tick(hipStream_t st, const LaunchCtx &c)
{
    Geo g = c.g;
    g.tiles_x = (g.W + TX - 1) / TX;
    g.ty_a = (g.r1 > g.r0) ? (g.r1 - g.r0 + TY - 1) / TY : 0;
    const int tiles_y = g.ty_a + ((g.rb1 > g.rb0) ? (g.rb1 - g.rb0 + TY - 1) / TY : 0);
    g.ntiles = g.tiles_x * tiles_y;
    if (g.ntiles <= 0) return hipSuccess;
    PtrTab<M::NVAR> pt;
    for (int v = 0; v < M::NVAR; ++v) {
        pt.in[v] = c.in[v];
        pt.out[v] = c.out[v];
    }
    const int grid = ((g.ntiles + 7) / 8) * 8;       // xcd_tile() needs a multiple of 8
    hipLaunchKernelGGL((tick_kernel<M, P, MODE, K, TX, TY, NT, PHASE>), dim3(grid), dim3(NT), 0, st, g, pt, c.ph,
                       *static_cast<const typename M::Consts *>(c.consts), c.sub0);
    return hipGetLastError();
}

template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE>
static hipError_t launch_strip(hipStream_t st, const LaunchCtx &c)
{
    Geo g = c.g;
    g.tiles_x = (g.W + TX - 1) / TX;
    g.ty_a = (g.r1 > g.r0) ? (g.r1 - g.r0 + TY - 1) / TY : 0;
    const int tiles_y = g.ty_a + ((g.rb1 > g.rb0) ? (g.rb1 - g.rb0 + TY - 1) / TY : 0);
    g.ntiles = g.tiles_x * tiles_y;
    if (g.ntiles <= 0) return hipSuccess;
    PtrTab<M::NVAR> pt;
    for (int v = 0; v < M::NVAR; ++v) {
        pt.in[v] = c.in[v];
        pt.out[v] = c.out[v];
    }
    constexpr int NT = 64 * ((TY + 2 * (K - 1) + R - 1) / R);
    const int grid = ((g.ntiles + 7) / 8) * 8;
    hipLaunchKernelGGL((strip_kernel<M, P, MODE, K, TX, TY, R, PHASE>), dim3(grid), dim3(NT), 0, st, g, pt, c.ph,
                       *static_cast<const typename M::Consts *>(c.consts), c.sub0);
    return hipGetLastError();
}

// the strip kernel advancing c.mt.nticks ticks in one launch; the caller has checked that all tiles can be resident
template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE>
static hipError_t launch_strip_mt(hipStream_t st, const LaunchCtx &c)
{
    Geo g = c.g;
    g.tiles_x = (g.W + TX - 1) / TX;
    g.ty_a = (g.r1 > g.r0) ? (g.r1 - g.r0 + TY - 1) / TY : 0;
    g.ntiles = g.tiles_x * g.ty_a;                  // (one band: the whole, unsharded grid)
    if (g.ntiles <= 0) return hipSuccess;
    PtrTab<M::NVAR> pt;
    for (int v = 0; v < M::NVAR; ++v) {
        pt.in[v] = c.in[v];
        pt.out[v] = c.out[v];
    }
    constexpr int NT = 64 * ((TY + 2 * (K - 1) + R - 1) / R);
    const int grid = ((g.ntiles + 7) / 8) * 8;
    hipLaunchKernelGGL((strip_mt_kernel<M, P, MODE, K, TX, TY, R, PHASE>), dim3(grid), dim3(NT), 0, st, g, pt, c.ph,
                       *static_cast<const typename M::Consts *>(c.consts), c.sub0, c.mt);
    return hipGetLastError();
}

template <class M, class P, int MODE, int K, int TX, int TY, int R, bool PHASE>
static hipError_t launch_rows(hipStream_t st, const LaunchCtx &c)
{
    Geo g = c.g;
    g.tiles_x = (g.W + TX - 1) / TX;
    g.ty_a = (g.r1 > g.r0) ? (g.r1 - g.r0 + TY - 1) / TY : 0;
    const int tiles_y = g.ty_a + ((g.rb1 > g.rb0) ? (g.rb1 - g.rb0 + TY - 1) / TY : 0);
    g.ntiles = g.tiles_x * tiles_y;
    if (g.ntiles <= 0) return hipSuccess;
    PtrTab<M::NVAR> pt;
    for (int v = 0; v < M::NVAR; ++v) {
        pt.in[v] = c.in[v];
        pt.out[v] = c.out[v];
    }
    constexpr int NT = 64 * ((TY + 2 * (K - 1) + R - 1) / R);
    const int grid = ((g.ntiles + 7) / 8) * 8;
    hipLaunchKernelGGL((rows_kernel<M, P, MODE, K, TX, TY, R, PHASE>), dim3(grid), dim3(NT), 0, st, g, pt, c.ph,
                       *static_cast<const typename M::Consts *>(c.consts), c.sub0);
    return hipGetLastError();
}

template <class M, class P, int MODE>
static hipError_t launch_pointwise(hipStream_t st, const LaunchCtx &c)
{
    PtrTab<M::NVAR> pt;
    for (int v = 0; v < M::NVAR; ++v) {
        pt.in[v] = c.in[v];
        pt.out[v] = c.out[v];
    }
    const long n = (long)(c.g.r1 - c.g.r0) * c.g.W;
    if (n <= 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL((pointwise_kernel<M, P, MODE>), dim3(grid), dim3(256), 0, st, c.g, pt,
                       *static_cast<const typename M::Consts *>(c.consts));
    return hipGetLastError();
}

// One launcher for every kernel of a run-time module (a traced model compiled in-process by hiprtc): the same grids
// as launch_tick / launch_strip / launch_pointwise, the kernel arguments laid out by hand as the compiler lays out
// (Geo, PtrTab<NVAR>, PhaseTab, Consts, int) — every argument at its natural alignment, in order.
enum { MK_TICK = 0, MK_STRIP = 1, MK_POINTWISE = 2, MK_STRIP_MT = 3 };
static hipError_t launch_module(hipStream_t st, const LaunchCtx &c)
{
    Geo g = c.g;
    int threads, grid;
    if (c.kind == MK_POINTWISE) {
        const long n = (long)(g.r1 - g.r0) * g.W;
        if (n <= 0) return hipSuccess;
        threads = 256;
        grid = (int)((n + 255) / 256);
    } else {
        g.tiles_x = (g.W + c.TX - 1) / c.TX;
        g.ty_a = (g.r1 > g.r0) ? (g.r1 - g.r0 + c.TY - 1) / c.TY : 0;
        const int tiles_y = g.ty_a + ((g.rb1 > g.rb0) ? (g.rb1 - g.rb0 + c.TY - 1) / c.TY : 0);
        g.ntiles = g.tiles_x * tiles_y;
        if (g.ntiles <= 0) return hipSuccess;
        threads = c.kind == MK_TICK ? c.NT : 64 * ((c.TY + 2 * (c.K - 1) + (-c.NT) - 1) / (-c.NT));
        grid = ((g.ntiles + 7) / 8) * 8;
        if (c.kind == MK_STRIP_MT) g.ntiles = g.tiles_x * g.ty_a;       // (one band: the whole, unsharded grid)
    }
    alignas(8) char buf[sizeof(Geo) + 8 + 2 * FIB_MAXVAR * sizeof(void *) + sizeof(PhaseTab) + 64 + 16 + sizeof(MtArgs) + 8];
    size_t off = 0;
    auto put = [&](const void *p, size_t n, size_t align) {
        off = (off + align - 1) & ~(align - 1);
        memcpy(buf + off, p, n);
        off += n;
    };
    put(&g, sizeof g, alignof(Geo));
    off = (off + 7) & ~(size_t)7;                                   // PtrTab<NVAR>: in[NVAR] then out[NVAR]
    memcpy(buf + off, c.in, (size_t)c.nvar * sizeof(void *));
    off += (size_t)c.nvar * sizeof(void *);
    memcpy(buf + off, c.out, (size_t)c.nvar * sizeof(void *));
    off += (size_t)c.nvar * sizeof(void *);
    if (c.kind != MK_POINTWISE) put(&c.ph, sizeof c.ph, alignof(PhaseTab));
    const char zeros[64] = {0};
    if (c.consts_bytes > 0) put(c.consts ? c.consts : zeros, (size_t)c.consts_bytes, 4);
    if (c.kind != MK_POINTWISE) put(&c.sub0, sizeof c.sub0, alignof(int));
    if (c.kind == MK_STRIP_MT) put(&c.mt, sizeof c.mt, alignof(MtArgs));      // strip_mt_kernel's last argument
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &off, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(c.kern, grid, 1, 1, threads, 1, 1, 0, st, nullptr, config);
}

constexpr int VM_FENTON_ZP = 100;   // variant-table id of FentonZP (not a fibhip_model: selected by FIBHIP_ZEROPAD)
constexpr int VM_COURT_AGG = 101;   // variant-table id of CourtAgg (Courtemanche, fast policy: fibhip_ctx::use_agg)

struct Variant {
    int model, mode, fast, phase;
    int K, TX, TY, NT;
    launch_fn fn;
    hipFunction_t kern = nullptr;   // run-time module kernels only (fn == launch_module)
    int kind = 0;
    launch_fn fn_mt = nullptr;      // the same shape advancing several ticks per launch (strip_mt_kernel), or null
    hipFunction_t kern_mt = nullptr;        // ... of a run-time module (fn_mt == launch_module, kind MK_STRIP_MT)
};

#define V4(MODEL, MID, MODE, K, TX, TY, NT)                                                        \
    {MID, MODE, 0, 0, K, TX, TY, NT, launch_tick<MODEL, Exact, MODE, K, TX, TY, NT, false>},       \
    {MID, MODE, 0, 1, K, TX, TY, NT, launch_tick<MODEL, Exact, MODE, K, TX, TY, NT, true>},        \
    {MID, MODE, 1, 0, K, TX, TY, NT, launch_tick<MODEL, Fast, MODE, K, TX, TY, NT, false>},        \
    {MID, MODE, 1, 1, K, TX, TY, NT, launch_tick<MODEL, Fast, MODE, K, TX, TY, NT, true>}

// fast-policy-only models (CourtAgg)
#define F2(MODEL, MID, MODE, K, TX, TY, NT)                                                        \
    {MID, MODE, 1, 0, K, TX, TY, NT, launch_tick<MODEL, Fast, MODE, K, TX, TY, NT, false>},        \
    {MID, MODE, 1, 1, K, TX, TY, NT, launch_tick<MODEL, Fast, MODE, K, TX, TY, NT, true>}

#define FS2(MODEL, MID, MODE, K, TX, TY, R)                                                        \
    {MID, MODE, 1, 0, K, TX, TY, -(R), launch_strip<MODEL, Fast, MODE, K, TX, TY, R, false>},      \
    {MID, MODE, 1, 1, K, TX, TY, -(R), launch_strip<MODEL, Fast, MODE, K, TX, TY, R, true>}

// strip kernels are listed with NT = -R (rows per wave)
#define S4(MODEL, MID, MODE, K, TX, TY, R)                                                         \
    {MID, MODE, 0, 0, K, TX, TY, -(R), launch_strip<MODEL, Exact, MODE, K, TX, TY, R, false>},     \
    {MID, MODE, 0, 1, K, TX, TY, -(R), launch_strip<MODEL, Exact, MODE, K, TX, TY, R, true>},      \
    {MID, MODE, 1, 0, K, TX, TY, -(R), launch_strip<MODEL, Fast, MODE, K, TX, TY, R, false>},      \
    {MID, MODE, 1, 1, K, TX, TY, -(R), launch_strip<MODEL, Fast, MODE, K, TX, TY, R, true>}

// strip kernels that also exist as multi-tick launches (K = the model's sub-steps per tick)
#define S4M(MODEL, MID, MODE, K, TX, TY, R)                                                        \
    {MID, MODE, 0, 0, K, TX, TY, -(R), launch_strip<MODEL, Exact, MODE, K, TX, TY, R, false>, nullptr, 0,  \
     launch_strip_mt<MODEL, Exact, MODE, K, TX, TY, R, false>},                                    \
    {MID, MODE, 0, 1, K, TX, TY, -(R), launch_strip<MODEL, Exact, MODE, K, TX, TY, R, true>, nullptr, 0,   \
     launch_strip_mt<MODEL, Exact, MODE, K, TX, TY, R, true>},                                     \
    {MID, MODE, 1, 0, K, TX, TY, -(R), launch_strip<MODEL, Fast, MODE, K, TX, TY, R, false>, nullptr, 0,   \
     launch_strip_mt<MODEL, Fast, MODE, K, TX, TY, R, false>},                                     \
    {MID, MODE, 1, 1, K, TX, TY, -(R), launch_strip<MODEL, Fast, MODE, K, TX, TY, R, true>, nullptr, 0,    \
     launch_strip_mt<MODEL, Fast, MODE, K, TX, TY, R, true>}

// rows kernels (potential in registers, DPP taps) are listed with NT = -(32 + R)
#define W4(MODEL, MID, MODE, K, TX, TY, R)                                                         \
    {MID, MODE, 0, 0, K, TX, TY, -(32 + (R)), launch_rows<MODEL, Exact, MODE, K, TX, TY, R, false>},  \
    {MID, MODE, 0, 1, K, TX, TY, -(32 + (R)), launch_rows<MODEL, Exact, MODE, K, TX, TY, R, true>},   \
    {MID, MODE, 1, 0, K, TX, TY, -(32 + (R)), launch_rows<MODEL, Fast, MODE, K, TX, TY, R, false>},   \
    {MID, MODE, 1, 1, K, TX, TY, -(32 + (R)), launch_rows<MODEL, Fast, MODE, K, TX, TY, R, true>}

// The first matching entry with the wanted K is the default; FIBHIP_VARIANT="K,TX,TY,NT" overrides
// (tuning sweeps).  Tile shapes: K=1 tiles are wide (coalesced 256-B rows); K>1 tiles are square-ish
// to keep the redundant rim small.
static const Variant g_variants[] = {
#ifdef FIB_CUSTOM_MODEL_INC
    // ---- the traced model this copy of the library was built for (constants from the generated header) ----
    V4(Custom, FIBHIP_CUSTOM, 0, 1, 64, 4, 256),
#if FIB_CUSTOM_K > 1
    S4(Custom, FIBHIP_CUSTOM, 0, FIB_CUSTOM_K, FIB_CUSTOM_TX, FIB_CUSTOM_TY, FIB_CUSTOM_R),
#if FIB_CUSTOM_TYB > 0
    S4(Custom, FIBHIP_CUSTOM, 0, FIB_CUSTOM_K, FIB_CUSTOM_TX, FIB_CUSTOM_TYB, FIB_CUSTOM_R),
#endif
#endif
#if FIB_CUSTOM_K2 > 1 && FIB_CUSTOM_K2 != FIB_CUSTOM_K
    S4(Custom, FIBHIP_CUSTOM, 0, FIB_CUSTOM_K2, FIB_CUSTOM_TX2, FIB_CUSTOM_TY2, FIB_CUSTOM_R2),
#endif
#endif
#ifndef FIB_CUSTOM_ONLY
#ifndef FIB_ONLY_BR
    // ---- Fenton 4v ----
    // K = 10 (the whole tick in one launch) and K = 5 strips of growing tile height: build_plan picks the shape
    // that gives every CU at most one tile (or the fewest rounds) for the grid at hand
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 25, 3),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 28, 3),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 27, 3),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 30, 3),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 32, 4),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 36, 4),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 40, 4),
    S4M(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 44, 4),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 21, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 23, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 22, 4),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 27, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 25, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 28, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 31, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 34, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 32, 4),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 40, 3),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 44, 4),
    S4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 56, 4),
    S4(Fenton, FIBHIP_FENTON4V, 0, 2, 60, 18, 4),
    // the same blocking with the potential in registers and DPP taps (rows_kernel): selectable, not a default
    W4(Fenton, FIBHIP_FENTON4V, 0, 10, 44, 25, 3),
    W4(Fenton, FIBHIP_FENTON4V, 0, 5, 54, 21, 3),
    V4(Fenton, FIBHIP_FENTON4V, 0, 10, 32, 32, 512),
    V4(Fenton, FIBHIP_FENTON4V, 0, 10, 32, 32, 1024),
    V4(Fenton, FIBHIP_FENTON4V, 0, 10, 32, 32, 256),
    V4(Fenton, FIBHIP_FENTON4V, 0, 5, 32, 32, 256),
    V4(Fenton, FIBHIP_FENTON4V, 0, 5, 32, 32, 512),
    V4(Fenton, FIBHIP_FENTON4V, 0, 5, 32, 16, 256),
    V4(Fenton, FIBHIP_FENTON4V, 0, 2, 64, 16, 256),
    V4(Fenton, FIBHIP_FENTON4V, 0, 2, 32, 32, 256),
    V4(Fenton, FIBHIP_FENTON4V, 0, 1, 64, 4, 256),
    V4(Fenton, FIBHIP_FENTON4V, 0, 1, 64, 16, 256),
    // ---- Fenton 4v with the zero-padded convolution Laplacian (FIBHIP_ZEROPAD), flat kernels only ----
    V4(FentonZP, VM_FENTON_ZP, 0, 10, 32, 32, 1024),
    V4(FentonZP, VM_FENTON_ZP, 0, 5, 32, 32, 512),
    V4(FentonZP, VM_FENTON_ZP, 0, 2, 64, 16, 256),
    V4(FentonZP, VM_FENTON_ZP, 0, 1, 64, 4, 256),
#endif
    // ---- Beeler-Reuter (mode 0 direct gates, 1 Chebyshev) ----
    S4M(BeelerReuter, FIBHIP_BR, 0, 5, 54, 21, 2),
    S4M(BeelerReuter, FIBHIP_BR, 0, 5, 54, 24, 2),
    S4M(BeelerReuter, FIBHIP_BR, 0, 5, 54, 27, 3),
    S4M(BeelerReuter, FIBHIP_BR, 0, 5, 54, 28, 3),
    S4M(BeelerReuter, FIBHIP_BR, 0, 5, 54, 16, 2),
    S4M(BeelerReuter, FIBHIP_BR, 0, 5, 54, 40, 3),
    V4(BeelerReuter, FIBHIP_BR, 0, 1, 64, 4, 256),
#ifndef FIB_ONLY_BR                     // tuning alternatives (tools/sweep.py); a specialised build keeps the two defaults
    S4(BeelerReuter, FIBHIP_BR, 0, 5, 54, 21, 3),
    S4(BeelerReuter, FIBHIP_BR, 0, 3, 58, 19, 2),
    S4(BeelerReuter, FIBHIP_BR, 0, 2, 60, 19, 2),
    S4(BeelerReuter, FIBHIP_BR, 0, 2, 60, 19, 3),
    V4(BeelerReuter, FIBHIP_BR, 0, 5, 32, 32, 256),
    V4(BeelerReuter, FIBHIP_BR, 0, 5, 32, 32, 512),
    V4(BeelerReuter, FIBHIP_BR, 0, 1, 64, 16, 256),
#endif
    S4M(BeelerReuter, FIBHIP_BR, 1, 5, 54, 21, 2),
    S4M(BeelerReuter, FIBHIP_BR, 1, 5, 54, 24, 2),
    S4M(BeelerReuter, FIBHIP_BR, 1, 5, 54, 27, 3),
    S4M(BeelerReuter, FIBHIP_BR, 1, 5, 54, 28, 3),
    S4M(BeelerReuter, FIBHIP_BR, 1, 5, 54, 16, 2),
    S4M(BeelerReuter, FIBHIP_BR, 1, 5, 54, 40, 3),
    V4(BeelerReuter, FIBHIP_BR, 1, 1, 64, 4, 256),
#ifndef FIB_ONLY_BR                     // tuning alternatives (tools/sweep.py); a specialised build keeps the two defaults
    S4(BeelerReuter, FIBHIP_BR, 1, 5, 54, 21, 3),
    S4(BeelerReuter, FIBHIP_BR, 1, 3, 58, 19, 2),
    S4(BeelerReuter, FIBHIP_BR, 1, 2, 60, 19, 2),
    S4(BeelerReuter, FIBHIP_BR, 1, 2, 60, 19, 3),
    V4(BeelerReuter, FIBHIP_BR, 1, 5, 32, 32, 256),
    V4(BeelerReuter, FIBHIP_BR, 1, 5, 32, 32, 512),
    V4(BeelerReuter, FIBHIP_BR, 1, 1, 64, 16, 256),
#endif
#ifndef FIB_ONLY_BR
    // ---- Courtemanche (mode 0 fast set, 2 all variables) ----
    V4(Courtemanche, FIBHIP_COURT, Courtemanche::MODE_FAST, 1, 64, 4, 256),
    V4(Courtemanche, FIBHIP_COURT, Courtemanche::MODE_FAST, 1, 64, 8, 256),
    V4(Courtemanche, FIBHIP_COURT, Courtemanche::MODE_ALL, 1, 64, 4, 256),
    V4(Courtemanche, FIBHIP_COURT, Courtemanche::MODE_FASTSLOW, 1, 64, 4, 256),
    // the fast tick on the five per-cell aggregates of the slow variables (models.hpp CourtAgg)
    F2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 1, 64, 4, 256),
    F2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 1, 64, 8, 256),
    F2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FASTSLOW, 1, 64, 4, 256),
    // two and three consecutive fast ticks in one launch (fibhip_step defers ticks: see tick_multi); first entry of
    // each K = default, the others for tools/sweep.py (FIBHIP_COURT_MULTI2 / FIBHIP_COURT_MULTI3 = "TX,TY,NT")
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 20, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 14, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 16, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 18, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 22, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 24, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 25, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 28, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 26, 3),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 58, 12, 1),
    F2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 3, 32, 32, 256),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 60, 14, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 60, 12, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 60, 16, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 60, 18, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 60, 22, 2),
    FS2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 60, 30, 2),
    F2(CourtAgg, VM_COURT_AGG, CourtAgg::MODE_FAST, 2, 64, 16, 256),
    // ---- court_ultra.py with the ultra-slow `_us_` gate: 22 variables, single rate ----
    V4(CourtemancheUS, FIBHIP_COURT_US, CourtemancheUS::MODE_ALL, 1, 64, 4, 256),
#endif
#endif
};
static const int g_nvariants = (int)(sizeof g_variants / sizeof g_variants[0]);

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct PlanItem {
    int K;
    launch_fn fn;
    int TY, TX;
    const Variant *v = nullptr;     // the table entry it came from (run-time module kernels carry their launch data there)
};

// a traced model's device code loaded at run time (fibhip_module_load)
struct fibhip_module {
    hipModule_t mod = nullptr;
    int device = 0;
    int nvar = 0, spt = 1, nmodes = 1, consts_bytes = 4;
    unsigned masks[8] = {0};
    int K = 1, TX = 64, TY = 4, R = 3, TYB = 0, K2 = 1, TX2 = 64, TY2 = 4, R2 = 4;   // plan hints of the generated header
    std::vector<Variant> variants;
};

struct fibhip_ctx {
    fibhip_desc d;
    int nvar, spt, mode;
    size_t cells;
    int pitch;              // floats between rows of one state array (W planar, nvar*W row-interleaved)
    size_t vstride;         // floats between the first rows of consecutive state arrays (cells / W)
    hipStream_t s0, s1;
    bool own_s0;
    float *slab[2];
    bool own_slab;
    float *phase3;          // dpy | dpx | q4 | r4 | dpy*r4 | dpx*r4, each `cells` floats
    float *phi_dev;
    bool has_phase;
    int cur[FIB_MAXVAR];            // which slab holds variable v
    int nxt[FIB_MAXVAR];            // where the tick in flight writes it (valid between edges and commit)
    bool has_consts;
    Fenton::Consts kf;
    BeelerReuter::Consts kb;
    Courtemanche::Consts kc;
#ifdef FIB_CUSTOM_MODEL_INC
    Custom::Consts ku;
#endif
    std::vector<PlanItem> plan;
    hipEvent_t ev_main, ev_int, ev_t0, ev_t1;
    int phase_of_tick;      // 0 idle, 1 edges issued, 2 interior issued
    long launches, t_launches0;
    int own0, own1;         // owned local rows
    bool whole_in_edges;    // this tick's last launch was issued entirely by step_edges
    int pending;            // ticks fibhip_step has accepted but not launched yet (see fibhip_step)
    int multi_max;          // up to this many consecutive ticks go into one launch (1 = never; CourtAgg: 3)
    std::vector<PlanItem> plan_multi[4];    // [T]: the one-launch plan of T ticks, T = 2..multi_max
    // several TICKS per launch (strip_mt_kernel): grids whose tiles are all resident at once, one device, planar slab
    float *xbuf;            // exchange buffer of 16-byte cells [2][nvar/4][cells], allocated on first use
    unsigned *epochs;       // one epoch word per tile, 256 bytes apart, + the give-up word behind them
    unsigned epoch_base;    // value of every epoch word between two launches
    bool epochs_stale;      // the tiling may have changed since the words were last written: zero them first
    int mt_max;             // most ticks one launch advances (<= 1: never)
    int mt_cur;             // ticks the next launch waits for: 1 after any observation of the state, then see fibhip_step
    long long n_ticks, n_mt_launches, n_mt_ticks;   // fibhip_launch_stats
    long long n_spec_kept, n_spec_redone;           // series launched ahead that the caller cut short: stopped in time / recomputed
    int mt_run, mt_run_prev;        // ticks launched since the last observation of the state / between the two before it
    // run-ahead: a caller that alternates series of n ticks with ONE read-back (run() with image() every n ticks) gets the
    // next n ticks launched BEFORE the read-back's copy is waited for; see fibhip_get_state_direct
    int spec_n, spec_used;          // ticks computed ahead of the caller / how many of them fibhip_step has handed out
    int spec_nxt[FIB_MAXVAR];       // where the state lives once all of them are handed out
    int mt_run_prev2;               // the series before mt_run_prev
    int hist[8], nhist;             // lengths of the last series of ticks, oldest first (predict_series)
    bool series_fresh;              // ticks have run since the last observation of the state
    bool ahead_ok;                  // FIBHIP_AHEAD != 0
    unsigned spec_id;               // ... of the launch that ran ahead
    unsigned *host_word;            // page-locked (behind snap_flags), read by tile 0 over PCIe: {launch id << 16 | n}, see flush()
    unsigned *done_word, *done_word_dev;    // wait_s0: a word of page-locked memory of its own (host / device address) ...
    unsigned done_seq;              // ... and the value the stream writes into it when it has got that far
    bool no_stream_write;
    unsigned *snap_flags_dev;       // device address of snap_flags
    unsigned mt_ids;                // launch ids cycle through 1 .. mt_ids
    unsigned mt_seq;                // id of the last multi-tick launch (the host's word names the launch it is meant for)
    bool spec_trust;                // the caller has not broken a predicted series since its last two equal ones
    hipEvent_t ev_spec;
    unsigned *snap_flags;           // page-locked: one word per tile, raised by the tiles of a launch that carries a read-back
    unsigned snap_seq;
    bool mt_inflight;       // a multi-tick launch has been issued since the give-up word was last read
    bool dead;              // a multi-tick launch gave up waiting and the state could not be restored: void
    // A multi-tick launch that gives up must not cost the run (ionic.py:202-204 has no such failure).  Every such launch since
    // the stream was last known good is remembered with the buffers it READ: a launch writes the other slab only and the
    // launches queued behind a failed one find the give-up word at their first boundary and leave without writing, so the
    // state the FIRST failed launch started from is intact when the host finds out (`recover`).
    struct MtRec {
        unsigned id;        // the launch's id (the give-up word names it)
        int T;              // ticks it advances (a launch that ran ahead and was stopped in time: the ticks it did)
        bool counted;       // the handle's state has moved past these ticks (false: a run-ahead not handed out yet)
        int src[FIB_MAXVAR];
    };
    std::vector<MtRec> journal;
    long long n_fallbacks, n_replayed;      // launches that gave up and were recovered / ticks recomputed one launch per tick
    unsigned mt_wait_ms;    // a tile's bound on its wait for a neighbour (FIBHIP_MT_WAIT_MS, fibhip_set_mt_wait_ms); 0 = 2 s
    long fake_giveup_at, fake_seen;         // test switch FIBHIP_MT_FAKE_GIVEUP=n: the n-th multi-tick launch finds the give-up word raised
    bool recovering;
    int expect;             // ticks the caller has DECLARED to come in one series (fibhip_expect) and that have not been asked for yet, or 0
    bool expect_fresh;      // ... none of them has been asked for yet: the observation the caller makes first does not end the series
    bool ptr_exposed;       // fibhip_state_ptr has handed out a raw pointer: the caller may write the state at any time
    int ncu;                // compute units of the device
    // fibhip_trace_begin / _end: the launches in between, each between two HIP events
    struct TraceRec {
        hipEvent_t e0, e1;
        char name[96];
        int K, TX, TY, R, ticks;
    };
    std::vector<TraceRec> trace;
    bool tracing;
    fibhip_module *mod;     // FIBHIP_CUSTOM on a run-time module (fibhip_module_load), or null
    bool tuned;             // the plan has been checked against the other tile shapes on this very geometry (autotune)
    launch_fn fused_fn;     // Courtemanche: tick + 'slow' in one launch, or null
    int cycle, cpos;        // ghost zone = cycle * steps_per_tick rows: the halo is exchanged every `cycle` ticks;
                            // cpos = ticks done since the last exchange
    int span;               // ticks the launch being issued covers (1; T while tick_multi fuses T Courtemanche ticks)
    void *comm;             // ncclComm_t of the direct halo path (fibhip_comm_*), or null
    float *probe_host;      // pinned
    float *stage;           // pinned staging buffer for get_state/set_state (one array), allocated on first use
    // Courtemanche, fast policy: the fast tick reads five per-cell aggregates of the slow
    // variables (models.hpp CourtAgg) instead of the variables themselves.  'slow' rewrites them; any other write to the
    // state (set_state) marks them stale and the next tick recomputes them first.
    float *agg;             // CourtAgg::NAGG arrays laid out like the state arrays (planar, or row-interleaved at the
    size_t agg_stride;      // slab's pitch on row-block shards), `agg_stride` floats apart; or null
    bool use_agg, agg_dirty;
    bool agg_ghost_dirty;   // row-block shards: a halo exchange has rewritten the ghost rows' slow variables
};

static const void *consts_of(fibhip_ctx *h)
{
    if (h->mod) return nullptr;                    // generated models carry their constants as literals
    switch (h->d.model) {
    case FIBHIP_FENTON4V: return &h->kf;
    case FIBHIP_BR: return &h->kb;
#ifdef FIB_CUSTOM_MODEL_INC
    case FIBHIP_CUSTOM: return &h->ku;
#endif
    default: return &h->kc;
    }
}

extern "C" int fibhip_nvar(int model)
{
    switch (model) {
    case FIBHIP_FENTON4V: return Fenton::NVAR;
    case FIBHIP_BR: return BeelerReuter::NVAR;
    case FIBHIP_COURT: return Courtemanche::NVAR;
    case FIBHIP_COURT_US: return CourtemancheUS::NVAR;
#ifdef FIB_CUSTOM_MODEL_INC
    case FIBHIP_CUSTOM: return Custom::NVAR;
#endif
    default: return fail(FIBHIP_EINVAL, "unknown model %d", model);
    }
}

extern "C" int fibhip_default_steps_per_tick(int model)
{
    switch (model) {
    case FIBHIP_FENTON4V: return Fenton::DEFAULT_STEPS;
    case FIBHIP_BR: return BeelerReuter::DEFAULT_STEPS;
    case FIBHIP_COURT: return Courtemanche::DEFAULT_STEPS;
    case FIBHIP_COURT_US: return CourtemancheUS::DEFAULT_STEPS;
#ifdef FIB_CUSTOM_MODEL_INC
    case FIBHIP_CUSTOM: return Custom::DEFAULT_STEPS;
#endif
    default: return fail(FIBHIP_EINVAL, "unknown model %d", model);
    }
}

extern "C" int fibhip_abi_version(void) { return FIBHIP_ABI_VERSION; }

extern "C" int fibhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static const Variant *find_variant(const fibhip_ctx *h, int K, const int *want /*TX,TY,NT or null*/, int mode = -1)
{
    const int fast = (h->d.flags & FIBHIP_FAST) ? 1 : 0, phase = h->has_phase ? 1 : 0;
    if (mode < 0) mode = h->mode;
    // fenton_simple.py's Laplacian is a property of the kernel's model type (FentonZP): its own rows of the table
    const int vmodel = (h->d.model == FIBHIP_FENTON4V && (h->d.flags & FIBHIP_ZEROPAD)) ? VM_FENTON_ZP
                       : (h->use_agg ? VM_COURT_AGG : h->d.model);
    const Variant *tab = h->mod ? h->mod->variants.data() : g_variants;
    const int ntab = h->mod ? (int)h->mod->variants.size() : g_nvariants;
    for (int i = 0; i < ntab; ++i) {
        const Variant &v = tab[i];
        if (v.kind == MK_POINTWISE) continue;
        if (v.model != vmodel || v.mode != mode || v.fast != fast || v.phase != phase || v.K != K) continue;
        if (want && (v.TX != want[0] || v.TY != want[1] || v.NT != want[2])) continue;
        return &v;
    }
    return nullptr;
}

// Decompose one tick of `spt` sub-steps into launches.  Default fusion depth per model comes from
// the measurements recorded in DESIGN.md; FIBHIP_K / FIBHIP_VARIANT override it for sweeps.
static int build_plan(fibhip_ctx *h)
{
    h->plan.clear();
    h->tuned = false;
    h->epochs_stale = true;
    int prefK = 0, want[3], nwant = 0;
    if (const char *e = getenv("FIBHIP_VARIANT")) {
        int k = 0;
        if (sscanf(e, "%d,%d,%d,%d", &k, &want[0], &want[1], &want[2]) == 4) {
            prefK = k;
            nwant = 1;
        }
    }
    if (!prefK)
        if (const char *e = getenv("FIBHIP_K")) prefK = atoi(e);
    if (!prefK) {
        // Measured on MI355X (tools/sweep.py).  Beeler-Reuter / Courtemanche spend their time
        // in the transcendental pipe (64 / ~70 per cell-step): redundant rim cells cost more than the
        // launches they save, so one sub-step per launch.  Fenton is cheap per cell: fuse — as deep as the
        // tick when the grid gives each CU about one tile (launch/latency-bound), 5 sub-steps with a
        // smaller rim when there are many tiles per CU (throughput-bound), fatter waves when there are
        // very many (occupancy).
        prefK = 1;
        if (h->mod) {
            // the same rule as the FIB_CUSTOM_* block below, with the generated header's numbers at run time
            const fibhip_module &m = *h->mod;
            const int ext = (h->cycle - 1) * h->spt;
            const int rows = (h->own1 - h->own0) + (h->d.ghost_top ? ext : 0) + (h->d.ghost_bottom ? ext : 0);
            const long tiles = (long)((h->d.width + m.TX - 1) / m.TX) * ((rows + m.TY - 1) / m.TY);
            prefK = tiles <= (m.K2 > 1 ? 256 : 512) ? m.K : m.K2;
            if (m.TYB > 0 && prefK == m.K && prefK > 1) {
                const long tiles_b = (long)((h->d.width + m.TX - 1) / m.TX) * ((rows + m.TYB - 1) / m.TYB);
                want[0] = m.TX; want[1] = tiles_b <= 256 ? m.TYB : m.TY; want[2] = -m.R;
                nwant = 1;
            }
        }
#ifdef FIB_CUSTOM_MODEL_INC
        if (h->d.model == FIBHIP_CUSTOM) {
            // one tile per CU or two: latency-bound, fuse the whole tick; more: throughput-bound (see the generator)
            const int ext = (h->cycle - 1) * h->spt;
            const int rows = (h->own1 - h->own0) + (h->d.ghost_top ? ext : 0) + (h->d.ghost_bottom ? ext : 0);
            const long tiles = (long)((h->d.width + FIB_CUSTOM_TX - 1) / FIB_CUSTOM_TX) *
                               ((rows + FIB_CUSTOM_TY - 1) / FIB_CUSTOM_TY);
            // a cheap graph has a shallower fusion to fall back to (K2 > 1): deep fusion only while every CU has at
            // most one tile (the measured Fenton rule); a heavy graph (K2 == 1) keeps it up to two tiles per CU
            // (the measured Beeler-Reuter rule)
            prefK = tiles <= (FIB_CUSTOM_K2 > 1 ? 256 : 512) ? FIB_CUSTOM_K : FIB_CUSTOM_K2;
#if FIB_CUSTOM_TYB > 0
            // the 15-wave tile when it still gives every CU at most one tile (measured on Fenton: 18.5 vs 19.6 us)
            const long tiles_b = (long)((h->d.width + FIB_CUSTOM_TX - 1) / FIB_CUSTOM_TX) *
                                 ((rows + FIB_CUSTOM_TYB - 1) / FIB_CUSTOM_TYB);
            if (prefK == FIB_CUSTOM_K && prefK > 1) {
                want[0] = FIB_CUSTOM_TX; want[1] = tiles_b <= 256 ? FIB_CUSTOM_TYB : FIB_CUSTOM_TY; want[2] = -FIB_CUSTOM_R;
                nwant = 1;
            }
#endif
        }
#endif
        if (h->d.model == FIBHIP_BR) {
            // up to two tiles per CU: the tick is launch/latency-bound, so all 5 sub-steps in one launch;
            // large grids are bound by the transcendental pipe, where the redundant rim costs more than launches
            const int ext = (h->cycle - 1) * h->spt;
            const int rows = (h->own1 - h->own0) + (h->d.ghost_top ? ext : 0) + (h->d.ghost_bottom ? ext : 0);
            const long tiles = (long)((h->d.width + 53) / 54) * ((rows + 20) / 21);
            if (tiles <= 512 && h->spt == 5) {
                prefK = 5; want[0] = 54; want[1] = 21; want[2] = -2;      // measured: profiles/r01_sweep_br512.txt
                nwant = 1;
            }
        }
        if (h->d.model == FIBHIP_FENTON4V) {
            // rows of the largest launch: the first tick of an exchange cycle also advances the ghost rows
            const int ext = (h->cycle - 1) * h->spt;
            const int rows = (h->own1 - h->own0) + (h->d.ghost_top ? ext : 0) + (h->d.ghost_bottom ? ext : 0);
            const int W = h->d.width;
            const long tx10 = (W + 43) / 44, tx5 = (W + 53) / 54;
            const long tiles10 = tx10 * ((rows + 24) / 25), t28 = tx10 * ((rows + 27) / 28);
            const long t21 = tx5 * ((rows + 20) / 21), t23 = tx5 * ((rows + 22) / 23);
            const long r21 = (t21 + 255) / 256, r23 = (t23 + 255) / 256;       // tiles per CU, rounded up
            const bool sharded = h->d.ghost_top || h->d.ghost_bottom;
            // Measured (tools/sweep_sizes.py, profiles/r01_sweep_sizes.txt): what matters is how many tiles a CU gets.
            // K=10: 18 us with <= 1 tile per CU, ~34 us with 2.  K=5 (two launches), R=3: 27 us with <= 2 per CU, 38 us
            // with 3; the 23-row tile fills its 11 waves exactly (33 rows) and is taken when it saves a whole round
            // of tiles.  Beyond that the fatter R=4 waves win, with the wave-exact 22-row tile.
            if (h->d.flags & FIBHIP_ZEROPAD) {
                // fenton_simple.py's Laplacian exists in the flat tick_kernel only (measured at 512^2: 10 x 32x32 x 1024
                // threads 20.8 us per 10 steps, 5 x 32x32 x 512 25.4)
                const long t32 = (long)((W + 31) / 32) * ((rows + 31) / 32);
                prefK = t32 <= 512 ? 10 : 5; want[0] = 32; want[1] = 32; want[2] = t32 <= 512 ? 1024 : 512;
            } else if (tiles10 <= 256 || t28 <= 256) {
                prefK = 10; want[0] = 44; want[1] = tiles10 <= 256 ? 25 : 28; want[2] = -3;
            } else if (sharded && tiles10 <= 512) {
                // row blocks: the launch is sized for the first tick of an exchange cycle, later ticks have fewer
                // ghost rows to advance (measured 21.6 us per tick for 512 + 2 x 40 rows)
                prefK = 10; want[0] = 44; want[1] = 25; want[2] = -3;
            } else if (r21 <= 2 || r23 <= 2) {
                prefK = 5; want[0] = 54; want[1] = r21 <= 2 ? 21 : 23; want[2] = -3;
            } else if (tiles10 <= 512 || t28 <= 512) {
                prefK = 10; want[0] = 44; want[1] = tiles10 <= 512 ? 25 : 28; want[2] = -3;
            } else if (r21 <= 3 || r23 <= 3) {
                prefK = 5; want[0] = 54; want[1] = r21 <= 3 ? 21 : 23; want[2] = -3;
            } else {
                prefK = 5; want[0] = 54; want[1] = 22; want[2] = -4;      // measured best at 1024^2 .. 4096^2
            }
            nwant = 1;
        }
    }
    const int maxghost = (h->d.ghost_top > 0 || h->d.ghost_bottom > 0)
                             ? (h->d.ghost_top > 0 && h->d.ghost_bottom > 0
                                    ? (h->d.ghost_top < h->d.ghost_bottom ? h->d.ghost_top : h->d.ghost_bottom)
                                    : (h->d.ghost_top > 0 ? h->d.ghost_top : h->d.ghost_bottom))
                             : 1 << 30;
    int rem = h->spt;
    while (rem > 0) {
        const Variant *best = nullptr;
        for (int K = (prefK < rem ? prefK : rem); K >= 1 && !best; --K) {
            if (K > maxghost) continue;
            best = find_variant(h, K, nwant ? want : nullptr);
            if (!best && nwant) best = find_variant(h, K, nullptr);
        }
        if (!best) return fail(FIBHIP_EINVAL, "no kernel variant for model %d mode %d", h->d.model, h->mode);
        h->plan.push_back({best->K, best->fn, best->TY, best->TX, best});
        rem -= best->K;
    }
    // Courtemanche: the reference's driver fires 'slow' right after every 10th tick (court.py:612-617).  When the
    // last tick of a fibhip_step call is still pending at that moment, both run as ONE launch (MODE_FASTSLOW): the
    // 21 arrays are read once instead of twice.  Requirements: a single K=1 launch per tick, no ghost rows, and
    // every border cell's inward neighbour inside the border cell's own tile.
    h->fused_fn = nullptr;
#if !defined(FIB_CUSTOM_ONLY) && !defined(FIB_ONLY_BR)
    if (h->d.model == FIBHIP_COURT && h->mode == Courtemanche::MODE_FAST && h->plan.size() == 1 && h->plan[0].K == 1 &&
        !h->d.ghost_top && !h->d.ghost_bottom && !getenv("FIBHIP_NO_LAZY")) {
        const int want1[3] = {h->plan[0].TX, h->plan[0].TY, 256};
        const Variant *v = find_variant(h, 1, want1, Courtemanche::MODE_FASTSLOW);
        if (v && (h->d.height - 1) % v->TY != 0 && (h->d.width - 1) % v->TX != 0) h->fused_fn = v->fn;
    }
    // Courtemanche on aggregates: one tick is one sub-step and moves 16 arrays for ~300 instructions per cell, so
    // two or three consecutive ticks are blocked in time like the sub-steps of a Fenton tick (fibhip_step defers
    // ticks until a launch is full; every entry point that observes the state launches what is pending first)
    h->multi_max = 1;
    for (int T = 2; T <= 3; ++T) h->plan_multi[T].clear();
    // (row blocks: the ticks between two halo exchanges are fused the same way — the ghost zone must be deep enough for the
    // fused ticks to stay inside the exchange cycle; the tick that ends the cycle stays a launch of its own)
    const bool shard = h->d.ghost_top || h->d.ghost_bottom;
    if (h->use_agg && h->mode == CourtAgg::MODE_FAST && h->plan.size() == 1 && h->plan[0].K == 1 && (!shard || h->cycle >= 3) &&
        !getenv("FIBHIP_NO_MULTI")) {
        for (int T = 2; T <= (shard ? imin(3, h->cycle - 1) : 3); ++T) {
            int w[3];
            const char *e = getenv(T == 2 ? "FIBHIP_COURT_MULTI2" : "FIBHIP_COURT_MULTI3");
            const bool have = e && sscanf(e, "%d,%d,%d", &w[0], &w[1], &w[2]) == 3;
            const Variant *v = find_variant(h, T, have ? w : nullptr, CourtAgg::MODE_FAST);
            if (!v) break;
            h->plan_multi[T].push_back({v->K, v->fn, v->TY, v->TX, v});
            h->multi_max = T;
        }
    }
#endif
    return 0;
}

static int create_impl(const fibhip_desc *desc, fibhip_ctx *&h);

extern "C" int fibhip_create(const fibhip_desc *desc, fibhip_t *out)
{
    if (!desc || !out) return fail(FIBHIP_EINVAL, "null argument");
    fibhip_ctx *h = nullptr;
    const int rc = create_impl(desc, h);
    if (rc) {
        if (h) fibhip_destroy(h);              // releases whatever had been acquired before the failure
        return rc;
    }
    *out = h;
    return 0;
}

static int create_impl(const fibhip_desc *desc, fibhip_ctx *&h)
{
    if (desc->struct_size != (int)sizeof(fibhip_desc))
        return fail(FIBHIP_EINVAL, "fibhip_desc size mismatch: caller %d, library %d", desc->struct_size,
                    (int)sizeof(fibhip_desc));
    fibhip_module *mod = (fibhip_module *)desc->module;
    if (mod && desc->model != FIBHIP_CUSTOM) return fail(FIBHIP_EINVAL, "a run-time module serves FIBHIP_CUSTOM only");
    if (mod && mod->device != desc->device)
        return fail(FIBHIP_EINVAL, "the module was loaded on device %d, the handle asks for device %d", mod->device, desc->device);
    const int nv = mod ? mod->nvar : fibhip_nvar(desc->model);
    if (nv < 0) return nv;
    const int Hg = desc->global_height ? desc->global_height : desc->height;
    if (desc->height < 3 || desc->width < 3 || Hg < 3)
        return fail(FIBHIP_EINVAL, "grid must be at least 3x3 (got %dx%d)", desc->height, desc->width);
    if (desc->row_offset < 0 || desc->row_offset + desc->height > Hg || desc->ghost_top < 0 || desc->ghost_bottom < 0 ||
        desc->ghost_top + desc->ghost_bottom >= desc->height)
        return fail(FIBHIP_EINVAL, "inconsistent row-block description");
    if ((desc->ghost_top > 0) != (desc->row_offset > 0) ||
        (desc->ghost_bottom > 0) != (desc->row_offset + desc->height < Hg))
        return fail(FIBHIP_EINVAL, "ghost rows must exist exactly on the sides that have a neighbour");
    {   // the kernels index with 32-bit ints: rows * (floats between rows) of one array view must stay below 2^31
        const long long pitch = (desc->flags & FIBHIP_ROW_INTERLEAVED) ? (long long)nv * desc->width : desc->width;
        if ((long long)desc->height * pitch >= (1LL << 31))
            return fail(FIBHIP_EINVAL, "grid too large for 32-bit indexing: %d rows x %lld floats per row", desc->height, pitch);
    }
    if (!(desc->dt > 0.0)) return fail(FIBHIP_EINVAL, "dt must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FIBHIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (desc->device < 0 || desc->device >= ndev) return fail(FIBHIP_EINVAL, "device %d out of range", desc->device);
    HIPCHK(hipSetDevice(desc->device));

    h = new (std::nothrow) fibhip_ctx();
    if (!h) return fail(FIBHIP_ENOMEM, "out of host memory");
    h->d = *desc;
    h->d.global_height = Hg;
    h->nvar = nv;
    h->mod = mod;
    h->spt = desc->steps_per_tick > 0 ? desc->steps_per_tick : (mod ? mod->spt : fibhip_default_steps_per_tick(desc->model));
    h->cells = (size_t)desc->height * desc->width;
    const bool interleaved = (desc->flags & FIBHIP_ROW_INTERLEAVED) != 0;
    h->pitch = interleaved ? nv * desc->width : desc->width;
    h->vstride = interleaved ? (size_t)desc->width : h->cells;
    h->own0 = desc->ghost_top;
    h->own1 = desc->height - desc->ghost_bottom;
    h->mode = 0;
    if (desc->model == FIBHIP_BR && (desc->flags & FIBHIP_CHEBY)) h->mode = BeelerReuter::MODE_CHEBY;
    if (desc->model == FIBHIP_COURT)
        h->mode = (desc->flags & FIBHIP_ALLVARS) ? Courtemanche::MODE_ALL : Courtemanche::MODE_FAST;
    if (desc->model == FIBHIP_COURT_US) h->mode = CourtemancheUS::MODE_ALL;
    const int ming = (desc->ghost_top && desc->ghost_bottom)
                         ? (desc->ghost_top < desc->ghost_bottom ? desc->ghost_top : desc->ghost_bottom)
                         : (desc->ghost_top ? desc->ghost_top : desc->ghost_bottom);
    if ((desc->flags & FIBHIP_ZEROPAD) && desc->model != FIBHIP_FENTON4V)
        return fail(FIBHIP_EINVAL, "FIBHIP_ZEROPAD exists for the Fenton 4v model only (fenton_simple.py)");
    if ((desc->flags & FIBHIP_ZEROPAD) && (desc->ghost_top || desc->ghost_bottom))
        return fail(FIBHIP_EINVAL, "FIBHIP_ZEROPAD is a single-device option (no row blocks)");
    if ((desc->ghost_top || desc->ghost_bottom) && ming < h->spt)
        return fail(FIBHIP_EINVAL, "ghost width %d < steps_per_tick %d", ming, h->spt);
    h->cycle = (desc->ghost_top || desc->ghost_bottom) ? ming / h->spt : 1;
    h->cpos = 0;
    h->span = 1;

    // scalars: every Python-float product is formed in double and rounded once
    const double dt = desc->dt, diff = desc->diff;
    h->kf.dt = (float)dt;
    h->kf.ddt = (float)(diff * dt);
    h->kf.cvp = (float)(1.0 - dt / 3.33);          // tau_vp, tau_vn, tau_wp, tau_wn (fenton.py:60-63)
    h->kf.cvn = (float)(1.0 - dt / 19.2);
    h->kf.dvn = (float)(dt / 19.2);
    h->kf.cwp = (float)(1.0 - dt / 160.0);
    h->kf.cwn = (float)(1.0 - dt / 75.0);
    h->kf.dwn = (float)(dt / 75.0);
    h->kb.dt = (float)dt;
    h->kb.ddt = (float)(diff * dt);
    h->kb.mdt = (float)(-dt);
    h->kb.mdt_skip = (float)(-(dt * 5));
    h->kb.skip = (desc->flags & FIBHIP_HOLD) ? 2 : ((desc->flags & FIBHIP_SKIP) ? 1 : 0);
    memset(h->kb.cheb, 0, sizeof h->kb.cheb);
    {
        const bool all = (desc->flags & FIBHIP_ALLVARS) != 0 || desc->model == FIBHIP_COURT_US;
        const double dts = all ? dt : dt * 10;                     // court.py:118-122
        const double chronic = (desc->flags & FIBHIP_CHRONIC) ? 1.0 : 0.0;
        h->kc.dtf = (float)dt;
        h->kc.dts = (float)dts;
        h->kc.mdt_f = (float)(-dt);
        h->kc.mdt_s = (float)(-dts);
        h->kc.ddt = (float)(diff * dt);
        h->kc.em1_fCa = expm1f((float)(-dts / 2.0));               // tau_f_Ca = 2.0, court.py:160,189
        h->kc.em1_u = expm1f((float)(-dts / 8.0));                 // tau_u = 8.0,   court.py:161,243
        h->kc.chronic = (float)chronic;
        h->kc.c_to = (float)((1.0 - 0.5 * chronic) * 100 * 0.1652);   // court.py:193
        h->kc.c_Kur = (float)((1.0 - 0.5 * chronic) * 100);           // court.py:194
        h->kc.c_CaL = (float)((1.0 - 0.7 * chronic) * 100 * 0.12375); // court.py:218
    }
    h->has_consts = !(desc->model == FIBHIP_BR && (desc->flags & FIBHIP_CHEBY));

    if (desc->stream) {
        h->s0 = (hipStream_t)desc->stream;
        h->own_s0 = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&h->s0, hipStreamNonBlocking));
        h->own_s0 = true;
    }
    HIPCHK(hipStreamCreateWithFlags(&h->s1, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_int, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&h->ev_t0));
    HIPCHK(hipEventCreate(&h->ev_t1));
    const size_t slab_bytes = (size_t)nv * h->cells * sizeof(float);
    if (desc->ext_slab[0] && desc->ext_slab[1]) {
        h->slab[0] = (float *)desc->ext_slab[0];
        h->slab[1] = (float *)desc->ext_slab[1];
        h->own_slab = false;
    } else {
        h->own_slab = true;                       // before the loop: a partial allocation is still ours to free
        for (int i = 0; i < 2; ++i) {
            if (hipMalloc((void **)&h->slab[i], slab_bytes) != hipSuccess) {
                h->slab[i] = nullptr;
                return fail(FIBHIP_ENOMEM, "hipMalloc of %zu bytes failed", slab_bytes);
            }
            HIPCHK(hipMemsetAsync(h->slab[i], 0, slab_bytes, h->s0));
        }
    }
    HIPCHK(hipMalloc((void **)&h->phase3, 6 * h->cells * sizeof(float)));
    HIPCHK(hipMalloc((void **)&h->phi_dev, h->cells * sizeof(float)));
    HIPCHK(hipHostMalloc((void **)&h->probe_host, 64, hipHostMallocDefault));
    h->has_phase = false;
    for (int v = 0; v < FIB_MAXVAR; ++v) h->cur[v] = h->nxt[v] = 0;
    h->phase_of_tick = 0;
    h->launches = 0;
    h->pending = 0;
    h->multi_max = 1;
    h->fused_fn = nullptr;
    h->comm = nullptr;
    h->tuned = false;
    h->tracing = false;
    h->xbuf = nullptr;
    h->epochs = nullptr;
    h->epoch_base = 0;
    h->epochs_stale = true;
    h->mt_cur = 1;
    h->mt_run = h->mt_run_prev = h->mt_run_prev2 = 0;
    h->nhist = 0;
    h->spec_n = h->spec_used = 0;
    h->series_fresh = false;
    h->spec_trust = true;
    h->mt_seq = h->spec_id = 0;
    {
        const char *e = getenv("FIBHIP_MT_IDS");
        h->mt_ids = (e && atoi(e) >= 2 && atoi(e) <= 65535) ? (unsigned)atoi(e) : 65535u;
    }
    h->host_word = h->snap_flags_dev = nullptr;
    h->done_seq = 0;
    h->done_word = h->done_word_dev = nullptr;
    {
        const char *e = getenv("FIBHIP_STREAM_WRITE");             // 0: notice the end of the stream's work through hipStreamQuery, as before
        h->no_stream_write = e && atoi(e) == 0;
    }
    h->snap_flags = nullptr;
    h->snap_seq = 0;
    {
        // Run-ahead starts the caller's NEXT ticks before it has asked for them.  A caller that owns the slabs
        // (desc->ext_slab: it can write them between two calls without the library knowing) never gets it; a caller that has
        // been handed a raw pointer (fibhip_state_ptr) loses it from then on — the same rule the aggregates follow.
        const char *e = getenv("FIBHIP_AHEAD");
        h->ahead_ok = !(e && atoi(e) == 0) && h->own_slab;
    }
    h->n_fallbacks = h->n_replayed = 0;
    h->recovering = false;
    h->expect = 0;
    h->expect_fresh = false;
    h->ptr_exposed = false;
    {
        const char *e = getenv("FIBHIP_MT_WAIT_MS");
        h->mt_wait_ms = (e && atol(e) > 0) ? (unsigned)(atol(e) > 0xFFFFFFl ? 0xFFFFFFl : atol(e)) : 0u;
        const char *f = getenv("FIBHIP_MT_FAKE_GIVEUP");
        h->fake_giveup_at = (f && atol(f) > 0) ? atol(f) : 0;
        h->fake_seen = 0;
    }
    HIPCHK(hipEventCreateWithFlags(&h->ev_spec, hipEventDisableTiming));
    h->n_ticks = h->n_mt_launches = h->n_mt_ticks = h->n_spec_kept = h->n_spec_redone = 0;
    h->mt_inflight = false;
    h->dead = false;
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, desc->device));
        h->ncu = prop.multiProcessorCount;
        // FIBHIP_MT=0 switches multi-tick launches off, FIBHIP_MT_MAX bounds the ticks of one launch
        const char *e = getenv("FIBHIP_MT"), *em = getenv("FIBHIP_MT_MAX");
        h->mt_max = (e && atoi(e) == 0) ? 1 : (em && atoi(em) > 0 ? imin(atoi(em), 4096) : MT_MAX_TICKS);
        if (interleaved || desc->ghost_top || desc->ghost_bottom || (long long)h->cells * ((nv + 3) / 4 * 4) * 8 >= (1LL << 31))
            h->mt_max = 1;
        // a process-wide CU mask takes compute units away that multiProcessorCount still reports: the tiles of a grid
        // "that fits" would then not all be resident (a launch would give up after its bound and the handle fall back, §2d of
        // DESIGN.md — correct, but two seconds late): never start
        for (const char *var : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"}) {
            const char *m = getenv(var);
            if (m && *m) h->mt_max = 1;
        }
    }
    h->agg = nullptr;
    h->use_agg = false;
    h->agg_dirty = true;
    h->agg_ghost_dirty = false;
    h->agg_stride = 0;
#if !defined(FIB_CUSTOM_ONLY) && !defined(FIB_ONLY_BR)
    {
        // A caller-owned slab can be written behind the library's back — except on a row-block shard, whose contract
        // is that only the ghost rows change, and only in the exchange between step_edges and step_commit.
        const char *e = getenv("FIBHIP_COURT_AGG");
        const bool shard = desc->ghost_top || desc->ghost_bottom;
        if (desc->model == FIBHIP_COURT && (desc->flags & FIBHIP_FAST) && !(desc->flags & FIBHIP_ALLVARS) &&
            (h->own_slab || shard) && !(e && atoi(e) == 0)) {
            const bool il = h->pitch != desc->width;       // row-interleaved: the aggregates share the slab's row pitch
            const size_t floats = il ? (size_t)desc->height * h->pitch : (size_t)CourtAgg::NAGG * h->cells;
            HIPCHK(hipMalloc((void **)&h->agg, floats * sizeof(float)));
            h->agg_stride = il ? (size_t)desc->width : h->cells;
            h->use_agg = true;
        }
    }
#endif
    return build_plan(h);
}

extern "C" int fibhip_comm_free(fibhip_t h);
static void mt_forget(fibhip_ctx *h);

extern "C" int fibhip_destroy(fibhip_t h)
{
    if (!h) return 0;
    fibhip_comm_free(h);
    hipSetDevice(h->d.device);
    if (h->spec_n > 0 && h->host_word)                 // a launch that ran ahead of the caller: nobody wants its ticks any more
        __atomic_store_n(h->host_word, (h->spec_id << 16) | MT_CANCEL, __ATOMIC_RELEASE);
    if (h->s0) hipStreamSynchronize(h->s0);
    if (h->s1) hipStreamSynchronize(h->s1);
    if (h->own_slab) {
        if (h->slab[0]) hipFree(h->slab[0]);
        if (h->slab[1]) hipFree(h->slab[1]);
    }
    if (h->phase3) hipFree(h->phase3);
    if (h->phi_dev) hipFree(h->phi_dev);
    if (h->agg) hipFree(h->agg);
    if (h->xbuf) hipFree(h->xbuf);
    if (h->epochs) hipFree(h->epochs);
    for (auto &r : h->trace) {
        if (r.e0) hipEventDestroy(r.e0);
        if (r.e1) hipEventDestroy(r.e1);
    }
    mt_forget(h);
    if (h->snap_flags) hipHostFree(h->snap_flags);
    if (h->done_word) hipHostFree(h->done_word);
    if (h->probe_host) hipHostFree(h->probe_host);
    if (h->stage) hipHostFree(h->stage);
    if (h->ev_spec) hipEventDestroy(h->ev_spec);
    if (h->ev_main) hipEventDestroy(h->ev_main);
    if (h->ev_int) hipEventDestroy(h->ev_int);
    if (h->ev_t0) hipEventDestroy(h->ev_t0);
    if (h->ev_t1) hipEventDestroy(h->ev_t1);
    if (h->s1) hipStreamDestroy(h->s1);
    if (h->own_s0 && h->s0) hipStreamDestroy(h->s0);
    delete h;
    return 0;
}

// ---- host-visible waits ---------------------------------------------------------------------------------
// A blocking hipStreamSynchronize parks the thread on an interrupt: 5-10 us until it runs again, a tenth of a 20-tick
// region of the 512x512 benchmark and a third of one image() read-back.  Poll instead for as long as short waits last
// (FIBHIP_SPIN_US, default 2000 us; 0 = always block), then block.
static long spin_us()
{
    static const long v = [] {
        const char *e = getenv("FIBHIP_SPIN_US");
        return e ? atol(e) : 2000L;
    }();
    return v;
}
static hipError_t wait_stream(hipStream_t s)
{
    const long lim = spin_us();
    if (lim <= 0) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();                                  // hipErrorNotReady is not an error to report later
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(lim)) return hipStreamSynchronize(s);
    }
}
// The end of everything enqueued on the handle's stream (any handle: single device or shard, whatever the launch plan), noticed
// through a word of page-locked host memory that the stream itself
// writes when it gets there (hipStreamWriteValue32 behind the work) instead of through hipStreamQuery: the host spins on its own
// memory, and knows 3 us sooner (tools/ubench/notice.hip: launch call -> notice, minus the kernel: 8.5 us by hipStreamQuery spin,
// 7.6 by hipStreamSynchronize, 5.3 this way) — 1 % of a 20-tick region of the benchmark, and of every read-back of a driver loop.
static hipError_t wait_s0(fibhip_ctx *h)
{
    const long lim = spin_us();
    if (lim <= 0 || h->no_stream_write) return wait_stream(h->s0);
    if (!h->done_word) {                               // (first use: 64 bytes of page-locked memory per handle)
        if (hipHostMalloc((void **)&h->done_word, 64, hipHostMallocDefault) != hipSuccess ||
            hipHostGetDevicePointer((void **)&h->done_word_dev, h->done_word, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (h->done_word) hipHostFree(h->done_word);
            h->done_word = nullptr;
            h->no_stream_write = true;
            return wait_stream(h->s0);
        }
        *h->done_word = 0u;
    }
    const unsigned seq = ++h->done_seq;
    volatile unsigned *w = h->done_word;
    if (hipStreamWriteValue32(h->s0, h->done_word_dev, seq, 0) != hipSuccess) {
        (void)hipGetLastError();
        h->no_stream_write = true;                     // (a runtime or a stream that cannot: the old way from now on)
        return wait_stream(h->s0);
    }
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(lim)) return hipStreamSynchronize(h->s0);
    }
    return hipSuccess;
}
static hipError_t wait_event(hipEvent_t ev)
{
    const long lim = spin_us();
    if (lim <= 0) return hipEventSynchronize(ev);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(lim)) return hipEventSynchronize(ev);
    }
}

// wait for everything enqueued on the handle's stream; reports a multi-tick launch that gave up (strip_mt_kernel)
static int tick_now(fibhip_t h);

// A multi-tick launch gave up (the give-up word names it).  The stream is idle.  The launch wrote the other slab only, and
// every multi-tick launch queued behind it left at its first boundary without writing: the state it STARTED from is where its
// journal record says.  Go back there, switch multi-tick launches off for this handle, and recompute — one launch per tick,
// bit-identical by construction — the ticks the handle's state had already moved past.
static int recover(fibhip_ctx *h, unsigned id)
{
    size_t i = 0;
    while (i < h->journal.size() && h->journal[i].id != id) ++i;
    if (i == h->journal.size() || h->recovering) {
        h->dead = true;
        return fail(FIBHIP_EHIP, "%s", MT_DEAD_MSG);
    }
    int lost = 0;
    for (size_t j = i; j < h->journal.size(); ++j)
        if (h->journal[j].counted) lost += h->journal[j].T;
    memcpy(h->cur, h->journal[i].src, sizeof h->cur);
    h->journal.clear();
    h->mt_max = 1;                                    // (mt_variant() is null from here on: no run-ahead, no series either)
    h->mt_cur = 1;
    h->epochs_stale = true;
    HIPCHK(hipMemsetAsync(h->epochs + (size_t)MT_MAX_TILES * MT_EPOCH_STRIDE, 0, 3 * MT_EPOCH_STRIDE * sizeof(unsigned), h->s0));
    __atomic_store_n(h->host_word + MT_GIVEUP_WORD, 0u, __ATOMIC_RELEASE);
    h->n_fallbacks++;
    h->n_replayed += lost;
    h->recovering = true;
    int rc = 0;
    for (int t = 0; t < lost && rc == 0; ++t) rc = tick_now(h);
    h->recovering = false;
    if (rc) {
        h->dead = true;
        return rc;
    }
    HIPCHK(wait_stream(h->s0));
    return 0;
}

static int sync_s0(fibhip_ctx *h)
{
    HIPCHK(wait_s0(h));
    if (h->mt_inflight && h->epochs) {
        // the tile that gave up first has written its launch's id into HOST memory (page-locked, behind the host's own word):
        // nothing is copied from the device behind every launch (a 4-byte device-to-host copy at the end of every
        // synchronising call cost a 20-tick benchmark region 5-7 us of its 250)
        h->mt_inflight = false;
        const unsigned gave_up = __atomic_load_n(h->host_word + MT_GIVEUP_WORD, __ATOMIC_ACQUIRE);
        if (gave_up) return recover(h, gave_up);
        h->journal.clear();                           // every launch so far has ended, and ended well
    }
    return 0;
}
// Nothing but another multi-tick launch is ever queued behind a multi-tick launch that has not been confirmed: a launch that
// gave up leaves the state it started from intact only as long as whatever follows it writes nothing — multi-tick launches find
// the give-up word and leave; a plain tick, a pace, a host write would not.  So those wait for the stream first.
static int confirm(fibhip_ctx *h)
{
    return (h->mt_inflight && h->epochs) ? sync_s0(h) : 0;
}
#define CONFIRM(h)                                                                                 \
    do {                                                                                           \
        if (int rc_ = confirm(h)) return rc_;                                                      \
    } while (0)
#define SYNC_S0(h)                                                                                 \
    do {                                                                                           \
        if (int rc_ = sync_s0(h)) return rc_;                                                      \
    } while (0)

static Geo base_geo(const fibhip_ctx *h)
{
    Geo g;
    g.H = h->d.height;
    g.W = h->d.width;
    g.pitch = h->pitch;
    g.Hg = h->d.global_height;
    g.row_off = h->d.row_offset;
    g.r0 = 0;
    g.r1 = h->d.height;
    g.rb0 = g.rb1 = 0;
    g.ty_a = 0;
    g.tiles_x = g.ntiles = 0;
    return g;
}

#define NEED(h)                                                  \
    do {                                                         \
        if (!(h)) return fail(FIBHIP_EINVAL, "null handle");     \
        if ((h)->dead) return fail(FIBHIP_EHIP, "%s", MT_DEAD_MSG);  \
        HIPCHK(hipSetDevice((h)->d.device));                     \
    } while (0)

// launches the tick fibhip_step may have left pending (defined with fibhip_step); every entry point that observes
// or changes the state starts with it
static int flush(fibhip_t h);
static int predict_series(const fibhip_ctx *h, bool *repeat);
static int journal_bound(fibhip_ctx *h);
struct Variant;
static const Variant *mt_variant(const fibhip_ctx *h);
static int mt_launch(fibhip_t h, const Variant *v, int T, bool commit, int *nxt_out, float *snap = nullptr, int snap_var = 0);
#define FLUSH(h)                                                                                   \
    do {                                                                                           \
        if (int rc_ = flush(h)) return rc_;                                                        \
    } while (0)

extern "C" int fibhip_set_phase(fibhip_t h, const float *phi)
{
    NEED(h);
    FLUSH(h);
    CONFIRM(h);
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "set_phase inside an open tick");
    if (!phi) {
        h->has_phase = false;
        return build_plan(h);
    }
    HIPCHK(hipMemcpyAsync(h->phi_dev, phi, h->cells * sizeof(float), hipMemcpyHostToDevice, h->s0));
    const Geo g = base_geo(h);
    hipLaunchKernelGGL(phase_prep_kernel, dim3(1024), dim3(256), 0, h->s0, g, h->phi_dev, h->phase3,
                       h->phase3 + h->cells, h->phase3 + 2 * h->cells, h->phase3 + 3 * h->cells, h->phase3 + 4 * h->cells,
                       h->phase3 + 5 * h->cells);
    HIPCHK(hipGetLastError());
    SYNC_S0(h);                                   // `phi` may be a temporary of the caller
    h->has_phase = true;
    return build_plan(h);
}

extern "C" int fibhip_set_state(fibhip_t h, int var, const float *src)
{
    NEED(h);
    FLUSH(h);
    CONFIRM(h);
    if (!src || var < -1 || var >= h->nvar) return fail(FIBHIP_EINVAL, "set_state: bad var %d", var);
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "set_state inside an open tick");
    const int v0 = var < 0 ? 0 : var, v1 = var < 0 ? h->nvar : var + 1;
    for (int v = v0; v < v1; ++v) {
        const float *s = src + (size_t)(v - v0) * h->cells;
        for (int b = 0; b < 2; ++b)               // both slabs, so either may become current
            HIPCHK(hipMemcpy2DAsync(h->slab[b] + (size_t)v * h->vstride, (size_t)h->pitch * sizeof(float), s,
                                    (size_t)h->d.width * sizeof(float), (size_t)h->d.width * sizeof(float),
                                    (size_t)h->d.height, hipMemcpyHostToDevice, h->s0));
    }
    SYNC_S0(h);
    // The whole slab restarts the exchange cycle: the caller supplied fresh ghost rows of every array.  ONE array
    // does not: mid-cycle the other arrays' outer ghost rows are stale, so the cycle position stays and the rows
    // of `var` that are still live at this position are the ones the caller's copy has to be right in.
    if (var < 0) h->cpos = 0;
    h->agg_dirty = true;
    return 0;
}

// Host memory for fibhip_get_state_direct destinations: page-locked, so the device writes it at PCIe rate.
extern "C" int fibhip_host_alloc(size_t nbytes, void **out)
{
    if (!out || nbytes == 0) return fail(FIBHIP_EINVAL, "host_alloc: bad argument");
    if (hipHostMalloc(out, nbytes, hipHostMallocDefault) != hipSuccess) {
        *out = nullptr;
        return fail(FIBHIP_ENOMEM, "hipHostMalloc of %zu bytes failed", nbytes);
    }
    return 0;
}

extern "C" int fibhip_host_free(void *p)
{
    if (p) HIPCHK(hipHostFree(p));
    return 0;
}

// get_state straight into the caller's buffer, which should come from fibhip_host_alloc: no staging copy (a pageable
// destination works too, at the ~1 GB/s of an unpinned device-to-host copy)
extern "C" int fibhip_get_state_direct(fibhip_t h, int var, float *dst)
{
    NEED(h);
    FLUSH(h);
    if (!dst || var < -1 || var >= h->nvar) return fail(FIBHIP_EINVAL, "get_state_direct: bad var %d", var);
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "get_state inside an open tick");
    // Run-ahead.  A caller that alternates series of n ticks with one read-back — IonicModel.run() with image() every n
    // ticks, fenton.py:184-185 — would leave the device idle for the whole read-back (34 us of a 125 us series at
    // 512x512).  When the lengths of the last series repeat (predict_series), the next n ticks are launched HERE, before the frame is waited
    // for (the launch reads the slab the frame comes from and writes the other one).  fibhip_step then hands those ticks out
    // without launching anything; any other call first makes the state what the caller has been told it is (flush()).
    // The frame itself travels INSIDE that launch when the destination is page-locked memory the device can write
    // (fibhip_host_alloc: what the Python binding hands in): every tile stores its cells of the array straight into it
    // while it starts computing and raises a word in host memory at its first tick boundary; this thread polls those words.
    // No copy engine, no blit kernel (which beside a grid that holds every compute unit would crawl: measured), no gap
    // between two series.  Any other destination: the copy goes first on the same stream and the launch right behind it.
    if (int rc = journal_bound(h)) return rc;
    bool repeats = false;
    int L_next = predict_series(h, &repeats);
    if (h->expect > 0 && h->expect_fresh) {           // the caller has said how many ticks it will ask for next (fibhip_expect)
        L_next = imin(h->expect, h->mt_max);
        repeats = true;
    }
    if (var >= 0 && h->ahead_ok && !h->tracing && (h->series_fresh || (h->expect > 0 && h->expect_fresh)) && repeats && L_next >= 2 &&
        L_next <= h->mt_max && h->pitch == h->d.width && h->tuned) {
        if (const Variant *mv = mt_variant(h)) {
            const int L = L_next;
            void *dev_dst = nullptr;
            const bool in_launch = hipHostGetDevicePointer(&dev_dst, dst, 0) == hipSuccess && dev_dst != nullptr;
            if (!in_launch) (void)hipGetLastError();
            if (in_launch) {
                h->snap_seq++;
                if (int rc = mt_launch(h, mv, L, false, h->spec_nxt, (float *)dev_dst, var)) return rc;
            } else {
                HIPCHK(hipMemcpyAsync(dst, h->slab[h->cur[var]] + (size_t)var * h->vstride, h->cells * sizeof(float), hipMemcpyDeviceToHost, h->s0));
                HIPCHK(hipEventRecord(h->ev_spec, h->s0));
                if (int rc = mt_launch(h, mv, L, false, h->spec_nxt)) return rc;
            }
            h->spec_n = L;
            h->spec_used = 0;
            h->spec_id = h->mt_seq;
            h->series_fresh = false;
            if (!in_launch) {
                HIPCHK(wait_event(h->ev_spec));
                return 0;
            }
            // every tile's word at this read-back's sequence number = every cell of the frame has landed
            const int ntiles = ((h->d.width + mv->TX - 1) / mv->TX) * ((h->d.height + mv->TY - 1) / mv->TY);
            volatile unsigned *fl = h->snap_flags;
            const unsigned want = h->snap_seq;
            const auto t0 = std::chrono::steady_clock::now();
            int next = 0;
            long spins = 0;
            bool delivered = true;
            while (next < ntiles) {
                if (fl[(size_t)next * MT_SNAP_STRIDE] == want) {
                    ++next;
                    continue;
                }
                if ((++spins & 1023) == 0) {
                    // a launch that has ended without raising every word gave up (or was never resident): report it
                    if (hipStreamQuery(h->s0) == hipSuccess && fl[(size_t)next * MT_SNAP_STRIDE] != want) {
                        // the launch gave up (or found the give-up word raised): nothing of it counts, the state it started
                        // from stands (sync_s0 -> recover), and the frame comes the plain way, below
                        h->spec_n = 0;
                        SYNC_S0(h);
                        if (mt_variant(h)) return fail(FIBHIP_EHIP, "the launch that carried the read-back ended without delivering it");
                        delivered = false;
                        break;
                    }
                    (void)hipGetLastError();
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                        h->dead = true;
                        return fail(FIBHIP_EHIP, "%s", MT_DEAD_MSG);
                    }
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            // The frame is the state the launch STARTED from.  If a launch in front of it gave up, that state is void — and the
            // tiles of that launch have said so in the host's memory before this launch's tiles could raise their words (one
            // stream: this launch started after that one had ended).  Then nothing of this launch counts: the state is restored
            // and recomputed (sync_s0 -> recover) and the frame comes the plain way, below.  (Found by the stress run of
            // tests/test_gpu_recovery.py: 7 of 400 random call sequences returned a frame of a void state.)
            if (delivered && __atomic_load_n(h->host_word + MT_GIVEUP_WORD, __ATOMIC_ACQUIRE) != 0u) {
                h->spec_n = 0;
                SYNC_S0(h);
                delivered = false;
            }
            if (delivered) return 0;
        }
    }
    const int v0 = var < 0 ? 0 : var, v1 = var < 0 ? h->nvar : var + 1;
    for (int pass = 0; pass < 2; ++pass) {
        const long long fb0 = h->n_fallbacks;
        for (int v = v0; v < v1; ++v) {
            const float *src = h->slab[h->cur[v]] + (size_t)v * h->vstride;
            float *d = dst + (size_t)(v - v0) * h->cells;
            if (h->pitch == h->d.width)
                HIPCHK(hipMemcpyAsync(d, src, h->cells * sizeof(float), hipMemcpyDeviceToHost, h->s0));
            else
                HIPCHK(hipMemcpy2DAsync(d, (size_t)h->d.width * sizeof(float), src, (size_t)h->pitch * sizeof(float),
                                        (size_t)h->d.width * sizeof(float), (size_t)h->d.height, hipMemcpyDeviceToHost, h->s0));
        }
        SYNC_S0(h);
        if (h->n_fallbacks == fb0) break;           // (else: a launch in front of the copy had given up — the state has been
    }                                               // restored and recomputed meanwhile, and the copy is taken again)
    return 0;
}

extern "C" int fibhip_get_state(fibhip_t h, int var, float *dst)
{
    NEED(h);
    FLUSH(h);
    if (!dst || var < -1 || var >= h->nvar) return fail(FIBHIP_EINVAL, "get_state: bad var %d", var);
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "get_state inside an open tick");
    const int v0 = var < 0 ? 0 : var, v1 = var < 0 ? h->nvar : var + 1;
    // through a pinned staging buffer: a D2H copy into pageable memory runs at ~1 GB/s, pinned at PCIe
    // rate; the reference driver reads the potential back 100 times per simulated second (fenton.py:184-185)
    if (!h->stage) HIPCHK(hipHostMalloc((void **)&h->stage, h->cells * sizeof(float), hipHostMallocDefault));
    for (int v = v0; v < v1; ++v) {
        for (int pass = 0; pass < 2; ++pass) {
            const long long fb0 = h->n_fallbacks;
            HIPCHK(hipMemcpy2DAsync(h->stage, (size_t)h->d.width * sizeof(float),
                                    h->slab[h->cur[v]] + (size_t)v * h->vstride, (size_t)h->pitch * sizeof(float),
                                    (size_t)h->d.width * sizeof(float), (size_t)h->d.height, hipMemcpyDeviceToHost, h->s0));
            SYNC_S0(h);
            if (h->n_fallbacks == fb0) break;       // (a launch in front of the copy had given up: recovered, copy again)
        }
        memcpy(dst + (size_t)(v - v0) * h->cells, h->stage, h->cells * sizeof(float));
    }
    return 0;
}

extern "C" int fibhip_set_consts(fibhip_t h, const float *tbl, int n)
{
    NEED(h);
    FLUSH(h);
    CONFIRM(h);
    if (h->d.model != FIBHIP_BR || !(h->d.flags & FIBHIP_CHEBY))
        return fail(FIBHIP_EINVAL, "set_consts: only the Beeler-Reuter Chebyshev path takes a table");
    if (!tbl || n != 12 * 9) return fail(FIBHIP_EINVAL, "set_consts: expected 108 coefficients, got %d", n);
    memcpy(h->kb.cheb, tbl, sizeof h->kb.cheb);
    h->has_consts = true;
    return 0;
}

// ------------------------------------------------------------------------------------------
// stepping
// ------------------------------------------------------------------------------------------
// Buffer rule for one launch: the potential always ping-pongs (its neighbours are read by other
// workgroups).  When the launch fuses K > 1 sub-steps every variable ping-pongs, because the halo
// cells of a tile are owned (and rewritten) by a neighbouring tile.  With K == 1 the pointwise
// variables are read and written by the same thread only, so they are updated in place — which is
// also what lets Courtemanche's fast tick assign 4 of its 21 arrays and leave the rest untouched.
// which variables the tick op of this handle assigns (M::mask(mode))
static unsigned tick_mask(const fibhip_ctx *h)
{
    if (h->mod) return h->mod->masks[h->mode];
    switch (h->d.model) {
    case FIBHIP_FENTON4V: return Fenton::mask(h->mode);
    case FIBHIP_BR: return BeelerReuter::mask(h->mode);
    case FIBHIP_COURT: return h->use_agg ? CourtAgg::mask(h->mode) : Courtemanche::mask(h->mode);
    case FIBHIP_COURT_US: return CourtemancheUS::mask(h->mode);
#ifdef FIB_CUSTOM_MODEL_INC
    case FIBHIP_CUSTOM: return Custom::mask(h->mode);
#endif
    default: return ~0u;
    }
}

// the aggregate arrays follow the state arrays in the pointer table of the CourtAgg kernels (read and written in place)
static void agg_ptrs(const fibhip_ctx *h, LaunchCtx &c)
{
#if !defined(FIB_CUSTOM_ONLY) && !defined(FIB_ONLY_BR)
    if (!h->use_agg) return;
    for (int a = 0; a < CourtAgg::NAGG; ++a)
        c.in[Courtemanche::NVAR + a] = c.out[Courtemanche::NVAR + a] = h->agg + (size_t)a * h->agg_stride;
#endif
}

static void fill_ptrs(fibhip_ctx *h, LaunchCtx &c, int K, const int *cur, int *nxt)
{
    const unsigned wmask = tick_mask(h);
    for (int v = 0; v < h->nvar; ++v) {
        // a variable the op never assigns is read-only for the whole launch: it stays where it is (Courtemanche's
        // fast tick: 17 of 21 arrays)
        const bool flip = (v == 0) || (K > 1 && ((wmask >> v) & 1u));
        nxt[v] = flip ? (cur[v] ^ 1) : cur[v];
        c.in[v] = h->slab[cur[v]] + (size_t)v * h->vstride;
        c.out[v] = h->slab[nxt[v]] + (size_t)v * h->vstride;
    }
    agg_ptrs(h, c);
    c.ph.dpy = h->phase3;
    c.ph.dpx = h->phase3 + h->cells;
    c.ph.q4 = h->phase3 + 2 * h->cells;
    c.ph.r4 = h->phase3 + 3 * h->cells;
    c.ph.pyr = h->phase3 + 4 * h->cells;
    c.ph.pxr = h->phase3 + 5 * h->cells;
    c.ph.phi = h->phi_dev;
    c.consts = consts_of(h);
}

// ---- timeline (fibhip_trace_begin / _end) ------------------------------------------------------------------------------
static int trace_open(fibhip_ctx *h, hipStream_t st, const char *family, int K, int TX, int TY, int NT, int ticks)
{
    if (!h->tracing) return 0;
    fibhip_ctx::TraceRec r;
    r.e0 = r.e1 = nullptr;
    HIPCHK(hipEventCreate(&r.e0));
    HIPCHK(hipEventCreate(&r.e1));
    r.K = K; r.TX = TX; r.TY = TY; r.R = NT < 0 ? (NT < -32 ? -NT - 32 : -NT) : 0; r.ticks = ticks;
    if (TX > 0 && NT < 0)
        snprintf(r.name, sizeof r.name, "%s<K=%d, tile %dx%d, %d rows per wave%s>", family, K, TX, TY, r.R,
                 ticks > 1 ? ", several ticks" : "");
    else if (TX > 0)
        snprintf(r.name, sizeof r.name, "%s<K=%d, tile %dx%d, %d threads>", family, K, TX, TY, NT);
    else
        snprintf(r.name, sizeof r.name, "%s", family);
    HIPCHK(hipEventRecord(r.e0, st));
    h->trace.push_back(r);
    return 0;
}
static int trace_close(fibhip_ctx *h, hipStream_t st)
{
    if (!h->tracing || h->trace.empty()) return 0;
    HIPCHK(hipEventRecord(h->trace.back().e1, st));
    return 0;
}
static const char *family_of(const Variant *v)
{
    if (!v) return "kernel";
    if (v->kern) return v->kind == MK_POINTWISE ? "pointwise_kernel (generated)" : (v->kind == MK_STRIP ? "strip_kernel (generated)" : "tick_kernel (generated)");
    return v->NT < -32 ? "rows_kernel" : (v->NT < 0 ? "strip_kernel" : "tick_kernel");
}

// one launch over the rows [r0, r1) and, optionally, a second band [rb0, rb1)
static int launch_range(fibhip_ctx *h, hipStream_t st, const PlanItem &it, LaunchCtx &c, int r0, int r1, int rb0 = 0,
                        int rb1 = 0)
{
    if (r1 <= r0 && rb1 <= rb0) return 0;
    c.g = base_geo(h);
    c.g.r0 = r0;
    c.g.r1 = r1 > r0 ? r1 : r0;
    c.g.rb0 = rb0;
    c.g.rb1 = rb1 > rb0 ? rb1 : rb0;
    if (it.v && it.v->kern) {                                     // a kernel of a run-time module
        c.kern = it.v->kern;
        c.kind = it.v->kind;
        c.K = it.v->K; c.TX = it.v->TX; c.TY = it.v->TY; c.NT = it.v->NT;
        c.nvar = h->nvar;
        c.consts_bytes = h->mod ? h->mod->consts_bytes : 0;
    }
    if (int rc = trace_open(h, st, family_of(it.v), it.K, it.TX, it.TY, it.v ? it.v->NT : 0, it.K > h->spt ? it.K / h->spt : 1)) return rc;
    HIPCHK(it.fn(st, c));
    if (int rc = trace_close(h, st)) return rc;
    h->launches++;
    return 0;
}


// rows launch `l` of the plan has to produce: the owned rows grown by the sub-steps still to come
// (those rows are the halo of the later launches of the same tick), clipped to the slab
// Communication-avoiding ghost zone: with ghost = cycle * steps_per_tick rows the neighbours' rows are
// exchanged only every `cycle` ticks; tick j of a cycle also advances the (cycle-1-j) * spt ghost rows
// next to the owned block, which are the halo of the ticks still to come.
// (a launch that fuses `span` ticks leaves the rows the ticks AFTER it still need)
static inline int ext_rows(const fibhip_ctx *h) { return (h->cycle - h->cpos - h->span) * h->spt; }
static inline bool ends_cycle(const fibhip_ctx *h)
{
    return (h->d.ghost_top || h->d.ghost_bottom) && h->cpos + h->span == h->cycle;
}
// On the tick that ends a cycle the strips the neighbours wait for can be launched first (main stream) and
// the rest of the block on a second stream, so that the messages overlap the interior.  A fused launch is
// latency-bound (~20 us however few tiles it has; measured: the split costs a 512-row block 44 us per tick
// instead of 22), so it only pays when the interior is several rounds of CUs long.
static inline bool split_tick(const fibhip_ctx *h, const PlanItem &it)
{
    if (!ends_cycle(h)) return false;
    if (const char *e = getenv("FIBHIP_SPLIT")) return atoi(e) != 0;
    const int hw = imax(h->d.ghost_top, h->d.ghost_bottom);
    const int edge = ((hw + it.TY - 1) / it.TY) * it.TY;
    const long interior_rows = (long)(h->own1 - h->own0) - ((h->d.ghost_top ? edge : 0) + (h->d.ghost_bottom ? edge : 0));
    const long tiles = ((h->d.width + it.TX - 1) / it.TX) * ((interior_rows + it.TY - 1) / it.TY);
    return interior_rows > 0 && tiles >= 4 * 256;
}

static void rows_of_launch(const fibhip_ctx *h, size_t l, int &r0, int &r1)
{
    int rem = ext_rows(h);
    for (size_t m = l + 1; m < h->plan.size(); ++m) rem += h->plan[m].K;
    r0 = imax(0, h->own0 - (h->d.ghost_top ? rem : 0));
    r1 = imin(h->d.height, h->own1 + (h->d.ghost_bottom ? rem : 0));
}

static int check_ready(fibhip_ctx *h)
{
    if (!h->has_consts) return fail(FIBHIP_EINVAL, "Chebyshev table not set (fibhip_set_consts)");
    return 0;
}

// ---- several ticks per launch (strip_mt_kernel) -----------------------------------------------------------------
// The tile program of a tick loops over T ticks inside one launch and re-reads only the rim of its compute box from its
// eight neighbours between two ticks (kernels.hpp, MtArgs).  That needs every tile resident at the same time: the plan
// must be ONE strip launch per tick whose tiles number at most the device's compute units — and no second such launch
// of this process on the device at the same time (two half-resident grids would wait for each other until both
// give up), which g_mt below guarantees.
// How many ticks will the caller's next series (the ticks between two observations of the state) have?  From the lengths of
// its last series: the same again if the last two were equal; if the lengths repeat with a period of 2, 3 or 4 — run() with an
// image() every 10 ticks inside benchmark regions of 20 ticks that start 6 ticks before a read-back: 6, 10, 4, 6, 10, 4, ... —
// the one that followed the last series' twin a period ago; else the last length (one sample).  `*repeat`: the prediction rests
// on a repetition, not on one sample.  Wrong predictions cost little: too long, the launch is stopped at the tick the caller
// reached (flush()); too short, the remaining ticks are launched the ordinary way.
static int predict_series(const fibhip_ctx *h, bool *repeat)
{
    const int n = h->nhist;
    if (repeat) *repeat = false;
    if (n == 0) return 0;
    const int *e = h->hist + n;                         // e[-1] = the last series
    if (n >= 2 && e[-1] == e[-2]) {
        if (repeat) *repeat = true;
        return e[-1];
    }
    for (int p = 2; p <= 4; ++p)
        if (n >= p + 1 && e[-1] == e[-1 - p]) {         // (ONE match is enough: a wrong guess is stopped or topped up)
            if (repeat) *repeat = true;
            return e[-p];
        }
    return e[-1];
}

static bool mt_eligible(const fibhip_ctx *h, const Variant *v)
{
    if (h->mt_max <= 1 || !v || !v->fn_mt || v->K != h->spt || h->use_agg) return false;
    const long tiles = (long)((h->d.width + v->TX - 1) / v->TX) * ((h->d.height + v->TY - 1) / v->TY);
    return tiles <= h->ncu && tiles <= MT_MAX_TILES && h->d.device < 16;
}
static const Variant *mt_variant(const fibhip_ctx *h)
{
    if (h->plan.size() != 1 || h->fused_fn || !mt_eligible(h, h->plan[0].v)) return nullptr;
    return h->plan[0].v;
}

static struct {
    std::mutex mu;
    fibhip_ctx *owner[16] = {nullptr};              // per device: the handle whose stream carries the last such launch
} g_mt;

static void mt_forget(fibhip_ctx *h)
{
    std::lock_guard<std::mutex> lock(g_mt.mu);
    for (fibhip_ctx *&o : g_mt.owner)
        if (o == h) o = nullptr;                     // (fibhip_destroy has drained the stream)
}

// one launch advancing T >= 2 ticks from the current slab into the other one; `commit`: the handle's state moves with it
// (autotune times such launches without moving the state)
static int mt_launch(fibhip_t h, const Variant *v, int T, bool commit, int *nxt_out, float *snap, int snap_var)
{
    if (!h->xbuf) {
        if (hipMalloc((void **)&h->xbuf, 2 * (size_t)((h->nvar + 3) / 4 * 4) * h->cells * sizeof(float)) != hipSuccess) {
            h->xbuf = nullptr;
            return fail(FIBHIP_ENOMEM, "hipMalloc of the tick-exchange buffer failed");
        }
        const size_t words = (size_t)MT_MAX_TILES * MT_EPOCH_STRIDE + 3 * MT_EPOCH_STRIDE;
        if (hipMalloc((void **)&h->epochs, words * sizeof(unsigned)) != hipSuccess) {
            h->epochs = nullptr;
            return fail(FIBHIP_ENOMEM, "hipMalloc of the epoch words failed");
        }
        h->epochs_stale = true;
        // page-locked: the tiles' words of a read-back inside a launch, and behind them the host's word (flush())
        if (hipHostMalloc((void **)&h->snap_flags, ((size_t)MT_HOST_WORD_AT + 16) * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
            h->snap_flags = nullptr;
            return fail(FIBHIP_ENOMEM, "hipHostMalloc of the host-side words failed");
        }
        memset(h->snap_flags, 0, ((size_t)MT_HOST_WORD_AT + 16) * sizeof(unsigned));
        HIPCHK(hipHostGetDevicePointer((void **)&h->snap_flags_dev, h->snap_flags, 0));
        h->host_word = h->snap_flags + MT_HOST_WORD_AT;
    }
    if (h->epochs_stale) {                            // first use, or the tiling may have changed: all words equal again
        HIPCHK(hipMemsetAsync(h->epochs, 0, ((size_t)MT_MAX_TILES * MT_EPOCH_STRIDE + 3 * MT_EPOCH_STRIDE) * sizeof(unsigned), h->s0));
        h->epoch_base = 0;
        h->epochs_stale = false;
    }
    LaunchCtx c;
    int nxt[FIB_MAXVAR];
    fill_ptrs(h, c, v->K, h->cur, nxt);
    c.sub0 = 0;
    c.g = base_geo(h);
    c.mt.xb = h->xbuf;
    c.mt.epoch = h->epochs;
    c.mt.err = h->epochs + (size_t)MT_MAX_TILES * MT_EPOCH_STRIDE;
    c.mt.epoch0 = h->epoch_base;
    h->mt_seq = h->mt_seq % h->mt_ids + 1u;           // 1 .. 65535 (FIBHIP_MT_IDS: a smaller cycle, for the tests)
    // (the host's word keeps naming the last launch it was written for: that id is not given out again while it stands there)
    if (h->host_word && (__atomic_load_n(h->host_word, __ATOMIC_RELAXED) >> 16) == h->mt_seq) h->mt_seq = h->mt_seq % h->mt_ids + 1u;
    c.mt.ticks_id = (unsigned)T | (h->mt_seq << 16);
    if (v->kern_mt) {                                 // a kernel of a run-time module (launch_module lays the arguments out)
        c.kern = v->kern_mt;
        c.kind = MK_STRIP_MT;
        c.K = v->K; c.TX = v->TX; c.TY = v->TY; c.NT = v->NT;
        c.nvar = h->nvar;
        c.consts_bytes = h->mod ? h->mod->consts_bytes : 0;
    }
    c.mt.snap = snap;
    c.mt.snap_flag = h->snap_flags_dev;
    c.mt.snap_seq = h->snap_seq;
    c.mt.snap_var = (snap_var & 0xFF) | (int)(h->mt_wait_ms << 8);
    const bool trial = !commit && !nxt_out;           // (autotune: timed and checked on the spot, never part of the state)
    if (!trial) {
        fibhip_ctx::MtRec rec;
        rec.id = h->mt_seq;
        rec.T = T;
        rec.counted = commit;
        memcpy(rec.src, h->cur, sizeof rec.src);
        h->journal.push_back(rec);
        if (h->fake_giveup_at > 0 && ++h->fake_seen == h->fake_giveup_at) {
            // test switch: this launch finds the give-up word raised in its name — what its tiles would have written had one
            // of them waited out its bound — and leaves at its first boundary, like every launch behind it
            unsigned *w = (unsigned *)(h->probe_host + 12);
            *w = h->mt_seq;
            HIPCHK(hipMemcpyAsync(c.mt.err, w, sizeof(unsigned), hipMemcpyHostToDevice, h->s0));
            __atomic_store_n(h->host_word + MT_GIVEUP_WORD, h->mt_seq, __ATOMIC_RELEASE);      // (what that tile would also have written)
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_mt.mu);
        fibhip_ctx *&owner = g_mt.owner[h->d.device];
        if (owner && owner != h) {                    // behind the other handle's launches, never beside them
            HIPCHK(hipEventRecord(owner->ev_main, owner->s0));
            HIPCHK(hipStreamWaitEvent(h->s0, owner->ev_main, 0));
        }
        if (int rc = trace_open(h, h->s0, "strip_mt_kernel", v->K, v->TX, v->TY, v->NT, T)) return rc;
        HIPCHK(v->fn_mt(h->s0, c));
        if (int rc = trace_close(h, h->s0)) return rc;
        owner = h;
    }
    h->launches++;
    if (commit) {
        h->n_mt_launches++;
        h->n_mt_ticks += T;
        h->n_ticks += T;
    }
    h->mt_inflight = true;
    h->epoch_base += (unsigned)(T - 1);               // every tile raised its word once per tick boundary
    if (commit) memcpy(h->cur, nxt, sizeof nxt);
    if (nxt_out) memcpy(nxt_out, nxt, sizeof nxt);
    return 0;
}

// A caller that never synchronises must not grow the journal without bound: every 256 multi-tick launches the stream is
// drained once (20 us in 100 ms of work) and the launches so far are confirmed — or the first that gave up is found.
static int journal_bound(fibhip_ctx *h)
{
    if (h->journal.size() < 256 || h->spec_n > 0) return 0;
    return sync_s0(h);
}

static int tick_mt(fibhip_t h, const Variant *v, int T)
{
    if (T <= 1) {
        CONFIRM(h);
        return tick_now(h);
    }
    if (h->phase_of_tick != 0) return fail(FIBHIP_EINVAL, "step: previous tick not committed");
    if (int rc = check_ready(h)) return rc;
    if (int rc = journal_bound(h)) return rc;
    if (h->mt_max <= 1) {                             // (a launch among those just confirmed had given up: one launch per tick now)
        for (int t = 0; t < T; ++t)
            if (int rc = tick_now(h)) return rc;
        return 0;
    }
    return mt_launch(h, v, T, true, nullptr);
}

// Plan selection by measurement (Fenton 4v and Beeler-Reuter).  The K-fused kernels exist in a family of tile shapes
// (Fenton: K = 10 or 5, tile heights 21..56; Beeler-Reuter: K = 5, heights 21..40, or one sub-step per launch): which
// one is fastest depends on how the grid's tiles land on the 256 CUs — a launch costs about the
// same whether a CU gets one tile or none, and nearly twice as much with two — so fixed size thresholds leave cliffs
// (576^2: 26.7 us per tick with the 512^2 choice, 18.9 with a taller tile).  The first tick of a handle therefore
// times every candidate ONCE on the handle's own geometry (its real launch: same buffers, same rows; a candidate
// writes what the real launch overwrites) and keeps the fastest.  All candidates are bit-identical in their results
// (tests/test_gpu_parity.py::test_fenton_fusion_depths_bit_identical), so the choice changes speed only — ranks of
// a sharded grid may choose differently.  FIBHIP_AUTOTUNE=0, FIBHIP_VARIANT or FIBHIP_K switch it off.
static int run_pointwise_mode(fibhip_t h, launch_fn fn, const Variant *mv = nullptr, int row0 = -1, int row1 = -1);

// Courtemanche on aggregates: recompute them if the state was written from outside since they were formed
static int refresh_agg(fibhip_t h)
{
#if !defined(FIB_CUSTOM_ONLY) && !defined(FIB_ONLY_BR)
    if (h->use_agg && h->agg_dirty) {
        if (int rc = run_pointwise_mode(h, launch_pointwise<CourtAgg, Fast, CourtAgg::MODE_AGG>, nullptr)) return rc;
        h->agg_dirty = false;
    }
    if (h->use_agg && h->agg_ghost_dirty) {               // the rows a neighbour's message has just replaced
        const launch_fn fn = launch_pointwise<CourtAgg, Fast, CourtAgg::MODE_AGG>;
        if (h->d.ghost_top)
            if (int rc = run_pointwise_mode(h, fn, nullptr, 0, h->d.ghost_top)) return rc;
        if (h->d.ghost_bottom)
            if (int rc = run_pointwise_mode(h, fn, nullptr, h->d.height - h->d.ghost_bottom, h->d.height)) return rc;
        h->agg_ghost_dirty = false;
    }
#endif
    return 0;
}

// Courtemanche on aggregates: which tile shape for the launches of two and of three ticks, on this very geometry
static int autotune_multi(fibhip_ctx *h)
{
#if !defined(FIB_CUSTOM_ONLY) && !defined(FIB_ONLY_BR)
    if (const char *e = getenv("FIBHIP_AUTOTUNE"))
        if (atoi(e) == 0) return 0;
    if (int rc = refresh_agg(h)) return rc;                       // time the kernels on real values
    const int phase = h->has_phase ? 1 : 0;
    const long launches0 = h->launches;
    for (int T = 2; T <= h->multi_max; ++T) {
        if (getenv(T == 2 ? "FIBHIP_COURT_MULTI2" : "FIBHIP_COURT_MULTI3")) continue;
        std::vector<PlanItem> cand;
        for (int i = 0; i < g_nvariants; ++i) {
            const Variant &v = g_variants[i];
            if (v.model != VM_COURT_AGG || v.mode != CourtAgg::MODE_FAST || v.fast != 1 || v.phase != phase || v.K != T) continue;
            cand.push_back({v.K, v.fn, v.TY, v.TX, &v});
        }
        std::vector<float> best_of(cand.size(), 1e30f);
        for (int round = 0; round < 4; ++round)                   // rounds over all candidates: see autotune()
            for (size_t t = 0; t < cand.size(); ++t) {
                HIPCHK(hipEventRecord(h->ev_t0, h->s0));
                LaunchCtx c;
                int nxt[FIB_MAXVAR];
                fill_ptrs(h, c, T, h->cur, nxt);                  // current slab -> other slab: the state stays put
                c.sub0 = 0;
                if (int rc = launch_range(h, h->s0, cand[t], c, 0, h->d.height)) return rc;
                HIPCHK(hipEventRecord(h->ev_t1, h->s0));
                HIPCHK(wait_event(h->ev_t1));
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
                if (round > 0 && ms < best_of[t]) best_of[t] = ms;
            }
        float best_ms = 1e30f;
        for (size_t t = 0; t < cand.size(); ++t)
            if (best_of[t] < best_ms) {
                best_ms = best_of[t];
                h->plan_multi[T].assign(1, cand[t]);
            }
        if (getenv("FIBHIP_PRINT_PLAN") && !h->plan_multi[T].empty())
            fprintf(stderr, "fibhip: %dx%d Courtemanche on aggregates: %d ticks per launch in tiles of %dx%d (%.2f us when chosen)\n",
                    h->d.height, h->d.width, T, h->plan_multi[T][0].TX, h->plan_multi[T][0].TY, best_ms * 1e3f);
    }
    h->launches = launches0;
#endif
    return 0;
}

static int autotune(fibhip_ctx *h)
{
    h->tuned = true;
    if (h->use_agg && h->multi_max > 1) return autotune_multi(h);
    if ((h->d.model != FIBHIP_FENTON4V && h->d.model != FIBHIP_BR && h->d.model != FIBHIP_CUSTOM) ||
        (h->d.flags & FIBHIP_ZEROPAD) || h->spt < 2)
        return 0;
    if (getenv("FIBHIP_VARIANT") || getenv("FIBHIP_K")) return 0;
    if (const char *e = getenv("FIBHIP_AUTOTUNE"))
        if (atoi(e) == 0) return 0;
    const int maxghost = (h->d.ghost_top > 0 || h->d.ghost_bottom > 0)
                             ? imin(h->d.ghost_top > 0 ? h->d.ghost_top : 1 << 30, h->d.ghost_bottom > 0 ? h->d.ghost_bottom : 1 << 30)
                             : 1 << 30;
    const int fast = (h->d.flags & FIBHIP_FAST) ? 1 : 0, phase = h->has_phase ? 1 : 0;
    const std::vector<PlanItem> heuristic = h->plan;
    const long launches0 = h->launches;
    // (a traced model on a run-time module brings its own, short, table: the shapes its generated header asked for)
    const Variant *tab = h->mod ? h->mod->variants.data() : g_variants;
    const int ntab = h->mod ? (int)h->mod->variants.size() : g_nvariants;
    std::vector<std::vector<PlanItem>> trials;
    trials.push_back(heuristic);                                  // the rule-based plan is a candidate like any other
    for (int i = 0; i < ntab; ++i) {
        const Variant &v = tab[i];
        if (v.kind == MK_POINTWISE) continue;
        if (v.model != h->d.model || v.mode != h->mode || v.fast != fast || v.phase != phase) continue;
        // strip kernels of every fusion depth, and the one-sub-step-per-launch tiles
        const bool strip = v.NT < 0 && v.NT > -32 && v.K >= 2, single = v.NT > 0 && v.K == 1;
        if (!(strip || single) || h->spt % v.K != 0 || v.K > maxghost) continue;
        if (!heuristic.empty() && heuristic[0].fn == v.fn) continue;
        std::vector<PlanItem> trial;
        for (int n = 0; n < h->spt / v.K; ++n) trial.push_back({v.K, v.fn, v.TY, v.TX, &v});
        trials.push_back(trial);
    }
    // One tick of a candidate, back to back between one pair of events (the gaps between its launches are part of its
    // cost).  The candidates are timed in ROUNDS — every candidate once per round, the first round a warm-up (code
    // objects, caches), the minimum over the other rounds kept: the clocks of a GPU that has just been idle rise for
    // many milliseconds, and timing the candidates one after the other would favour whichever come last.
    std::vector<float> best_of(trials.size(), 1e30f);
    std::vector<bool> failed(trials.size(), false);
    for (int round = 0; round < 4; ++round) {
        for (size_t t = 0; t < trials.size(); ++t) {
            if (failed[t] || trials[t].empty()) continue;
            const std::vector<PlanItem> &trial = trials[t];
            h->plan = trial;
            // a shape that will run several ticks per launch is timed as that: AT_MT_TICKS ticks in one launch, per tick
            const bool as_mt = trial.size() == 1 && mt_eligible(h, trial[0].v);
            HIPCHK(hipEventRecord(h->ev_t0, h->s0));
            int sub = 0;
            if (as_mt) {
                if (mt_launch(h, trial[0].v, AT_MT_TICKS, false, nullptr)) failed[t] = true;
            } else
            for (size_t l = 0; l < trial.size(); ++l) {           // every launch with the rows edges_impl gives it
                LaunchCtx c;
                int nxt[FIB_MAXVAR];
                fill_ptrs(h, c, trial[l].K, h->cur, nxt);         // ALWAYS current slab -> other slab: the state stays put
                for (int v = 0; v < h->nvar; ++v)                 // (a K = 1 launch would update the pointwise arrays in place)
                    c.out[v] = h->slab[h->cur[v] ^ 1] + (size_t)v * h->vstride;
                c.sub0 = sub;
                int r0, r1;
                rows_of_launch(h, l, r0, r1);
                if (launch_range(h, h->s0, trial[l], c, r0, r1)) { failed[t] = true; break; }
                sub += trial[l].K;
            }
            if (failed[t]) {
                // a shape that cannot be launched here is dropped — audibly, and a failure of the rule-based plan itself is
                // the caller's error to see
                (void)hipGetLastError();
                if (getenv("FIBHIP_PRINT_PLAN"))
                    fprintf(stderr, "fibhip: %dx%d model %d: candidate K=%d tile %dx%d could not be launched: %s\n", h->d.height,
                            h->d.width, h->d.model, trial[0].K, trial[0].TX, trial[0].TY, g_err);
                if (t == 0) {
                    h->plan = heuristic;
                    return FIBHIP_EHIP;
                }
                HIPCHK(wait_stream(h->s0));
                continue;
            }
            HIPCHK(hipEventRecord(h->ev_t1, h->s0));
            HIPCHK(wait_event(h->ev_t1));
            if (as_mt) {
                // a candidate whose tiles could not all become resident (a CU mask, another process on the device) gave up
                // waiting: it is dropped like one that could not be launched — the state is untouched, a trial writes the
                // other slab only — and the words are cleared for the next candidate
                const unsigned gave_up = __atomic_load_n(h->host_word + MT_GIVEUP_WORD, __ATOMIC_ACQUIRE);
                if (gave_up) {
                    failed[t] = true;
                    h->epochs_stale = true;
                    __atomic_store_n(h->host_word + MT_GIVEUP_WORD, 0u, __ATOMIC_RELEASE);
                    if (getenv("FIBHIP_PRINT_PLAN"))
                        fprintf(stderr, "fibhip: %dx%d model %d: candidate K=%d tile %dx%d gave up waiting as a multi-tick launch: dropped\n",
                                h->d.height, h->d.width, h->d.model, trial[0].K, trial[0].TX, trial[0].TY);
                    continue;
                }
            }
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
            if (as_mt) ms /= (float)AT_MT_TICKS;
            if (round > 0 && ms < best_of[t]) best_of[t] = ms;
        }
    }
    std::vector<PlanItem> best_plan = heuristic;
    float best_ms = 1e30f;
    for (size_t t = 0; t < trials.size(); ++t)
        if (!failed[t] && !trials[t].empty() && best_of[t] < best_ms) {
            best_ms = best_of[t];
            best_plan = trials[t];
        }
    h->plan = best_plan;
    h->epochs_stale = true;
    h->launches = launches0;
    if (getenv("FIBHIP_PRINT_PLAN") && !best_plan.empty())
        fprintf(stderr, "fibhip: %dx%d model %d: %zu launch(es) per tick of K=%d, tile %dx%d, %s (%.2f us per tick when chosen)\n",
                h->d.height, h->d.width, h->d.model, best_plan.size(), best_plan[0].K, best_plan[0].TX, best_plan[0].TY,
                best_plan[0].v ? (best_plan[0].v->NT < 0 ? (best_plan.size() == 1 && mt_eligible(h, best_plan[0].v)
                                                                ? "strips, several ticks per launch" : "strips") : "flat tiles") : "rule-based",
                best_ms * 1e3f);
    return 0;
}

static int edges_impl(fibhip_t h)
{
    if (h->phase_of_tick != 0) return fail(FIBHIP_EINVAL, "step_edges: previous tick not committed");
    if (int rc = check_ready(h)) return rc;
    if (!h->tuned)
        if (int rc = autotune(h)) return rc;
    if (int rc = refresh_agg(h)) return rc;
    int cur[FIB_MAXVAR];
    memcpy(cur, h->cur, sizeof cur);
    int sub = 0;
    for (size_t l = 0; l < h->plan.size(); ++l) {
        const PlanItem &it = h->plan[l];
        LaunchCtx c;
        int nxt[FIB_MAXVAR];
        fill_ptrs(h, c, it.K, cur, nxt);
        c.sub0 = sub;
        int r0, r1;
        rows_of_launch(h, l, r0, r1);
        if (l + 1 < h->plan.size()) {
            if (int rc = launch_range(h, h->s0, it, c, r0, r1)) return rc;
            memcpy(cur, nxt, sizeof cur);
            sub += it.K;
            continue;
        }
        // last launch: only the strips a neighbour is waiting for — and only on the tick that ends a cycle.
        // The interior part (step_interior, second stream) depends on everything enqueued on s0 up to HERE
        // — the earlier launches of this tick and the previous tick's halo refresh — but not on the strips.
        const bool split = split_tick(h, it);
        h->whole_in_edges = ends_cycle(h) && !split;      // the caller exchanges right after step_edges:
        if (h->whole_in_edges) {                          // everything it sends must be computed by then
            if (int rc = launch_range(h, h->s0, it, c, r0, r1)) return rc;
        } else {
            if (split) HIPCHK(hipEventRecord(h->ev_main, h->s0));
            const int hw = split ? imax(h->d.ghost_top, h->d.ghost_bottom) : 0;
            const int e = hw > 0 ? ((hw + it.TY - 1) / it.TY) * it.TY : 0;
            int t1 = (hw && h->d.ghost_top) ? imin(r0 + e, r1) : r0;          // top strip [r0, t1)
            int b0 = (hw && h->d.ghost_bottom) ? imax(r1 - e, t1) : r1;        // bottom strip [b0, r1)
            if (int rc = launch_range(h, h->s0, it, c, r0, t1, b0, r1)) return rc;   // both strips, one launch
        }
        memcpy(h->nxt, nxt, sizeof nxt);
    }
    h->phase_of_tick = 1;
    return 0;
}

static int interior_impl(fibhip_t h)
{
    if (h->phase_of_tick != 1) return fail(FIBHIP_EINVAL, "step_interior: call step_edges first");
    if (h->whole_in_edges) {                              // step_edges already launched the whole block
        h->phase_of_tick = 2;
        return 0;
    }
    // recompute the last launch's geometry (same arithmetic as step_edges)
    int cur[FIB_MAXVAR];
    memcpy(cur, h->cur, sizeof cur);
    int sub = 0;
    for (size_t l = 0; l + 1 < h->plan.size(); ++l) {
        LaunchCtx tmp;
        int nxt[FIB_MAXVAR];
        fill_ptrs(h, tmp, h->plan[l].K, cur, nxt);
        memcpy(cur, nxt, sizeof cur);
        sub += h->plan[l].K;
    }
    const PlanItem &it = h->plan.back();
    LaunchCtx c;
    int nxt[FIB_MAXVAR];
    fill_ptrs(h, c, it.K, cur, nxt);
    c.sub0 = sub;
    int r0, r1;
    rows_of_launch(h, h->plan.size() - 1, r0, r1);
    const bool split = split_tick(h, it);
    const int hw = split ? imax(h->d.ghost_top, h->d.ghost_bottom) : 0;
    const int e = hw > 0 ? ((hw + it.TY - 1) / it.TY) * it.TY : 0;
    const int t1 = (hw && h->d.ghost_top) ? imin(r0 + e, r1) : r0;
    const int b0 = (hw && h->d.ghost_bottom) ? imax(r1 - e, t1) : r1;
    hipStream_t st = split ? h->s1 : h->s0;
    if (split) HIPCHK(hipStreamWaitEvent(h->s1, h->ev_main, 0));   // recorded in step_edges, before the strips
    if (int rc = launch_range(h, st, it, c, t1, b0)) return rc;
    if (split) HIPCHK(hipEventRecord(h->ev_int, h->s1));
    h->phase_of_tick = 2;
    return 0;
}

static int commit_impl(fibhip_t h)
{
    if (h->phase_of_tick != 2) return fail(FIBHIP_EINVAL, "step_commit: call step_interior first");
    if (split_tick(h, h->plan.back())) HIPCHK(hipStreamWaitEvent(h->s0, h->ev_int, 0));
    memcpy(h->cur, h->nxt, sizeof h->cur);
    h->n_ticks += h->plan.empty() ? 1 : (h->plan[0].K > h->spt ? h->plan[0].K / h->spt : 1);
    if (h->use_agg && ends_cycle(h)) h->agg_ghost_dirty = true;   // the exchange of this tick replaced the ghost rows
    h->cpos = (h->cpos + h->span) % h->cycle;
    h->phase_of_tick = 0;
    return 0;
}

static int tick_now(fibhip_t h)
{
    if (int rc = edges_impl(h)) return rc;
    if (int rc = interior_impl(h)) return rc;
    return commit_impl(h);
}

// T consecutive ticks as one launch (T <= multi_max)
static int tick_multi(fibhip_t h, int T)
{
    // on a row block a fused launch stays inside the exchange cycle: the tick that ends it is the caller's
    // step_edges / exchange / step_interior / step_commit
    if (h->d.ghost_top || h->d.ghost_bottom) T = imax(1, imin(T, h->cycle - 1 - h->cpos));
    if (T <= 1) return tick_now(h);
    h->plan.swap(h->plan_multi[T]);
    h->span = T;
    const int rc = tick_now(h);
    h->span = 1;
    h->plan.swap(h->plan_multi[T]);
    return rc;
}

// launch `n` of the ticks fibhip_step has deferred, the fewest launches first
static int launch_pending(fibhip_t h, int n)
{
    if (n > 0 && !h->tracing)
        if (const Variant *v = mt_variant(h)) {
            while (n > 0) {
                const int T = imin(h->mt_max, n);
                h->pending -= T;
                n -= T;
                if (int rc = tick_mt(h, v, T)) return rc;
                h->mt_run += T;
            }
            return 0;
        }
    while (n > 0) {
        int T = imin(h->multi_max, n);
        if ((h->d.ghost_top || h->d.ghost_bottom) && T > 1) T = imax(1, imin(T, h->cycle - 1 - h->cpos));
        h->pending -= T;
        n -= T;
        if (int rc = tick_multi(h, T)) return rc;
    }
    return 0;
}

// launch the ticks fibhip_step left pending; every entry point that observes or changes the state calls this first
static int flush(fibhip_t h)
{
    if (h->spec_n > 0) {
        // The caller did not go on as predicted.  The launch that ran ahead is told so through the host's word (page-locked
        // host memory; ONE thread of the grid reads it at the start of every tick and passes it on at the tick's end):
        //  * some of its ticks have been handed out: "stop after spec_used ticks" — a tile leaves through its write-back at
        //    that boundary, and counts itself.  The interpreter hands ticks out faster than the device computes them, so the
        //    boundary is normally still ahead of every tile and nothing is computed twice; if a tile was past it already (it
        //    then leaves without writing) the count falls short and the ticks are recomputed from the state the launch
        //    started from — still intact: the launch writes the other slab only;
        //  * none has: the launch is simply cancelled.
        const int redo = h->spec_used;
        bool kept = false;
        if (h->epochs) {
            // (the word names the launch: earlier launches of this handle may still be queued or running)
            unsigned word = (h->spec_id << 16) | (redo > 0 ? (unsigned)redo : MT_CANCEL);
            // (a plain store: the tiles read this word over PCIe.  A copy through the second stream does not reach a device
            // whose compute units are all taken before the launch has ended: measured at 512x512, 238-387 us)
            __atomic_store_n(h->host_word, word, __ATOMIC_RELEASE);
            unsigned *base = h->epochs + (size_t)MT_MAX_TILES * MT_EPOCH_STRIDE;
            h->epochs_stale = true;
            if (redo > 0) {
                HIPCHK(hipMemcpyAsync(h->probe_host + 10, base + 2 * MT_EPOCH_STRIDE, sizeof(unsigned), hipMemcpyDeviceToHost, h->s0));
                SYNC_S0(h);
                unsigned stopped;
                memcpy(&stopped, h->probe_host + 10, sizeof stopped);
                const Variant *v = mt_variant(h);
                const long tiles = v ? (long)((h->d.width + v->TX - 1) / v->TX) * ((h->d.height + v->TY - 1) / v->TY) : -1;
                kept = (long)stopped == tiles;
            }
        }
        h->spec_n = h->spec_used = 0;
        for (auto &r : h->journal)                  // (the launch's journal record: what it did, if it did it)
            if (r.id == h->spec_id) {
                r.T = redo;
                r.counted = kept;
            }
        if (kept) {                                 // the state after `redo` ticks is where the launch wrote it
            memcpy(h->cur, h->spec_nxt, sizeof h->cur);
            h->n_mt_launches++;
            h->n_mt_ticks += redo;
            h->n_ticks += redo;
            h->n_spec_kept++;
        } else {
            h->mt_run -= redo;
            h->pending += redo;
            if (redo > 0) h->n_spec_redone++;
        }
        if (!kept && redo > 0) h->spec_trust = false;   // ONE sample is not believed again until two equal series were seen
    }
    const int rc = launch_pending(h, h->pending);
    h->series_fresh = h->mt_run > 0;
    if (h->mt_run > 0) {                            // the caller is about to look: the next tick starts a new series
        h->mt_run_prev2 = h->mt_run_prev;
        h->mt_run_prev = h->mt_run;
        if (h->nhist == 8) {
            memmove(h->hist, h->hist + 1, 7 * sizeof(int));
            h->nhist = 7;
        }
        h->hist[h->nhist++] = h->mt_run;
        h->mt_run = 0;
        if (h->mt_run_prev == h->mt_run_prev2) h->spec_trust = true;
    }
    h->mt_cur = 1;
    if (!h->expect_fresh) h->expect = 0;            // an observation inside a declared series ends the declaration
    return rc;
}

extern "C" int fibhip_step_edges(fibhip_t h)
{
    NEED(h);
    FLUSH(h);
    CONFIRM(h);
    return edges_impl(h);
}

extern "C" int fibhip_step_interior(fibhip_t h)
{
    NEED(h);
    return interior_impl(h);
}

extern "C" int fibhip_step_commit(fibhip_t h)
{
    NEED(h);
    return commit_impl(h);
}

extern "C" int fibhip_step(fibhip_t h, int nticks)
{
    NEED(h);
    if (nticks < 0) return fail(FIBHIP_EINVAL, "negative tick count");
    if (h->phase_of_tick != 0) return fail(FIBHIP_EINVAL, "step inside an open tick");
    // Ticks are accepted here and launched when a launch is full: up to multi_max ticks go into one kernel (CourtAgg),
    // and the last accepted tick is held back when the next call may be a step_slow, which then rides on its launch
    // (fused_fn).  Whatever is held back is launched by the next entry point that observes or changes the state.
    // Fenton / Beeler-Reuter on a grid whose tiles are all resident at once: consecutive ticks become ONE launch whose
    // tiles hand their rims to each other (tick_mt).  A launch goes out as soon as `mt_cur` ticks are waiting and takes
    // every waiting tick, up to mt_max.  mt_cur is 1 after any call that observes the state, so the device starts at once;
    // then the rest of the series if the caller works in series of equal length (run() with an image() every n ticks, a
    // benchmark region: the ticks between the last two observations), else 2, 4, ... mt_max while the caller keeps stepping.
    // A series is launched WHOLE at its first tick when the caller's last series had that length (a benchmark region, run()
    // with a probe or a sync every n ticks): one launch of n ticks instead of the first tick at once + the other n-1 when the
    // last of them has arrived (at 512x512: the device idle while the interpreter makes its 19 other calls, and two launch
    // prologues instead of one — 270 -> 247 us per 20-tick region).  It is the run-ahead of fibhip_get_state_direct started
    // from here: the ticks are handed out below call by call, and a caller that does anything else first gets them recomputed
    // / cancelled by flush() — after which ONE sample is not believed again until two equal series have been seen.
    // A caller that KNOWS its series says so (fibhip_expect: IonicModel.run() does, from its frame period and tick count) and
    // nothing is guessed: the declared ticks are launched at the first of them, mt_max at a time.
    if (int rc = journal_bound(h)) return rc;
    bool repeats = false;
    int L_next = 0;
    if (nticks > 0 && h->spec_n == 0 && h->mt_max > 1) {
        if (h->expect > 0) {
            L_next = imin(h->expect, h->mt_max);
            repeats = true;
        } else if (h->mt_run == 0) {
            L_next = predict_series(h, &repeats);
        }
    }
    if (L_next >= 2 && !h->tracing && h->ahead_ok && (repeats || h->spec_trust) && h->tuned && h->pending == 0 && L_next <= h->mt_max &&
        nticks < L_next && h->pitch == h->d.width && h->phase_of_tick == 0 && h->has_consts) {
        if (const Variant *v = mt_variant(h)) {
            const int L = L_next;
            if (int rc = mt_launch(h, v, L, false, h->spec_nxt)) return rc;
            h->spec_n = L;
            h->spec_used = 0;
            h->spec_id = h->mt_seq;
        }
    }
    if (h->expect > 0 && nticks > 0) {                 // (the declared series has begun / goes on)
        h->expect = imax(0, h->expect - nticks);
        h->expect_fresh = false;
    }
    if (h->spec_n > 0 && nticks > 0) {                 // ticks that have been computed ahead already
        const int take = imin(nticks, h->spec_n - h->spec_used);
        h->spec_used += take;
        h->mt_run += take;
        nticks -= take;
        if (h->spec_used == h->spec_n) {             // all handed out: the state moves to where the launch put it
            for (auto &r : h->journal)
                if (r.id == h->spec_id) r.counted = true;
            memcpy(h->cur, h->spec_nxt, sizeof h->cur);
            h->n_mt_launches++;
            h->n_mt_ticks += h->spec_n;
            h->n_ticks += h->spec_n;
            h->spec_n = h->spec_used = 0;
        }
        if (nticks == 0) return 0;
    }
    if (h->mt_max > 1 && nticks > 0 && !h->tracing) {
        if (int rc = check_ready(h)) return rc;
        if (!h->tuned)
            if (int rc = autotune(h)) return rc;
        if (const Variant *v = mt_variant(h)) {
            h->pending += nticks;
            while (h->pending >= h->mt_cur) {
                const int T = imin(h->pending, h->mt_max);
                h->pending -= T;
                if (int rc = tick_mt(h, v, T)) return rc;
                const bool first = h->mt_run == 0;
                h->mt_run += T;
                const int rest = predict_series(h, nullptr) - h->mt_run;
                h->mt_cur = (first && rest >= 2) ? imin(rest, h->mt_max) : imin(2 * h->mt_cur, h->mt_max);
            }
            return 0;
        }
    }
    const int reserve = (h->fused_fn && !h->tracing) ? 1 : 0;
    const int cap = ((h->multi_max > 1 && !h->tracing) ? h->multi_max - 1 : 0) + reserve;
    if (cap > 0 && nticks > 0) {
        if (int rc = check_ready(h)) return rc;           // a deferred tick must not fail later, in someone else's call
        if (!h->tuned)                                    // (here, not inside a launch: the plans are being chosen)
            if (int rc = autotune(h)) return rc;
    }
    h->pending += nticks;
    while (h->pending > cap) {
        int T = h->tracing ? 1 : imin(h->multi_max, h->pending - reserve);
        if ((h->d.ghost_top || h->d.ghost_bottom) && T > 1) T = imax(1, imin(T, h->cycle - 1 - h->cpos));
        h->pending -= T;
        if (int rc = tick_multi(h, T)) return rc;
    }
    return 0;
}

// re-evaluation of the model on the current state, in place, without the stencil: assigns mask(mode)
static int run_pointwise_mode(fibhip_t h, launch_fn fn, const Variant *mv, int row0, int row1)
{
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "step_mode inside an open tick");
    LaunchCtx c;
    if (mv) {                                             // a pointwise kernel of a run-time module
        c.kern = mv->kern;
        c.kind = MK_POINTWISE;
        c.K = 1; c.TX = c.TY = c.NT = 0;
        c.nvar = h->nvar;
        c.consts_bytes = h->mod->consts_bytes;
    }
    for (int v = 0; v < h->nvar; ++v) {
        c.in[v] = h->slab[h->cur[v]] + (size_t)v * h->vstride;
        c.out[v] = h->slab[h->cur[v]] + (size_t)v * h->vstride;   // in place
    }
    agg_ptrs(h, c);
    c.consts = consts_of(h);
    c.g = base_geo(h);
    // the ghost rows that later ticks of this cycle still advance must get the update too
    const int live = (h->cpos == 0 ? h->cycle : h->cycle - h->cpos) * h->spt;
    c.g.r0 = imax(0, h->own0 - (h->d.ghost_top ? live : 0));
    c.g.r1 = imin(h->d.height, h->own1 + (h->d.ghost_bottom ? live : 0));
    if (row0 >= 0) {                                      // an explicit band of rows instead
        c.g.r0 = row0;
        c.g.r1 = row1;
    }
    c.sub0 = 0;
    if (int rc = trace_open(h, h->s0, mv ? "pointwise_kernel (generated)" : "pointwise_kernel", 1, 0, 0, 0, 1)) return rc;
    HIPCHK(fn(h->s0, c));
    if (int rc = trace_close(h, h->s0)) return rc;
    h->launches++;
    return 0;
}

#ifdef FIB_CUSTOM_MODEL_INC
template <class P, int M_>
static launch_fn custom_mode_fn(int mode)
{
    if constexpr (M_ >= Custom::NMODES) {
        return nullptr;
    } else {
        if (mode == M_) return launch_pointwise<Custom, P, M_>;
        return custom_mode_fn<P, M_ + 1>(mode);
    }
}
#endif

extern "C" int fibhip_step_mode(fibhip_t h, int mode)
{
    NEED(h);
    const bool fast = (h->d.flags & FIBHIP_FAST) != 0;
    (void)fast;
    if (h->mod) {
        FLUSH(h);
        for (const Variant &v : h->mod->variants)
            if (v.kind == MK_POINTWISE && v.mode == mode && v.fast == (fast ? 1 : 0) && mode >= 1)
                return run_pointwise_mode(h, launch_module, &v);
        return fail(FIBHIP_EINVAL, "step_mode: the traced model has no mode %d", mode);
    }
#ifdef FIB_CUSTOM_MODEL_INC
    if (h->d.model == FIBHIP_CUSTOM) {
        FLUSH(h);
        launch_fn fn = mode >= 1 ? (fast ? custom_mode_fn<Fast, 1>(mode) : custom_mode_fn<Exact, 1>(mode)) : nullptr;
        if (!fn) return fail(FIBHIP_EINVAL, "step_mode: the traced model has no mode %d", mode);
        return run_pointwise_mode(h, fn);
    }
#endif
#if !defined(FIB_CUSTOM_ONLY) && !defined(FIB_ONLY_BR)
    if (h->d.model == FIBHIP_COURT && mode == Courtemanche::MODE_SLOW) {
        if (h->d.flags & FIBHIP_ALLVARS) return fail(FIBHIP_EINVAL, "step_slow: handle was created with FIBHIP_ALLVARS");
        if (h->pending && h->fused_fn) {                  // the last deferred tick + slow as one launch
            if (int rc = launch_pending(h, h->pending - 1)) return rc;
            h->pending = 0;
            const launch_fn plain = h->plan[0].fn;
            h->plan[0].fn = h->fused_fn;
            const int rc = tick_now(h);
            h->plan[0].fn = plain;
            return rc;
        }
        FLUSH(h);
        if (h->use_agg) {                                 // 'slow' on the state as it stands; the aggregates follow it
            h->agg_dirty = false;
            return run_pointwise_mode(h, launch_pointwise<CourtAgg, Fast, CourtAgg::MODE_SLOW>);
        }
        return run_pointwise_mode(h, fast ? launch_pointwise<Courtemanche, Fast, Courtemanche::MODE_SLOW>
                                          : launch_pointwise<Courtemanche, Exact, Courtemanche::MODE_SLOW>);
    }
#endif
    return fail(FIBHIP_EINVAL, "step_mode: model %d has no mode %d", h->d.model, mode);
}

extern "C" int fibhip_step_slow(fibhip_t h)
{
    NEED(h);
    if (h->d.model != FIBHIP_COURT) return fail(FIBHIP_EINVAL, "step_slow: Courtemanche only");
    return fibhip_step_mode(h, Courtemanche::MODE_SLOW);
}

extern "C" int fibhip_pace(fibhip_t h, int r0, int r1, int c0, int c1, float v, float min_v)
{
    NEED(h);
    FLUSH(h);
    CONFIRM(h);
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "pace inside an open tick");
    const Geo g = base_geo(h);
    if (int rc = trace_open(h, h->s0, "pace_kernel", 1, 0, 0, 0, 1)) return rc;
    hipLaunchKernelGGL(pace_kernel, dim3(1024), dim3(256), 0, h->s0, g, h->slab[h->cur[0]], r0, r1, c0, c1, v, min_v);   // variable 0 starts at the slab base in both layouts
    HIPCHK(hipGetLastError());
    if (int rc = trace_close(h, h->s0)) return rc;
    h->launches++;
    return 0;
}

extern "C" int fibhip_probe(fibhip_t h, int var, int row, int col, float *out)
{
    NEED(h);
    FLUSH(h);
    if (!out || var < 0 || var >= h->nvar || row < 0 || row >= h->d.height || col < 0 || col >= h->d.width)
        return fail(FIBHIP_EINVAL, "probe: out of range");
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "probe inside an open tick");
    for (int pass = 0; pass < 2; ++pass) {
        const long long fb0 = h->n_fallbacks;
        HIPCHK(hipMemcpyAsync(h->probe_host, h->slab[h->cur[var]] + (size_t)var * h->vstride + (size_t)row * h->pitch + col,
                              sizeof(float), hipMemcpyDeviceToHost, h->s0));
        SYNC_S0(h);
        if (h->n_fallbacks == fb0) break;           // (a launch in front of the copy had given up: recovered, copy again)
    }
    *out = *h->probe_host;
    return 0;
}

extern "C" int fibhip_sync(fibhip_t h)
{
    NEED(h);
    FLUSH(h);
    HIPCHK(wait_stream(h->s1));
    SYNC_S0(h);
    return 0;
}

extern "C" int fibhip_time_steps(fibhip_t h, int nticks, float *elapsed_ms, int *launches)
{
    NEED(h);
    FLUSH(h);
    const long l0 = h->launches;
    HIPCHK(hipEventRecord(h->ev_t0, h->s0));
    if (int rc = fibhip_step(h, nticks)) return rc;
    FLUSH(h);                                             // the timed region ends after the LAST tick's launch
    HIPCHK(hipEventRecord(h->ev_t1, h->s0));
    SYNC_S0(h);
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
    if (elapsed_ms) *elapsed_ms = ms;
    if (launches) *launches = (int)(h->launches - l0);
    return 0;
}

// HIP-event bracket on the handle's stream around whatever the caller enqueues in between (ticks, 'slow' ops,
// pacing): bench.py times the reference driver's real tick mix with it.
extern "C" int fibhip_time_begin(fibhip_t h)
{
    NEED(h);
    FLUSH(h);
    h->t_launches0 = h->launches;
    HIPCHK(hipEventRecord(h->ev_t0, h->s0));
    return 0;
}

extern "C" int fibhip_time_end(fibhip_t h, float *elapsed_ms, int *launches)
{
    NEED(h);
    FLUSH(h);
    HIPCHK(hipEventRecord(h->ev_t1, h->s0));
    SYNC_S0(h);
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
    if (elapsed_ms) *elapsed_ms = ms;
    if (launches) *launches = (int)(h->launches - h->t_launches0);
    return 0;
}

// IonicModel's building blocks as array ops on HOST arrays (copied through the device), for the
// unit-level parity tests: op 0 enforce_boundary(a), 1 laplace(a [, phi]), 2 phase_field(pad(a), phi),
// 3 rush_larsen(a=g, b=g_inf, c=tau, dt).
extern "C" int fibhip_unit_op(int device, int op, int H, int W, const float *a, const float *b, const float *c,
                              const float *phi, double dt, int fast, float *out)
{
    if (!a || !out || H < 3 || W < 3 || op < 0 || op > 3) return fail(FIBHIP_EINVAL, "unit_op: bad argument");
    if (op == OP_RUSH_LARSEN && (!b || !c)) return fail(FIBHIP_EINVAL, "unit_op: rush_larsen needs g_inf and tau");
    if (op == OP_PHASE && !phi) return fail(FIBHIP_EINVAL, "unit_op: phase_field needs phi");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FIBHIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    const size_t n = (size_t)H * W, B = n * sizeof(float);
    float *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, 9 * B));          // a b c phi ph3[4] out
    float *da = d, *db = d + n, *dc = d + 2 * n, *dphi = d + 3 * n, *dph3 = d + 4 * n, *dout = d + 8 * n;
    int rc = 0;
    do {
        if (hipMemcpy(da, a, B, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(FIBHIP_EHIP, "unit_op: H2D failed"); break; }
        if (b && hipMemcpy(db, b, B, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(FIBHIP_EHIP, "unit_op: H2D failed"); break; }
        if (c && hipMemcpy(dc, c, B, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(FIBHIP_EHIP, "unit_op: H2D failed"); break; }
        if (phi) {
            if (hipMemcpy(dphi, phi, B, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(FIBHIP_EHIP, "unit_op: H2D failed"); break; }
            Geo g;
            g.H = g.Hg = H; g.W = W; g.pitch = W; g.row_off = 0; g.r0 = 0; g.r1 = H; g.rb0 = g.rb1 = g.ty_a = 0; g.tiles_x = g.ntiles = 0;
            hipLaunchKernelGGL(phase_prep_kernel, dim3(256), dim3(256), 0, 0, g, dphi, dph3, dph3 + n, dph3 + 2 * n, dph3 + 3 * n,
                               (float *)nullptr, (float *)nullptr);
        }
        const float mdt = (float)(-dt);
        if (fast)
            hipLaunchKernelGGL(unit_op_kernel<Fast>, dim3(256), dim3(256), 0, 0, op, H, W, da, db, dc, phi ? dph3 : nullptr, mdt, dout);
        else
            hipLaunchKernelGGL(unit_op_kernel<Exact>, dim3(256), dim3(256), 0, 0, op, H, W, da, db, dc, phi ? dph3 : nullptr, mdt, dout);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, B, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(FIBHIP_EHIP, "unit_op: kernel or D2H failed");
            break;
        }
    } while (0);
    hipFree(d);
    return rc;
}

extern "C" int fibhip_court_inter(int device, int n, const float *V, int fast, float *out)
{
    if (!V || !out || n <= 0) return fail(FIBHIP_EINVAL, "court_inter: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FIBHIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    float *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, (size_t)(1 + COURT_NINTER) * n * sizeof(float)));
    int rc = 0;
    do {
        if (hipMemcpy(d, V, (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(FIBHIP_EHIP, "court_inter: H2D failed"); break; }
        const int blocks = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
        if (fast)
            hipLaunchKernelGGL(court_inter_kernel<Fast>, dim3(blocks), dim3(256), 0, 0, n, d, d + n);
        else
            hipLaunchKernelGGL(court_inter_kernel<Exact>, dim3(blocks), dim3(256), 0, 0, n, d, d + n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpy(out, d + n, (size_t)COURT_NINTER * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(FIBHIP_EHIP, "court_inter: kernel or D2H failed");
            break;
        }
    } while (0);
    hipFree(d);
    return rc;
}

// ------------------------------------------------------------------------------------------
// Direct halo exchange: ncclSend / ncclRecv issued from here, on the handle's own stream, grouped into one RCCL
// kernel per exchange — no torch enqueue path (55-70 us of host time per exchange, tools/p2p_overhead.py) and no
// hop to a communication stream and back.  RCCL is bound at run time (dlopen) so that the library has no link-time
// dependency on it; the caller passes the path of the librccl its process already uses (torch's).  Opt-in:
// fib_tf_amd/sharded.py uses this path when FIBTF_HALO=direct.
// ------------------------------------------------------------------------------------------
struct FibNcclId {
    char internal[128];                                     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
};
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, FibNcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}  // namespace
static RcclApi g_rccl;
constexpr int FIB_NCCL_FLOAT = 7;                           // ncclFloat32 (rccl.h)

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        const int rc_ = (expr);                                                                    \
        if (rc_ != 0)                                                                              \
            return fail(FIBHIP_EHIP, "RCCL: %s failed: %s", #expr,                                 \
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(rc_) : "?");                 \
    } while (0)

extern "C" int fibhip_comm_open(const char *librccl_path)
{
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);         // the copy this process already runs
    if (!lib && librccl_path) lib = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(FIBHIP_EINVAL, "comm_open: librccl not loaded and not found at %s", librccl_path ? librccl_path : "(null)");
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, FibNcclId, int))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    a.GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
    a.GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
    a.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclSend");
    a.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclRecv");
    a.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv)
        return fail(FIBHIP_EINVAL, "comm_open: librccl lacks the point-to-point API");
    g_rccl = a;
    return 0;
}

extern "C" int fibhip_comm_unique_id(char *out128)
{
    if (!out128) return fail(FIBHIP_EINVAL, "comm_unique_id: null argument");
    if (!g_rccl.lib) return fail(FIBHIP_EINVAL, "comm_unique_id: call fibhip_comm_open first");
    FibNcclId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(out128, id.internal, sizeof id.internal);
    return 0;
}

// everything fibhip_comm_init can refuse WITHOUT talking to another rank: callers run it on every rank and agree
// on the outcome before any of them enters the collective ncclCommInitRank (a rank that failed alone would
// leave the others blocked inside it)
extern "C" int fibhip_comm_check(fibhip_t h, int rank, int nranks)
{
    if (!h) return fail(FIBHIP_EINVAL, "null handle");
    if (rank < 0 || rank >= nranks) return fail(FIBHIP_EINVAL, "comm_init: bad argument");
    if (!g_rccl.lib) return fail(FIBHIP_EINVAL, "comm_init: call fibhip_comm_open first");
    if (h->comm) return fail(FIBHIP_EINVAL, "comm_init: this handle already has a communicator");
    if (!(h->d.flags & FIBHIP_ROW_INTERLEAVED))
        return fail(FIBHIP_EINVAL, "comm_init: the direct exchange needs the row-interleaved slab (one block per message)");
    if ((rank > 0) != (h->d.ghost_top > 0) || (rank < nranks - 1) != (h->d.ghost_bottom > 0))
        if (nranks > 1)
            return fail(FIBHIP_EINVAL, "comm_init: rank %d of %d does not match the ghost rows of this block", rank, nranks);
    return 0;
}

extern "C" int fibhip_comm_init(fibhip_t h, const char *id128, int rank, int nranks)
{
    NEED(h);
    if (!id128) return fail(FIBHIP_EINVAL, "comm_init: bad argument");
    if (int rc = fibhip_comm_check(h, rank, nranks)) return rc;
    FibNcclId id;
    memcpy(id.internal, id128, sizeof id.internal);
    void *comm = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&comm, nranks, id, rank));               // collective over the `nranks` callers
    h->comm = comm;
    return 0;
}

// The messages of one halo exchange, as offsets into the slab the open tick writes: my outermost owned ghost-depth
// rows of ALL arrays go to the neighbours, theirs arrive in my ghost rows, in place.  One description serves both
// transports (fibhip_comm_exchange below; the caller's own library through fibhip_halo_plan), so that the row
// arithmetic the multi-rank tests verify is the arithmetic RCCL executes.
extern "C" int fibhip_halo_plan(fibhip_t h, int up_rank, int down_rank, fibhip_halo_msg *out, int *slab_index)
{
    NEED(h);
    if (!out) return fail(FIBHIP_EINVAL, "halo_plan: null argument");
    if (h->phase_of_tick != 1) return fail(FIBHIP_EINVAL, "halo_plan: call it between step_edges and step_commit");
    if (!(h->d.flags & FIBHIP_ROW_INTERLEAVED))
        return fail(FIBHIP_EINVAL, "halo_plan: needs the row-interleaved slab (one block per message)");
    if ((up_rank >= 0) != (h->d.ghost_top > 0) || (down_rank >= 0) != (h->d.ghost_bottom > 0))
        return fail(FIBHIP_EINVAL, "halo_plan: neighbours do not match the ghost rows of this block");
    for (int v = 1; v < h->nvar; ++v)
        if (h->nxt[v] != h->nxt[0])
            return fail(FIBHIP_EINVAL, "halo_plan: the arrays of this tick live in different slabs (one sub-step per "
                                       "launch without a multi-tick ghost zone): use the packed exchange");
    const long long row = h->pitch;                                        // floats per grid row, all arrays
    int n = 0;
    if (up_rank >= 0) {
        const long long cnt = (long long)h->d.ghost_top * row;
        out[n++] = {(long long)h->own0 * row, cnt, up_rank, 1};
        out[n++] = {0, cnt, up_rank, 0};
    }
    if (down_rank >= 0) {
        const long long cnt = (long long)h->d.ghost_bottom * row;
        out[n++] = {(long long)(h->own1 - h->d.ghost_bottom) * row, cnt, down_rank, 1};
        out[n++] = {(long long)h->own1 * row, cnt, down_rank, 0};
    }
    if (slab_index) *slab_index = h->nxt[0];
    return n;
}

// the same messages as ONE grouped RCCL kernel on the handle's stream
extern "C" int fibhip_comm_exchange(fibhip_t h, int up_rank, int down_rank)
{
    NEED(h);
    if (!h->comm) return fail(FIBHIP_EINVAL, "comm_exchange: no communicator (fibhip_comm_init)");
    fibhip_halo_msg msg[4];
    int idx = 0;
    const int n = fibhip_halo_plan(h, up_rank, down_rank, msg, &idx);
    if (n < 0) return n;
    float *slab = h->slab[idx];
    NCCLCHK(g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        if (msg[i].send)
            NCCLCHK(g_rccl.Send(slab + msg[i].offset, (size_t)msg[i].count, FIB_NCCL_FLOAT, msg[i].peer, h->comm, h->s0));
        else
            NCCLCHK(g_rccl.Recv(slab + msg[i].offset, (size_t)msg[i].count, FIB_NCCL_FLOAT, msg[i].peer, h->comm, h->s0));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return 0;
}

extern "C" int fibhip_comm_free(fibhip_t h)
{
    if (!h) return 0;
    if (h->comm && g_rccl.CommDestroy) {
        hipSetDevice(h->d.device);
        if (h->s0) hipStreamSynchronize(h->s0);
        g_rccl.CommDestroy(h->comm);
    }
    h->comm = nullptr;
    return 0;
}

// ------------------------------------------------------------------------------------------
// run-time modules: the device code of ONE traced model, compiled in-process by the caller (hiprtc), loaded here and
// launched through the module API by the same host logic that drives the built-in models
// ------------------------------------------------------------------------------------------
extern "C" int fibhip_module_load(int device, const void *code, size_t nbytes, const fibhip_module_desc *d, fibhip_module_t *out)
{
    if (!code || !nbytes || !d || !out) return fail(FIBHIP_EINVAL, "module_load: null argument");
    if (d->struct_size != (int)sizeof(fibhip_module_desc))
        return fail(FIBHIP_EINVAL, "fibhip_module_desc size mismatch: caller %d, library %d", d->struct_size, (int)sizeof(fibhip_module_desc));
    if (d->nvar < 1 || d->nvar > FIB_MAXVAR || d->nmodes < 1 || d->nmodes > 8 || d->steps_per_tick < 1 || d->consts_bytes < 0 ||
        d->consts_bytes > 64 || d->nkernels < 1 || !d->kernels)
        return fail(FIBHIP_EINVAL, "module_load: inconsistent description");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FIBHIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FIBHIP_EINVAL, "device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    fibhip_module *m = new (std::nothrow) fibhip_module();
    if (!m) return fail(FIBHIP_ENOMEM, "out of host memory");
    m->device = device;
    m->nvar = d->nvar; m->spt = d->steps_per_tick; m->nmodes = d->nmodes; m->consts_bytes = d->consts_bytes;
    for (int i = 0; i < 8; ++i) m->masks[i] = d->masks[i];
    m->K = d->K; m->TX = d->TX; m->TY = d->TY; m->R = d->R; m->TYB = d->TYB;
    m->K2 = d->K2; m->TX2 = d->TX2; m->TY2 = d->TY2; m->R2 = d->R2;
    if (hipModuleLoadData(&m->mod, code) != hipSuccess) {
        delete m;
        return fail(FIBHIP_EHIP, "module_load: hipModuleLoadData refused the code object");
    }
    for (int i = 0; i < d->nkernels; ++i) {
        const fibhip_module_kernel &k = d->kernels[i];
        Variant v;
        v.model = FIBHIP_CUSTOM; v.mode = k.mode; v.fast = k.fast; v.phase = k.phase;
        v.K = k.K; v.TX = k.TX; v.TY = k.TY; v.NT = k.NT;
        v.fn = launch_module;
        v.kind = k.kind;
        if (k.kind == MK_STRIP_MT) {                    // the multi-tick form of a strip kernel listed before it
            Variant *base = nullptr;
            for (Variant &b : m->variants)
                if (b.kind == MK_STRIP && b.mode == k.mode && b.fast == k.fast && b.phase == k.phase && b.K == k.K && b.TX == k.TX &&
                    b.TY == k.TY && b.NT == k.NT)
                    base = &b;
            if (!base || !k.symbol || hipModuleGetFunction(&base->kern_mt, m->mod, k.symbol) != hipSuccess) {
                hipModuleUnload(m->mod);
                delete m;
                return fail(FIBHIP_EINVAL, "module_load: kernel %d (%s): no strip kernel of that shape before it, or missing from the code object", i,
                            k.symbol ? k.symbol : "(null)");
            }
            base->fn_mt = launch_module;
            continue;
        }
        const bool ok_shape = k.kind == MK_POINTWISE ||
                              (k.kind == MK_TICK && k.NT >= 64 && k.NT <= 1024 && k.K >= 1 && k.TX >= 1 && k.TY >= 1) ||
                              (k.kind == MK_STRIP && k.NT < 0 && k.NT > -32 && k.K >= 2 && k.TX + 2 * (k.K - 1) <= 62 &&
                               (k.TY + 2 * (k.K - 1) + (-k.NT) - 1) / (-k.NT) <= 16);
        if (!k.symbol || !ok_shape || hipModuleGetFunction(&v.kern, m->mod, k.symbol) != hipSuccess) {
            hipModuleUnload(m->mod);
            delete m;
            return fail(FIBHIP_EINVAL, "module_load: kernel %d (%s) is missing from the code object or has an impossible shape", i,
                        k.symbol ? k.symbol : "(null)");
        }
        m->variants.push_back(v);
    }
    *out = m;
    return 0;
}

extern "C" int fibhip_module_unload(fibhip_module_t m)
{
    if (!m) return 0;
    hipSetDevice(m->device);
    if (m->mod) hipModuleUnload(m->mod);
    delete m;
    return 0;
}

extern "C" int fibhip_warm(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FIBHIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FIBHIP_EINVAL, "device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(256), 0, 0, (const fib_v4f *)nullptr, (fib_v4f *)nullptr, (size_t)0);   // n = 0: touches nothing
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(0));
    return 0;
}

extern "C" int fibhip_copy_bandwidth(int device, size_t nbytes, int reps, float *gbs)
{
    if (!gbs || nbytes < (1u << 20) || reps < 1) return fail(FIBHIP_EINVAL, "copy_bandwidth: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FIBHIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    const size_t n = nbytes / sizeof(fib_v4f);
    fib_v4f *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    do {
        if (hipMalloc((void **)&a, n * sizeof(fib_v4f)) != hipSuccess || hipMalloc((void **)&b, n * sizeof(fib_v4f)) != hipSuccess ||
            hipMemset(a, 0, n * sizeof(fib_v4f)) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
            hipEventCreate(&e1) != hipSuccess) {
            rc = fail(FIBHIP_EHIP, "copy_bandwidth: allocation failed");
            break;
        }
        const unsigned grid = (unsigned)((n + 255) / 256);  // one 16-byte element per thread
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);   // warm-up
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
            hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(FIBHIP_EHIP, "copy_bandwidth: kernel failed"); break; }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (!rc) *gbs = (float)(2.0 * (double)(n * sizeof(fib_v4f)) / (best * 1e-3) / 1e9);   // bytes read + written
    } while (0);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (a) hipFree(a);
    if (b) hipFree(b);
    return rc;
}

extern "C" int fibhip_state_ptr(fibhip_t h, int var, void **dev_ptr)
{
    if (!h || !dev_ptr || var < 0 || var >= h->nvar) return fail(FIBHIP_EINVAL, "state_ptr: bad argument");
    FLUSH(h);
    // the caller may write the state through this pointer at any time, between any two calls: nothing may run ahead of it
    // any more (a launch started before the caller asked for its ticks would read the state before such a write, or race it)
    h->ptr_exposed = true;
    h->ahead_ok = false;
    if (h->use_agg && !h->d.ghost_top && !h->d.ghost_bottom) {   // the caller may write through the pointer at any time: back to
                                                                 // the plain kernels (a shard's ghost rows: see create_impl)
        h->use_agg = false;
        if (int rc = build_plan(h)) return rc;
    }
    *dev_ptr = h->slab[h->cur[var]] + (size_t)var * h->vstride;
    return h->cur[var];
}

extern "C" int fibhip_next_ptr(fibhip_t h, int var, void **dev_ptr)
{
    if (!h || !dev_ptr || var < 0 || var >= h->nvar) return fail(FIBHIP_EINVAL, "next_ptr: bad argument");
    if (h->phase_of_tick == 0) return fail(FIBHIP_EINVAL, "next_ptr: no tick in flight");
    *dev_ptr = h->slab[h->nxt[var]] + (size_t)var * h->vstride;
    return h->nxt[var];
}

extern "C" int fibhip_halo_vars(fibhip_t h)
{
    if (!h) return fail(FIBHIP_EINVAL, "null handle");
    return (h->spt > 1 || h->cycle > 1) ? h->nvar : 1;
}

extern "C" int fibhip_halo_due(fibhip_t h)
{
    if (!h) return fail(FIBHIP_EINVAL, "null handle");
    // (ticks fibhip_step has accepted but not launched yet count: the NEXT tick is the one asked about)
    return ((h->d.ghost_top || h->d.ghost_bottom) && (h->cpos + h->pending) % h->cycle == h->cycle - 1) ? 1 : 0;
}

extern "C" int fibhip_plan_tile(fibhip_t h, int *tile_w, int *tile_h, int *rows_per_wave)
{
    if (!h) return fail(FIBHIP_EINVAL, "null handle");
    if (h->plan.empty()) return fail(FIBHIP_EINVAL, "no plan");
    const PlanItem &it = (h->multi_max > 1 && !h->plan_multi[h->multi_max].empty()) ? h->plan_multi[h->multi_max][0] : h->plan[0];
    if (tile_w) *tile_w = it.TX;
    if (tile_h) *tile_h = it.TY;
    if (rows_per_wave) *rows_per_wave = it.v ? -it.v->NT : 0;
    return 0;
}

extern "C" int fibhip_ticks_per_launch(fibhip_t h)
{
    if (!h) return fail(FIBHIP_EINVAL, "null handle");
    if (mt_variant(h)) return h->mt_max;
    return h->multi_max;
}

extern "C" int fibhip_trace_begin(fibhip_t h)
{
    NEED(h);
    FLUSH(h);
    if (h->phase_of_tick) return fail(FIBHIP_EINVAL, "trace_begin inside an open tick");
    for (auto &r : h->trace) {
        if (r.e0) hipEventDestroy(r.e0);
        if (r.e1) hipEventDestroy(r.e1);
    }
    h->trace.clear();
    h->tracing = true;
    return 0;
}

extern "C" int fibhip_trace_end(fibhip_t h, fibhip_trace_event *out, int max_events)
{
    NEED(h);
    if (!h->tracing) return fail(FIBHIP_EINVAL, "trace_end without trace_begin");
    if (!out || max_events < 1) return fail(FIBHIP_EINVAL, "trace_end: bad argument");
    FLUSH(h);
    h->tracing = false;
    HIPCHK(wait_stream(h->s1));
    SYNC_S0(h);
    int n = 0;
    for (const auto &r : h->trace) {
        if (n >= max_events) {                        // (like snprintf: the return value says how many there were)
            n = (int)h->trace.size();
            break;
        }
        float t0 = 0.f, dur = 0.f;
        HIPCHK(hipEventElapsedTime(&t0, h->trace[0].e0, r.e0));
        HIPCHK(hipEventElapsedTime(&dur, r.e0, r.e1));
        fibhip_trace_event &e = out[n++];
        memcpy(e.name, r.name, sizeof e.name);
        e.start_us = t0 * 1e3;
        e.dur_us = dur * 1e3;
        e.K = r.K; e.tile_w = r.TX; e.tile_h = r.TY; e.rows_per_wave = r.R; e.ticks = r.ticks;
    }
    return n;
}

extern "C" int fibhip_launch_stats(fibhip_t h, long long out[4])
{
    if (!h || !out) return fail(FIBHIP_EINVAL, "launch_stats: null argument");
    out[0] = h->launches;
    out[1] = h->n_ticks;
    out[2] = h->n_mt_launches;
    out[3] = h->n_mt_ticks;
    return 0;
}

extern "C" int fibhip_spec_stats(fibhip_t h, long long out[2])
{
    if (!h || !out) return fail(FIBHIP_EINVAL, "spec_stats: null argument");
    out[0] = h->n_spec_kept;
    out[1] = h->n_spec_redone;
    return 0;
}

extern "C" int fibhip_fallbacks(fibhip_t h, long long out[2])
{
    if (!h || !out) return fail(FIBHIP_EINVAL, "fallbacks: null argument");
    out[0] = h->n_fallbacks;
    out[1] = h->n_replayed;
    return 0;
}

extern "C" int fibhip_set_mt_wait_ms(fibhip_t h, int ms)
{
    if (!h || ms < 0 || ms > 0xFFFFFF) return fail(FIBHIP_EINVAL, "set_mt_wait_ms: 0 (the default, 2000) .. 16777215");
    h->mt_wait_ms = (unsigned)ms;
    return 0;
}

extern "C" int fibhip_expect(fibhip_t h, int nticks)
{
    if (!h || nticks < 0) return fail(FIBHIP_EINVAL, "expect: bad argument");
    h->expect = nticks;
    h->expect_fresh = nticks > 0;
    return 0;
}

extern "C" int fibhip_launch_plan(fibhip_t h, int *fused_steps, int *launches_per_tick)
{
    if (!h) return fail(FIBHIP_EINVAL, "null handle");
    if (!h->tuned && h->phase_of_tick == 0 && !h->pending && check_ready(h) == 0) {   // report the plan that will run
        HIPCHK(hipSetDevice(h->d.device));
        if (int rc = autotune(h)) return rc;
    }
    if (fused_steps) *fused_steps = h->plan.empty() ? 0 : h->plan[0].K;
    if (launches_per_tick) *launches_per_tick = (int)h->plan.size();
    return 0;
}
