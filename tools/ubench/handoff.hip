// handoff.hip — what a tick boundary costs INSIDE a launch when a tile only synchronises with its 8 neighbours.
//
// The shape of the Fenton 512x512 launch (252 tiles of 44x25 cells, 15 waves each, compute box 62x43) without the
// arithmetic: per tick a workgroup "computes" for `busy` shader cycles (+ a per-tile, per-tick jitter), publishes its
// 44x25 cells x 4 variables, raises its epoch word, waits for the epoch words of its neighbours and reloads the 9-deep
// rim of its compute box from what they published.  Every reloaded value is checked (the value a cell must hold after
// tick t is a function of (row, col, variable, t)), so a stale line shows up as a count, not as a timing.
//
//   mode 0  exchange buffer [2][H][W] of 16-byte cells, buffer_store/load_dwordx4 ... sc1, flag = sc1 store, sc1 poll
//   mode 1  four planar arrays [2][4][H][W], global_store/load_dword ... sc1 (the slab's own layout)
//   mode 2  planar arrays, plain stores + agent release fence / agent acquire fence + plain loads
//   mode 3  the word travels WITH the data: 8-byte {tag, value} granules (32 bytes per cell, two dwordx4 sc1 stores), no
//           drain, no epoch word, no barrier: every wave re-reads its rim cells until every tag says `this tick`
//
// build: hipcc -O3 --offload-arch=gfx950 handoff.hip -o handoff ;  run: ./handoff [ticks] [busy_cycles]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                             \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int H = 512, W = 512, TX = 44, TY = 25, K = 10, R = 3;
constexpr int CX = TX + 2 * (K - 1), CY = TY + 2 * (K - 1), NW = (CY + R - 1) / R;
constexpr int TILES_X = (W + TX - 1) / TX, TILES_Y = (H + TY - 1) / TY, NTILES = TILES_X * TILES_Y;
constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ unsigned expect(int gy, int gx, int v, int t) { return (unsigned)((gy * W + gx) * 4 + v) * 64u + (unsigned)t; }

__device__ __forceinline__ int xcd_tile(int b, int ntiles)
{
    const int per = (ntiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

template <int MODE>
__global__ void __launch_bounds__(64 * NW)
handoff(unsigned *xbuf, unsigned *flags, unsigned *err, unsigned long long *cyc, int T, int busy, int fs)
{
    const int tile = xcd_tile(blockIdx.x, NTILES);
    if (tile >= NTILES) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int by = tile / TILES_X, bx = tile - by * TILES_X;
    const int y0 = by * TY, x0 = bx * TX, cy0 = y0 - (K - 1), cx0 = x0 - (K - 1);
    const int gx = cx0 - 1 + lane;
    bool own[R], rim[R];
    int gy[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        gy[r] = cy0 + wave * R + r;
        const bool in = lane >= 1 && lane <= CX && gx >= 0 && gx < W && gy[r] >= 0 && gy[r] < H && wave * R + r < CY;
        own[r] = in && gx >= x0 && gx < x0 + TX && gy[r] >= y0 && gy[r] < y0 + TY;
        rim[r] = in && !own[r];
    }
    const size_t plane = (size_t)H * W;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, (int)(2 * plane * 16), 0x00020000);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned bad = 0;
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0}, ts = 0;
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc[i] += n_ - ts; ts = n_; } while (0)
    for (int t = 0; t < T; ++t) {
        if (MODE == 3) {
            // ---- "compute" ----
            if (busy > 0) {
                const unsigned j = (unsigned)(tile * 2654435761u + t * 40503u) >> 22;
                const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)busy + j;
                while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(2);
            }
            const int par = t & 1;
            const unsigned tag = (unsigned)(t + 1);
            auto rs3 = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, (int)(2 * plane * 32), 0x00020000);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (own[r]) {
                    const size_t cell = (size_t)gy[r] * W + gx;
                    v4u a = {tag, expect(gy[r], gx, 0, t + 1), tag, expect(gy[r], gx, 1, t + 1)};
                    v4u b = {tag, expect(gy[r], gx, 2, t + 1), tag, expect(gy[r], gx, 3, t + 1)};
                    __builtin_amdgcn_raw_buffer_store_b128(a, rs3, (int)((par * plane + cell) * 32), 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(b, rs3, (int)((par * plane + cell) * 32 + 16), 0, 16);
                }
            }
            if (t + 1 == T) break;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                unsigned wrong = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (rim[r]) {
                        const size_t cell = (size_t)gy[r] * W + gx;
                        const v4u a = __builtin_amdgcn_raw_buffer_load_b128(rs3, (int)((par * plane + cell) * 32), 0, 16);
                        const v4u b = __builtin_amdgcn_raw_buffer_load_b128(rs3, (int)((par * plane + cell) * 32 + 16), 0, 16);
                        ok = ok && a.x == tag && a.z == tag && b.x == tag && b.z == tag;
                        wrong += (a.y != expect(gy[r], gx, 0, t + 1)) + (a.w != expect(gy[r], gx, 1, t + 1)) +
                                 (b.y != expect(gy[r], gx, 2, t + 1)) + (b.w != expect(gy[r], gx, 3, t + 1));
                    }
                }
                if (__all(ok)) {
                    bad += wrong;
                    break;
                }
                if (++spins > (SPIN_MAX >> 6)) {
                    if (lane == 0) atomicOr(err + 1, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();                                          // (the real kernel publishes the rim to LDS here)
            continue;
        }
        // ---- "compute" ----
        if (busy > 0) {
            const unsigned j = (unsigned)(tile * 2654435761u + t * 40503u) >> 22;      // 0..1023 cycles of jitter
            const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)busy + j;
            while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(2);
        }
        const int par = t & 1;
        ts = __builtin_amdgcn_s_memtime();
        // ---- publish the tile ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (own[r]) {
                const size_t cell = (size_t)gy[r] * W + gx;
                if (MODE == 0) {
                    v4u v = {expect(gy[r], gx, 0, t + 1), expect(gy[r], gx, 1, t + 1), expect(gy[r], gx, 2, t + 1), expect(gy[r], gx, 3, t + 1)};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)((par * plane + cell) * 16), 0, 16);
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        unsigned *p = xbuf + ((size_t)par * 4 + v) * plane + cell;
                        if (MODE == 1)
                            __hip_atomic_store(p, expect(gy[r], gx, v, t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else
                            *p = expect(gy[r], gx, v, t + 1);
                    }
                }
            }
        }
        STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(1);
        __syncthreads();
        STAMP(2);
        if (threadIdx.x == 0) {
            if (MODE == 2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(flags + (size_t)tile * fs, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t + 1 == T) break;
        // ---- wait for the 8 neighbours ----
        if (wave == 0) {
            const int d = lane < 4 ? lane : lane + 1;                      // 0..8 without the centre
            const int ny = by + d / 3 - 1, nx = bx + d % 3 - 1;
            const bool need = lane < 8 && ny >= 0 && ny < TILES_Y && nx >= 0 && nx < TILES_X;
            const unsigned *f = flags + (size_t)(need ? ny * TILES_X + nx : tile) * fs;
            unsigned spins = 0;
            for (;;) {
                const unsigned e = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!need || e >= (unsigned)(t + 1))) break;
                if (++spins > SPIN_MAX) {
                    if (lane == 0) atomicOr(err + 1, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (MODE == 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);
        // ---- reload the rim ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (rim[r]) {
                const size_t cell = (size_t)gy[r] * W + gx;
                if (MODE == 0) {
                    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((par * plane + cell) * 16), 0, 16);
                    bad += (v.x != expect(gy[r], gx, 0, t + 1)) + (v.y != expect(gy[r], gx, 1, t + 1)) +
                           (v.z != expect(gy[r], gx, 2, t + 1)) + (v.w != expect(gy[r], gx, 3, t + 1));
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const unsigned *p = xbuf + ((size_t)par * 4 + v) * plane + cell;
                        const unsigned x = MODE == 1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
                        bad += x != expect(gy[r], gx, v, t + 1);
                    }
                }
            }
        }
        asm volatile("" : "+v"(bad));
        STAMP(5);
    }
    if (bad) atomicAdd(err, bad);
    if (lane == 0 && (wave == 0 || wave == 7)) {
        const int o = (wave ? NTILES : 0) + tile;
        cyc[o * 8] = __builtin_amdgcn_s_memtime() - t0;
        for (int i = 0; i < 6; ++i) cyc[o * 8 + 1 + i] = acc[i];
    }
}

template <int MODE>
static void run(const char *name, int T, int busy, int fs, unsigned *xbuf, unsigned *flags, unsigned *err, unsigned long long *cyc)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr[2] = {0, 0};
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(flags, 0, (size_t)NTILES * 1024 * sizeof(unsigned)));
        CK(hipMemset(err, 0, 2 * sizeof(unsigned)));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(handoff<MODE>, dim3((NTILES + 7) / 8 * 8), dim3(64 * NW), 0, 0, xbuf, flags, err, cyc, T, busy, fs);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
        unsigned h2[2];
        CK(hipMemcpy(h2, err, sizeof h2, hipMemcpyDeviceToHost));
        herr[0] += h2[0];
        herr[1] |= h2[1];
    }
    std::vector<unsigned long long> hc(2 * NTILES * 8);
    CK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
    double m[7] = {0}, m7[7] = {0};
    for (int i = 0; i < NTILES; ++i)
        for (int k = 0; k < 7; ++k) {
            m[k] += (double)hc[i * 8 + k] / NTILES / T;
            m7[k] += (double)hc[(NTILES + i) * 8 + k] / NTILES / T;
        }
    const double us_tick = best * 1e3 / T, busy_us = busy / 2400.0;
    printf("%-44s T=%4d busy=%6d cyc (%.2f us): %.3f us per tick, %.3f beyond busy; wrong values %u, timeouts %u\n", name, T, busy,
           busy_us, us_tick, us_tick - busy_us, herr[0], herr[1]);
    printf("    wave 0 cycles per tick (mean over tiles): total %.0f = store issue %.0f + drain %.0f + barrier %.0f + flag/poll %.0f + barrier %.0f + rim loads %.0f\n",
           m[0], m[1], m[2], m[3], m[4], m[5], m[6]);
    printf("    wave 7 cycles per tick (mean over tiles): total %.0f = store issue %.0f + drain %.0f + barrier %.0f + (flag/poll) %.0f + barrier %.0f + rim loads %.0f\n",
           m7[0], m7[1], m7[2], m7[3], m7[4], m7[5], m7[6]);
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 200;
    unsigned *xbuf, *flags, *err;
    unsigned long long *cyc;
    CK(hipMalloc((void **)&xbuf, 2ull * H * W * 32));
    CK(hipMalloc((void **)&flags, (size_t)NTILES * 1024 * sizeof(unsigned)));
    CK(hipMalloc((void **)&err, 2 * sizeof(unsigned)));
    CK(hipMalloc((void **)&cyc, 2 * NTILES * 8 * sizeof(unsigned long long)));
    CK(hipMemset(xbuf, 0xff, 2ull * H * W * 32));
    const int busys[] = {0, 24000};
    const int strides[] = {1, 32, 64, 1024};
    for (int b : busys) {
        if (argc > 2 && atoi(argv[2]) != b) continue;
        for (int fs : strides) {
            printf("---- epoch words %d bytes apart\n", fs * 4);
            run<0>("16-byte cells, dwordx4 sc1", T, b, fs, xbuf, flags, err, cyc);
            run<1>("planar arrays, dword sc1", T, b, fs, xbuf, flags, err, cyc);
        }
        run<2>("planar arrays, plain + release/acquire fences", T, b, 1024, xbuf, flags, err, cyc);
        CK(hipMemset(xbuf, 0xff, 2ull * H * W * 32));
        run<3>("8-byte {tag, value} granules, no word, no drain", T, b, 1, xbuf, flags, err, cyc);
    }
    // the launch boundary this replaces: T launches of one tick each
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipMemset(flags, 0, (size_t)NTILES * 1024 * sizeof(unsigned)));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int t = 0; t < T; ++t)
                hipLaunchKernelGGL(handoff<0>, dim3((NTILES + 7) / 8 * 8), dim3(64 * NW), 0, 0, xbuf, flags, err, cyc, 1, 0, 1);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%d launches of one tick (publish only, no wait): %.3f us per launch\n", T, ms * 1e3 / T);
        }
    }
    return 0;
}
