// Diagnostic (never shipped): where does a TICK BOUNDARY inside a multi-tick launch spend its cycles?
// Builds kernels.hpp with -DFIB_STAMPS: s_memtime stamps around every phase of the boundary (kernels.hpp FIB_BSTAMP) and
// around every sub-step (FIB_STAMP), last tick / last boundary of the launch kept; means over all tiles per wave position.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize stamp_mt.hip -o stamp_mt ; ./stamp_mt [ticks]
#define FIB_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../../fib_tf_amd/csrc/kernels.hpp"
using namespace fib;

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    constexpr int K = 10, TX = 44, TY = 25, R = 3, H = 512, W = 512;
    const size_t n = (size_t)H * W;
    float *buf, *xb;
    unsigned *ep;
    hipMalloc(&buf, 14 * n * sizeof(float));
    hipMalloc(&xb, 8 * n * sizeof(float));
    hipMalloc(&ep, (1024 * 64 + 128) * sizeof(unsigned));
    hipMemset(ep, 0, (1024 * 64 + 128) * sizeof(unsigned));
    std::vector<float> h(14 * n);
    for (size_t i = 0; i < 14 * n; ++i) h[i] = 0.3f + 0.4f * ((i * 2654435761u) % 1000) / 1000.f;
    hipMemcpy(buf, h.data(), 14 * n * sizeof(float), hipMemcpyHostToDevice);
    Geo g{H, W, W, H, 0, 0, H, 0, 0, (H + TY - 1) / TY, (W + TX - 1) / TX, 0};
    g.ntiles = g.tiles_x * ((H + TY - 1) / TY);
    PtrTab<4> pt;
    for (int v = 0; v < 4; ++v) { pt.in[v] = buf + v * n; pt.out[v] = buf + (4 + v) * n; }
    PhaseTab ph{buf + 8 * n, buf + 9 * n, buf + 10 * n, buf + 11 * n, buf + 12 * n, buf + 13 * n, buf + 10 * n};
    Fenton::Consts k{0.1f, 0.15f, 1.f - 0.1f / 3.33f, 1.f - 0.1f / 19.2f, 0.1f / 19.2f, 1.f - 0.1f / 160.f, 1.f - 0.1f / 75.f, 0.1f / 75.f};
    constexpr int NW = (TY + 2 * (K - 1) + R - 1) / R;
    const int grid = ((g.ntiles + 7) / 8) * 8;
    unsigned epoch0 = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    unsigned *hostw, *hostw_dev;                              // the host's word (never raised here)
    hipHostMalloc((void **)&hostw, (MT_HOST_WORD_AT + 16) * sizeof(unsigned), hipHostMallocDefault);
    memset(hostw, 0, (MT_HOST_WORD_AT + 16) * sizeof(unsigned));
    hipHostGetDevicePointer((void **)&hostw_dev, hostw, 0);
    for (int rep = 0; rep < 5; ++rep) {
        MtArgs mt{xb, ep, ep + 1024 * 64, epoch0, (unsigned)T | ((unsigned)rep + 1u) << 16, nullptr, hostw_dev, 0, (int)(1526u << 8)};
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((strip_mt_kernel<Fenton, Fast, 0, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0, mt);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        epoch0 += T - 1;
    }
    printf("strip_mt_kernel<Fenton, Fast, K=10, 44x25, R=3>, %d ticks per launch: %.2f us per launch, %.2f us per tick (stamped build)\n", T, ms * 1e3, ms * 1e3 / T);
    std::vector<unsigned long long> st(4096 * 16), bs(4096 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(fib_stamps), st.size() * 8);
    hipMemcpyFromSymbol(bs.data(), HIP_SYMBOL(fib_bstamps), bs.size() * 8);
    const char *names[9] = {"store issue", "drain", "barrier", "epoch+poll", "barrier", "rim loads", "LDS publish", "barrier", "window"};
    printf("cycles (2.4 GHz: 2400 = 1 us); mean over the %d tiles, last boundary of the launch\n", g.ntiles);
    printf("%-8s", "wave");
    for (int q = 0; q < 9; ++q) printf(" %12s", names[q]);
    printf(" %10s | sub-steps of the last tick (sum)\n", "boundary");
    for (int w = 0; w < NW; ++w) {
        double acc[9] = {0}, tot = 0, steps = 0;
        int cnt = 0;
        for (int b = 0; b < grid; ++b) {
            const int tile = (b & 7) * ((g.ntiles + 7) >> 3) + (b >> 3);
            if (tile >= g.ntiles) continue;
            const unsigned long long *s = &bs[(size_t)((b * 16 + w) % 4096) * 16];
            const unsigned long long *t = &st[(size_t)((b * 16 + w) % 4096) * 16];
            for (int q = 0; q < 9; ++q) acc[q] += (double)(s[q + 1] - s[q]);
            tot += (double)(s[9] - s[0]);
            steps += (double)(t[2 + K] - s[9]);                     // window read of the last boundary -> end of the last sub-step
            ++cnt;
        }
        printf("wave %-3d", w);
        for (int q = 0; q < 9; ++q) printf(" %12.0f", acc[q] / cnt);
        printf(" %10.0f | %8.0f\n", tot / cnt, steps / cnt);
    }
    // round 4: the sub-steps of the last tick, per wave — cycles from the end of the previous sub-step (barrier passed, window
    // read) to the arrival at this sub-step's barrier (arithmetic + LDS write: "work") and from there to the end of the sub-step
    // (barrier wait + window read: "wait").  The last sub-step has no barrier.
    std::vector<unsigned long long> ws(4096 * 16);
    hipMemcpyFromSymbol(ws.data(), HIP_SYMBOL(fib_wstamps), ws.size() * 8);
    printf("\nsub-steps of the last tick: work / wait cycles per wave (mean over the tiles)\n%-8s", "wave");
    for (int st = 0; st < K; ++st) printf("   step %d      ", st);
    printf("  sum work  sum wait\n");
    for (int w = 0; w < NW; ++w) {
        double work[K] = {0}, wait[K] = {0};
        int cnt = 0;
        for (int b = 0; b < grid; ++b) {
            const int tile = (b & 7) * ((g.ntiles + 7) >> 3) + (b >> 3);
            if (tile >= g.ntiles) continue;
            const size_t at = (size_t)((b * 16 + w) % 4096) * 16;
            const unsigned long long *s = &bs[at], *t = &st[at], *u = &ws[at];
            for (int q = 0; q < K; ++q) {
                const unsigned long long begin = q == 0 ? s[9] : t[2 + q];
                if (q + 1 < K) {
                    work[q] += (double)(u[q] - begin);
                    wait[q] += (double)(t[3 + q] - u[q]);
                } else {
                    work[q] += (double)(t[3 + q] - begin);
                }
            }
            ++cnt;
        }
        double sw = 0, sa = 0;
        printf("wave %-3d", w);
        for (int q = 0; q < K; ++q) {
            printf(" %6.0f/%-6.0f", work[q] / cnt, wait[q] / cnt);
            sw += work[q] / cnt;
            sa += wait[q] / cnt;
        }
        printf(" %9.0f %9.0f\n", sw, sa);
    }
    return 0;
}
