// Diagnostic (never shipped): the multi-tick Fenton kernel alone, timed by HIP events, for same-box A/B of two versions of
// kernels.hpp:   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DKH='"path/kernels.hpp"' [-DNEW_ARGS] mt_ab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include KH
using namespace fib;
#ifndef POLICY
#define POLICY Fast
#endif

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 30;
#ifndef SUB0_ARG
#define SUB0_ARG 0
#endif
#ifndef TILE_TY
#define TILE_TY 25
#endif
#ifndef STRIP_R
#define STRIP_R 3
#endif
    constexpr int K = 10, TX = 44, TY = TILE_TY, R = STRIP_R, H = 512, W = 512;
    const size_t n = (size_t)H * W;
    float *buf, *xb;
    unsigned *ep;
    hipMalloc(&buf, 14 * n * sizeof(float));
    hipMalloc(&xb, 8 * n * sizeof(float));
    hipMalloc(&ep, (1024 * 64 + 256) * sizeof(unsigned));
    hipMemset(ep, 0, (1024 * 64 + 256) * sizeof(unsigned));
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
    unsigned *hostw, *hostw_dev;
    hipHostMalloc((void **)&hostw, (MT_HOST_WORD_AT + 16) * sizeof(unsigned), hipHostMallocDefault);
    memset(hostw, 0, (MT_HOST_WORD_AT + 16) * sizeof(unsigned));
    hipHostGetDevicePointer((void **)&hostw_dev, hostw, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int rep = 0; rep < reps + 5; ++rep) {
#ifdef NEW_ARGS
        MtArgs mt{xb, ep, ep + 1024 * 64, epoch0, (unsigned)T | ((unsigned)rep + 1u) << 16, nullptr, hostw_dev, 0, (int)(1526u << 8)};
#else
        MtArgs mt{xb, ep, ep + 1024 * 64, epoch0, T, nullptr, nullptr, 0, 0};
#endif
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((strip_mt_kernel<Fenton, POLICY, 0, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, SUB0_ARG, mt);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        epoch0 += T - 1;
        if (rep >= 5) us.push_back(ms * 1e3f / T);
    }
    std::sort(us.begin(), us.end());
    printf("%s: %d ticks per launch, us per tick: min %.3f median %.3f max %.3f\n", argv[0], T, us.front(), us[us.size() / 2], us.back());
    return 0;
}
