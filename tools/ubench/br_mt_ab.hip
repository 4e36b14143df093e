// Diagnostic (never shipped): the multi-tick Beeler-Reuter kernel alone (Chebyshev gates, table baked in), timed by HIP events,
// for same-box A/B of versions of kernels.hpp / br_step.inc:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DFIB_ONLY_BR \
//         -DFIB_BR_TABLE_INC='"<abs path>/fib_tf_amd/_spec/br_table_<tag>.inc"' -DKH='"<path>/kernels.hpp"' [-DTY=21 -DR=2] br_mt_ab.hip
// (the same flags + `-S --cuda-device-only` give the kernel's ISA)
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
#ifndef TY
#define TY 21
#endif
#ifndef R
#define R 2
#endif

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 30;
    constexpr int K = 5, TX = 54, H = 512, W = 512, NV = 8;
    const size_t n = (size_t)H * W;
    float *buf, *xb;
    unsigned *ep;
    hipMalloc(&buf, (2 * NV + 6) * n * sizeof(float));
    hipMalloc(&xb, 2 * NV * n * sizeof(float));
    hipMalloc(&ep, (1024 * 64 + 256) * sizeof(unsigned));
    hipMemset(ep, 0, (1024 * 64 + 256) * sizeof(unsigned));
    // a resting state with a depolarised band, so that every branch of the kinetics sees ordinary arguments
    std::vector<float> h((2 * NV + 6) * n);
    const float rest[NV] = {-84.624f, 1e-4f, 0.01f, 0.988f, 0.975f, 0.003f, 0.994f, 0.0001f};
    for (int v = 0; v < NV; ++v)
        for (size_t i = 0; i < n; ++i) h[v * n + i] = h[(NV + v) * n + i] = rest[v];
    for (int r = 0; r < H; ++r)
        for (int c = 100; c < 140; ++c) h[(size_t)r * W + c] = 10.0f - 0.5f * (c - 100);
    for (size_t i = 0; i < 6 * n; ++i) h[2 * NV * n + i] = 0.3f + 0.4f * ((i * 2654435761u) % 1000) / 1000.f;
    hipMemcpy(buf, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    Geo g{H, W, W, H, 0, 0, H, 0, 0, (H + TY - 1) / TY, (W + TX - 1) / TX, 0};
    g.ntiles = g.tiles_x * ((H + TY - 1) / TY);
    PtrTab<NV> pt;
    for (int v = 0; v < NV; ++v) { pt.in[v] = buf + v * n; pt.out[v] = buf + (NV + v) * n; }
    float *p0 = buf + 2 * NV * n;
    PhaseTab ph{p0, p0 + n, p0 + 2 * n, p0 + 3 * n, p0 + 4 * n, p0 + 5 * n, p0 + 2 * n};
    BeelerReuter::Consts k;
    memset(&k, 0, sizeof k);
    k.dt = 0.1f; k.ddt = 0.0809f; k.mdt = -0.1f; k.mdt_skip = -0.5f; k.skip = 0;
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
        MtArgs mt{xb, ep, ep + 1024 * 64, epoch0, (unsigned)T | ((unsigned)rep + 1u) << 16, nullptr, hostw_dev, 0, (int)(1526u << 8)};
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((strip_mt_kernel<BeelerReuter, POLICY, 1, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0, mt);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        epoch0 += T - 1;
        if (rep >= 5) us.push_back(ms * 1e3f / T);
    }
    unsigned gave = 0;
    hipMemcpy(&gave, ep + 1024 * 64, 4, hipMemcpyDeviceToHost);
    std::sort(us.begin(), us.end());
    printf("%s: %d tiles, %d ticks per launch, us per tick: min %.3f median %.3f max %.3f%s\n", argv[0], g.ntiles, T, us.front(),
           us[us.size() / 2], us.back(), gave ? "  (A TILE GAVE UP)" : "");
    return 0;
}
