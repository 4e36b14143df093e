// Diagnostic (never shipped): the Courtemanche fast-policy kernel on aggregates alone (three ticks per launch, tile 58x20, 2 rows per
// wave: what configs[4] runs), 1024x1024, timed by HIP events, for same-box A/B of versions of models.hpp:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -DKH='"<path>/kernels.hpp"' [-D...] court_ab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include KH
using namespace fib;

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 40, reps = argc > 2 ? atoi(argv[2]) : 9;
    constexpr int K = 3, TX = 58, TY = 20, R = 2, H = 1024, W = 1024, NV = CourtAgg::NVAR;
    const size_t n = (size_t)H * W;
    float *buf;
    hipMalloc(&buf, (2 * NV + 6) * n * sizeof(float));
    // the resting state of the model (court.py:124-160), aggregates from plausible values: only the time matters
    const float rest[NV] = {-81.18f, 11.17f, 0.002908f, 0.9649f, 0.9775f, 139.0f, 0.03043f, 0.9992f, 0.004966f, 0.9986f, 3.296e-5f,
                            0.01869f, 1.013e-4f, 1.367e-4f, 0.9996f, 0.7755f, 1.488f, 2.35e-112f, 1.0f, 0.9992f, 1.488f,
                            0.05f, 0.0001f, -86.0f, 0.0001f, 0.03f};
    std::vector<float> h((2 * NV + 6) * n);
    for (int v = 0; v < NV; ++v)
        for (size_t i = 0; i < n; ++i) h[v * n + i] = h[(NV + v) * n + i] = rest[v] * (1.0f + 1e-3f * ((i * 2654435761u) % 100) / 100.f);
    for (size_t i = 0; i < 6 * n; ++i) h[2 * NV * n + i] = 0.001f * (((i * 2654435761u) % 1000) / 1000.f);
    hipMemcpy(buf, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    Geo g{H, W, W, H, 0, 0, H, 0, 0, 0x7fffffff, (W + TX - 1) / TX, 0};
    g.ntiles = g.tiles_x * ((H + TY - 1) / TY);
    PtrTab<NV> pt;
    for (int v = 0; v < NV; ++v) { pt.in[v] = buf + v * n; pt.out[v] = buf + (NV + v) * n; }
    float *p0 = buf + 2 * NV * n;
    PhaseTab ph{p0, p0 + n, p0 + 2 * n, p0 + 3 * n, p0 + 4 * n, p0 + 5 * n, p0 + 2 * n};
    CourtConsts k;
    memset(&k, 0, sizeof k);
    k.dtf = 0.02f; k.dts = 0.2f; k.mdt_f = -0.02f; k.mdt_s = -0.2f; k.ddt = 0.02f; k.em1_fCa = -0.01f; k.em1_u = -0.0025f;
    k.chronic = 0.f; k.c_to = 16.52f; k.c_Kur = 100.f; k.c_CaL = 12.375f;
    constexpr int NW = (TY + 2 * (K - 1) + R - 1) / R;
    const int grid = ((g.ntiles + 7) / 8) * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int rep = 0; rep < reps + 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int l = 0; l < launches; ++l)
            hipLaunchKernelGGL((strip_kernel<CourtAgg, Fast, CourtAgg::MODE_FAST, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) us.push_back(ms * 1e3f / launches);
    }
    std::sort(us.begin(), us.end());
    printf("%s: %d tiles, us per launch of three ticks: min %.3f median %.3f max %.3f (%s)\n", argv[0], g.ntiles, us.front(), us[us.size() / 2],
           us.back(), hipGetErrorString(hipGetLastError()));
    return 0;
}
