// Diagnostic (never shipped): where does one launch of the Fenton strip kernel spend its cycles?
// Builds kernels.hpp with -DFIB_STAMPS and prints, per wave position, the s_memtime deltas between
// launch start / loads issued / first barrier / each sub-step / write-back.
#define FIB_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../fib_tf_amd/csrc/kernels.hpp"
using namespace fib;

template <class P, int K, int TX, int TY, int R>
void run_br(const char *name)
{
    const int H = 512, W = 512;
    const size_t n = (size_t)H * W;
    float *buf;
    hipMalloc(&buf, 20 * n * sizeof(float));
    std::vector<float> h(20 * n);
    for (size_t i = 0; i < 20 * n; ++i) h[i] = 0.3f + 0.4f * ((i * 2654435761u) % 1000) / 1000.f;
    for (size_t i = 0; i < n; ++i) h[i] = -80.f + 90.f * ((i * 2654435761u) % 1000) / 1000.f;       // V
    for (size_t i = n; i < 2 * n; ++i) h[i] = 1e-6f;                                                   // C
    hipMemcpy(buf, h.data(), 20 * n * sizeof(float), hipMemcpyHostToDevice);
    Geo g{H, W, W, H, 0, 0, H, 0, 0, (H + TY - 1) / TY, (W + TX - 1) / TX, 0};
    g.ntiles = g.tiles_x * ((H + TY - 1) / TY);
    PtrTab<8> pt;
    for (int v = 0; v < 8; ++v) { pt.in[v] = buf + v * n; pt.out[v] = buf + (8 + v) * n; }
    PhaseTab ph{buf + 16 * n, buf + 17 * n, buf + 18 * n, buf + 19 * n, buf + 18 * n};
    BeelerReuter::Consts k{};
    k.dt = 0.1f; k.ddt = 0.0809f; k.mdt = -0.1f; k.mdt_skip = -0.5f; k.skip = 0;
    constexpr int NW = (TY + 2 * (K - 1) + R - 1) / R;
    const int grid = ((g.ntiles + 7) / 8) * 8;
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL((strip_kernel<BeelerReuter, P, 0, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(4096 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(fib_stamps), st.size() * sizeof(unsigned long long));
    printf("== %s: grid %d x %d threads (%d waves)\n", name, grid, 64 * NW, NW);
    for (int b : {grid / 2 + 3}) {
        for (int w = 0; w < NW; ++w) {
            const unsigned long long *s = &st[(size_t)((b * 16 + w) % 4096) * 16];
            printf("  wave %2d: load-issue %5llu  wait+barrier %5llu | steps:", w, s[1] - s[0], s[2] - s[1]);
            for (int q = 0; q < K; ++q) printf(" %5llu", s[3 + q] - s[2 + q]);
            printf(" | store %5llu  total %6llu\n", s[14] - s[2 + K], s[14] - s[0]);
        }
    }
    hipFree(buf);
}

template <class P, int K, int TX, int TY, int R>
void run(const char *name)
{
    const int H = 512, W = 512;
    const size_t n = (size_t)H * W;
    float *buf;
    hipMalloc(&buf, 12 * n * sizeof(float));
    std::vector<float> h(12 * n);
    for (size_t i = 0; i < 12 * n; ++i) h[i] = 0.3f + 0.4f * ((i * 2654435761u) % 1000) / 1000.f;
    hipMemcpy(buf, h.data(), 12 * n * sizeof(float), hipMemcpyHostToDevice);
    Geo g{H, W, W, H, 0, 0, H, 0, 0, (H + TY - 1) / TY, (W + TX - 1) / TX, 0};
    g.ntiles = g.tiles_x * ((H + TY - 1) / TY);
    PtrTab<4> pt;
    for (int v = 0; v < 4; ++v) { pt.in[v] = buf + v * n; pt.out[v] = buf + (4 + v) * n; }
    PhaseTab ph{buf + 8 * n, buf + 9 * n, buf + 10 * n, buf + 11 * n, buf + 10 * n};
    Fenton::Consts k{0.1f, 0.15f, 0.97f, 0.995f, 0.005f, 0.999f, 0.9987f, 0.0013f};
    constexpr int NW = (TY + 2 * (K - 1) + R - 1) / R;
    const int grid = ((g.ntiles + 7) / 8) * 8;
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL((strip_kernel<Fenton, P, 0, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(4096 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(fib_stamps), st.size() * sizeof(unsigned long long));
    printf("== %s: grid %d x %d threads (%d waves)\n", name, grid, 64 * NW, NW);
    // block in the middle of the domain
    for (int b : {grid / 2 + 3, 3}) {
        printf(" block %d\n", b);
        for (int w = 0; w < NW; ++w) {
            const unsigned long long *s = &st[(size_t)((b * 16 + w) % 4096) * 16];
            printf("  wave %2d: load-issue %5llu  wait+barrier %5llu | steps:", w, s[1] - s[0], s[2] - s[1]);
            for (int q = 0; q < K; ++q) printf(" %5llu", s[3 + q] - s[2 + q]);
            printf(" | store %5llu  total %6llu\n", s[14] - s[2 + K], s[14] - s[0]);
        }
    }
    // block start skew
    unsigned long long t0 = ~0ull, t1 = 0, e1 = 0;
    for (int b = 0; b < g.ntiles && b < 256; ++b) { t0 = std::min(t0, st[(size_t)(b * 16) * 16]); t1 = std::max(t1, st[(size_t)(b * 16) * 16]); e1 = std::max(e1, st[(size_t)(b * 16) * 16 + 14]); }
    printf(" first block start -> last block start: %llu ticks; first start -> last end: %llu ticks\n", t1 - t0, e1 - t0);
    hipFree(buf);
}

template <class P, int K, int TX, int TY, int R>
void run_rows(const char *name)
{
    const int H = 512, W = 512;
    const size_t n = (size_t)H * W;
    float *buf;
    hipMalloc(&buf, 12 * n * sizeof(float));
    std::vector<float> h(12 * n);
    for (size_t i = 0; i < 12 * n; ++i) h[i] = 0.3f + 0.4f * ((i * 2654435761u) % 1000) / 1000.f;
    hipMemcpy(buf, h.data(), 12 * n * sizeof(float), hipMemcpyHostToDevice);
    Geo g{H, W, W, H, 0, 0, H, 0, 0, (H + TY - 1) / TY, (W + TX - 1) / TX, 0};
    g.ntiles = g.tiles_x * ((H + TY - 1) / TY);
    PtrTab<4> pt;
    for (int v = 0; v < 4; ++v) { pt.in[v] = buf + v * n; pt.out[v] = buf + (4 + v) * n; }
    PhaseTab ph{buf + 8 * n, buf + 9 * n, buf + 10 * n, buf + 11 * n, buf + 10 * n};
    Fenton::Consts k{0.1f, 0.15f, 0.97f, 0.995f, 0.005f, 0.999f, 0.9987f, 0.0013f};
    constexpr int NW = (TY + 2 * (K - 1) + R - 1) / R;
    const int grid = ((g.ntiles + 7) / 8) * 8;
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL((rows_kernel<Fenton, P, 0, K, TX, TY, R, true>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(4096 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(fib_stamps), st.size() * sizeof(unsigned long long));
    printf("== %s: grid %d x %d threads (%d waves)\n", name, grid, 64 * NW, NW);
    // block in the middle of the domain
    for (int b : {grid / 2 + 3, 3}) {
        printf(" block %d\n", b);
        for (int w = 0; w < NW; ++w) {
            const unsigned long long *s = &st[(size_t)((b * 16 + w) % 4096) * 16];
            printf("  wave %2d: load-issue %5llu  wait+barrier %5llu | steps:", w, s[1] - s[0], s[2] - s[1]);
            for (int q = 0; q < K; ++q) printf(" %5llu", s[3 + q] - s[2 + q]);
            printf(" | store %5llu  total %6llu\n", s[14] - s[2 + K], s[14] - s[0]);
        }
    }
    // block start skew
    unsigned long long t0 = ~0ull, t1 = 0, e1 = 0;
    for (int b = 0; b < g.ntiles && b < 256; ++b) { t0 = std::min(t0, st[(size_t)(b * 16) * 16]); t1 = std::max(t1, st[(size_t)(b * 16) * 16]); e1 = std::max(e1, st[(size_t)(b * 16) * 16 + 14]); }
    printf(" first block start -> last block start: %llu ticks; first start -> last end: %llu ticks\n", t1 - t0, e1 - t0);
    hipFree(buf);
}

int main()
{
    run<Fast, 10, 44, 25, 3>("strip fenton fast K=10 44x25 R=3");
    run_rows<Fast, 10, 44, 25, 3>("ROWS fenton fast K=10 44x25 R=3");
    run_rows<Fast, 10, 44, 25, 4>("ROWS fenton fast K=10 44x25 R=4");
    return 0;
}
