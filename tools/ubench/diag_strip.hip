// Diagnostic (never shipped): what bounds one launch of the Fenton strip kernel?  The same kernel built with one
// ingredient removed at a time (results are then wrong; only the time matters), event-timed back to back.
//   hipcc ... -DFIB_DIAG_NO_BARRIER | -DFIB_DIAG_NO_RELOAD | -DFIB_DIAG_NOTRANS | -DDIAG_LITE (kinetics removed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../fib_tf_amd/csrc/kernels.hpp"
using namespace fib;

struct FentonLite : Fenton {       // stencil + update only: the cost of everything that is not kinetics
    template <class P, int MODE, int R>
    static FIB_DEV void stepN(float (&s)[R][NVAR], const float (&U0)[R], const float (&lap)[R], const Consts &k, int)
    {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r][0] = __builtin_fmaf(lap[r], k.ddt, U0[r]);
    }
    template <class P, int MODE>
    static FIB_DEV void step(float (&s)[NVAR], float U0, float lap, const Consts &k, int) { s[0] = __builtin_fmaf(lap, k.ddt, U0); }
};
#ifdef DIAG_LITE
typedef FentonLite Model;
#else
typedef Fenton Model;
#endif

template <class P, int K, int TX, int TY, int R, bool PH = true>
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
    PhaseTab ph{buf + 8 * n, buf + 9 * n, buf + 10 * n, buf + 11 * n, buf + 10 * n};
    Fenton::Consts k{0.1f, 0.15f, 0.97f, 0.995f, 0.005f, 0.999f, 0.9987f, 0.0013f};
    constexpr int NW = (TY + 2 * (K - 1) + R - 1) / R;
    const int grid = ((g.ntiles + 7) / 8) * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) {
            for (int v = 0; v < 4; ++v) { pt.in[v] = buf + ((i & 1) * 4 + v) * n; pt.out[v] = buf + (((i & 1) ^ 1) * 4 + v) * n; }
            hipLaunchKernelGGL((strip_kernel<Model, P, 0, K, TX, TY, R, PH>), dim3(grid), dim3(64 * NW), 0, 0, g, pt, ph, k, 0);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-40s %7.2f us per launch\n", name, best * 1000.f / 200);
    hipFree(buf);
}

int main()
{
    run<Fast, 10, 44, 25, 3>(DIAG_NAME " fast K=10 44x25 R=3");
    run<Fast, 10, 44, 25, 4>(DIAG_NAME " fast K=10 44x25 R=4");
    run<Exact, 10, 44, 25, 3>(DIAG_NAME " exact K=10 44x25 R=3");
    run<Fast, 10, 44, 25, 3, false>(DIAG_NAME " fast K=10 44x25 R=3 NO PHASE FIELD");
    run<Fast, 2, 44, 25, 3>(DIAG_NAME " fast K=2 44x25 R=3 (fixed cost + 2 steps)");
    run<Fast, 2, 44, 25, 3, false>(DIAG_NAME " fast K=2 44x25 R=3 NO PHASE FIELD");
    return 0;
}
