// Diagnostic (never shipped): what ONE sub-step's arithmetic costs when nothing else is in its way.  The Laplacian with its phase
// term and the model's update, exactly the device functions the fused kernels call (kernels.hpp: lap9, PhaseCoef::add, M::stepN),
// on registers alone — no LDS, no barrier, no memory inside the loop — in workgroups of NW waves, one per compute unit, like a tile
// of the multi-tick kernel.  The window's twelve values that would come from other lanes are made opaque after every step (the
// compiler must treat them as new: nothing of the stencil is hoisted), the centre column of the strip's own rows takes the new
// potential.  Cycles per step here = the instruction-issue bound of a sub-step with all NW strips live; the kernel's own
// sub-step (tools/ubench/stamp_mt.hip) minus this = what LDS, barriers and strips of unequal length cost.
//   Fenton:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DKH='"<path>/kernels.hpp"' [-DPOLICY=Exact] issue_bound.hip
//   BR:      ... -DMODEL_BR -DFIB_ONLY_BR -DFIB_BR_TABLE_INC='"<abs path>/br_table_<tag>.inc"' issue_bound.hip
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
#ifdef MODEL_BR
typedef BeelerReuter MODEL;
constexpr int MODE_ = 1, R_ = 2;
#else
typedef Fenton MODEL;
#ifndef RROWS
#define RROWS 3
#endif
constexpr int MODE_ = 0, R_ = RROWS;
#endif
#ifndef PAD_KB                  // LDS per workgroup: 96 KB = one workgroup per compute unit, 40 KB = two (waves per SIMD doubled)
#define PAD_KB 96
#endif
#ifndef NWAVES
#define NWAVES 15
#endif

template <class M, class P, int MODE, int R, int NW>
__global__ __launch_bounds__(64 * NW) void issue_kernel(float *buf, const typename M::Consts k, int n_iter, unsigned long long *cyc)
{
    constexpr int NV = M::NVAR;
    __shared__ float pad[PAD_KB * 256];                             // (PAD_KB: how many workgroups share a compute unit)
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)gridDim.x * blockDim.x;
    if (n_iter < 0) pad[threadIdx.x] = buf[tid];                    // (never: keeps the array)
    float s[R][NV], win[R + 2][3];
    PhaseCoef<P> pc[R];
    PhaseTab ph{buf + 40 * n, buf + 41 * n, buf + 42 * n, buf + 43 * n, buf + 44 * n, buf + 45 * n, buf + 42 * n};
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int v = 0; v < NV; ++v) s[r][v] = buf[(size_t)(r * NV + v) * n + tid];
        pc[r].load(ph, tid);
    }
#pragma unroll
    for (int j = 0; j < R + 2; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) win[j][c] = buf[(size_t)(30 + j) * n + tid] + 0.001f * c;
    const auto kk = M::pinned(k);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    auto one_step = [&](int it) {
        float lp[R], cc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float l = lap9<P>(win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2], win[r][0], win[r + 2][0], win[r][2], win[r + 2][2],
                              win[r + 1][1]);
            l = pc[r].add(l, win[r][1], win[r + 2][1], win[r + 1][0], win[r + 1][2]);
            lp[r] = l;
            cc[r] = win[r + 1][1];
        }
        M::template stepN<P, MODE, R>(s, cc, lp, kk, it);
#pragma unroll
        for (int r = 0; r < R; ++r) win[r + 1][1] = s[r][0];
#pragma unroll
        for (int j = 0; j < R + 2; ++j) asm volatile("" : "+v"(win[j][0]), "+v"(win[j][1]), "+v"(win[j][2]));
    };
#pragma unroll 1
    for (int it = 0; it < n_iter; it += 2) {                        // (two steps per pass, like the kernels' step loops)
        one_step(it);
        one_step(it + 1);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) buf[(size_t)(r * NV + v) * n + tid] = s[r][v];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (n_iter < 0) buf[tid] = pad[threadIdx.x ^ 1];
}

int main(int argc, char **argv)
{
    const int n_iter = argc > 1 ? atoi(argv[1]) : 20000, blocks = argc > 2 ? atoi(argv[2]) : 252;
    constexpr int NW = NWAVES;
    const size_t n = (size_t)blocks * 64 * NW;
    float *buf;
    unsigned long long *cyc;
    hipMalloc(&buf, 46 * n * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    std::vector<float> h(46 * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.3f + 0.4f * ((i * 2654435761u) % 1000) / 1000.f;
#ifdef MODEL_BR
    const float rest[8] = {-84.624f, 1e-4f, 0.01f, 0.988f, 0.975f, 0.003f, 0.994f, 0.0001f};
    for (int r = 0; r < R_; ++r)
        for (int v = 0; v < 8; ++v)
            for (size_t i = 0; i < n; ++i) h[(size_t)(r * 8 + v) * n + i] = rest[v];
    for (int j = 0; j < R_ + 2; ++j)
        for (size_t i = 0; i < n; ++i) h[(size_t)(30 + j) * n + i] = -84.624f;
    BeelerReuter::Consts k;
    memset(&k, 0, sizeof k);
    k.dt = 0.1f; k.ddt = 0.0809f; k.mdt = -0.1f; k.mdt_skip = -0.5f; k.skip = 0;
#else
    Fenton::Consts k{0.1f, 0.15f, 1.f - 0.1f / 3.33f, 1.f - 0.1f / 19.2f, 0.1f / 19.2f, 1.f - 0.1f / 160.f, 1.f - 0.1f / 75.f, 0.1f / 75.f};
#endif
    hipMemcpy(buf, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> us, cy;
    std::vector<unsigned long long> c(blocks);
    for (int rep = 0; rep < 7; ++rep) {
        hipMemcpy(buf, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((issue_kernel<MODEL, POLICY, MODE_, R_, NW>), dim3(blocks), dim3(64 * NW), 0, 0, buf, k, n_iter, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(c.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        if (rep >= 2) {
            us.push_back(ms * 1e3 / n_iter);
            cy.push_back((double)c[blocks / 2] / n_iter);
        }
    }
    std::sort(us.begin(), us.end());
    std::sort(cy.begin(), cy.end());
    printf("%s: %d waves per workgroup, %d workgroups, %d steps: %.4f us per step by events = %.0f cycles at 2.4 GHz; cycle counter per step (median workgroup) %.0f\n",
           argv[0], NW, blocks, n_iter, us[us.size() / 2], us[us.size() / 2] * 2400.0, cy[cy.size() / 2]);
    return 0;
}
