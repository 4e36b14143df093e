// What does the chip do under full VALU load?  Per wave: s_memtime (shader clock) and s_memrealtime (100 MHz) at the
// start and the end of a fixed instruction stream; reported: shader clock = d(memtime)/d(realtime), per-wave duration,
// and whether all waves ran concurrently (span of all waves vs the duration of one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define N_ITER 20000
#define REP4(a) a a a a
__global__ void k(float *out, unsigned long long *st, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, y0 = b, y1 = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < N_ITER; ++i)
        asm volatile(REP4("v_add_f32 %0, 0x3f8ccccd, %0\n v_add_f32 %1, 0x3f8ccccd, %1\n v_add_f32 %2, 0x3f8ccccd, %2\n v_add_f32 %3, 0x3f8ccccd, %3\n")
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *p = st + 4 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        p[0] = t0; p[1] = t1; p[2] = r0; p[3] = r1;
    }
}
int main()
{
    float *out; unsigned long long *st;
    hipMalloc(&out, 512 * 1024 * sizeof(float));
    hipMalloc(&st, 512 * 16 * 4 * sizeof(unsigned long long));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, st, 0.999f, 0.001f);
    hipDeviceSynchronize();
    for (int blocks = 64; blocks <= 512; blocks *= 2)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps, nw = blocks * threads / 64;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, st, 0.999f, 0.001f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, st, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(nw * 4);
            hipMemcpy(h.data(), st, nw * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double dt = 0, dr = 0; unsigned long long rmin = ~0ull, rmax = 0;
            for (int w = 0; w < nw; ++w) {
                dt += h[4 * w + 1] - h[4 * w]; dr += h[4 * w + 3] - h[4 * w + 2];
                rmin = std::min(rmin, h[4 * w + 2]); rmax = std::max(rmax, h[4 * w + 3]);
            }
            dt /= nw; dr /= nw;
            printf("blocks %3d x %4d threads (%d waves/SIMD where resident): event %.1f us | per wave: %.0f shader ticks, %.1f us real -> clock %.2f GHz, "
                   "%.2f ticks per instr per wave | all waves span %.1f us real\n",
                   blocks, threads, wps, ms * 1e3, dt, dr / 100.0, dt / (dr * 10.0), dt / (N_ITER * 16.0), (rmax - rmin) / 100.0);
        }
    return 0;
}
