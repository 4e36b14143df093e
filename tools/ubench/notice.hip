// Diagnostic (never shipped): how long after a kernel has ended does the host know?  A kernel that spins for ~200 us, then
//   A: the host spins on hipStreamQuery (what fibhip's wait_stream does);
//   B: a hipStreamWriteValue32 to page-locked host memory is enqueued behind the kernel and the host spins on that word;
//   C: hipStreamSynchronize.
// Reported: host time from just before the launch call to the notice, minus the kernel's own duration (HIP events in a separate pass).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void spin_kernel(unsigned long long ticks, unsigned *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (sink && threadIdx.x == 12345) *sink = 1;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    const unsigned long long ticks = 20000;                         // 100 MHz: 200 us
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *word, *word_dev;
    hipHostMalloc((void **)&word, 64, hipHostMallocDefault);
    *word = 0;
    hipHostGetDevicePointer((void **)&word_dev, word, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // the kernel's own duration
    std::vector<float> kd;
    for (int i = 0; i < 50; ++i) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(spin_kernel, dim3(252), dim3(960), 0, s, ticks, (unsigned *)nullptr);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        kd.push_back(ms * 1e3f);
    }
    std::sort(kd.begin(), kd.end());
    const double kernel_us = kd[kd.size() / 2];
    unsigned seq = 0;
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> t;
        for (int i = 0; i < reps + 20; ++i) {
            hipStreamSynchronize(s);
            const double t0 = now_us();
            hipLaunchKernelGGL(spin_kernel, dim3(252), dim3(960), 0, s, ticks, (unsigned *)nullptr);
            if (mode == 0) {
                while (hipStreamQuery(s) == hipErrorNotReady) {}
            } else if (mode == 1) {
                ++seq;
                hipStreamWriteValue32(s, word_dev, seq, 0);
                while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {}
            } else {
                hipStreamSynchronize(s);
            }
            const double t1 = now_us();
            if (i >= 20) t.push_back(t1 - t0 - kernel_us);
        }
        std::sort(t.begin(), t.end());
        printf("%s: launch call to notice minus the kernel's %.1f us: median %.2f us, 10th percentile %.2f, 90th %.2f (%s)\n",
               mode == 0 ? "A hipStreamQuery spin     " : (mode == 1 ? "B stream write + host spin" : "C hipStreamSynchronize    "), kernel_us,
               t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
