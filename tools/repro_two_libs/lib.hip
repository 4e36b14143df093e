// one kernel + one C entry point that launches it; built twice (-DNAME=a / -DNAME=b) into liba.so / libb.so.
// Minimal reproducer for the order-dependent SIGSEGV under rocprofv3: no fibhip code at all.
#include <hip/hip_runtime.h>
#define CAT(x, y) x##y
#define XCAT(x, y) CAT(x, y)
__global__ void XCAT(kernel_, NAME)(int *p) { if (p) p[threadIdx.x] = threadIdx.x; }
extern "C" __attribute__((visibility("default"))) int XCAT(launch_, NAME)(void)
{
    hipLaunchKernelGGL(XCAT(kernel_, NAME), dim3(1), dim3(64), 0, 0, (int *)nullptr);
    return (int)hipDeviceSynchronize();
}
