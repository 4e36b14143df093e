// second VALU microbenchmark: which instruction kinds of the kernels' mix issue slower than v_fma?
// inline asm so that the compiler cannot rewrite the streams; 4 waves per SIMD, 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 1000
#define REP8(X) X X X X X X X X

template <int MODE>
__global__ void k(float *out, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
#pragma unroll 1
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) {        // v_add_f32 e32, VGPR operands, 8 independent regs  (16 instr)
            asm volatile(REP8("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (MODE == 1) { // v_add_f32 with a 32-bit literal (8-byte encoding)
            asm volatile(REP8("v_add_f32_e32 %0, 0x3f8ccccd, %0\n v_add_f32_e32 %1, 0x3f8ccccd, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (MODE == 2) { // v_cmp_gt_f32 e32 -> vcc ; v_cndmask_b32 e32 (reads vcc)
            asm volatile(REP8("v_cmp_gt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %0, vcc\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        } else if (MODE == 3) { // 8 cmps then 8 cndmasks through different SGPR pairs (e64)
            asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %8\n v_cmp_gt_f32_e64 s[22:23], %1, %8\n v_cmp_gt_f32_e64 s[24:25], %2, %8\n v_cmp_gt_f32_e64 s[26:27], %3, %8\n"
                         "v_cmp_gt_f32_e64 s[28:29], %4, %8\n v_cmp_gt_f32_e64 s[30:31], %5, %8\n v_cmp_gt_f32_e64 s[32:33], %6, %8\n v_cmp_gt_f32_e64 s[34:35], %7, %8\n"
                         "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[22:23]\n v_cndmask_b32_e64 %2, %2, %3, s[24:25]\n v_cndmask_b32_e64 %3, %3, %4, s[26:27]\n"
                         "v_cndmask_b32_e64 %4, %4, %5, s[28:29]\n v_cndmask_b32_e64 %5, %5, %6, s[30:31]\n v_cndmask_b32_e64 %6, %6, %7, s[32:33]\n v_cndmask_b32_e64 %7, %7, %0, s[34:35]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
        } else if (MODE == 4) { // v_mul_f32 e32 independent
            asm volatile(REP8("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (MODE == 5) { // only cndmask e32 with a fixed vcc
            asm volatile(REP8("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (MODE == 6) { // only v_cmp e32 -> vcc
            asm volatile(REP8("v_cmp_gt_f32_e32 vcc, %0, %8\n v_cmp_gt_f32_e32 vcc, %1, %8\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        } else if (MODE == 7) { // v_add with an SGPR operand
            asm volatile(REP8("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
        } else if (MODE == 8) { // v_fma_f32 (VOP3, 8 bytes) independent
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 9) { // v_fmac_f32 e32 (4 bytes) independent
            asm volatile(REP8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int wps : {1, 4}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)N_ITER * 16 * wps;
        printf("%-44s waves/SIMD %d: %.2f ns per wave-instr per SIMD\n", name, wps, (ms * 1e6 - 3000) / n);
    }
    hipFree(out);
}

int main()
{
    run<0>("v_add_f32 e32 vgpr");
    run<4>("v_mul_f32 e32 vgpr");
    run<9>("v_fmac_f32 e32");
    run<8>("v_fma_f32 (VOP3)");
    run<1>("v_add_f32 e32 + 32-bit literal");
    run<7>("v_add_f32 e32 sgpr operand");
    run<6>("v_cmp_gt_f32 e32 -> vcc");
    run<5>("v_cndmask_b32 e32 (vcc)");
    run<2>("v_cmp e32 ; v_cndmask e32 (dependent via vcc)");
    run<3>("8x v_cmp e64 ; 8x v_cndmask e64 (sgpr pairs)");
    return 0;
}
