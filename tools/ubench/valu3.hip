// Instruction-form price list for gfx950 (wave64): ns per wave-instruction per SIMD, from wall time, at 1/2/4 waves
// per SIMD, one 256*wps-thread workgroup per CU, hand-written instruction streams (asm volatile, VGPR operands and
// inline/literal constants only, four independent dependency chains unless the name says otherwise).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 20000

__device__ unsigned long long g_st[2 * 4096];
#define REP4(a) a a a a
#define REP16(a) REP4(a) REP4(a) REP4(a) REP4(a)

// each body = 16 instructions on v[x0..x3] (4 chains); CLOB lists what it clobbers
#define KERNEL(NAME, BODY, CLOB...)                                                                  \
    __global__ void NAME(float *out, float a, float b)                                               \
    {                                                                                                \
        float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, y0 = b, y1 = a; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        _Pragma("unroll 1") for (int i = 0; i < N_ITER; ++i)                                         \
            asm volatile(BODY : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1) : : CLOB);   \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1;                     \
        if (threadIdx.x == 0) { g_st[2 * blockIdx.x] = t1 - t0; g_st[2 * blockIdx.x + 1] = r1 - r0; }  \
    }

KERNEL(k_fma, REP4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"), "memory")
KERNEL(k_fma_dep, REP16("v_fma_f32 %0, %0, %4, %5\n"), "memory")
KERNEL(k_add_lit, REP4("v_add_f32 %0, 0x3f8ccccd, %0\n v_add_f32 %1, 0x3f8ccccd, %1\n v_add_f32 %2, 0x3f8ccccd, %2\n v_add_f32 %3, 0x3f8ccccd, %3\n"), "memory")
KERNEL(k_mul_inl, REP4("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 0.5, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 0.5, %3\n"), "memory")
KERNEL(k_fma_sgpr, REP4("v_fma_f32 %0, %0, s20, %5\n v_fma_f32 %1, %1, s20, %5\n v_fma_f32 %2, %2, s20, %5\n v_fma_f32 %3, %3, s20, %5\n"), "memory", "s20")
KERNEL(k_dpp_wshr, REP4("v_add_f32_dpp %0, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                        "v_add_f32_dpp %2, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"), "memory")
KERNEL(k_dpp_rshr, REP4("v_add_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                        "v_add_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"), "memory")
KERNEL(k_dpp_quad, REP4("v_add_f32_dpp %0, %0, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                        "v_add_f32_dpp %2, %2, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"), "memory")
KERNEL(k_movdpp_wshr, REP4("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                           "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"), "memory")
KERNEL(k_exp, REP4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"), "memory")
KERNEL(k_rcp, REP4("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"), "memory")
KERNEL(k_fma_exp_1in8, REP4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_exp_f32 %3, %3\n")
                       , "memory")
KERNEL(k_med3, REP4("v_med3_f32 %0, %0, 0, 1.0\n v_med3_f32 %1, %1, 0, 1.0\n v_med3_f32 %2, %2, 0, 1.0\n v_med3_f32 %3, %3, 0, 1.0\n"), "memory")
KERNEL(k_fma_clamp, REP4("v_fma_f32 %0, %0, %4, %5 clamp\n v_fma_f32 %1, %1, %4, %5 clamp\n v_fma_f32 %2, %2, %4, %5 clamp\n v_fma_f32 %3, %3, %4, %5 clamp\n"), "memory")
KERNEL(k_max, REP4("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n"), "memory")
KERNEL(k_cmp_cnd_vcc, REP4("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %1, %2, vcc\n v_cmp_gt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %1, %2, vcc\n"), "memory", "vcc")
KERNEL(k_cmp_cnd_sgpr, REP4("v_cmp_gt_f32 s[20:21], %0, %4\n v_cmp_gt_f32 s[22:23], %3, %4\n v_cndmask_b32 %0, %1, %2, s[20:21]\n v_cndmask_b32 %3, %1, %2, s[22:23]\n"), "memory", "s20", "s21", "s22", "s23")
KERNEL(k_cmp_only, REP4("v_cmp_gt_f32 s[20:21], %0, %4\n v_cmp_gt_f32 s[22:23], %1, %4\n v_cmp_gt_f32 s[24:25], %2, %4\n v_cmp_gt_f32 s[26:27], %3, %4\n"), "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
KERNEL(k_cnd_only, REP4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"), "memory")
KERNEL(k_swizzle, REP4("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %1, %1 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %2, %2 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %3, %3 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)\n")
                  , "memory")
KERNEL(k_bperm, REP4("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n"), "memory")
KERNEL(k_fma_salu, REP4("v_fma_f32 %0, %0, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %3, %4, %5\n s_add_u32 s23, s23, 1\n"),
       "memory", "s20", "s21", "s22", "s23", "scc")

typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, BODY)                                                                          \
    __global__ void NAME(float *out, float a, float b)                                               \
    {                                                                                                \
        f2 x0 = {threadIdx.x * 1e-3f + a, a}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, y0 = {b, b}, y1 = {a, a}; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        _Pragma("unroll 1") for (int i = 0; i < N_ITER; ++i)                                         \
            asm volatile(BODY : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1) : : "memory");   \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0.x + x1.x + x2.y + x3.y + y0.x + y1.y;         \
        if (threadIdx.x == 0) { g_st[2 * blockIdx.x] = t1 - t0; g_st[2 * blockIdx.x + 1] = r1 - r0; }  \
    }
KERNEL2(k_pk_fma, REP4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"))
KERNEL2(k_pk_mul, REP4("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"))
KERNEL2(k_pk_add, REP4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"))
KERNEL(k_add_vv, REP4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"), "memory")
KERNEL(k_mul_vv, REP4("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"), "memory")
KERNEL(k_mac, REP4("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n"), "memory")
KERNEL(k_salu, REP16("s_add_u32 s20, s20, 1\n"), "memory", "s20", "scc")
KERNEL(k_nop, REP16("s_nop 0\n"), "memory")
typedef void (*kfn)(float *, float, float);
static void run(const char *name, kfn k, int ipi)
{
    static float *out = nullptr;
    if (!out) hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    printf("%-18s", name);
    const int nbs[] = {1, 2, 4, 8};
    for (int nb : nbs) {                              // nb workgroups of 4 waves per CU = nb waves per SIMD, evenly placed
        hipLaunchKernelGGL(k, dim3(256 * nb), dim3(256), 0, 0, out, 0.999f, 0.001f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * nb), dim3(256), 0, 0, out, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> h(2 * 256 * nb);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_st), h.size() * sizeof(unsigned long long));
        double dt = 0, dr = 0;
        for (int b = 0; b < 256 * nb; ++b) { dt += h[2 * b]; dr += h[2 * b + 1]; }
        const double n = (double)N_ITER * ipi * nb;
        // per SIMD: shader cycles of one wave's run / instructions issued on the SIMD meanwhile; clock; wall check
        printf(" | w%d %5.2f (%.2fGHz, wall %5.2f)", nb, dt / (256 * nb) / n, dt / (dr * 10.0), best * 1e6 / n * (dt / (dr * 10.0)));
    }
    printf("\n");
}

int main()
{
    {   // clocks up: ~2 s of back-to-back launches before anything is timed
        float *w; hipMalloc(&w, 256 * 1024 * sizeof(float));
        for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_fma, dim3(256), dim3(1024), 0, 0, w, 0.999f, 0.001f);
        hipDeviceSynchronize();
    }
#define R(k, n) run(#k, k, n)
    R(k_nop, 16); R(k_add_lit, 16); R(k_add_vv, 16); R(k_fma, 16); R(k_mac, 16); R(k_pk_fma, 16); R(k_pk_add, 16);
    R(k_fma_sgpr, 16); R(k_dpp_wshr, 16); R(k_exp, 16); R(k_med3, 16); R(k_max, 16); R(k_cmp_cnd_sgpr, 16); R(k_salu, 16);
    R(k_fma_salu, 32); R(k_swizzle, 16);
    return 0;
}
