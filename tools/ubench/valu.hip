// VALU issue-rate microbenchmark for gfx950: how many cycles does a SIMD need per wave64 VALU
// instruction at 1/2/4/8 waves per SIMD, for plain f32 ops, packed f32 ops and transcendentals?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 2000
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) {   // 8 independent fma chains (16 instr / iter)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {   // one dependent chain (16 instr / iter)
#pragma unroll
            for (int r = 0; r < 16; ++r) x0 = __builtin_fmaf(x0, a, b);
        } else if (MODE == 2) {   // packed fma, 4 independent chains (16 instr / iter = 32 fma)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            }
        } else if (MODE == 3) {   // mul+add separately (no contraction), 8 chains: 16 instr / iter
            x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
            x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
        } else if (MODE == 4) {   // v_exp_f32, 8 chains: 16 / iter
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
                x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
            }
        } else if (MODE == 6) {   // 2 independent chains, 16 instr / iter
#pragma unroll
            for (int r = 0; r < 8; ++r) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); }
        } else if (MODE == 7) {   // 3 independent chains, 15 instr / iter
#pragma unroll
            for (int r = 0; r < 5; ++r) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); }
        } else if (MODE == 8) {   // 4 independent chains, 16 instr / iter
#pragma unroll
            for (int r = 0; r < 4; ++r) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
        } else if (MODE == 9) {   // 3 chains of DPP adds (wave_shr:1 folded into v_add_f32_dpp), 15 instr / iter
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                x0 = b + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x138, 0xF, 0xF, true));
                x1 = b + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x1), 0x130, 0xF, 0xF, true));
                x2 = b + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x2), 0x138, 0xF, 0xF, true));
            }
        } else if (MODE == 10) {  // 3 chains, one v_exp_f32 + one v_rcp_f32 per 15 instr (the Fenton mix: 4 of 63)
#pragma unroll
            for (int r = 0; r < 4; ++r) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); }
            x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_fmaf(x2, a, b);
        } else if (MODE == 11) {  // 3 chains, fmed3 clamp + mul + add mix, literal constants: 15 instr / iter
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                x0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(x0, 134217728.0f, 0.5f), 0.0f, 1.0f); x1 = x1 * 0.123f; x2 = x2 + 0.77f;
            }
        } else if (MODE == 5) {   // cndmask / compare mix: 16 instr / iter
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                x0 = x0 > a ? x1 : x2; x1 = x1 > b ? x2 : x3; x2 = x2 > a ? x3 : x4; x3 = x3 > b ? x4 : x5;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int instr_per_iter)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    hipMalloc(&cyc, 4096 * sizeof(unsigned long long));
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD
        const int threads = 256 * wps;                // 4 SIMDs x wps waves, one block per CU
        if (threads > 1024) {                         // 2 blocks of 1024 per CU for wps = 8
            hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(1024), 0, 0, out, cyc, 1.0001f, 0.5f);
        } else
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int nb = threads > 1024 ? 512 : 256, nt = threads > 1024 ? 1024 : threads;
        hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(nt), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nb);
        hipMemcpy(h.data(), cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= nb;
        // s_memtime ticks at 100 MHz? report both: ticks and wall-derived
        const double instr_per_simd = (double)N_ITER * instr_per_iter * wps;
        printf("%-28s waves/SIMD %d: kernel %.1f us, memtime ticks/wave %.0f, instr/SIMD %.0f -> %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz); by s_memtime: %.2f cyc per wave-instr per SIMD\n",
               name, wps, ms * 1e3, avg, instr_per_simd, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, avg / instr_per_simd);
    }
}

int main()
{
    run<0>("fma x8 independent", 16);
    run<1>("fma dependent chain", 16);
    run<2>("pk_fma x4 independent", 16);
    run<3>("mul,add x8 independent", 16);
    run<4>("v_exp_f32 x8 independent", 16);
    run<5>("cmp+cndmask", 8);
    run<6>("fma x2 chains", 16);
    run<7>("fma x3 chains", 15);
    run<8>("fma x4 chains", 16);
    run<9>("add_dpp wave_shr x3 chains", 15);
    run<10>("fma x3 + exp + rcp per 15", 15);
    run<11>("fmed3(fma)+mul+add x3", 20);
    return 0;
}
