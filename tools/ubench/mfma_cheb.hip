// Diagnostic (never shipped): can v_mfma_f32_4x4x1_16B_f32 with the A operand broadcast from one block (cbsz = 4) serve as
// "four multiply-adds per lane on the lane's own value" — D[i][lane] = A[4*abid + i] * B[lane] + C[i][lane] — for the twelve
// degree-8 sums of Beeler-Reuter's Chebyshev gates (br.py:329-331), and is that bit-identical to v_fma_f32?
//   1. bitwise: MFMA against fmaf on random, denormal, huge, signed-zero and NaN operands, abid = 0, 5, 15;
//   2. under a partial EXEC mask (divergent branch): do active lanes still get the right A (source lanes inactive)?  are
//      inactive lanes' destination registers written?
//   3. issue: a loop of 248 VALU + 192 v_fmac per pass against 248 VALU + 54 MFMA against 248 VALU alone, 15 waves per
//      workgroup, one workgroup per compute unit (the shape of the Beeler-Reuter multi-tick kernel).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/ubench/mfma_cheb.hip -o tools/ubench/mfma_cheb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void k_check(const float *a, const float *b, const float *c, float *out_mfma, float *out_fma, int n)
{
    const int lane = threadIdx.x & 63;
    const float A = a[lane];
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const float B = b[t * 64 + lane];
        v4f C;
        for (int i = 0; i < 4; ++i) C[i] = c[(t * 64 + lane) * 4 + i];
        const v4f D = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, C, 4, ABID, 0);
        for (int i = 0; i < 4; ++i) {
            const float ai = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, A), 4 * ABID + i));
            out_mfma[(t * 64 + lane) * 4 + i] = D[i];
            out_fma[(t * 64 + lane) * 4 + i] = __builtin_fmaf(ai, B, C[i]);
        }
    }
}

// partial EXEC: lanes with (lane % 3 == 0) stay out of the branch (lanes 0 and 3 of block 0 among them: A sources)
__global__ void k_partial(const float *a, const float *b, float *out, float *ref)
{
    const int lane = threadIdx.x & 63;
    const float A = a[lane], B = b[lane];
    v4f D = {-777.0f, -777.0f, -777.0f, -777.0f};
    if (lane % 3 != 0) {
        const v4f C = {1.0f, 2.0f, 3.0f, 4.0f};
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, C, 4, 0, 0);
    }
    for (int i = 0; i < 4; ++i) {
        const float ai = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, A), i));
        out[lane * 4 + i] = D[i];
        ref[lane * 4 + i] = lane % 3 != 0 ? __builtin_fmaf(ai, B, (float)(i + 1)) : -777.0f;
    }
}

// ---- issue test -------------------------------------------------------------------------------------------------------
// MODE 0: NV VALU + 192 v_fmac (12 chains x 8 terms x 2 cells); 1: NV VALU + 54 MFMA (3 groups x 9 terms x 2 cells);
// 2: NV VALU alone.  The "other" VALU work is 8 independent fma chains.
template <int MODE, int NV>
__global__ void __launch_bounds__(960) k_issue(float *out, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = seed + 0.001f * (float)(threadIdx.x + j);
    float s0[9], s1[9];
    for (int j = 0; j < 9; ++j) {
        s0[j] = seed * (float)(j + 1) + 1e-3f * lane;
        s1[j] = seed * (float)(j + 2) + 2e-3f * lane;
    }
    const float A0 = 0.01f * (float)(lane + 1), A1 = 0.02f * (float)(lane + 1);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < NV / 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
        if (MODE == 0) {
            float r0[12], r1[12];
#pragma unroll
            for (int f = 0; f < 12; ++f) {
                r0[f] = 0.1f * (float)f;
                r1[f] = 0.2f * (float)f;
#pragma unroll
                for (int k = 1; k <= 8; ++k) {
                    r0[f] = __builtin_fmaf(s0[k], 0.001f * (float)(f * 9 + k), r0[f]);
                    r1[f] = __builtin_fmaf(s1[k], 0.001f * (float)(f * 9 + k), r1[f]);
                }
            }
#pragma unroll
            for (int f = 0; f < 12; ++f) acc += r0[f] + r1[f];
        } else if (MODE == 1) {
            v4f c0[3], c1[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                c0[g] = v4f{0.f, 0.f, 0.f, 0.f};
                c1[g] = v4f{0.f, 0.f, 0.f, 0.f};
            }
            // 27 (g, k) combinations: blocks 0..15 of A0, 0..10 of A1
#define MF(G, K, AR, AB)                                                                           \
    c0[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(AR, s0[K], c0[G], 4, AB, 0);                        \
    c1[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(AR, s1[K], c1[G], 4, AB, 0);
            MF(0, 0, A0, 0) MF(1, 0, A0, 1) MF(2, 0, A0, 2) MF(0, 1, A0, 3) MF(1, 1, A0, 4) MF(2, 1, A0, 5)
            MF(0, 2, A0, 6) MF(1, 2, A0, 7) MF(2, 2, A0, 8) MF(0, 3, A0, 9) MF(1, 3, A0, 10) MF(2, 3, A0, 11)
            MF(0, 4, A0, 12) MF(1, 4, A0, 13) MF(2, 4, A0, 14) MF(0, 5, A0, 15) MF(1, 5, A1, 0) MF(2, 5, A1, 1)
            MF(0, 6, A1, 2) MF(1, 6, A1, 3) MF(2, 6, A1, 4) MF(0, 7, A1, 5) MF(1, 7, A1, 6) MF(2, 7, A1, 7)
            MF(0, 8, A1, 8) MF(1, 8, A1, 9) MF(2, 8, A1, 10)
#undef MF
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += c0[g][i] + c1[g][i];
        }
        // keep the sums' inputs moving so that nothing is hoisted out of the loop
#pragma unroll
        for (int j = 1; j < 9; ++j) {
            s0[j] = __builtin_fmaf(acc, 1e-9f, s0[j]);
            s1[j] = __builtin_fmaf(acc, 1e-9f, s1[j]);
        }
    }
    float r = acc;
    for (int j = 0; j < 8; ++j) r += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float rnd_special(unsigned &st, int kind)
{
    st = st * 1664525u + 1013904223u;
    const float u = (float)(st >> 8) / 16777216.0f;
    switch (kind) {
    case 0: return (u - 0.5f) * 4.0f;                       // ordinary
    case 1: return (u - 0.5f) * 2e-39f;                     // denormal
    case 2: return (u - 0.5f) * 1e-20f;                     // products underflow
    case 3: return (u - 0.5f) * 3e38f;                      // near overflow
    case 4: return u < 0.5f ? 0.0f : -0.0f;
    case 5: return (u - 0.5f) * 1e3f;
    default: return u < 0.1f ? NAN : (u < 0.2f ? INFINITY : (u - 0.5f));
    }
}

template <int ABID>
static void check(int n)
{
    std::vector<float> a(64), b((size_t)n * 64), c((size_t)n * 256), om((size_t)n * 256), of((size_t)n * 256);
    unsigned st = 12345u + ABID;
    long bad_total = 0;
    for (int kind = 0; kind < 7; ++kind) {
        for (auto &v : a) v = rnd_special(st, kind == 6 ? 0 : kind);
        for (auto &v : b) v = rnd_special(st, kind);
        for (auto &v : c) v = rnd_special(st, kind == 1 ? 1 : (kind == 2 ? 1 : (kind == 6 ? 6 : 0)));
        float *da, *db, *dc, *dm, *df;
        hipMalloc(&da, 256); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, c.size() * 4); hipMalloc(&dm, c.size() * 4); hipMalloc(&df, c.size() * 4);
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check<ABID>, dim3(64), dim3(64), 0, 0, da, db, dc, dm, df, n);
        hipMemcpy(om.data(), dm, c.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(of.data(), df, c.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0, nanmis = 0;
        int shown = 0;
        for (size_t i = 0; i < om.size(); ++i) {
            unsigned x, y;
            memcpy(&x, &om[i], 4);
            memcpy(&y, &of[i], 4);
            if (x != y) {
                if (std::isnan(om[i]) && std::isnan(of[i])) { ++nanmis; continue; }
                ++bad;
                if (shown < 4) {
                    const size_t cell = i / 4;
                    printf("    mismatch kind %d: b=%a c=%a mfma=%a fma=%a\n", kind, b[cell], c[i], om[i], of[i]);
                    ++shown;
                }
            }
        }
        printf("  abid %2d kind %d: %zu results, %ld differ bitwise (%ld NaN payloads differ)\n", ABID, kind, om.size(), bad, nanmis);
        bad_total += bad;
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dm); hipFree(df);
    }
    printf("abid %d: %s\n", ABID, bad_total ? "NOT bit-identical to fmaf" : "bit-identical to fmaf on every operand class");
}

template <int MODE, int NV>
static void issue(const char *what, int iters)
{
    float *out;
    hipMalloc(&out, 256 * 960 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_issue<MODE, NV>), dim3(256), dim3(960), 0, 0, out, iters, 0.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    // 15 waves per CU = 3.75 per SIMD; cycles per pass per SIMD at 2.4 GHz
    printf("%-44s %9.1f us for %d passes = %7.1f ns per pass (%.0f cycles at 2.4 GHz)\n", what, us[us.size() / 2], iters,
           us[us.size() / 2] * 1e3f / iters, us[us.size() / 2] * 1e3f / iters * 2.4f);
    hipFree(out);
}

int main()
{
    printf("== 1. v_mfma_f32_4x4x1_16B_f32 (cbsz 4) against v_fma_f32, bitwise ==\n");
    check<0>(2048);
    check<5>(2048);
    check<15>(2048);

    printf("== 2. partial EXEC (lanes with lane %% 3 == 0 outside the branch; A comes from lanes 0..3) ==\n");
    {
        std::vector<float> a(64), b(64), o(256), r(256);
        for (int i = 0; i < 64; ++i) { a[i] = 0.5f + i; b[i] = 1.0f + 0.25f * i; }
        float *da, *db, *dout, *dref;
        hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024); hipMalloc(&dref, 1024);
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_partial, dim3(1), dim3(64), 0, 0, da, db, dout, dref);
        hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(r.data(), dref, 1024, hipMemcpyDeviceToHost);
        int act_bad = 0, inact_written = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                if (l % 3 != 0) act_bad += o[l * 4 + i] != r[l * 4 + i];
                else inact_written += o[l * 4 + i] != -777.0f;
            }
        printf("  active lanes wrong: %d of %d; inactive lanes' destination overwritten: %d of %d\n", act_bad, 4 * 42, inact_written, 4 * 22);
        printf("  lane 1: %g %g %g %g (want %g %g %g %g)\n", o[4], o[5], o[6], o[7], r[4], r[5], r[6], r[7]);
        printf("  lane 0 (inactive): %g %g %g %g\n", o[0], o[1], o[2], o[3]);
    }

    printf("== 3. issue: 15 waves per workgroup, 256 workgroups ==\n");
    issue<2, 248>("248 VALU", 2000);
    issue<0, 248>("248 VALU + 192 v_fmac", 2000);
    issue<1, 248>("248 VALU + 54 v_mfma_f32_4x4x1", 2000);
    issue<2, 8>("8 VALU", 2000);
    issue<0, 8>("8 VALU + 192 v_fmac", 2000);
    issue<1, 8>("8 VALU + 54 v_mfma_f32_4x4x1", 2000);
    return 0;
}
