// Does the MFMA pipe overlap with ordinary / transcendental VALU work on gfx950 — inside one wave, and across the waves
// of one SIMD?  Shapes follow attn_kernel<48,64> per 64-key tile: 14 x v_mfma_f32_32x32x16_bf16 against
// 32 v_exp_f32 + 32 v_fma_f32 + 22 v_max + 16 v_cvt_pk per lane.
//   mode 0: MFMA only          mode 1: VALU only
//   mode 2: 14 MFMA, then the VALU block (program order as in the kernel today)
//   mode 3: one MFMA, 1/14 of the VALU block, ... (fine interleave)
// build: hipcc --offload-arch=gfx950 -O3 -o overlapbench tools/overlapbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int N>
__device__ __forceinline__ void valu_chunk(float (&x)[32], float& m, int base) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int j = (base + i) & 31;
        float t = __builtin_fmaf(x[j], 0.5f, -1.0f);
        t = __builtin_amdgcn_exp2f(t);
        m = fmaxf(m, t);
        x[j] = t;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) overlap_kernel(float* out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0.01f * (threadIdx.x + i);
    float m = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 1 || MODE == 2) {
            valu_chunk<32>(x, m, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                valu_chunk<2>(x, m, j * 4);
                __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                valu_chunk<2>(x, m, j * 4 + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            valu_chunk<4>(x, m, 28);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = m;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 32; ++i) s += x[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
static float run(int waves_per_simd, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL(overlap_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(overlap_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / waves_per_simd;   // ns per (iteration x wave) of one SIMD
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 20000;
    printf("ns per tile-equivalent per wave slot of a SIMD (lower = more throughput); 14 MFMA = 448 cycles\n");
    printf("%-28s %10s %10s %10s\n", "mode", "1 wave", "2 waves", "3 waves");
    const char* names[4] = {"0 MFMA only", "1 VALU only", "2 MFMA block ; VALU block", "3 fine interleave"};
    for (int mode = 0; mode < 4; ++mode) {
        float r[3];
        for (int w = 1; w <= 3; ++w)
            r[w - 1] = mode == 0 ? run<0>(w, iters, out) : mode == 1 ? run<1>(w, iters, out) : mode == 2 ? run<2>(w, iters, out) : run<3>(w, iters, out);
        printf("%-28s %10.1f %10.1f %10.1f\n", names[mode], r[0], r[1], r[2]);
    }
    return 0;
}
