// Issue cost of the VALU instructions the attention softmax and the GEGLU epilogue are made of, per wave-instruction on one SIMD
// (clocks at the clock the chip sustains while running them), alone and beside v_mfma_f32_32x32x16_bf16 in the same wave.
//   build: hipcc --offload-arch=gfx950 -O3 -o valubench tools/valubench.hip ; run: ./valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

enum { K_FMA, K_EXP, K_MAX3, K_CVTPK, K_PKFMA32, K_PKMUL32, K_PKFMAH, K_EXPH, K_RCP, K_LDEXP, K_N };
static const char* names[K_N] = {"v_fma_f32", "v_exp_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_fma_f16",
                                 "v_exp_f16", "v_rcp_f32", "v_ldexp_f32"};

template <int KIND>
__device__ __forceinline__ void op(float (&x)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_PKFMAH) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_EXPH) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        if constexpr (KIND == K_LDEXP) asm volatile("v_ldexp_f32 %0, %0, %0" : "+v"(x[i]));
    }
    if constexpr (KIND == K_PKFMA32 || KIND == K_PKMUL32) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            f32x2 v = {x[i], x[i + 1]};
            if constexpr (KIND == K_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
            else asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v));
            // two independent values per instruction: issue it twice so that the instruction count per call stays 16
            if constexpr (KIND == K_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
            else asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v));
            x[i] = v.x; x[i + 1] = v.y;
        }
    }
}

// MODE 0: VALU only (16 independent chains x 4 = 64 instructions per iteration). MODE 1: the same 64 instructions spread between 8
// MFMAs (two independent accumulators). MODE 2: 8 MFMAs only.
template <int KIND, int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 1.0f + 0.001f * (threadIdx.x + i);
    f32x16 a0 = {0}, a1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            op<KIND>(x); op<KIND>(x); op<KIND>(x); op<KIND>(x);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 1) {
                    float y[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) y[i] = x[i];
                    // 8 of the 16 chains behind this MFMA, 8 behind the next
                    if constexpr (KIND == K_FMA) { for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_EXP) { for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_MAX3) { for (int i = 0; i < 8; ++i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_CVTPK) { for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_PKFMAH) { for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(x[i])); }
                    (void)y;
                }
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 1) {
                    if constexpr (KIND == K_FMA) { for (int i = 8; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_EXP) { for (int i = 8; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_MAX3) { for (int i = 8; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_CVTPK) { for (int i = 8; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i])); }
                    else if constexpr (KIND == K_PKFMAH) { for (int i = 8; i < 16; ++i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(x[i])); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + a0[i] + a1[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND, int MODE>
static float run(int wps, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wps;
    hipLaunchKernelGGL((k<KIND, MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / wps;   // ns per (iteration x wave) of one SIMD
}

template <int KIND>
static void row(float* out) {
    const int iters = 20000;
    float v[3], m[3];
    for (int w = 1; w <= 3; ++w) v[w - 1] = run<KIND, 0>(w == 3 ? 4 : w, iters, out);
    printf("%-20s alone: ns per instruction per SIMD at 1 / 2 / 4 waves: %6.2f %6.2f %6.2f", names[KIND], v[0] / 64, v[1] / 64, v[2] / 64);
    if (KIND == K_FMA || KIND == K_EXP || KIND == K_MAX3 || KIND == K_CVTPK || KIND == K_PKFMAH) {
        for (int w = 1; w <= 3; ++w) m[w - 1] = run<KIND, 1>(w == 3 ? 4 : w, iters, out);
        printf("   | 64 of them between 8 MFMAs, ns per iteration at 1 / 2 / 4 waves: %7.1f %7.1f %7.1f", m[0], m[1], m[2]);
    }
    printf("\n");
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    float mf[3];
    for (int w = 1; w <= 3; ++w) mf[w - 1] = run<K_FMA, 2>(w == 3 ? 4 : w, 20000, out);
    printf("8 x v_mfma_f32_32x32x16_bf16 alone (256 clocks): ns per iteration at 1 / 2 / 4 waves per SIMD: %.1f %.1f %.1f\n", mf[0], mf[1], mf[2]);
    row<K_FMA>(out); row<K_EXP>(out); row<K_MAX3>(out); row<K_CVTPK>(out); row<K_PKFMA32>(out); row<K_PKMUL32>(out);
    row<K_PKFMAH>(out); row<K_EXPH>(out); row<K_RCP>(out); row<K_LDEXP>(out);
    return 0;
}
