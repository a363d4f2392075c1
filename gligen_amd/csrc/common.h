// gligen_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GL_OK 0
#define GL_ERR_HIP 1
#define GL_ERR_ARG 2
#define GL_ERR_STATE 3
#define GL_ERR_MISSING 4
#define GL_ERR_UNSUPPORTED 5

namespace gl {

int set_error(int code, const char* fmt, ...);
const char* last_error();

#define GL_HIP(expr)                                                               \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess)                                                      \
            return gl::set_error(GL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, \
                                 #expr, hipGetErrorString(_e));                    \
    } while (0)

#define GL_TRY(expr)               \
    do {                           \
        int _r = (expr);           \
        if (_r != GL_OK) return _r; \
    } while (0)

#define GL_LAUNCH_CHECK() GL_HIP(hipGetLastError())

// Developer A/B switches (GL_GEMM_*, GL_CONV_HALO, GL_ATTN_V2, GL_LN_FOLD, ...; tools/README.md). The product library does not
// let the environment pick kernels behind the caller's back: the switches are read only when GL_DEV_SWITCHES=1 is set (the
// tools/gpu_*.sh scripts export it), i.e. the library itself consults exactly one environment variable, once.
inline const char* dev_env(const char* name) {
    static const bool on = [] {
        const char* e = getenv("GL_DEV_SWITCHES");
        return e && atoi(e) != 0;
    }();
    return on ? getenv(name) : nullptr;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// x * sigmoid(x); exp2 + hardware reciprocal (1 ulp) instead of an IEEE division: the result is rounded to bf16
__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
// erf GELU, as torch F.gelu default (reference ldm/modules/attention.py:44). erf by Abramowitz & Stegun
// 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 output's 4e-3 relative step): 2 transcendental +
// ~10 plain VALU ops instead of libm erff's ~30 on the GEGLU epilogue's critical path.
__device__ __forceinline__ float gelu_erf_f(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float erf_abs = fmaf(-p, e, 1.f);
    return 0.5f * v * (1.f + copysignf(erf_abs, v));
}

// Two GEGLU outputs at once, val * gelu_erf(gate), written on 2-vectors so that hipcc emits packed fp32 VALU
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue): ~11 VALU instructions per output instead of ~21 for the
// scalar form -- the GEGLU epilogue of the K = 320 feed-forward GEMM spends more time in this function than in stores.
// Same Abramowitz & Stegun 7.1.26 erf as gelu_erf_f, algebraically rearranged:
//   gelu(v) = v/2 + |v|/2 * erf(|v|/sqrt2),  erf = 1 - p(t) * exp(-v^2/2),  t = 1 / (1 + 0.3275911 |v| / sqrt2)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 geglu2(f32x2 val, f32x2 gate) {
    f32x2 a;
    a.x = __builtin_fabsf(gate.x);
    a.y = __builtin_fabsf(gate.y);
    const f32x2 den = __builtin_elementwise_fma(a, f32x2{0.23164189f, 0.23164189f}, f32x2{1.f, 1.f});   // 0.3275911 / sqrt2
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(den.x);
    t.y = __builtin_amdgcn_rcpf(den.y);
    f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(t, p, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(t, p, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(t, p, f32x2{0.254829592f, 0.254829592f});
    p = p * t;
    const f32x2 w = a * 0.84932180f;   // sqrt(log2(e) / 2): exp(-v^2/2) = exp2(-w^2)
    const f32x2 w2 = w * w;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(-w2.x);
    e.y = __builtin_amdgcn_exp2f(-w2.y);
    const f32x2 ha = a * 0.5f;
    const f32x2 r = __builtin_elementwise_fma(-(p * e), ha, ha);   // |v|/2 * erf
    return (gate * 0.5f + r) * val;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

union U4BF8 {
    uint4 u;
    bf16x8 v;
    bf16 e[8];
};
union U2BF4 {
    uint2 u;
    bf16x4 v;
    bf16 e[4];
};

}  // namespace gl
