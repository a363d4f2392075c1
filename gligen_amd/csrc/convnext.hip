// Spatial-map grounding tokenizer pieces for gfx950 (see convnext.h). Once-per-prompt work of a few GFLOP: the matrix
// products go through the MFMA GEMM, the kernels here are plain HBM/latency-bound gathers.
#include "convnext.h"

namespace gl {

static inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
    int64_t g = cdiv64(n, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ void patchify_f32_kernel(const float* __restrict__ x, bf16* __restrict__ out, int B, int Cin, int H, int W, int k, int Kpad) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * Ho * Wo * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kpad);
        const int64_t m = i / Kpad;
        float v = 0.f;
        if (col < k * k * Cin) {
            const int c = col % Cin, tap = col / Cin;
            const int ky = tap / k, kx = tap - ky * k;
            const int ox = (int)(m % Wo);
            const int oy = (int)((m / Wo) % Ho);
            const int b = (int)(m / ((int64_t)Wo * Ho));
            v = x[(((size_t)b * Cin + c) * H + oy * k + ky) * W + ox * k + kx];
        }
        out[i] = f2bf(v);
    }
}
int patchify_f32_launch(const float* x, bf16* out, int B, int Cin, int H, int W, int k, int Kpad, hipStream_t stream) {
    if (H % k || W % k || Kpad < k * k * Cin) return set_error(GL_ERR_ARG, "patchify: %dx%d image, kernel %d, Kpad %d", H, W, k, Kpad);
    hipLaunchKernelGGL(patchify_f32_kernel, dim3(grid_for((int64_t)B * (H / k) * (W / k) * Kpad)), dim3(256), 0, stream, x, out, B, Cin, H, W, k, Kpad);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// one thread per 8 channels of one (output pixel, tap)
__global__ void patchify_bf16_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int B, int H, int W, int C, int ld, int k) {
    const int Ho = H / k, Wo = W / k, C8 = C >> 3;
    const int64_t total = (int64_t)B * Ho * Wo * k * k * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        int64_t r = i / C8;
        const int tap = (int)(r % (k * k));
        r /= k * k;
        const int ox = (int)(r % Wo);
        const int oy = (int)((r / Wo) % Ho);
        const int b = (int)(r / ((int64_t)Wo * Ho));
        const int ky = tap / k, kx = tap - ky * k;
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + oy * k + ky) * W + ox * k + kx) * ld + c8 * 8);
        *reinterpret_cast<uint4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * (size_t)(k * k * C) + (size_t)tap * C + c8 * 8) = v;
    }
}
int patchify_bf16_launch(const bf16* x, bf16* out, int B, int H, int W, int C, int ld, int k, hipStream_t stream) {
    if (H % k || W % k || C % 8 || ld % 8) return set_error(GL_ERR_ARG, "patchify: %dx%dx%d (ld %d), kernel %d", H, W, C, ld, k);
    hipLaunchKernelGGL(patchify_bf16_kernel, dim3(grid_for((int64_t)B * (H / k) * (W / k) * k * k * (C / 8))), dim3(256), 0, stream, x, out, B, H, W, C, ld, k);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void pack_patch_weight_kernel(const float* __restrict__ w, bf16* __restrict__ out, int O, int I, int k, int Kpad) {
    const int64_t total = (int64_t)O * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kpad);
        const int o = (int)(i / Kpad);
        float v = 0.f;
        if (col < k * k * I) {
            const int c = col % I, tap = col / I;
            v = w[((size_t)o * I + c) * k * k + tap];
        }
        out[i] = f2bf(v);
    }
}
int pack_patch_weight_launch(const float* w, bf16* out, int O, int I, int k, int Kpad, hipStream_t stream) {
    hipLaunchKernelGGL(pack_patch_weight_kernel, dim3(grid_for((int64_t)O * Kpad)), dim3(256), 0, stream, w, out, O, I, k, Kpad);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// one thread per (pixel, 8 channels): 49 taps of 16-byte loads, fp32 accumulate
__global__ void __launch_bounds__(256) dwconv7_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      bf16* __restrict__ y, int B, int H, int W, int C, int ld) {
    const int C8 = C >> 3;
    const int64_t total = (int64_t)B * H * W * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % C8) * 8;
        const int64_t p = i / C8;
        const int ox = (int)(p % W);
        const int oy = (int)((p / W) % H);
        const int b = (int)(p / ((int64_t)W * H));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias[c0 + e];
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = oy + ky - 3;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = ox + kx - 3;
                if (ix < 0 || ix >= W) continue;
                U4BF8 v;
                v.u = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * ld + c0);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(bf2f(v.e[e]), w[(size_t)(c0 + e) * 49 + ky * 7 + kx], acc[e]);
            }
        }
        U4BF8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = f2bf(acc[e]);
        *reinterpret_cast<uint4*>(y + (size_t)p * ld + c0) = o.u;
    }
}
int dwconv7_launch(const bf16* x, const float* w, const float* bias, bf16* y, int B, int H, int W, int C, int ld, hipStream_t stream) {
    if (C % 8 || ld % 8 || ld < C) return set_error(GL_ERR_ARG, "dwconv7: C=%d ld=%d", C, ld);
    hipLaunchKernelGGL(dwconv7_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, stream, x, w, bias, y, B, H, W, C, ld);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void conv3x3_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                   int B, int Cin, int Cout, int H, int W) {
    const int64_t total = (int64_t)B * Cout * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W);
        const int oy = (int)((i / W) % H);
        const int o = (int)((i / ((int64_t)W * H)) % Cout);
        const int b = (int)(i / ((int64_t)W * H * Cout));
        float acc = bias[o];
        for (int c = 0; c < Cin; ++c) {
            const float* xc = x + ((size_t)b * Cin + c) * H * W;
            const float* wc = w + ((size_t)o * Cin + c) * 9;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    if (ix < 0 || ix >= W) continue;
                    acc = fmaf(wc[ky * 3 + kx], xc[(size_t)iy * W + ix], acc);
                }
            }
        }
        y[i] = acc;
    }
}
int conv3x3_f32_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, hipStream_t stream) {
    hipLaunchKernelGGL(conv3x3_f32_kernel, dim3(grid_for((int64_t)B * Cout * H * W)), dim3(256), 0, stream, x, w, bias, y, B, Cin, Cout, H, W);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void token_mix_kernel(const bf16* __restrict__ feat, int ld, const float* __restrict__ mask, const float* __restrict__ null_feat,
                                 const float* __restrict__ pos, bf16* __restrict__ out, int B, int T, int C) {
    const int64_t total = (int64_t)B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        const float m = mask[b];
        out[i] = f2bf(bf2f(feat[(size_t)r * ld + c]) * m + null_feat[c] * (1.f - m) + pos[(size_t)t * C + c]);
    }
}
int token_mix_launch(const bf16* feat, int ld, const float* mask, const float* null_feat, const float* pos, bf16* out, int B, int T, int C,
                     hipStream_t stream) {
    hipLaunchKernelGGL(token_mix_kernel, dim3(grid_for((int64_t)B * T * C)), dim3(256), 0, stream, feat, ld, mask, null_feat, pos, out, B, T, C);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void scale_rows_kernel(float* w, float* b, const float* __restrict__ gamma, int O, int K) {
    const int64_t total = (int64_t)O * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i / K);
        w[i] *= gamma[o];
        if (i % K == 0) b[o] *= gamma[o];
    }
}
int scale_rows_launch(float* w, float* b, const float* gamma, int O, int K, hipStream_t stream) {
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for((int64_t)O * K)), dim3(256), 0, stream, w, b, gamma, O, K);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace gl
