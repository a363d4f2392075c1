// Small latency-bound kernels of the GLIGEN denoising path (gfx950): timestep embedding,
// grounding-token MLP input (Fourier features + null mixing), small-channel im2col, weight
// packing, fuser gates, the CFG + PLMS update, inpainting blend, uint8 image epilogue.
#include "misc.h"

#include <algorithm>
#include <type_traits>

namespace gl {

static inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = cdiv64(n, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- timestep embedding (reference util.py:160-180): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half)
__global__ void timestep_embed_kernel(const int64_t* __restrict__ t, bf16* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * half; idx += gridDim.x * blockDim.x) {
        const int b = idx / half, i = idx - b * half;
        const float freq = expf(-9.210340371976184f * (float)i / (float)half);
        const float arg = (float)t[b] * freq;
        out[(size_t)b * dim + i] = f2bf(cosf(arg));
        out[(size_t)b * dim + half + i] = f2bf(sinf(arg));
    }
}
int timestep_embed_launch(const int64_t* t, bf16* out, int B, int dim, hipStream_t stream) {
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(grid_for((int64_t)B * dim / 2)), dim3(256), 0, stream, t, out, B, dim);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- grounding tokenizer input
__global__ void posnet_input_kernel(PosNetIn p) {
    const int PD = 16 * p.P;
    const int64_t total = (int64_t)p.rows * p.ld_out;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / p.ld_out);
        const int c = (int)(idx - (int64_t)r * p.ld_out);
        float v = 0.f;
        if (c < p.F) {
            const float m = p.fmask ? p.fmask[r] : p.mask[r];
            const int fr = p.feat_mod > 0 ? r % p.feat_mod : r;
            v = p.feat[(size_t)fr * p.F + c] * m + (1.f - m) * p.null_feat[c];
        } else if (c < p.F + PD) {
            const int e = c - p.F;
            const int k = e / (2 * p.P);
            const int rem = e - k * 2 * p.P;
            const int sc = rem / p.P;
            const int j = rem - sc * p.P;
            const float freq = powf(100.f, (float)k / 8.f);
            const float a = freq * p.pos[(size_t)r * p.P + j];
            const float f = sc ? cosf(a) : sinf(a);
            const float m = p.mask[r];
            v = f * m + (1.f - m) * p.null_pos[e];
        }
        p.out[idx] = f2bf(v);
    }
}
int posnet_input_launch(const PosNetIn& p, hipStream_t stream) {
    if (p.ld_out % 64 != 0 || p.ld_out < p.F + 16 * p.P) return set_error(GL_ERR_ARG, "posnet_input: bad ld_out %d", p.ld_out);
    hipLaunchKernelGGL(posnet_input_kernel, dim3(grid_for((int64_t)p.rows * p.ld_out)), dim3(256), 0, stream, p);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- small-channel im2col
__global__ void im2col_small_kernel(Im2colParams p) {
    const int Cin = p.C0 + p.C1;
    const int HW = p.H * p.W;
    const int64_t total = (int64_t)p.B * HW * p.Kpad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / p.Kpad;
        const int k = (int)(idx - m * p.Kpad);
        float v = 0.f;
        if (k < 9 * Cin) {
            const int tap = k / Cin, c = k - tap * Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int b = (int)(m / HW);
            const int pix = (int)(m - (int64_t)b * HW);
            const int oy = pix / p.W, ox = pix - oy * p.W;
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const int sp = iy * p.W + ix;
                if (c < p.C0) {
                    if (p.pre_w) {
                        float acc = 0.f;
                        for (int j = 0; j < p.C0; ++j)
                            acc += p.pre_w[c * p.C0 + j] * (p.pre_scale * p.x0[((size_t)b * p.C0 + j) * HW + sp]);
                        v = acc + p.pre_b[c];
                    } else {
                        v = p.x0[((size_t)b * p.C0 + c) * HW + sp];
                    }
                } else {
                    v = p.x1[((size_t)b * p.C1 + (c - p.C0)) * HW + sp];
                }
            }
        }
        p.out[idx] = f2bf(v);
    }
}
int im2col_small_launch(const Im2colParams& p, hipStream_t stream) {
    if (p.Kpad % 64 != 0 || p.Kpad < 9 * (p.C0 + p.C1)) return set_error(GL_ERR_ARG, "im2col_small: bad Kpad %d", p.Kpad);
    hipLaunchKernelGGL(im2col_small_kernel, dim3(grid_for((int64_t)p.B * p.H * p.W * p.Kpad, 256, 65535)), dim3(256), 0, stream, p);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- casts / packing
__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = f2bf(s[i]);
}
int cast_f32_bf16_launch(const float* src, bf16* dst, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, n);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void pad_rows_cast_kernel(const float* __restrict__ s, bf16* __restrict__ d, int B, int rows, int rows_pad, int cols) {
    const int64_t total = (int64_t)B * rows_pad * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int64_t rr = i / cols;
        const int r = (int)(rr % rows_pad);
        const int b = (int)(rr / rows_pad);
        d[i] = r < rows ? f2bf(s[((size_t)b * rows + r) * cols + c]) : f2bf(0.f);
    }
}
int pad_rows_cast_launch(const float* src, bf16* dst, int B, int rows, int rows_pad, int cols, hipStream_t stream) {
    hipLaunchKernelGGL(pad_rows_cast_kernel, dim3(grid_for((int64_t)B * rows_pad * cols)), dim3(256), 0, stream, src, dst, B, rows, rows_pad, cols);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void cast_pad_cols_kernel(const float* __restrict__ s, bf16* __restrict__ d, int N, int K, int Kpad) {
    const int64_t total = (int64_t)N * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const int64_t n = i / Kpad;
        d[i] = k < K ? f2bf(s[n * K + k]) : f2bf(0.f);
    }
}
int cast_pad_cols_launch(const float* src, bf16* dst, int N, int K, int Kpad, hipStream_t stream) {
    hipLaunchKernelGGL(cast_pad_cols_kernel, dim3(grid_for((int64_t)N * Kpad)), dim3(256), 0, stream, src, dst, N, K, Kpad);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void pad_rows_bf16_kernel(const bf16* __restrict__ s, bf16* __restrict__ d, int B, int rows, int rows_pad, int cols) {
    const int64_t total = (int64_t)B * rows_pad * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int64_t rr = i / cols;
        const int r = (int)(rr % rows_pad);
        const int b = (int)(rr / rows_pad);
        d[i] = r < rows ? s[((size_t)b * rows + r) * cols + c] : f2bf(0.f);
    }
}
int pad_rows_bf16_launch(const bf16* src, bf16* dst, int B, int rows, int rows_pad, int cols, hipStream_t stream) {
    hipLaunchKernelGGL(pad_rows_bf16_kernel, dim3(grid_for((int64_t)B * rows_pad * cols)), dim3(256), 0, stream, src, dst, B, rows, rows_pad, cols);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// bf16 [B][src_rows][cols] (first `rows` rows of every sample) -> fp32 [B][rows][cols]
__global__ void bf16_rows_to_f32_kernel(const bf16* __restrict__ s, float* __restrict__ d, int B, int rows, int src_rows, int cols) {
    const int64_t total = (int64_t)B * rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int64_t rr = i / cols;
        const int r = (int)(rr % rows);
        const int b = (int)(rr / rows);
        d[i] = bf2f(s[((size_t)b * src_rows + r) * cols + c]);
    }
}
int bf16_rows_to_f32_launch(const bf16* src, float* dst, int B, int rows, int src_rows, int cols, hipStream_t stream) {
    hipLaunchKernelGGL(bf16_rows_to_f32_kernel, dim3(grid_for((int64_t)B * rows * cols)), dim3(256), 0, stream, src, dst, B, rows, src_rows, cols);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void set_f32_kernel(float* d, float v) { d[0] = v; }
int set_f32_launch(float* dst, float v, hipStream_t stream) {
    hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(1), 0, stream, dst, v);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ s, bf16* __restrict__ d, int O, int I, int KH, int KW, int O_pad) {
    const int taps = KH * KW;
    const int64_t total = (int64_t)O_pad * taps * I;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % I);
        const int64_t r = i / I;
        const int tap = (int)(r % taps);
        const int o = (int)(r / taps);
        d[i] = o < O ? f2bf(s[((size_t)o * I + c) * taps + tap]) : f2bf(0.f);
    }
}
int pack_conv_weight_launch(const float* src, bf16* dst, int O, int I, int KH, int KW, int O_pad, hipStream_t stream) {
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for((int64_t)O_pad * KH * KW * I)), dim3(256), 0, stream, src, dst, O, I, KH, KW, O_pad);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// I_src <= I: the source weight has fewer input channels than the packed layout (the SD first conv restored into a
// 4 + k channel GLIGEN first conv); the missing channels get zero weights
__global__ void pack_conv_small_kernel(const float* __restrict__ s, bf16* __restrict__ d, int O, int I, int I_src, int Kpad) {
    const int64_t total = (int64_t)O * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const int o = (int)(i / Kpad);
        float v = 0.f;
        if (k < 9 * I) {
            const int tap = k / I, c = k - tap * I;
            if (c < I_src) v = s[((size_t)o * I_src + c) * 9 + tap];
        }
        d[i] = f2bf(v);
    }
}
int pack_conv_small_launch(const float* src, bf16* dst, int O, int I, int Kpad, hipStream_t stream, int I_src) {
    if (I_src <= 0) I_src = I;
    hipLaunchKernelGGL(pack_conv_small_kernel, dim3(grid_for((int64_t)O * Kpad)), dim3(256), 0, stream, src, dst, O, I, I_src, Kpad);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// packed row p (tile = p>>5, i = p&31): quad q = i>>3, half g = (i>>2)&1, e = i&3
//   -> feature j = tile*16 + e + 4g + 8(q>>1) ; original row = (q&1 ? gate : value) half of proj
__global__ void pack_geglu_kernel(const float* __restrict__ w, const float* __restrict__ b, bf16* __restrict__ wp, float* __restrict__ bp, int C4, int K, int layout) {
    const int64_t total = (int64_t)2 * C4 * K;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K);
        const int p = (int)(idx / K);
        const int tile = p >> 5, i = p & 31;
        int orig;
        if (layout == 1) {  // 16 value rows then the 16 matching gate rows (gemm.hip epi_geglu4_t16)
            orig = (i >> 4) * C4 + tile * 16 + (i & 15);
        } else {            // 32x32-tile interleave (gemm.hip epi_geglu4)
            const int q = i >> 3, g = (i >> 2) & 1, e = i & 3;
            const int j = tile * 16 + e + 4 * g + 8 * (q >> 1);
            orig = (q & 1) * C4 + j;
        }
        wp[idx] = f2bf(w[(size_t)orig * K + k]);
        if (k == 0) bp[p] = b[orig];
    }
}
int pack_geglu_launch(const float* w, const float* b, bf16* wp, float* bp, int C4, int K, int layout, hipStream_t stream) {
    if (C4 % 16 != 0) return set_error(GL_ERR_ARG, "pack_geglu: inner dim %d must be a multiple of 16", C4);
    hipLaunchKernelGGL(pack_geglu_kernel, dim3(grid_for((int64_t)2 * C4 * K)), dim3(256), 0, stream, w, b, wp, bp, C4, K, layout);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- LayerNorm folded into the linear layer that consumes it (gemm.h, Epilogue::ln_stats; reference attention.py:333-338):
//   wf[n][k] = w[n][k] gamma[k]          bf[n] = b[n] + sum_k w[n][k] beta[k]          (fp32; one block per output row)
__global__ void __launch_bounds__(256) ln_fold_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ wf, float* __restrict__ bf, int K) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = w[(size_t)n * K + k];
        wf[(size_t)n * K + k] = v * gamma[k];
        acc += v * beta[k];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bf[n] = (b ? b[n] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}
int ln_fold_launch(const float* w, const float* b, const float* gamma, const float* beta, float* wf, float* bf, int N, int K, hipStream_t stream) {
    hipLaunchKernelGGL(ln_fold_kernel, dim3(N), dim3(256), 0, stream, w, b, gamma, beta, wf, bf, K);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
// csum[n] = sum_k w[n][k] over the PACKED bf16 rows (what the matrix core multiplies), fp32, fixed order
__global__ void __launch_bounds__(256) rowsum_bf16_kernel(const bf16* __restrict__ w, float* __restrict__ csum, int K) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) acc += bf2f(w[(size_t)n * K + k]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) csum[n] = (red[0] + red[1]) + (red[2] + red[3]);
}
int rowsum_bf16_launch(const bf16* w, float* csum, int N, int K, hipStream_t stream) {
    hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(N), dim3(256), 0, stream, w, csum, K);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// scale: one value per fuser (gates 2i and 2i+1 belong to fuser i)
__global__ void gates_kernel(const float* const* alpha_ptrs, const float* scale, float* gates, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gates[i] = scale[i >> 1] * tanhf(alpha_ptrs[i][0]);
}
__global__ void fill_f32_kernel(float* d, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = v;
}
int fill_f32_launch(float* dst, float v, int n, hipStream_t stream) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(cdiv(n, 64)), dim3(64), 0, stream, dst, v, n);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
int gates_launch(const float* const* alpha_ptrs, const float* scale, float* gates, int n, hipStream_t stream) {
    hipLaunchKernelGGL(gates_kernel, dim3(cdiv(n, 64)), dim3(64), 0, stream, alpha_ptrs, scale, gates, n);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- CFG + PLMS
__global__ void plms_update_kernel(PlmsParams p) {
    const float sa = sqrtf(p.a_t), s1a = sqrtf(1.f - p.a_t);
    const float sp = sqrtf(p.a_prev), s1p = sqrtf(1.f - p.a_prev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        float e = p.eps_pair[i];
        if (p.has_uncond) {
            const float eu = p.eps_pair[p.n + i];
            e = eu + p.guidance * (e - eu);
        }
        p.e_t_out[i] = e;
        float ep = p.c0 * e;
        if (p.o1) ep += p.c1 * p.o1[i];
        if (p.o2) ep += p.c2 * p.o2[i];
        if (p.o3) ep += p.c3 * p.o3[i];
        const float x = p.x[i];
        const float pred = (x - s1a * ep) / sa;
        p.x_out[i] = sp * pred + s1p * ep;
    }
}
int plms_update_launch(const PlmsParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(plms_update_kernel, dim3(grid_for(p.n)), dim3(256), 0, stream, p);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// x0 / noise / mask may carry batch 1 (broadcast over the latent batch, as the reference's tensor broadcasting does for the
// single encoded input image of gligen_inference.run) or batch B
__global__ void inpaint_blend_kernel(float* img, const float* x0, const float* noise, const float* mask, float sa, float s1, int B, int C, int HW,
                                     int x0B, int noiseB, int maskB) {
    const int64_t total = (int64_t)B * C * HW;
    const int64_t per = (int64_t)C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const int b = (int)(i / per);
        const int64_t r = i - (int64_t)b * per;
        const float m = mask[(size_t)(maskB == 1 ? 0 : b) * HW + pix];
        const float orig = sa * x0[(x0B == 1 ? 0 : (int64_t)b * per) + r] + s1 * noise[(noiseB == 1 ? 0 : (int64_t)b * per) + r];
        img[i] = orig * m + (1.f - m) * img[i];
    }
}
int inpaint_blend_launch(float* img, const float* x0, const float* noise, const float* mask, float sqrt_ac, float sqrt_1mac, int B, int C, int HW,
                         int x0B, int noiseB, int maskB, hipStream_t stream) {
    hipLaunchKernelGGL(inpaint_blend_kernel, dim3(grid_for((int64_t)B * C * HW)), dim3(256), 0, stream, img, x0, noise, mask, sqrt_ac, sqrt_1mac, B, C, HW,
                       x0B, noiseB, maskB);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// GatedSelfAttentionDense2 residual (reference attention.py:289-295): the projected attention outputs at the sg x sg
// grid of grounding tokens, resized to the sv x sv visual grid by torch's bicubic (F.interpolate mode='bicubic',
// align_corners=False: src = (dst + 0.5) * sg / sv - 0.5, cubic convolution with A = -0.75, taps clamped to the grid),
// then y = x + gate * residual.  tok: [B][row_stride][C] with the grid at rows [grid_off, grid_off + sg*sg);  x, y: [B][sv*sv][C].
__device__ __forceinline__ void cubic_taps(float t, float w[4]) {
    const float A = -0.75f;
    const float a = t + 1.f, b = 1.f - t, c = 2.f - t;
    w[0] = ((A * a - 5.f * A) * a + 8.f * A) * a - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * b - (A + 3.f)) * b * b + 1.f;
    w[3] = ((A * c - 5.f * A) * c + 8.f * A) * c - 4.f * A;
}
__global__ void fuser_resize_kernel(const bf16* __restrict__ tok, const bf16* __restrict__ x, const float* __restrict__ gate,
                                    bf16* __restrict__ y, int B, int row_stride, int grid_off, int sg, int sv, int C) {
    const int C2 = C >> 1;
    const int64_t total = (int64_t)B * sv * sv * C2;
    const float g = *gate;
    const float sc = (float)sg / (float)sv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C2) * 2;
        const int64_t r = i / C2;
        const int pix = (int)(r % (sv * sv));
        const int b = (int)(r / (sv * sv));
        const int oy = pix / sv, ox = pix - oy * sv;
        const float fy = sc * (oy + 0.5f) - 0.5f, fx = sc * (ox + 0.5f) - 0.5f;
        const float fly = floorf(fy), flx = floorf(fx);
        float wy[4], wx[4];
        cubic_taps(fy - fly, wy);
        cubic_taps(fx - flx, wx);
        const int iy = (int)fly, ix = (int)flx;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int yy = min(max(iy - 1 + p, 0), sg - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xx = min(max(ix - 1 + q, 0), sg - 1);
                const uint32_t v = *reinterpret_cast<const uint32_t*>(tok + ((size_t)b * row_stride + grid_off + yy * sg + xx) * C + c);
                const float w = wy[p] * wx[q];
                a0 += w * __uint_as_float(v << 16);
                a1 += w * __uint_as_float(v & 0xffff0000u);
            }
        }
        const size_t o = ((size_t)b * sv * sv + pix) * C + c;
        const uint32_t xv = *reinterpret_cast<const uint32_t*>(x + o);
        union { uint32_t u; bf16 e[2]; } out;
        out.e[0] = f2bf(__uint_as_float(xv << 16) + g * a0);
        out.e[1] = f2bf(__uint_as_float(xv & 0xffff0000u) + g * a1);
        *reinterpret_cast<uint32_t*>(y + o) = out.u;
    }
}
int fuser_resize_launch(const bf16* tok, const bf16* x, const float* gate, bf16* y, int B, int row_stride, int grid_off, int sg, int sv,
                        int C, hipStream_t stream) {
    if (C & 1) return set_error(GL_ERR_ARG, "fuser_resize: odd channel count");
    hipLaunchKernelGGL(fuser_resize_kernel, dim3(grid_for((int64_t)B * sv * sv * (C >> 1))), dim3(256), 0, stream, tok, x, gate, y, B,
                       row_stride, grid_off, sg, sv, C);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void to_uint8_kernel(const float* __restrict__ s, uint8_t* __restrict__ d, int B, int C, int HW) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int pix = (int)(r % HW);
        const int b = (int)(r / HW);
        float v = s[((size_t)b * C + c) * HW + pix];
        v = fminf(fmaxf(v, -1.f), 1.f) * 0.5f + 0.5f;
        d[i] = (uint8_t)(v * 255.f);
    }
}
int to_uint8_launch(const float* src, uint8_t* dst, int B, int C, int HW, hipStream_t stream) {
    hipLaunchKernelGGL(to_uint8_kernel, dim3(grid_for((int64_t)B * C * HW)), dim3(256), 0, stream, src, dst, B, C, HW);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// quant_conv (1x1, 2zc -> 2zc) + DiagonalGaussianDistribution.sample() * scale_factor, per latent pixel
// (reference autoencoder.py:34-38, distributions.py:24-37): moments = Wq h + bq ; mean, logvar = chunk(moments) ;
// z = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale
__global__ void vae_posterior_kernel(const float* __restrict__ h, const float* __restrict__ wq, const float* __restrict__ bq,
                                     const float* __restrict__ noise, float* __restrict__ z, int B, int zc, int HW, float scale) {
    const int64_t total = (int64_t)B * zc * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const int c = (int)((i / HW) % zc);
        const int b = (int)(i / ((int64_t)HW * zc));
        const float* hp = h + (size_t)b * 2 * zc * HW + pix;
        float mean = bq[c], logvar = bq[zc + c];
        for (int k = 0; k < 2 * zc; ++k) {
            const float v = hp[(size_t)k * HW];
            mean = fmaf(wq[c * 2 * zc + k], v, mean);
            logvar = fmaf(wq[(zc + c) * 2 * zc + k], v, logvar);
        }
        logvar = fminf(fmaxf(logvar, -30.f), 20.f);
        z[i] = (mean + expf(0.5f * logvar) * noise[i]) * scale;
    }
}
int vae_posterior_launch(const float* h, const float* wq, const float* bq, const float* noise, float* z, int B, int zc, int HW,
                         float scale, hipStream_t stream) {
    hipLaunchKernelGGL(vae_posterior_kernel, dim3(grid_for((int64_t)B * zc * HW)), dim3(256), 0, stream, h, wq, bq, noise, z, B, zc, HW, scale);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void fill_i64_kernel(int64_t* d, int64_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = v;
}
int fill_i64_launch(int64_t* dst, int64_t v, int n, hipStream_t stream) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(n, 64)), dim3(64), 0, stream, dst, v, n);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

__global__ void zero_kernel(uint4* d, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) d[i] = make_uint4(0, 0, 0, 0);
}
int zero_launch(void* dst, size_t bytes, hipStream_t stream) {
    if (bytes % 16 != 0) return set_error(GL_ERR_ARG, "zero: size %zu not a multiple of 16", bytes);
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, stream, (uint4*)dst, (int64_t)(bytes / 16));
    GL_LAUNCH_CHECK();
    return GL_OK;
}


// ---- GroundingDownsampler of the spatial-map modalities (reference canny/depth/normal/sem/hed_grounding_downsampler.py):
// F.interpolate of the first n channels of the conditioning image to R x R (bicubic, align_corners=False, A = -0.75, taps
// clamped -- torch's upsample_bicubic2d without antialias; or legacy 'nearest': src = floor(dst * in / out)), then
// Conv2d(k=4, s=2, p=1) -> SiLU -> Conv2d(k=4, s=2, p=1). Once per prompt, fp32 throughout, a few MFLOP: plain kernels.
__global__ void resize_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int Cimg, int n, int H, int W, int R, int nearest) {
    const int64_t total = (int64_t)B * n * R * R;
    const float sy = (float)H / (float)R, sx = (float)W / (float)R;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % R);
        const int oy = (int)((i / R) % R);
        const int c = (int)((i / ((int64_t)R * R)) % n);
        const int b = (int)(i / ((int64_t)R * R * n));
        const float* src = img + ((size_t)b * Cimg + c) * H * W;
        float v;
        if (nearest) {
            const int iy = min((int)floorf(oy * sy), H - 1), ix = min((int)floorf(ox * sx), W - 1);
            v = src[(size_t)iy * W + ix];
        } else {
            const float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
            const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            float wy[4], wx[4];
            cubic_taps(fy - y0, wy);
            cubic_taps(fx - x0, wx);
            v = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(y0 - 1 + a, 0), H - 1);
                float row = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xx = min(max(x0 - 1 + e, 0), W - 1);
                    row += wx[e] * src[(size_t)yy * W + xx];
                }
                v += wy[a] * row;
            }
        }
        out[i] = v;
    }
}
__global__ void conv4x4s2_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                     int B, int Cin, int Cout, int Hin, int Win, int silu) {
    const int Ho = Hin / 2, Wo = Win / 2;
    const int64_t total = (int64_t)B * Cout * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const int oy = (int)((i / Wo) % Ho);
        const int o = (int)((i / ((int64_t)Wo * Ho)) % Cout);
        const int b = (int)(i / ((int64_t)Wo * Ho * Cout));
        float acc = bias[o];
        for (int c = 0; c < Cin; ++c) {
            const float* xc = x + ((size_t)b * Cin + c) * Hin * Win;
            const float* wc = w + ((size_t)o * Cin + c) * 16;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int iy = 2 * oy - 1 + ky;
                if (iy < 0 || iy >= Hin) continue;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ix = 2 * ox - 1 + kx;
                    if (ix < 0 || ix >= Win) continue;
                    acc += wc[ky * 4 + kx] * xc[(size_t)iy * Win + ix];
                }
            }
        }
        y[i] = silu ? acc / (1.f + __expf(-acc)) : acc;
    }
}
int resize_f32_launch(const float* img, float* out, int B, int Cimg, int n, int H, int W, int R, int nearest, hipStream_t stream) {
    hipLaunchKernelGGL(resize_f32_kernel, dim3(grid_for((int64_t)B * n * R * R)), dim3(256), 0, stream, img, out, B, Cimg, n, H, W, R, nearest);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
int conv4x4s2_f32_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Hin, int Win, int silu, hipStream_t stream) {
    if ((Hin | Win) & 1) return set_error(GL_ERR_ARG, "conv4x4s2: odd input size %dx%d", Hin, Win);
    hipLaunchKernelGGL(conv4x4s2_f32_kernel, dim3(grid_for((int64_t)B * Cout * (Hin / 2) * (Win / 2))), dim3(256), 0, stream, x, w, bias, y, B, Cin, Cout, Hin, Win,
                       silu);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---------------------------------------------------------------- box calibration (gl_box_calibrate)
// Three numbers that say what THIS box delivers, next to the bench line (same commit: +-10 % images/s between boxes):
//   hbm_copy_gbs     float4 copy of `bytes` (read + written bytes / time), far beyond the 256 MB Infinity Cache
//   lds_dma_tbs      every CU streams a private, L2-resident 1-KiB-per-instruction window into LDS (buffer-less global_load_lds: the
//                    operand-delivery path of every GEMM / conv / attention / row-local kernel here), chip total
//   mfma_bf16_tflops all SIMDs issue independent v_mfma_f32_32x32x16_bf16 back to back (two waves per SIMD): the dense bf16 rate
//                    the chip sustains under load (the 2.5 PFLOP/s roof is 2.4 GHz x 256 CUs; a power-limited clock shows up here)
__global__ void __launch_bounds__(256) calib_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) calib_dma_kernel(const char* __restrict__ src, size_t region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    size_t base = ((size_t)blockIdx.x * 7919 * 8192) % (region - 65536);
    base &= ~(size_t)1023;
    for (int it = 0; it < iters; ++it) {
        const char* p = src + base + (size_t)wave * 8192 + lane * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)j * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + ((wave * 8 + j) & 31) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        base += 32768;
        if (base + 65536 > region) base = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 77 && iters < 0) sink[0] = 1;
}
__global__ void __launch_bounds__(256, 2) calib_mfma_kernel(int iters, float* sink) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = f2bf(0.001f * (threadIdx.x + e)); b[e] = f2bf(0.002f * (threadIdx.x - e)); }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}
// ---- the MFMA ceiling with the clock that explains it (gl_mfma_calibrate)
// The same back-to-back issue loop, by MFMA shape, independent accumulators per wave, waves per SIMD and operand data, for >= 20 ms,
// and the kernel measures ITS OWN shader clock over the loop: s_memtime ticks at the shader clock, s_memrealtime at the constant
// 100 MHz reference, so sclk = 100 MHz x d(memtime) / d(memrealtime) -- the clock the chip sustained while every SIMD issued MFMAs
// (a power-limited clock shows up here and nowhere in a sysfs sample taken before or after the loop).
template <int SHAPE, int NACC>
__global__ void __launch_bounds__(256, 2) calib_mfma2_kernel(int iters, int zero_data, unsigned long long* clk_out, float* sink) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // pseudo-random operands in [-1, 1) (power draw depends on the toggling data: all-zero operands clock higher, MI355X_MICROARCH.md DVFS)
        const unsigned h1 = (threadIdx.x * 2654435761u + e * 40503u + blockIdx.x * 97u) >> 8, h2 = (threadIdx.x * 40503u + e * 2654435761u + 17u) >> 8;
        a[e] = f2bf(zero_data ? 0.f : (float)(h1 & 0xffff) * (1.f / 32768.f) - 1.f);
        b[e] = f2bf(zero_data ? 0.f : (float)(h2 & 0xffff) * (1.f / 32768.f) - 1.f);
    }
    typedef typename std::conditional<SHAPE == 0, f32x16, f32x4>::type acc_t;
    acc_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < (SHAPE == 0 ? 16 : 4); ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(acc[i]));      // every chain has landed before the second clock read
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < (SHAPE == 0 ? 16 : 4); ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x < 8) {     // one wave per XCD (blocks 0..7 land on XCDs 0..7)
        clk_out[2 * blockIdx.x] = c1 - c0;
        clk_out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int mfma_calibrate_launch(int shape, int waves_per_simd, int n_acc, int zero_data, float target_ms, void* scratch, float* out4, hipStream_t stream) {
    if ((shape != 0 && shape != 1) || (waves_per_simd != 1 && waves_per_simd != 2) || (n_acc != 4 && n_acc != 8) || !scratch || !out4)
        return set_error(GL_ERR_ARG, "mfma_calibrate: shape 0|1 (32x32x16 | 16x16x32), waves per SIMD 1|2, accumulators 4|8");
    int dev = 0, cus = 256;
    GL_HIP(hipGetDevice(&dev));
    GL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    GL_HIP(hipEventCreate(&e0));
    GL_HIP(hipEventCreate(&e1));
    struct Guard { hipEvent_t a, b; ~Guard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{e0, e1};
    unsigned long long* clk = reinterpret_cast<unsigned long long*>(scratch);
    float* sink = reinterpret_cast<float*>(clk + 16);
    const dim3 grid(cus * waves_per_simd), block(256);      // 256 threads = one wave on each of the CU's four SIMDs
    auto launch = [&](int iters) {
#define GL_CAL(S_, N_) hipLaunchKernelGGL((calib_mfma2_kernel<S_, N_>), grid, block, 0, stream, iters, zero_data, clk, sink)
        if (shape == 0 && n_acc == 4) GL_CAL(0, 4);
        else if (shape == 0) GL_CAL(0, 8);
        else if (n_acc == 4) GL_CAL(1, 4);
        else GL_CAL(1, 8);
#undef GL_CAL
    };
    // 32 cycles per 32x32x16 (16 per 16x16x32) per SIMD at ~2.4 GHz: iterations for the requested duration, then one warm-up + one timed launch
    const double cyc = shape == 0 ? 32.0 : 16.0;
    int iters = (int)std::max(1.0, (double)target_ms * 1e-3 * 2.4e9 / (cyc * 4 * n_acc * waves_per_simd));
    launch(std::max(1, iters / 20));
    GL_HIP(hipEventRecord(e0, stream));
    launch(iters);
    GL_HIP(hipEventRecord(e1, stream));
    GL_HIP(hipEventSynchronize(e1));
    GL_LAUNCH_CHECK();
    float ms = 0.f;
    GL_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    GL_HIP(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    double sclk = 0;
    int n = 0;
    for (int i = 0; i < 8; ++i)
        if (h[2 * i + 1]) { sclk += 100.0 * (double)h[2 * i] / (double)h[2 * i + 1]; ++n; }
    const double flops_per = shape == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const double total = (double)cus * waves_per_simd * 4 /* waves per block */ * 4.0 * n_acc * iters * flops_per;
    out4[0] = (float)(total / (ms * 1e-3) / 1e12);          // TFLOP/s
    out4[1] = n ? (float)(sclk / n) : 0.f;                   // MHz, measured inside the loop (mean over one wave per XCD)
    out4[2] = ms;
    // issue cycles per MFMA per SIMD at that clock: (SIMD-cycles available) / (MFMAs issued per SIMD)
    const double mfma_per_simd = (double)waves_per_simd * 4.0 * n_acc * iters;
    out4[3] = n ? (float)((sclk / n) * 1e6 * ms * 1e-3 / mfma_per_simd) : 0.f;
    return GL_OK;
}

int box_calibrate_launch(void* scratch, size_t scratch_bytes, float* out3, hipStream_t stream) {
    if (!scratch || scratch_bytes < (size_t(64) << 20) || !out3) return set_error(GL_ERR_ARG, "box_calibrate: needs >= 64 MiB of scratch");
    int dev = 0, cus = 256;
    GL_HIP(hipGetDevice(&dev));
    GL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    GL_HIP(hipEventCreate(&e0));
    GL_HIP(hipEventCreate(&e1));
    struct Guard { hipEvent_t a, b; ~Guard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{e0, e1};
    auto timed = [&](auto&& launch, int reps, float* best_ms) -> int {
        *best_ms = 1e30f;
        launch();   // warm-up
        for (int r = 0; r < reps; ++r) {
            GL_HIP(hipEventRecord(e0, stream));
            launch();
            GL_HIP(hipEventRecord(e1, stream));
            GL_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            GL_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (ms < *best_ms) *best_ms = ms;
        }
        GL_LAUNCH_CHECK();
        return GL_OK;
    };
    float ms = 0.f;
    // 1. copy: half of the scratch to the other half
    const size_t half = (scratch_bytes / 2) & ~(size_t)4095;
    GL_TRY(timed([&] { hipLaunchKernelGGL(calib_copy_kernel, dim3(cus * 8), dim3(256), 0, stream, (const uint4*)scratch, (uint4*)((char*)scratch + half), half / 16); }, 3, &ms));
    out3[0] = (float)(2.0 * (double)half / (ms * 1e-3) / 1e9);
    // 2. LDS-DMA from an L2-resident window
    const size_t region = size_t(24) << 20;
    const int it_dma = 1000;
    GL_TRY(timed([&] { hipLaunchKernelGGL(calib_dma_kernel, dim3(cus), dim3(256), 32 * 1024, stream, (const char*)scratch, region, it_dma, (unsigned*)((char*)scratch + half)); }, 3, &ms));
    out3[1] = (float)((double)cus * 4 * 8 * 1024.0 * it_dma / (ms * 1e-3) / 1e12);
    // 3. MFMA issue rate
    const int it_mfma = 2000;
    GL_TRY(timed([&] { hipLaunchKernelGGL(calib_mfma_kernel, dim3(cus * 2), dim3(256), 0, stream, it_mfma, (float*)((char*)scratch + half)); }, 3, &ms));
    out3[2] = (float)((double)cus * 2 * 4 * 16.0 * it_mfma * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12);
    return GL_OK;
}

}  // namespace gl
