// Training slice for gfx950: forward + backward of one BasicTransformerBlock with a gatedSA fuser under the reference's loss.
//
// Reference: ldm/modules/attention.py:333-338 (BasicTransformerBlock._forward), :236-244 (GatedSelfAttentionDense.forward),
// :127-186 (CrossAttention / SelfAttention), :37-64 (GEGLU / FeedForward); trainer.py:353-371 (run_one_step: mse_loss(model_output,
// noise)), :217-245 (what is trainable: fuser.*, position_net, downsample_net), :375-392 (loss.backward(); opt.step()).
//
// What runs where: every matrix product of the forward and of the backward (dgrad  dX = dY W,  wgrad  dW = dY^T X) goes through
// the bf16 MFMA GEMM of gemm.hip with fp32 output (operands are cast / transposed to bf16 by the kernels below; accumulation and
// everything else is fp32, as the reference trains in fp32). LayerNorm, attention (forward with the row log-sum-exp saved,
// backward with the probabilities RECOMPUTED from q, k and that log-sum-exp -- nothing of size Nq x Nk is ever stored), GEGLU,
// the tanh gates and the loss have their own fp32 kernels. Gradients are produced for the fuser.* parameters only (weight
// gradients of the frozen SD layers are never formed -- trainer.py:217-245 leaves them out of the optimizer), for the block's
// input and for the grounding tokens, so the step chains into position_net and the blocks in front of this one.
// This is the first slice of the training path (DESIGN.md section 9 has the rest of the plan): op-level, held to gradients of the
// reference's own autograd (tests/golden/block_backward_gatedsa.npz), not yet tuned.
#include "train.h"

#include <map>
#include <tuple>
#include <unordered_set>

#include "engine.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace gl {

namespace {

std::string fmt(const char* f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

// The training path's matrix products run on the bf16 MFMA kernels at fp32-like precision: an fp32 operand x is the pair
// (hi = bf16(x), lo = bf16(x - hi)) and a product A W^T is  hi_A hi_W^T + lo_A hi_W^T + hi_A lo_W^T  (three GEMMs with fp32 accumulation;
// the dropped lo lo term is 2^-16 relative). The reference trains in fp32 (trainer.py: no autocast); single-pass bf16 is ~2^-9 per
// operand element, which through the 40-odd layers of the full UNet's backward reached 8e-4 rel-MSE on the deepest gradient
// (developer switch GL_TRAIN_BF16X1=1: one pass).
__global__ void transpose_f32_bf16_kernel(const float* __restrict__ src, int R, int Cc, int ld, bf16* __restrict__ dst, bf16* __restrict__ dst_lo, int Rpad) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < Cc) ? src[(size_t)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < Cc && r < Rpad) {
            const float v = tile[threadIdx.x][i];
            const bf16 h = f2bf(v);
            dst[(size_t)c * Rpad + r] = h;
            if (dst_lo) dst_lo[(size_t)c * Rpad + r] = f2bf(v - bf2f(h));
        }
    }
}
__global__ void split_bf16_kernel(const float* __restrict__ src, size_t n, bf16* __restrict__ hi, bf16* __restrict__ lo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = src[i];
    const bf16 h = f2bf(v);
    hi[i] = h;
    lo[i] = f2bf(v - bf2f(h));
}
// res = src - float(bf16(src))   (the low half of a conv weight, packed like the high half by pack_conv_weight_launch)
__global__ void bf16_residual_kernel(const float* __restrict__ src, size_t n, float* __restrict__ res) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) res[i] = src[i] - bf2f(f2bf(src[i]));
}
// The three passes of a split-precision product as ONE contraction (round 5): x = hi + lo in bf16, and
//     a w^T  ~  a_hi w_hi^T + a_lo w_hi^T + a_hi w_lo^T  =  [a_hi | a_lo | a_hi] [w_hi | w_hi | w_lo]^T
// -- the k dimension tripled, one GEMM launch with one fp32 accumulator instead of three launches, two fp32 temporaries and an add
// pass. side 0 (activation): segments (hi, lo, hi); side 1 (weight): (hi, hi, lo). Rows [R][3 K], columns K' = seg * K + c.
__global__ void cat3_rows_kernel(const float* __restrict__ src, size_t n, int K, int side, bf16* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t r = i / K;
    const int c = (int)(i - r * K);
    const float v = src[i];
    const bf16 h = f2bf(v), l = f2bf(v - bf2f(h));
    bf16* row = dst + r * 3 * (size_t)K;
    row[c] = h;
    row[K + c] = side ? h : l;
    row[2 * (size_t)K + c] = side ? l : h;
}
// [R][Cc] fp32 -> [Cc][3 Rpad] (the transposed operand of dgrad / wgrad), zero behind row R
__global__ void cat3_transposed_kernel(const float* __restrict__ src, int R, int Cc, int ld, int side, bf16* __restrict__ dst, int Rpad) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < Cc) ? src[(size_t)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < Cc && r < Rpad) {
            const float v = tile[threadIdx.x][i];
            const bf16 h = f2bf(v), l = f2bf(v - bf2f(h));
            bf16* row = dst + (size_t)c * 3 * Rpad;
            row[r] = h;
            row[Rpad + r] = side ? h : l;
            row[2 * (size_t)Rpad + r] = side ? l : h;
        }
    }
}
// conv activations [rows][C] fp32 -> [rows][2 C] = (hi | lo): the implicit-GEMM loader reads channels (hi | lo | hi) as a two-source
// concat of this buffer with its own first half
__global__ void cat2_rows_kernel(const float* __restrict__ src, size_t n, int C, bf16* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const float v = src[i];
    const bf16 h = f2bf(v);
    dst[r * 2 * (size_t)C + c] = h;
    dst[r * 2 * (size_t)C + C + c] = f2bf(v - bf2f(h));
}
// conv weight OIHW fp32 [O][I][9] -> [O][3 I][9] fp32 = (w | w | w - bf16(w)) along I: packed to bf16 it is (hi | hi | lo)
__global__ void conv_w_cat3_kernel(const float* __restrict__ w, int O, int I, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)O * I * 9) return;
    const int t = (int)(idx % 9), i = (int)((idx / 9) % I);
    const size_t o = idx / ((size_t)9 * I);
    const float v = w[idx];
    float* row = out + o * 3 * (size_t)I * 9;
    row[(size_t)i * 9 + t] = v;
    row[((size_t)I + i) * 9 + t] = v;
    row[((size_t)2 * I + i) * 9 + t] = v - bf2f(f2bf(v));
}
__global__ void add3_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += a[i] + b[i];
}

// out[c] = sum_r a[r][c] (* b[r][c]): 64 columns x 16 row lanes per workgroup, each row lane sums rows ry, ry + 16, ... in order and the
// sixteen partial sums are combined in a fixed order -- deterministic, and 16 x the parallelism of one thread per column (which was
// 8.5 % of a training iteration: profiles/r4_final/train_kernel_stats.csv is the profile before this change)
__global__ void __launch_bounds__(1024) colsum_kernel(const float* __restrict__ a, const float* __restrict__ b, int R, int Cc, float* __restrict__ out) {
    __shared__ float part[16][64];
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6, c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < Cc)
        for (int r = ry; r < R; r += 16) s += b ? a[(size_t)r * Cc + c] * b[(size_t)r * Cc + c] : a[(size_t)r * Cc + c];
    part[ry][tx] = s;
    __syncthreads();
    if (ry == 0 && c < Cc) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][tx];
        out[c] = t;
    }
}

// LayerNorm over C per row (eps 1e-5, nn.LayerNorm): one wave per row
__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, int R, int Cc,
                              float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd_out) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* xr = x + (size_t)row * Cc;
    float s = 0.f, ss = 0.f;
    for (int c = lane; c < Cc; c += 64) { const float v = xr[c]; s += v; ss += v * v; }
    s = wave_sum(s); ss = wave_sum(ss);
    const float mean = s / Cc, var = fmaxf(ss / Cc - mean * mean, 0.f), rstd = rsqrtf(var + 1e-5f);
    for (int c = lane; c < Cc; c += 64) {
        const float h = (xr[c] - mean) * rstd;
        xhat[(size_t)row * Cc + c] = h;
        y[(size_t)row * Cc + c] = h * g[c] + b[c];
    }
    if (lane == 0) rstd_out[row] = rstd;
}
// dx (+)= rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma
__global__ void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd, const float* __restrict__ gam,
                              int R, int Cc, float* __restrict__ dx, int accumulate) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* dr = dy + (size_t)row * Cc;
    const float* hr = xhat + (size_t)row * Cc;
    float m1 = 0.f, m2 = 0.f;
    for (int c = lane; c < Cc; c += 64) { const float gv = dr[c] * gam[c]; m1 += gv; m2 += gv * hr[c]; }
    m1 = wave_sum(m1) / Cc; m2 = wave_sum(m2) / Cc;
    const float rs = rstd[row];
    for (int c = lane; c < Cc; c += 64) {
        const float v = rs * (dr[c] * gam[c] - m1 - hr[c] * m2);
        float* o = dx + (size_t)row * Cc + c;
        *o = accumulate ? *o + v : v;
    }
}

// ---- attention on row-major q [B][Nq][H d], k / v [B][Nk][H d]. A (batch, head, query) -- or key, in attn_bwd_kv -- is owned by LPQ
// adjacent lanes, each holding a DC-wide chunk of the head dimension (d = DC * LPQ; DC <= 40 keeps a lane's q / accumulator chunks in
// registers: with one lane per query d = 160 spilled 320 floats per lane and was 3 ms per launch). Dot products over d are chunk sums
// combined across the LPQ lanes by xor shuffles.
template <int LPQ>
__device__ __forceinline__ float lpq_sum(float v) {
    if constexpr (LPQ >= 2) v += __shfl_xor(v, 1, 64);
    if constexpr (LPQ >= 4) v += __shfl_xor(v, 2, 64);
    return v;
}
template <int DC, int LPQ>
__global__ void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int H, int Nq, int Nk, float scale,
                                float* __restrict__ o, float* __restrict__ lse) {
    constexpr int D = DC * LPQ;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, gi = blockIdx.x * blockDim.x + threadIdx.x, part = gi % LPQ;
    const int i = min(gi / LPQ, Nq - 1);                  // (lanes past the end recompute the last query and do not store: the shuffles need them)
    const bool live = gi / LPQ < Nq;
    const int ld = H * D, c0 = h * D + part * DC;
    float qi[DC], acc[DC];
    const float* qp = q + ((size_t)b * Nq + i) * ld + c0;
#pragma unroll
    for (int c = 0; c < DC; ++c) { qi[c] = qp[c] * scale; acc[c] = 0.f; }
    float m = -1e30f, l = 0.f;
    for (int j = 0; j < Nk; ++j) {
        const float* kp = k + ((size_t)b * Nk + j) * ld + c0;
        const float* vp = v + ((size_t)b * Nk + j) * ld + c0;
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c) sc = fmaf(qi[c], kp[c], sc);
        sc = lpq_sum<LPQ>(sc);
        const float mn = fmaxf(m, sc), corr = __expf(m - mn), p = __expf(sc - mn);
        l = l * corr + p;
#pragma unroll
        for (int c = 0; c < DC; ++c) acc[c] = fmaf(acc[c], corr, p * vp[c]);
        m = mn;
    }
    if (!live) return;
    float* op = o + ((size_t)b * Nq + i) * ld + c0;
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < DC; ++c) op[c] = acc[c] * inv;
    if (part == 0) lse[(size_t)bh * Nq + i] = m + __logf(l);
}
// dq and the row terms delta_i = do_i . o_i; the probabilities are recomputed from q, k and the saved log-sum-exp
template <int DC, int LPQ>
__global__ void attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ o,
                                  const float* __restrict__ dout, const float* __restrict__ lse, int H, int Nq, int Nk, float scale,
                                  float* __restrict__ dq, float* __restrict__ delta) {
    constexpr int D = DC * LPQ;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, gi = blockIdx.x * blockDim.x + threadIdx.x, part = gi % LPQ;
    const int i = min(gi / LPQ, Nq - 1);
    const bool live = gi / LPQ < Nq;
    const int ld = H * D;
    const size_t off = ((size_t)b * Nq + i) * ld + h * D + part * DC;
    float qi[DC], di[DC], acc[DC];
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < DC; ++c) { qi[c] = q[off + c] * scale; di[c] = dout[off + c]; dl = fmaf(di[c], o[off + c], dl); acc[c] = 0.f; }
    dl = lpq_sum<LPQ>(dl);
    const float L = lse[(size_t)bh * Nq + i];
    for (int j = 0; j < Nk; ++j) {
        const float* kp = k + ((size_t)b * Nk + j) * ld + h * D + part * DC;
        const float* vp = v + ((size_t)b * Nk + j) * ld + h * D + part * DC;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c) { sc = fmaf(qi[c], kp[c], sc); dp = fmaf(di[c], vp[c], dp); }
        sc = lpq_sum<LPQ>(sc);
        dp = lpq_sum<LPQ>(dp);
        const float ds = __expf(sc - L) * (dp - dl);
#pragma unroll
        for (int c = 0; c < DC; ++c) acc[c] = fmaf(ds, kp[c], acc[c]);
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < DC; ++c) dq[off + c] = acc[c] * scale;
    if (part == 0) delta[(size_t)bh * Nq + i] = dl;
}
template <int DC, int LPQ>
__global__ void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ dout,
                                   const float* __restrict__ lse, const float* __restrict__ delta, int H, int Nq, int Nk, float scale,
                                   float* __restrict__ dk, float* __restrict__ dv) {
    constexpr int D = DC * LPQ;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, gj = blockIdx.x * blockDim.x + threadIdx.x, part = gj % LPQ;
    const int j = min(gj / LPQ, Nk - 1);
    const bool live = gj / LPQ < Nk;
    const int ld = H * D;
    const size_t off = ((size_t)b * Nk + j) * ld + h * D + part * DC;
    float kj[DC], vj[DC], ak[DC], av[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) { kj[c] = k[off + c]; vj[c] = v[off + c]; ak[c] = 0.f; av[c] = 0.f; }
    for (int i = 0; i < Nq; ++i) {
        const float* qp = q + ((size_t)b * Nq + i) * ld + h * D + part * DC;
        const float* dp_ = dout + ((size_t)b * Nq + i) * ld + h * D + part * DC;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c) { sc = fmaf(qp[c], kj[c], sc); dp = fmaf(dp_[c], vj[c], dp); }
        sc = lpq_sum<LPQ>(sc);
        dp = lpq_sum<LPQ>(dp);
        const float p = __expf(sc * scale - lse[(size_t)bh * Nq + i]);
        const float ds = p * (dp - delta[(size_t)bh * Nq + i]);
#pragma unroll
        for (int c = 0; c < DC; ++c) { av[c] = fmaf(p, dp_[c], av[c]); ak[c] = fmaf(ds, qp[c], ak[c]); }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < DC; ++c) { dk[off + c] = ak[c] * scale; dv[off + c] = av[c]; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same three attention passes on the matrix cores (round 5): flash-style, one wave per 32 queries (forward, dq) or 32 keys (dk, dv),
// every product as THREE bf16 MFMA passes over (hi, lo) splits of its fp32 operands (x = hi + lo; hi.hi + lo.hi + hi.lo with fp32
// accumulation, the same scheme as the training GEMMs: 2^-16 relative), softmax arithmetic in fp32 registers. A prep pass writes every
// operand once per call in the two forms the MFMA fragments read straight from memory (no LDS in these kernels: a wave's fragment is
// one 16-byte load per lane):
//   R  [bh][N_pad][DP]   token-major rows, head dim zero-padded to a multiple of 16        (operand with k-slots over the head dim)
//   T  [bh][DPO][N_pad]  transposed, tokens permuted inside groups of 16 as [0-3, 8-11, 4-7, 12-15] (k-slots over tokens: the B
//                        operand of those products comes straight out of 32x32 accumulator registers, attention.hip's S^T -> P^T trick)
// each as hi and lo bf16. v_mfma_f32_32x32x16_bf16 layouts: A lane L = row L & 31, k-slots 8 (L >> 5) .. + 7; B lane L = column L & 31,
// same k-slots; D lane L = column L & 31, register 4 j + e = row 8 j + 4 (L >> 5) + e.
// Head dims 32 / 40 / 64 / 80 (d = 160 -- the 16 x 16 level, 256 tokens -- stays on the VALU kernels above: 5 % of the attention time).
struct APrep {
    const bf16* r_hi; const bf16* r_lo; const bf16* t_hi; const bf16* t_lo;
    int Npad;
};
__device__ __forceinline__ int perm16_tok(int t) {
    const int g = (t >> 2) & 3;
    return (t & ~15) | ((((g & 1) << 1) | (g >> 1)) << 2) | (t & 3);
}
// x [B][N][H d] fp32 (times mul) -> R / T hi / lo for every (b, h); one thread per (bh, token, column)
__global__ void attn_prep_kernel(const float* __restrict__ x, int N, int H, int d, float mul, int Npad, int DP, int DPO, bf16* __restrict__ r_hi,
                                 bf16* __restrict__ r_lo, bf16* __restrict__ t_hi, bf16* __restrict__ t_lo, size_t total) {
    const int CW = DP > DPO ? DP : DPO;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CW);
        const int t = (int)((idx / CW) % Npad);
        const int bh = (int)(idx / ((size_t)CW * Npad));
        const int b = bh / H, h = bh - b * H;
        const float v = (t < N && c < d) ? x[(((size_t)b * N + t) * H + h) * d + c] * mul : 0.f;
        const bf16 hi = f2bf(v);
        const bf16 lo = f2bf(v - bf2f(hi));
        if (c < DP) {
            const size_t o = ((size_t)bh * Npad + t) * DP + c;
            r_hi[o] = hi; r_lo[o] = lo;
        }
        if (c < DPO) {
            const size_t o = ((size_t)bh * DPO + c) * Npad + perm16_tok(t);
            t_hi[o] = hi; t_lo[o] = lo;
        }
    }
}
// lse_pad [bh][Npad] (+1e30 behind the last query: its probabilities vanish), delta_pad [bh][Npad] = do_i . o_i (0 behind the last query)
__global__ void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout, const float* __restrict__ lse, int N, int H, int d, int Npad,
                                  float* __restrict__ lse_pad, float* __restrict__ delta_pad, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % Npad);
        const int bh = (int)(idx / Npad);
        const int b = bh / H, h = bh - b * H;
        float dl = 0.f, L = 1e30f;
        if (t < N) {
            const size_t off = (((size_t)b * N + t) * H + h) * d;
            for (int c = 0; c < d; ++c) dl = fmaf(dout[off + c], o[off + c], dl);
            L = lse[(size_t)bh * N + t];
        }
        lse_pad[idx] = L;
        delta_pad[idx] = dl;
    }
}
__device__ __forceinline__ float tr_max_xor32(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
// acc += (ah + al) (bh + bl) without the lo.lo term
__device__ __forceinline__ void mfma3(f32x16& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
}
// accumulator registers 8 kk .. 8 kk + 7 of a 32 x 32 tile -> (hi, lo) B operands of k-step kk (k-slots = the tile's rows, in the
// permuted order of the T layouts)
__device__ __forceinline__ void split8(const f32x16& v, int kk, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = kk ? v[8 + e] : v[e];
        hi[e] = f2bf(x);
        lo[e] = f2bf(x - bf2f(hi[e]));
    }
}

// forward: o = softmax(q' k^T) v (q' = q scale, folded by the prep pass), lse = log sum exp. grid (Nq_pad / 32, B H), one wave
template <int DP, int DPO>
__global__ void __launch_bounds__(64) attn_mfma_fwd_kernel(APrep Q, APrep K, APrep V, int H, int d, int Nq, int Nk, float* __restrict__ o, float* __restrict__ lse) {
    constexpr int KS = DP / 16, DT = DPO / 32;
    const int lane = threadIdx.x, lrow = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, q0 = blockIdx.x * 32;
    bf16x8 qh[KS], ql[KS];
    {
        const size_t off = ((size_t)bh * Q.Npad + q0 + lrow) * DP + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { qh[ks] = ld8(Q.r_hi + off + 16 * ks); ql[ks] = ld8(Q.r_lo + off + 16 * ks); }
    }
    f32x16 ot[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const bf16* krh = K.r_hi + ((size_t)bh * K.Npad + lrow) * DP + 8 * half;
    const bf16* krl = K.r_lo + ((size_t)bh * K.Npad + lrow) * DP + 8 * half;
    const bf16* vth = V.t_hi + ((size_t)bh * DPO + lrow) * V.Npad + 8 * half;
    const bf16* vtl = V.t_lo + ((size_t)bh * DPO + lrow) * V.Npad + 8 * half;
    for (int k0 = 0; k0 < Nk; k0 += 32) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) mfma3(s, ld8(krh + (size_t)k0 * DP + 16 * ks), ld8(krl + (size_t)k0 * DP + 16 * ks), qh[ks], ql[ks]);
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (k0 + 8 * (r >> 2) + 4 * half + (r & 3) >= Nk) s[r] = -1e30f;
            mx = fmaxf(mx, s[r]);
        }
        mx = tr_max_xor32(mx);
        const float mn = fmaxf(m, mx), alpha = __expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mn); ps += s[r]; }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ph, pl;
            split8(s, kk, ph, pl);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const size_t off = (size_t)32 * i * V.Npad + k0 + 16 * kk;
                mfma3(ot[i], ld8(vth + off), ld8(vtl + off), ph, pl);
            }
        }
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const int q = q0 + lrow;
    if (q >= Nq) return;
    const float inv = 1.f / lt;
    float* op = o + (((size_t)b * Nq + q) * H + h) * d;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dd = 32 * i + 8 * j + 4 * half;
            if (dd < d) *reinterpret_cast<float4*>(op + dd) = make_float4(ot[i][4 * j] * inv, ot[i][4 * j + 1] * inv, ot[i][4 * j + 2] * inv, ot[i][4 * j + 3] * inv);
        }
    if (half == 0) lse[(size_t)bh * Nq + q] = m + __logf(lt);
}

// dq = scale (dS K), dS = P (dP - delta), P = exp(q' k^T - lse), dP = do v^T. grid (Nq_pad / 32, B H), one wave
template <int DP, int DPO>
__global__ void __launch_bounds__(64) attn_mfma_bwd_q_kernel(APrep Q, APrep K, APrep V, APrep DO, const float* __restrict__ lse_pad,
                                                              const float* __restrict__ delta_pad, int H, int d, int Nq, int Nk, float scale,
                                                              float* __restrict__ dq) {
    constexpr int KS = DP / 16, DT = DPO / 32;
    const int lane = threadIdx.x, lrow = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, q0 = blockIdx.x * 32;
    bf16x8 qh[KS], ql[KS], gh[KS], gl_[KS];
    {
        const size_t off = ((size_t)bh * Q.Npad + q0 + lrow) * DP + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[ks] = ld8(Q.r_hi + off + 16 * ks); ql[ks] = ld8(Q.r_lo + off + 16 * ks);
            gh[ks] = ld8(DO.r_hi + off + 16 * ks); gl_[ks] = ld8(DO.r_lo + off + 16 * ks);
        }
    }
    const float L = lse_pad[(size_t)bh * Q.Npad + q0 + lrow], dl = delta_pad[(size_t)bh * Q.Npad + q0 + lrow];
    f32x16 acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const size_t rbase = ((size_t)bh * K.Npad + lrow) * DP + 8 * half;
    const size_t tbase = ((size_t)bh * DPO + lrow) * K.Npad + 8 * half;
    for (int k0 = 0; k0 < Nk; k0 += 32) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t off = rbase + (size_t)k0 * DP + 16 * ks;
            mfma3(s, ld8(K.r_hi + off), ld8(K.r_lo + off), qh[ks], ql[ks]);
            mfma3(dp, ld8(V.r_hi + off), ld8(V.r_lo + off), gh[ks], gl_[ks]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = k0 + 8 * (r >> 2) + 4 * half + (r & 3) < Nk;
            s[r] = ok ? __expf(s[r] - L) * (dp[r] - dl) : 0.f;       // dS
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 dh, dlo;
            split8(s, kk, dh, dlo);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const size_t off = tbase + (size_t)32 * i * K.Npad + k0 + 16 * kk;
                mfma3(acc[i], ld8(K.t_hi + off), ld8(K.t_lo + off), dh, dlo);
            }
        }
    }
    const int q = q0 + lrow;
    if (q >= Nq) return;
    float* op = dq + (((size_t)b * Nq + q) * H + h) * d;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dd = 32 * i + 8 * j + 4 * half;
            if (dd < d) *reinterpret_cast<float4*>(op + dd) = make_float4(acc[i][4 * j] * scale, acc[i][4 * j + 1] * scale, acc[i][4 * j + 2] * scale, acc[i][4 * j + 3] * scale);
        }
}

// dk = dS^T q' (= scale dS^T q), dv = P^T do. grid (Nk_pad / 32, B H), one wave owns 32 keys and walks the queries
template <int DP, int DPO>
__global__ void __launch_bounds__(64) attn_mfma_bwd_kv_kernel(APrep Q, APrep K, APrep V, APrep DO, const float* __restrict__ lse_pad,
                                                               const float* __restrict__ delta_pad, int H, int d, int Nq, int Nk, float* __restrict__ dk,
                                                               float* __restrict__ dv) {
    constexpr int KS = DP / 16, DT = DPO / 32;
    const int lane = threadIdx.x, lrow = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, kb0 = blockIdx.x * 32;
    bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
    {
        const size_t off = ((size_t)bh * K.Npad + kb0 + lrow) * DP + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kh[ks] = ld8(K.r_hi + off + 16 * ks); kl[ks] = ld8(K.r_lo + off + 16 * ks);
            vh[ks] = ld8(V.r_hi + off + 16 * ks); vl[ks] = ld8(V.r_lo + off + 16 * ks);
        }
    }
    f32x16 ak[DT], av[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[i][r] = 0.f; av[i][r] = 0.f; }
    const size_t rbase = ((size_t)bh * Q.Npad + lrow) * DP + 8 * half;
    const size_t tbase = ((size_t)bh * DPO + lrow) * Q.Npad + 8 * half;
    const float* Lp = lse_pad + (size_t)bh * Q.Npad + 4 * half;
    const float* Dp = delta_pad + (size_t)bh * Q.Npad + 4 * half;
    const int nq_pad32 = (Nq + 31) & ~31;
    for (int q0 = 0; q0 < nq_pad32; q0 += 32) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t off = rbase + (size_t)q0 * DP + 16 * ks;
            mfma3(s, ld8(Q.r_hi + off), ld8(Q.r_lo + off), kh[ks], kl[ks]);      // rows = queries, columns = this wave's keys
            mfma3(dp, ld8(DO.r_hi + off), ld8(DO.r_lo + off), vh[ks], vl[ks]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 L4 = *reinterpret_cast<const float4*>(Lp + q0 + 8 * j);
            const float4 D4 = *reinterpret_cast<const float4*>(Dp + q0 + 8 * j);
            const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __expf(s[4 * j + e] - Lr[e]);        // (queries behind the last one carry lse = 1e30: p = 0)
                s[4 * j + e] = p;
                dp[4 * j + e] = p * (dp[4 * j + e] - Dr[e]);          // dS
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ph, pl, dh, dlo;
            split8(s, kk, ph, pl);
            split8(dp, kk, dh, dlo);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const size_t off = tbase + (size_t)32 * i * Q.Npad + q0 + 16 * kk;
                mfma3(av[i], ld8(DO.t_hi + off), ld8(DO.t_lo + off), ph, pl);
                mfma3(ak[i], ld8(Q.t_hi + off), ld8(Q.t_lo + off), dh, dlo);
            }
        }
    }
    const int key = kb0 + lrow;
    if (key >= Nk) return;
    const size_t ob = (((size_t)b * Nk + key) * H + h) * d;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dd = 32 * i + 8 * j + 4 * half;
            if (dd < d) {
                *reinterpret_cast<float4*>(dk + ob + dd) = make_float4(ak[i][4 * j], ak[i][4 * j + 1], ak[i][4 * j + 2], ak[i][4 * j + 3]);
                *reinterpret_cast<float4*>(dv + ob + dd) = make_float4(av[i][4 * j], av[i][4 * j + 1], av[i][4 * j + 2], av[i][4 * j + 3]);
            }
        }
}

// GEGLU (attention.py:37-44): h = val * gelu(gate), u = [val | gate] of width 2 I (erf GELU, F.gelu default)
__global__ void geglu_fwd_kernel(const float* __restrict__ u, int R, int I, float* __restrict__ h) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * I) return;
    const size_t r = idx / I, c = idx % I;
    const float val = u[r * 2 * I + c], g = u[r * 2 * I + I + c];
    h[idx] = val * 0.5f * g * (1.f + erff(g * 0.70710678118654752440f));
}
__global__ void geglu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ u, int R, int I, float* __restrict__ du) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * I) return;
    const size_t r = idx / I, c = idx % I;
    const float val = u[r * 2 * I + c], g = u[r * 2 * I + I + c], d = dh[idx];
    const float Phi = 0.5f * (1.f + erff(g * 0.70710678118654752440f));
    const float phi = 0.3989422804014327f * __expf(-0.5f * g * g);
    du[r * 2 * I + c] = d * g * Phi;
    du[r * 2 * I + I + c] = d * val * (Phi + g * phi);
}

// out = a + gate * b, gate = scale * tanh(*alpha) (alpha null: gate 1)
__global__ void gated_add_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ alpha, float scale, size_t n,
                                 float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = alpha ? scale * tanhf(*alpha) : 1.f;
    out[i] = a[i] + g * b[i];
}
// out = gate * a
__global__ void gated_scale_kernel(const float* __restrict__ a, const float* __restrict__ alpha, float scale, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = scale * tanhf(*alpha) * a[i];
}
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
// out[0] = coef(alpha) * sum_i a[i] b[i]: one block, fixed order.  mode 0: coef = scale (1 - tanh^2 alpha) (the gate's derivative);
// mode 1: coef = 1 / n and out = mean((a - b)^2) (the loss)
__global__ void __launch_bounds__(1024) dot_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, const float* __restrict__ alpha,
                                                          float scale, int mode, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) s += mode ? (a[i] - b[i]) * (a[i] - b[i]) : a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        if (mode) out[0] = t / (float)n;
        else { const float th = tanhf(*alpha); out[0] = scale * (1.f - th * th) * t; }
    }
}
// the same reduction for long vectors (the tanh-gate gradients and the loss at the 64 x 64 level are 10^7-element dots: one block took
// 1.2 ms each, 4 % of an iteration): fixed chunks -> partial sums (one block per chunk) -> one wave adds the partials in order
__global__ void __launch_bounds__(1024) dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, size_t chunk, int mode,
                                                           float* __restrict__ partial) {
    __shared__ float red[16];
    const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    float s = 0.f;
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) s += mode ? (a[i] - b[i]) * (a[i] - b[i]) : a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}
__global__ void dot_final_kernel(const float* __restrict__ partial, int nb, size_t n, const float* __restrict__ alpha, float scale, int mode, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    float t = 0.f;
    for (int i = 0; i < nb; ++i) t += partial[i];
    if (mode) out[0] = t / (float)n;
    else { const float th = tanhf(*alpha); out[0] = scale * (1.f - th * th) * t; }
}
// dy = 2 (y - t) / n   (d mse_loss / dy)
__global__ void mse_grad_kernel(const float* __restrict__ y, const float* __restrict__ t, size_t n, float* __restrict__ dy) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dy[i] = 2.f * (y[i] - t[i]) / (float)n;
}

// torch.optim.AdamW (trainer.py:245, one step of opt.step()): decoupled weight decay, bias-corrected moments, in place
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                             float b1, float omb1, float b2, float omb2, float eps, float decay, float step_size, float bc2_sqrt) {
    // torch.optim.AdamW (_single_tensor_adamw): every scalar below is formed in double on the host, as torch forms them from Python
    // floats, and rounded to fp32 once: decay = 1 - lr wd, omb = 1 - beta, step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    (void)lr;
    const float gi = g[i];
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] * decay - step_size * (mi / denom);
}


// ---- ResBlock pieces (openaimodel.py:154-232): GroupNorm32 + SiLU over pixel rows [B][HW][C] (fp32), and their backward.
// One workgroup per (group, sample); the group's HW x cpg slab is read twice (statistics, then apply), sums in a fixed order.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// xhat = (x - mean) * rstd per (sample, group); a = silu(xhat * gamma + beta)   (util.py:223-226 GroupNorm32, eps 1e-5; nn.SiLU)
__global__ void __launch_bounds__(256) gn_silu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gam, const float* __restrict__ bet, int HW, int Cc,
                                                          float* __restrict__ xhat, float* __restrict__ rstd_out, float* __restrict__ a, int silu, float eps) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y, cpg = Cc / 32, n = HW * cpg;
    const float* xb = x + (size_t)b * HW * Cc + g * cpg;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = xb[(size_t)(i / cpg) * Cc + i % cpg]; s += v; q += v * v; }
    s = block_sum_256(s, red);
    q = block_sum_256(q, red);
    const float mean = s / n, rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = g * cpg + i % cpg;
        const size_t o = ((size_t)b * HW + i / cpg) * Cc + c;
        const float h = (x[o] - mean) * rstd, u = h * gam[c] + bet[c];
        xhat[o] = h;
        a[o] = silu ? u / (1.f + __expf(-u)) : u;
    }
    if (threadIdx.x == 0) rstd_out[b * 32 + g] = rstd;
}
// da (gradient w.r.t. a = silu(u), u = xhat gamma + beta)  ->  dx of the GroupNorm's input:
//   du = da sigma(u) (1 + u (1 - sigma(u))),  t = du gamma,  dx = rstd (t - mean_g(t) - xhat mean_g(t xhat))      (dx (+)= when accumulate)
__global__ void __launch_bounds__(256) gn_silu_bwd_kernel(const float* __restrict__ da, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                          const float* __restrict__ gam, const float* __restrict__ bet, int HW, int Cc,
                                                          float* __restrict__ dx, int accumulate, int silu) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y, cpg = Cc / 32, n = HW * cpg;
    auto t_of = [&](int i, float& h) {
        const int c = g * cpg + i % cpg;
        const size_t o = ((size_t)b * HW + i / cpg) * Cc + c;
        h = xhat[o];
        if (!silu) return da[o] * gam[c];
        const float u = h * gam[c] + bet[c], sg = 1.f / (1.f + __expf(-u));
        return da[o] * sg * (1.f + u * (1.f - sg)) * gam[c];
    };
    float m1 = 0.f, m2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { float h; const float t = t_of(i, h); m1 += t; m2 += t * h; }
    m1 = block_sum_256(m1, red) / n;
    m2 = block_sum_256(m2, red) / n;
    const float rs = rstd[b * 32 + g];
    for (int i = threadIdx.x; i < n; i += 256) {
        float h;
        const float t = t_of(i, h);
        const size_t o = ((size_t)b * HW + i / cpg) * Cc + g * cpg + i % cpg;
        const float v = rs * (t - m1 - h * m2);
        dx[o] = accumulate ? dx[o] + v : v;
    }
}
// out[b][p][c] = a[b][p][c] + e[b][c]   (h + emb_out[..., None, None], openaimodel.py:230)
__global__ void add_per_sample_kernel(const float* __restrict__ a, const float* __restrict__ e, int HW, int Cc, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a[i] + e[(i / ((size_t)HW * Cc)) * Cc + i % Cc];
}
__global__ void silu_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] / (1.f + __expf(-x[i]));
}
// The weights of the data-gradient conv: dX = conv3x3(dY, W'), W'[i][o][ky][kx] = W[o][i][2 - ky][2 - kx] (stride 1, pad 1: the
// transposed convolution of a 3x3 / pad 1 conv is a 3x3 / pad 1 conv with the filter flipped and the channel roles exchanged)
__global__ void conv_dgrad_weight_kernel(const float* __restrict__ w, int O, int I, float* __restrict__ wt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * I * 9) return;
    const int t = idx % 9, i = (idx / 9) % I, o = idx / (9 * I);
    wt[((size_t)i * O + o) * 9 + (8 - t)] = w[idx];
}

// Downsample backward: z [B][H][W][C] = dy [B][H/2][W/2][C] at the even positions, zeros elsewhere (the transposed stride-2 conv is
// the stride-1 conv of this with the flipped filter). Upsample backward: dx [B][H][W][C] = 2 x 2 block sums of du [B][2H][2W][C]
// (the adjoint of nearest-neighbour doubling).
__global__ void zero_insert2_kernel(const float* __restrict__ dy, int H, int W, int Cc, size_t n, float* __restrict__ z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % Cc), xw = (int)((i / Cc) % W), yh = (int)((i / ((size_t)Cc * W)) % H);
    const size_t b = i / ((size_t)Cc * W * H);
    z[i] = ((xw | yh) & 1) ? 0.f : dy[((b * (H / 2) + yh / 2) * (W / 2) + xw / 2) * Cc + c];
}
__global__ void sum2x2_kernel(const float* __restrict__ du, int H, int W, int Cc, size_t n, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % Cc), xw = (int)((i / Cc) % W), yh = (int)((i / ((size_t)Cc * W)) % H);
    const size_t b = i / ((size_t)Cc * W * H);
    const float* p = du + ((b * 2 * H + 2 * yh) * 2 * W + 2 * xw) * Cc + c;
    dx[i] = (p[0] + p[Cc]) + (p[(size_t)2 * W * Cc] + p[(size_t)2 * W * Cc + Cc]);
}

}  // namespace

// operand copies of frozen parameters, kept across training steps (train.h)
struct TrainWeightCache {
    struct Key {
        const void* p; int kind, a, b, c;
        bool operator<(const Key& o) const { return std::tie(p, kind, a, b, c) < std::tie(o.p, o.kind, o.a, o.b, o.c); }
    };
    std::map<Key, void*> m;
    size_t bytes = 0;
};
TrainWeightCache* train_cache_create() { return new TrainWeightCache(); }
void train_cache_destroy(TrainWeightCache* c) {
    if (!c) return;
    for (auto& kv : c->m) (void)hipFree(kv.second);
    delete c;
}
size_t train_cache_bytes(const TrainWeightCache* c) { return c ? c->bytes : 0; }

namespace {

struct Ctx {
    Arena& ar;
    float* ws;
    size_t ws_bytes;
    hipStream_t s;
    TrainWeightCache* wc = nullptr;                            // null: every operand copy is built per product in the arena
    const std::unordered_set<const void*>* frozen = nullptr;   // parameter tensors the caller does not update (no gradient asked for)
    // a bf16 operand copy of weight W: from the cache when W is frozen and a cache is attached (built on first use, on this stream), else
    // from the arena (released with the product's mark as before)
    template <class F>
    bf16* weight_operand(const float* W, int kind, int a, int b, int c, size_t n_elems, F&& build) const {
        if (wc && frozen && frozen->count(W)) {
            const TrainWeightCache::Key key{W, kind, a, b, c};
            auto it = wc->m.find(key);
            if (it != wc->m.end()) return reinterpret_cast<bf16*>(it->second);
            void* p = nullptr;
            hip(hipMalloc(&p, n_elems * sizeof(bf16)), "hipMalloc (training weight cache)");
            const size_t mk = ar.mark();
            build(reinterpret_cast<bf16*>(p));
            ar.release(mk);               // (fp32 temporaries of the build; stream order keeps their reuse safe)
            wc->m.emplace(key, p);
            wc->bytes += n_elems * sizeof(bf16);
            return reinterpret_cast<bf16*>(p);
        }
        bf16* d = ar.get<bf16>(n_elems);
        build(d);
        return d;
    }
    void ck(int rc) const { if (rc != GL_OK) throw GlError(rc, gl::last_error()); }
    void hip(hipError_t e, const char* what) const { if (e != hipSuccess) throw GlError(GL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
    float* f32(size_t n) const { return ar.get<float>(n); }
    static dim3 g1(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

    static bool split_precision() {
        static const bool one_pass = dev_env("GL_TRAIN_BF16X1") && atoi(dev_env("GL_TRAIN_BF16X1")) != 0;
        return !one_pass;
    }
    struct Split { bf16* hi; bf16* lo; };      // lo null: single-pass bf16
    Split to_bf16(const float* src, size_t n) const {
        Split d{ar.get<bf16>(n), nullptr};
        if (split_precision()) {
            d.lo = ar.get<bf16>(n);
            hipLaunchKernelGGL(split_bf16_kernel, g1(n), dim3(256), 0, s, src, n, d.hi, d.lo);
        } else {
            ck(cast_f32_bf16_launch(src, d.hi, (int64_t)n, s));
        }
        return d;
    }
    // [R][Cc] fp32 -> [Cc][Rpad] bf16, columns R..Rpad zero (Rpad: the contraction length of the GEMM that reads it, a multiple of 64)
    Split transposed(const float* src, int R, int Cc, int Rpad) const {
        Split d{ar.get<bf16>((size_t)Cc * Rpad), split_precision() ? ar.get<bf16>((size_t)Cc * Rpad) : nullptr};
        hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3(cdiv(Cc, 32), cdiv(Rpad, 32)), dim3(32, 8), 0, s, src, R, Cc, Cc, d.hi, d.lo, Rpad);
        return d;
    }
    void mm1(const bf16* a, const bf16* w, int M, int N, int K, const float* bias, float* out) const {
        AOperand A;
        aoperand_rows(A, a, K, K);
        Epilogue E;
        epilogue_defaults(E);
        E.out = out; E.ldo = N; E.out_f32 = 1; E.bias = bias;
        ck(gemm_launch(A, w, M, N, K, E, ws, ws_bytes, s));
    }
    // out [M][N] fp32 = a [M][K] w[N][K]^T (+ bias)
    void mm(const Split& a, const Split& w, int M, int N, int K, const float* bias, float* out) const {
        mm1(a.hi, w.hi, M, N, K, bias, out);
        if (a.lo && w.lo) {
            const size_t mk = ar.mark();
            float* t1 = f32((size_t)M * N);
            float* t2 = f32((size_t)M * N);
            mm1(a.lo, w.hi, M, N, K, nullptr, t1);
            mm1(a.hi, w.lo, M, N, K, nullptr, t2);
            hipLaunchKernelGGL(add3_kernel, g1((size_t)M * N), dim3(256), 0, s, out, (const float*)t1, (const float*)t2, (size_t)M * N);
            ar.release(mk);
        }
    }
    // ---- one-launch split-precision products (cat3_* above); GL_TRAIN_3LAUNCH=1 (developer A/B) keeps the three-launch form
    static bool one_launch() {
        static const bool three = dev_env("GL_TRAIN_3LAUNCH") && atoi(dev_env("GL_TRAIN_3LAUNCH")) != 0;
        return split_precision() && !three;
    }
    bf16* cat3_rows(const float* src, size_t R, int K, int side) const {
        bf16* d = ar.get<bf16>(R * 3 * (size_t)K);
        hipLaunchKernelGGL(cat3_rows_kernel, g1(R * (size_t)K), dim3(256), 0, s, src, R * (size_t)K, K, side, d);
        return d;
    }
    bf16* cat3_transposed(const float* src, int R, int Cc, int Rpad, int side) const {
        bf16* d = ar.get<bf16>((size_t)Cc * 3 * Rpad);
        hipLaunchKernelGGL(cat3_transposed_kernel, dim3(cdiv(Cc, 32), cdiv(Rpad, 32)), dim3(32, 8), 0, s, src, R, Cc, Cc, side, d, Rpad);
        return d;
    }
    // the weight side of a product (side = 1): cached for frozen parameters
    bf16* cat3_rows_w(const float* W, int N, int K) const {
        return weight_operand(W, 1, N, K, 0, (size_t)N * 3 * K, [&](bf16* d) {
            hipLaunchKernelGGL(cat3_rows_kernel, g1((size_t)N * K), dim3(256), 0, s, W, (size_t)N * K, K, 1, d);
        });
    }
    bf16* cat3_transposed_w(const float* W, int N, int K) const {
        return weight_operand(W, 2, N, K, 0, (size_t)K * 3 * N, [&](bf16* d) {
            hipLaunchKernelGGL(cat3_transposed_kernel, dim3(cdiv(K, 32), cdiv(N, 32)), dim3(32, 8), 0, s, W, N, K, K, 1, d, N);
        });
    }
    // y = x W^T + b
    float* lin_fwd(const float* x, int M, int K, const float* W, const float* b, int N) const {
        float* y = f32((size_t)M * N);
        const size_t mk = ar.mark();       // the bf16 operand copies live for this product only (stream order keeps their reuse safe)
        if (one_launch()) mm1(cat3_rows(x, M, K, 0), cat3_rows_w(W, N, K), M, N, 3 * K, b, y);
        else mm(to_bf16(x, (size_t)M * K), to_bf16(W, (size_t)N * K), M, N, K, b, y);
        ar.release(mk);
        return y;
    }
    // dgrad: dx [M][K] = dy [M][N] W [N][K]   (the GEMM's "weight" operand is W^T, contraction over N)
    float* lin_dgrad(const float* dy, int M, int N, const float* W, int K) const {
        float* dx = f32((size_t)M * K);
        const size_t mk = ar.mark();
        if (one_launch()) mm1(cat3_rows(dy, M, N, 0), cat3_transposed_w(W, N, K), M, K, 3 * N, nullptr, dx);
        else mm(to_bf16(dy, (size_t)M * N), transposed(W, N, K, N), M, K, N, nullptr, dx);
        ar.release(mk);
        return dx;
    }
    // wgrad: dW [N][K] = dy^T x (contraction over the M rows, zero-padded to a multiple of 64), db [N] = column sums of dy
    void lin_wgrad(const float* dy, const float* x, int M, int N, int K, float* dW, float* db) const {
        const int Mp = round_up(M, 64);
        const size_t mk = ar.mark();
        if (dW) {
            if (one_launch()) mm1(cat3_transposed(dy, M, N, Mp, 0), cat3_transposed(x, M, K, Mp, 1), N, K, 3 * Mp, nullptr, dW);
            else mm(transposed(dy, M, N, Mp), transposed(x, M, K, Mp), N, K, Mp, nullptr, dW);
        }
        ar.release(mk);
        if (db) hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64)), dim3(1024), 0, s, dy, (const float*)nullptr, M, N, db);
    }
    // out[0] = coef * sum a b (mode 0: the tanh gate's derivative) / mean (a - b)^2 (mode 1: the loss); fixed summation order
    void dot_reduce(const float* a, const float* b, size_t n, const float* alpha, float scale, int mode, float* out) const {
        if (n <= ((size_t)1 << 18)) {
            hipLaunchKernelGGL(dot_reduce_kernel, dim3(1), dim3(1024), 0, s, a, b, n, alpha, scale, mode, out);
            return;
        }
        const size_t chunk = (size_t)1 << 16;
        const int nb = (int)((n + chunk - 1) / chunk);
        const size_t mk = ar.mark();
        float* partial = f32(nb);
        hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(1024), 0, s, a, b, n, chunk, mode, partial);
        hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(64), 0, s, (const float*)partial, nb, n, alpha, scale, mode, out);
        ar.release(mk);
    }
    struct LN { float* y; float* xhat; float* rstd; };
    LN ln_fwd(const float* x, int R, int Cc, const float* g, const float* b) const {
        LN r{f32((size_t)R * Cc), f32((size_t)R * Cc), f32(R)};
        hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, x, g, b, R, Cc, r.y, r.xhat, r.rstd);
        return r;
    }
    void ln_bwd(const float* dy, const LN& f, const float* g, int R, int Cc, float* dx, bool accumulate, float* dgamma, float* dbeta) const {
        hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, dy, f.xhat, f.rstd, g, R, Cc, dx, accumulate ? 1 : 0);
        if (dgamma) hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(Cc, 64)), dim3(1024), 0, s, dy, (const float*)f.xhat, R, Cc, dgamma);
        if (dbeta) hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(Cc, 64)), dim3(1024), 0, s, dy, (const float*)nullptr, R, Cc, dbeta);
    }
    struct Attn { float* o; float* lse; };
    template <int DC, int LPQ>
    Attn attn_fwd_d(const float* q, const float* k, const float* v, int B, int H, int Nq, int Nk) const {
        constexpr int D = DC * LPQ;
        Attn a{f32((size_t)B * Nq * H * D), f32((size_t)B * H * Nq)};
        hipLaunchKernelGGL((attn_fwd_kernel<DC, LPQ>), dim3(cdiv(Nq * LPQ, 64), B * H), dim3(64), 0, s, q, k, v, H, Nq, Nk, 1.f / sqrtf((float)D), a.o, a.lse);
        return a;
    }
    template <int DC, int LPQ>
    void attn_bwd_d(const float* q, const float* k, const float* v, const Attn& f, const float* dout, int B, int H, int Nq, int Nk, float* dq,
                    float* dk, float* dv) const {
        constexpr int D = DC * LPQ;
        float* delta = f32((size_t)B * H * Nq);
        const float sc = 1.f / sqrtf((float)D);
        hipLaunchKernelGGL((attn_bwd_q_kernel<DC, LPQ>), dim3(cdiv(Nq * LPQ, 64), B * H), dim3(64), 0, s, q, k, v, (const float*)f.o, dout, (const float*)f.lse, H, Nq,
                           Nk, sc, dq, delta);
        if (dk && dv)
            hipLaunchKernelGGL((attn_bwd_kv_kernel<DC, LPQ>), dim3(cdiv(Nk * LPQ, 64), B * H), dim3(64), 0, s, q, k, v, dout, (const float*)f.lse,
                               (const float*)delta, H, Nq, Nk, sc, dk, dv);
    }
    // ---- MFMA attention (head dims 32 / 40 / 64 / 80): prep passes + the three kernels above
    static bool attn_on_mfma(int D) {
        static const bool valu = dev_env("GL_TRAIN_ATTN_VALU") && atoi(dev_env("GL_TRAIN_ATTN_VALU")) != 0;
        return !valu && (D == 32 || D == 40 || D == 64 || D == 80);
    }
    APrep attn_prep(const float* x, int B, int N, int H, int D, float mul) const {
        const int Npad = round_up(N, 64), DP = round_up(D, 16), DPO = round_up(D, 32);
        const size_t nr = (size_t)B * H * Npad * DP, nt = (size_t)B * H * DPO * Npad;
        bf16* rh = ar.get<bf16>(nr); bf16* rl = ar.get<bf16>(nr); bf16* th = ar.get<bf16>(nt); bf16* tl = ar.get<bf16>(nt);
        const size_t total = (size_t)B * H * Npad * std::max(DP, DPO);
        hipLaunchKernelGGL(attn_prep_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535 * 16)), dim3(256), 0, s, x, N, H, D, mul, Npad, DP, DPO, rh, rl, th, tl,
                           total);
        return APrep{rh, rl, th, tl, Npad};
    }
    template <int DP, int DPO>
    void attn_mfma_fwd_d(const APrep& Q, const APrep& K, const APrep& V, int B, int H, int D, int Nq, int Nk, float* o, float* lse) const {
        hipLaunchKernelGGL((attn_mfma_fwd_kernel<DP, DPO>), dim3(Q.Npad / 32, B * H), dim3(64), 0, s, Q, K, V, H, D, Nq, Nk, o, lse);
    }
    template <int DP, int DPO>
    void attn_mfma_bwd_d(const APrep& Q, const APrep& K, const APrep& V, const APrep& G, const float* lp, const float* dp, int B, int H, int D, int Nq, int Nk,
                         float sc, float* dq, float* dk, float* dv) const {
        hipLaunchKernelGGL((attn_mfma_bwd_q_kernel<DP, DPO>), dim3(Q.Npad / 32, B * H), dim3(64), 0, s, Q, K, V, G, lp, dp, H, D, Nq, Nk, sc, dq);
        if (dk && dv) hipLaunchKernelGGL((attn_mfma_bwd_kv_kernel<DP, DPO>), dim3(K.Npad / 32, B * H), dim3(64), 0, s, Q, K, V, G, lp, dp, H, D, Nq, Nk, dk, dv);
    }
    Attn attn_fwd_mfma(int D, const float* q, const float* k, const float* v, int B, int H, int Nq, int Nk) const {
        Attn a{f32((size_t)B * Nq * H * D), f32((size_t)B * H * Nq)};
        const size_t mk = ar.mark();
        const float sc = 1.f / sqrtf((float)D);
        const APrep Q = attn_prep(q, B, Nq, H, D, sc), K = attn_prep(k, B, Nk, H, D, 1.f), V = attn_prep(v, B, Nk, H, D, 1.f);
        switch (D) {
            case 32: attn_mfma_fwd_d<32, 32>(Q, K, V, B, H, D, Nq, Nk, a.o, a.lse); break;
            case 40: attn_mfma_fwd_d<48, 64>(Q, K, V, B, H, D, Nq, Nk, a.o, a.lse); break;
            case 64: attn_mfma_fwd_d<64, 64>(Q, K, V, B, H, D, Nq, Nk, a.o, a.lse); break;
            default: attn_mfma_fwd_d<80, 96>(Q, K, V, B, H, D, Nq, Nk, a.o, a.lse); break;
        }
        ar.release(mk);
        return a;
    }
    void attn_bwd_mfma(int D, const float* q, const float* k, const float* v, const Attn& f, const float* dout, int B, int H, int Nq, int Nk, float* dq,
                       float* dk, float* dv) const {
        const size_t mk = ar.mark();
        const float sc = 1.f / sqrtf((float)D);
        const APrep Q = attn_prep(q, B, Nq, H, D, sc), K = attn_prep(k, B, Nk, H, D, 1.f), V = attn_prep(v, B, Nk, H, D, 1.f), G = attn_prep(dout, B, Nq, H, D, 1.f);
        float* lp = f32((size_t)B * H * Q.Npad);
        float* dp = f32((size_t)B * H * Q.Npad);
        const size_t total = (size_t)B * H * Q.Npad;
        hipLaunchKernelGGL(attn_delta_kernel, g1(total), dim3(256), 0, s, (const float*)f.o, dout, (const float*)f.lse, Nq, H, D, Q.Npad, lp, dp, total);
        switch (D) {
            case 32: attn_mfma_bwd_d<32, 32>(Q, K, V, G, lp, dp, B, H, D, Nq, Nk, sc, dq, dk, dv); break;
            case 40: attn_mfma_bwd_d<48, 64>(Q, K, V, G, lp, dp, B, H, D, Nq, Nk, sc, dq, dk, dv); break;
            case 64: attn_mfma_bwd_d<64, 64>(Q, K, V, G, lp, dp, B, H, D, Nq, Nk, sc, dq, dk, dv); break;
            default: attn_mfma_bwd_d<80, 96>(Q, K, V, G, lp, dp, B, H, D, Nq, Nk, sc, dq, dk, dv); break;
        }
        ar.release(mk);
    }
    Attn attn_fwd(int D, const float* q, const float* k, const float* v, int B, int H, int Nq, int Nk) const {
        if (attn_on_mfma(D)) return attn_fwd_mfma(D, q, k, v, B, H, Nq, Nk);
        switch (D) {
            case 32: return attn_fwd_d<32, 1>(q, k, v, B, H, Nq, Nk);
            case 40: return attn_fwd_d<40, 1>(q, k, v, B, H, Nq, Nk);
            case 64: return attn_fwd_d<32, 2>(q, k, v, B, H, Nq, Nk);
            case 80: return attn_fwd_d<40, 2>(q, k, v, B, H, Nq, Nk);
            case 160: return attn_fwd_d<40, 4>(q, k, v, B, H, Nq, Nk);
            default: throw GlError(GL_ERR_UNSUPPORTED, fmt("training slice: head dim %d (32, 40, 64, 80, 160 are built)", D));
        }
    }
    void attn_bwd(int D, const float* q, const float* k, const float* v, const Attn& f, const float* dout, int B, int H, int Nq, int Nk, float* dq,
                  float* dk, float* dv) const {
        if (attn_on_mfma(D)) return attn_bwd_mfma(D, q, k, v, f, dout, B, H, Nq, Nk, dq, dk, dv);
        switch (D) {
            case 32: return attn_bwd_d<32, 1>(q, k, v, f, dout, B, H, Nq, Nk, dq, dk, dv);
            case 40: return attn_bwd_d<40, 1>(q, k, v, f, dout, B, H, Nq, Nk, dq, dk, dv);
            case 64: return attn_bwd_d<32, 2>(q, k, v, f, dout, B, H, Nq, Nk, dq, dk, dv);
            case 80: return attn_bwd_d<40, 2>(q, k, v, f, dout, B, H, Nq, Nk, dq, dk, dv);
            case 160: return attn_bwd_d<40, 4>(q, k, v, f, dout, B, H, Nq, Nk, dq, dk, dv);
            default: throw GlError(GL_ERR_UNSUPPORTED, "training slice: head dim");
        }
    }
    // 3x3 conv, stride 1, pad 1, over pixel rows a [B][H*W][Cin] (fp32; cast to bf16 for the implicit-GEMM kernel of gemm.hip) with
    // an OIHW fp32 weight -> [B][H*W][Cout] fp32 (+ bias). dgrad = true: the data gradient of that conv, a [..][Cout] -> [..][Cin].
    // stride 2 (Downsample: output (H/2) x (W/2)) and ups = 1 (Upsample: nearest 2x in the loader, output 2H x 2W) for the forward only
    float* conv3(const float* a, int B, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout, bool dgrad, int stride = 1, int ups = 0) const {
        const int Ho = stride == 2 ? H / 2 : H << ups, Wo = stride == 2 ? W / 2 : W << ups;
        const int Ci = dgrad ? Cout : Cin, Co = dgrad ? Cin : Cout, M = B * Ho * Wo;
        float* out = f32((size_t)M * Co);
        const size_t mk_ops = ar.mark();   // packed weights / bf16 activation copies: this product's, released behind it
        if (one_launch() && Ci % 64 == 0) {
            // one implicit-GEMM launch over 3 Ci input channels: activations (hi | lo | hi) as a two-source concat of the (hi | lo) buffer
            // with its own first half, weights (hi | hi | lo) along I (cached for frozen convs: flip / transpose for dgrad, split, pack -- once)
            const size_t nw = (size_t)Co * 9 * Ci;
            bf16* wp3 = weight_operand(w_oihw, dgrad ? 4 : 3, Cout, Cin, 0, 3 * nw, [&](bf16* d) {
                const float* src = w_oihw;
                if (dgrad) {
                    float* wt = f32((size_t)Cin * Cout * 9);
                    hipLaunchKernelGGL(conv_dgrad_weight_kernel, g1((size_t)Cin * Cout * 9), dim3(256), 0, s, w_oihw, Cout, Cin, wt);
                    src = wt;
                }
                float* w3 = f32(3 * nw);
                hipLaunchKernelGGL(conv_w_cat3_kernel, g1(nw), dim3(256), 0, s, src, Co, Ci, w3);
                ck(pack_conv_weight_launch(w3, d, Co, 3 * Ci, 3, 3, Co, s));
            });
            const size_t na = (size_t)B * H * W * Ci;
            bf16* a2 = ar.get<bf16>(2 * na);
            hipLaunchKernelGGL(cat2_rows_kernel, g1(na), dim3(256), 0, s, a, na, Ci, a2);
            AOperand A{};
            A.p0 = a2; A.C0 = 2 * Ci; A.ld0 = 2 * Ci; A.p1 = a2; A.C1 = Ci; A.ld1 = 2 * Ci; A.mode = A_CONV3;
            A.Hin = H; A.Win = W; A.Ho = Ho; A.Wo = Wo; A.stride = stride; A.ups = ups; A.pad_lo = 1;
            Epilogue E;
            epilogue_defaults(E);
            E.out = out; E.ldo = Co; E.out_f32 = 1; E.bias = bias; E.rows_per_b = Ho * Wo;
            ck(gemm_launch(A, wp3, M, Co, 27 * Ci, E, ws, ws_bytes, s));
            ar.release(mk_ops);
            return out;
        }
        const float* wsrc = w_oihw;
        if (dgrad) {
            float* wt = f32((size_t)Cin * Cout * 9);
            hipLaunchKernelGGL(conv_dgrad_weight_kernel, g1((size_t)Cin * Cout * 9), dim3(256), 0, s, w_oihw, Cout, Cin, wt);
            wsrc = wt;
        }
        bf16* wp = ar.get<bf16>((size_t)Co * 9 * Ci);
        ck(pack_conv_weight_launch(wsrc, wp, Co, Ci, 3, 3, Co, s));
        bf16* wp_lo = nullptr;
        if (split_precision()) {
            const size_t nw = (size_t)Co * 9 * Ci;
            float* wres = f32(nw);
            hipLaunchKernelGGL(bf16_residual_kernel, g1(nw), dim3(256), 0, s, wsrc, nw, wres);
            wp_lo = ar.get<bf16>(nw);
            ck(pack_conv_weight_launch(wres, wp_lo, Co, Ci, 3, 3, Co, s));
        }
        const Split av = to_bf16(a, (size_t)B * H * W * Ci);
        auto one = [&](const bf16* ap, const bf16* wq, const float* bb, float* o) {
            AOperand A{};
            A.p0 = ap; A.C0 = Ci; A.ld0 = Ci; A.mode = A_CONV3;
            A.Hin = H; A.Win = W; A.Ho = Ho; A.Wo = Wo; A.stride = stride; A.ups = ups; A.pad_lo = 1;
            Epilogue E;
            epilogue_defaults(E);
            E.out = o; E.ldo = Co; E.out_f32 = 1; E.bias = bb; E.rows_per_b = Ho * Wo;
            ck(gemm_launch(A, wq, M, Co, 9 * Ci, E, ws, ws_bytes, s));
        };
        one(av.hi, wp, bias, out);
        if (av.lo && wp_lo) {
            const size_t mk = ar.mark();
            float* t1 = f32((size_t)M * Co);
            float* t2 = f32((size_t)M * Co);
            one(av.lo, wp, nullptr, t1);
            one(av.hi, wp_lo, nullptr, t2);
            hipLaunchKernelGGL(add3_kernel, g1((size_t)M * Co), dim3(256), 0, s, out, (const float*)t1, (const float*)t2, (size_t)M * Co);
            ar.release(mk);
        }
        ar.release(mk_ops);
        return out;
    }
    struct GN { float* a; float* xhat; float* rstd; };
    GN gn_silu_fwd(const float* x, int B, int HW, int Cc, const float* g, const float* b, bool silu = true, float eps = 1e-5f) const {
        GN r{f32((size_t)B * HW * Cc), f32((size_t)B * HW * Cc), f32((size_t)B * 32)};
        hipLaunchKernelGGL(gn_silu_fwd_kernel, dim3(32, B), dim3(256), 0, s, x, g, b, HW, Cc, r.xhat, r.rstd, r.a, silu ? 1 : 0, eps);
        return r;
    }
    void gn_silu_bwd(const float* da, const GN& f, const float* g, const float* b, int B, int HW, int Cc, float* dx, bool accumulate, bool silu = true) const {
        hipLaunchKernelGGL(gn_silu_bwd_kernel, dim3(32, B), dim3(256), 0, s, da, (const float*)f.xhat, (const float*)f.rstd, g, b, HW, Cc, dx, accumulate ? 1 : 0,
                           silu ? 1 : 0);
    }
    void add(float* dst, const float* src, size_t n) const { hipLaunchKernelGGL(add_inplace_kernel, g1(n), dim3(256), 0, s, dst, src, n); }
    // rows [B][rows_per_b][Cc] of a [B][stride_rows][Cc] tensor starting at row0 -> a packed copy, and back
    float* slice_rows(const float* src, int B, int stride_rows, int row0, int rows, int Cc) const {
        float* d = f32((size_t)B * rows * Cc);
        hip(hipMemcpy2DAsync(d, (size_t)rows * Cc * 4, src + (size_t)row0 * Cc, (size_t)stride_rows * Cc * 4, (size_t)rows * Cc * 4, B, hipMemcpyDeviceToDevice, s),
            "hipMemcpy2DAsync");
        return d;
    }
    void put_rows(float* dst, int B, int stride_rows, int row0, const float* src, int rows, int Cc) const {
        hip(hipMemcpy2DAsync(dst + (size_t)row0 * Cc, (size_t)stride_rows * Cc * 4, src, (size_t)rows * Cc * 4, (size_t)rows * Cc * 4, B, hipMemcpyDeviceToDevice, s),
            "hipMemcpy2DAsync");
    }
};

}  // namespace

int adamw_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, double wd, int step, hipStream_t s) {
    if (step < 1) return set_error(GL_ERR_ARG, "adamw_step: step counts from 1");
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, (float)lr, (float)b1, (float)(1.0 - b1), (float)b2,
                       (float)(1.0 - b2), (float)eps, (float)(1.0 - lr * wd), (float)(lr / bc1), (float)sqrt(bc2));
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// Everything the backward of a BasicTransformerBlock needs from its forward (all in the arena)
struct BlockSaved {
    Ctx::LN n1, nf1, nf2, n2, n3;
    Ctx::Attn a1, af, a2;
    float *q1, *k1, *v1, *qf, *kf, *vf, *q2, *k2, *v2;
    float *af_vis, *of, *uf, *hf, *ff_f, *u3;
};

// y = BasicTransformerBlock(x, context, objs) (attention.py:333-338), everything the backward needs kept in the arena
static BlockSaved block_forward(const Ctx& c, const TrainBlockDims& d, const float* const* P, const float* x, const float* objs, const float* context, float* y) {
    hipStream_t s = c.s;
    const int B = d.B, N = d.N, Ng = d.Ng, C = d.C, H = d.heads, D = C / H, T = N + Ng, M = B * N, MT = B * T, MC = B * d.ctx_T, KD = d.ctx_dim;
    const size_t nx = (size_t)M * C;
    BlockSaved S;
    // x1 = attn1(norm1(x)) + x
    S.n1 = c.ln_fwd(x, M, C, P[TP_NORM1_W], P[TP_NORM1_B]);
    S.q1 = c.lin_fwd(S.n1.y, M, C, P[TP_A1_Q], nullptr, C);
    S.k1 = c.lin_fwd(S.n1.y, M, C, P[TP_A1_K], nullptr, C);
    S.v1 = c.lin_fwd(S.n1.y, M, C, P[TP_A1_V], nullptr, C);
    S.a1 = c.attn_fwd(D, S.q1, S.k1, S.v1, B, H, N, N);
    float* o1 = c.lin_fwd(S.a1.o, M, C, P[TP_A1_O], P[TP_A1_OB], C);
    float* x1 = c.f32(nx);
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(nx), dim3(256), 0, s, x, (const float*)o1, (const float*)nullptr, 1.f, nx, x1);
    // fuser (attention.py:236-244): x2 = x1 + scale tanh(alpha_attn) attn(norm1([x1 ; linear(objs)]))[:, :N]
    float* ol = c.lin_fwd(objs, B * Ng, KD, P[TP_F_LIN_W], P[TP_F_LIN_B], C);
    float* cat = c.f32((size_t)MT * C);
    c.put_rows(cat, B, T, 0, x1, N, C);
    c.put_rows(cat, B, T, N, ol, Ng, C);
    S.nf1 = c.ln_fwd(cat, MT, C, P[TP_F_N1_W], P[TP_F_N1_B]);
    S.qf = c.lin_fwd(S.nf1.y, MT, C, P[TP_F_Q], nullptr, C);
    S.kf = c.lin_fwd(S.nf1.y, MT, C, P[TP_F_K], nullptr, C);
    S.vf = c.lin_fwd(S.nf1.y, MT, C, P[TP_F_V], nullptr, C);
    S.af = c.attn_fwd(D, S.qf, S.kf, S.vf, B, H, T, T);
    S.af_vis = c.slice_rows(S.af.o, B, T, 0, N, C);
    S.of = c.lin_fwd(S.af_vis, M, C, P[TP_F_O], P[TP_F_OB], C);
    float* x2 = c.f32(nx);
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)x1, (const float*)S.of, P[TP_F_ALPHA_ATTN], d.fuser_scale, nx, x2);
    //        x3 = x2 + scale tanh(alpha_dense) ff(norm2(x2))
    S.nf2 = c.ln_fwd(x2, M, C, P[TP_F_N2_W], P[TP_F_N2_B]);
    S.uf = c.lin_fwd(S.nf2.y, M, C, P[TP_F_FF1_W], P[TP_F_FF1_B], 8 * C);
    S.hf = c.f32((size_t)M * 4 * C);
    hipLaunchKernelGGL(geglu_fwd_kernel, Ctx::g1((size_t)M * 4 * C), dim3(256), 0, s, (const float*)S.uf, M, 4 * C, S.hf);
    S.ff_f = c.lin_fwd(S.hf, M, 4 * C, P[TP_F_FF2_W], P[TP_F_FF2_B], C);
    float* x3 = c.f32(nx);
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)x2, (const float*)S.ff_f, P[TP_F_ALPHA_DENSE], d.fuser_scale, nx, x3);
    // x4 = attn2(norm2(x3), context) + x3
    S.n2 = c.ln_fwd(x3, M, C, P[TP_NORM2_W], P[TP_NORM2_B]);
    S.q2 = c.lin_fwd(S.n2.y, M, C, P[TP_A2_Q], nullptr, C);
    S.k2 = c.lin_fwd(context, MC, KD, P[TP_A2_K], nullptr, C);
    S.v2 = c.lin_fwd(context, MC, KD, P[TP_A2_V], nullptr, C);
    S.a2 = c.attn_fwd(D, S.q2, S.k2, S.v2, B, H, N, d.ctx_T);
    float* o2 = c.lin_fwd(S.a2.o, M, C, P[TP_A2_O], P[TP_A2_OB], C);
    float* x4 = c.f32(nx);
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)x3, (const float*)o2, (const float*)nullptr, 1.f, nx, x4);
    // y = ff(norm3(x4)) + x4
    S.n3 = c.ln_fwd(x4, M, C, P[TP_NORM3_W], P[TP_NORM3_B]);
    S.u3 = c.lin_fwd(S.n3.y, M, C, P[TP_FF1_W], P[TP_FF1_B], 8 * C);
    float* h3 = c.f32((size_t)M * 4 * C);
    hipLaunchKernelGGL(geglu_fwd_kernel, Ctx::g1((size_t)M * 4 * C), dim3(256), 0, s, (const float*)S.u3, M, 4 * C, h3);
    float* ff3 = c.lin_fwd(h3, M, 4 * C, P[TP_FF2_W], P[TP_FF2_B], C);
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)x4, (const float*)ff3, (const float*)nullptr, 1.f, nx, y);
    return S;
}

// g: dL/dy on entry, dL/dx on return (the running gradient of the residual stream). dobjs and G[slot] (fuser.* only) are written.
static void block_backward(const Ctx& c, const TrainBlockDims& d, const float* const* P, const BlockSaved& S, const float* objs, float* g, float* dobjs,
                           float* const* G) {
    hipStream_t s = c.s;
    const int B = d.B, N = d.N, Ng = d.Ng, C = d.C, H = d.heads, D = C / H, T = N + Ng, M = B * N, MT = B * T, KD = d.ctx_dim;
    const size_t nx = (size_t)M * C;
    {   // y = x4 + ff(norm3(x4)): frozen weights, data gradients only
        float* g_h3 = c.lin_dgrad(g, M, C, P[TP_FF2_W], 4 * C);
        float* g_u3 = c.f32((size_t)M * 8 * C);
        hipLaunchKernelGGL(geglu_bwd_kernel, Ctx::g1((size_t)M * 4 * C), dim3(256), 0, s, (const float*)g_h3, (const float*)S.u3, M, 4 * C, g_u3);
        float* g_n3 = c.lin_dgrad(g_u3, M, 8 * C, P[TP_FF1_W], C);
        c.ln_bwd(g_n3, S.n3, P[TP_NORM3_W], M, C, g, true, nullptr, nullptr);
    }
    {   // x4 = x3 + attn2(norm2(x3), context): the context comes from the frozen text encoder, no dK / dV
        float* g_a2 = c.lin_dgrad(g, M, C, P[TP_A2_O], C);
        float* g_q2 = c.f32(nx);
        c.attn_bwd(D, S.q2, S.k2, S.v2, S.a2, g_a2, B, H, N, d.ctx_T, g_q2, nullptr, nullptr);
        float* g_n2 = c.lin_dgrad(g_q2, M, C, P[TP_A2_Q], C);
        c.ln_bwd(g_n2, S.n2, P[TP_NORM2_W], M, C, g, true, nullptr, nullptr);
    }
    {   // x3 = x2 + g_d ff(norm2(x2)), g_d = scale tanh(alpha_dense): the fuser's feed-forward, TRAINABLE
        if (G[TP_F_ALPHA_DENSE])
            c.dot_reduce((const float*)g, (const float*)S.ff_f, nx, P[TP_F_ALPHA_DENSE], d.fuser_scale, 0,
                               G[TP_F_ALPHA_DENSE]);
        float* g_ff = c.f32(nx);
        hipLaunchKernelGGL(gated_scale_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)g, P[TP_F_ALPHA_DENSE], d.fuser_scale, nx, g_ff);
        c.lin_wgrad(g_ff, S.hf, M, C, 4 * C, G[TP_F_FF2_W], G[TP_F_FF2_B]);
        float* g_hf = c.lin_dgrad(g_ff, M, C, P[TP_F_FF2_W], 4 * C);
        float* g_uf = c.f32((size_t)M * 8 * C);
        hipLaunchKernelGGL(geglu_bwd_kernel, Ctx::g1((size_t)M * 4 * C), dim3(256), 0, s, (const float*)g_hf, (const float*)S.uf, M, 4 * C, g_uf);
        c.lin_wgrad(g_uf, S.nf2.y, M, 8 * C, C, G[TP_F_FF1_W], G[TP_F_FF1_B]);
        float* g_nf2 = c.lin_dgrad(g_uf, M, 8 * C, P[TP_F_FF1_W], C);
        c.ln_bwd(g_nf2, S.nf2, P[TP_F_N2_W], M, C, g, true, G[TP_F_N2_W], G[TP_F_N2_B]);
    }
    {   // x2 = x1 + g_a attn(norm1([x1 ; linear(objs)]))[:, :N]: the fuser's attention, TRAINABLE
        if (G[TP_F_ALPHA_ATTN])
            c.dot_reduce((const float*)g, (const float*)S.of, nx, P[TP_F_ALPHA_ATTN], d.fuser_scale, 0,
                               G[TP_F_ALPHA_ATTN]);
        float* g_of = c.f32(nx);
        hipLaunchKernelGGL(gated_scale_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)g, P[TP_F_ALPHA_ATTN], d.fuser_scale, nx, g_of);
        c.lin_wgrad(g_of, S.af_vis, M, C, C, G[TP_F_O], G[TP_F_OB]);
        float* g_af_vis = c.lin_dgrad(g_of, M, C, P[TP_F_O], C);
        float* g_af = c.f32((size_t)MT * C);      // the grounding-token rows of the attention output are dropped by [:, :N]: zero gradient
        c.hip(hipMemsetAsync(g_af, 0, (size_t)MT * C * 4, s), "hipMemsetAsync");
        c.put_rows(g_af, B, T, 0, g_af_vis, N, C);
        float* g_qf = c.f32((size_t)MT * C);
        float* g_kf = c.f32((size_t)MT * C);
        float* g_vf = c.f32((size_t)MT * C);
        c.attn_bwd(D, S.qf, S.kf, S.vf, S.af, g_af, B, H, T, T, g_qf, g_kf, g_vf);
        c.lin_wgrad(g_qf, S.nf1.y, MT, C, C, G[TP_F_Q], nullptr);
        c.lin_wgrad(g_kf, S.nf1.y, MT, C, C, G[TP_F_K], nullptr);
        c.lin_wgrad(g_vf, S.nf1.y, MT, C, C, G[TP_F_V], nullptr);
        float* g_nf1 = c.lin_dgrad(g_qf, MT, C, P[TP_F_Q], C);
        c.add(g_nf1, c.lin_dgrad(g_kf, MT, C, P[TP_F_K], C), (size_t)MT * C);
        c.add(g_nf1, c.lin_dgrad(g_vf, MT, C, P[TP_F_V], C), (size_t)MT * C);
        float* g_cat = c.f32((size_t)MT * C);
        c.ln_bwd(g_nf1, S.nf1, P[TP_F_N1_W], MT, C, g_cat, false, G[TP_F_N1_W], G[TP_F_N1_B]);
        c.add(g, c.slice_rows(g_cat, B, T, 0, N, C), nx);
        float* g_ol = c.slice_rows(g_cat, B, T, N, Ng, C);
        c.lin_wgrad(g_ol, objs, B * Ng, C, KD, G[TP_F_LIN_W], G[TP_F_LIN_B]);
        float* g_objs = c.lin_dgrad(g_ol, B * Ng, C, P[TP_F_LIN_W], KD);
        c.hip(hipMemcpyAsync(dobjs, g_objs, (size_t)B * Ng * KD * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    }
    {   // x1 = x + attn1(norm1(x)): frozen
        float* g_a1 = c.lin_dgrad(g, M, C, P[TP_A1_O], C);
        float* g_q1 = c.f32(nx);
        float* g_k1 = c.f32(nx);
        float* g_v1 = c.f32(nx);
        c.attn_bwd(D, S.q1, S.k1, S.v1, S.a1, g_a1, B, H, N, N, g_q1, g_k1, g_v1);
        float* g_n1 = c.lin_dgrad(g_q1, M, C, P[TP_A1_Q], C);
        c.add(g_n1, c.lin_dgrad(g_k1, M, C, P[TP_A1_K], C), nx);
        c.add(g_n1, c.lin_dgrad(g_v1, M, C, P[TP_A1_V], C), nx);
        c.ln_bwd(g_n1, S.n1, P[TP_NORM1_W], M, C, g, true, nullptr, nullptr);
    }
}

static void block_check(const TrainBlockDims& d, const float* const* P) {
    if (d.C % 64 || d.ctx_dim % 64 || d.C % d.heads || d.B < 1 || d.N < 1 || d.Ng < 1) throw GlError(GL_ERR_ARG, "block_train_step: C and ctx_dim must be multiples of 64");
    for (int i = 0; i < TP_COUNT; ++i)
        if (!P[i]) throw GlError(GL_ERR_ARG, fmt("block_train_step: parameter slot %d is null", i));
}

int block_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainBlockDims& d, const float* const* P, const float* x, const float* objs,
                     const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* G, hipStream_t s) {
    try {
        block_check(d, P);
        Ctx c{ar, ws, ws_bytes, s};
        const size_t nx = (size_t)d.B * d.N * d.C;
        const BlockSaved S = block_forward(c, d, P, x, objs, context, y);
        // loss (trainer.py:366) and its gradient
        c.dot_reduce((const float*)y, target, nx, (const float*)nullptr, 1.f, 1, loss);
        float* g = c.f32(nx);
        hipLaunchKernelGGL(mse_grad_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)y, target, nx, g);
        block_backward(c, d, P, S, objs, g, dobjs, G);
        c.hip(hipMemcpyAsync(dx, g, nx * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        c.hip(hipGetLastError(), "training slice kernel launch");
    } catch (const GlError& e) {
        return set_error(e.code, "%s", e.what());
    }
    return GL_OK;
}

// SpatialTransformer.forward (attention.py:366-376) around one BasicTransformerBlock: x + proj_out(block(proj_in(norm(x)))), norm =
// GroupNorm(32, eps 1e-6) without activation, proj_in / proj_out 1 x 1 convs = Linears over pixel rows. norm / proj_* are SD layers
// (frozen); gradients: the block's fuser.* parameters, dx, dobjs. P / G: [norm.w, norm.b, proj_in.w, proj_in.b, <37 block slots>,
// proj_out.w, proj_out.b].
struct STSaved {
    Ctx::GN n0;
    BlockSaved blk;
};
static void st_check(const TrainBlockDims& d, const float* const* P) {
    for (int i = 0; i < ST_COUNT; ++i)
        if (!P[i]) throw GlError(GL_ERR_ARG, fmt("st_train_step: parameter slot %d is null", i));
    block_check(d, P + ST_BLOCK0);
}
static STSaved st_forward(const Ctx& c, const TrainBlockDims& d, const float* const* P, const float* x, const float* objs, const float* context, float* y) {
    const int B = d.B, N = d.N, C = d.C, M = B * N;
    const size_t nx = (size_t)M * C;
    STSaved S;
    S.n0 = c.gn_silu_fwd(x, B, N, C, P[ST_NORM_W], P[ST_NORM_B], false, 1e-6f);
    float* t0 = c.lin_fwd(S.n0.a, M, C, P[ST_PIN_W], P[ST_PIN_B], C);
    float* yb = c.f32(nx);
    S.blk = block_forward(c, d, P + ST_BLOCK0, t0, objs, context, yb);
    float* po = c.lin_fwd(yb, M, C, P[ST_POUT_W], P[ST_POUT_B], C);
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(nx), dim3(256), 0, c.s, x, (const float*)po, (const float*)nullptr, 1.f, nx, y);
    return S;
}
// g: dL/dy on entry, dL/dx on return
static void st_backward(const Ctx& c, const TrainBlockDims& d, const float* const* P, const STSaved& S, const float* objs, float* g, float* dobjs,
                        float* const* G) {
    const int B = d.B, N = d.N, C = d.C, M = B * N;
    float* g_b = c.lin_dgrad(g, M, C, P[ST_POUT_W], C);                              // through proj_out
    block_backward(c, d, P + ST_BLOCK0, S.blk, objs, g_b, dobjs, G + ST_BLOCK0);     // g_b: dL/d(block output) -> dL/d(block input)
    float* g_a = c.lin_dgrad(g_b, M, C, P[ST_PIN_W], C);                             // through proj_in
    c.gn_silu_bwd(g_a, S.n0, P[ST_NORM_W], P[ST_NORM_B], B, N, C, g, true, false);   // + the residual x_in: g already holds dL/dy
}

int st_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainBlockDims& d, const float* const* P, const float* x, const float* objs,
                  const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* G, hipStream_t s) {
    try {
        st_check(d, P);
        Ctx c{ar, ws, ws_bytes, s};
        const size_t nx = (size_t)d.B * d.N * d.C;
        const STSaved S = st_forward(c, d, P, x, objs, context, y);
        c.dot_reduce((const float*)y, target, nx, (const float*)nullptr, 1.f, 1, loss);
        float* g = c.f32(nx);
        hipLaunchKernelGGL(mse_grad_kernel, Ctx::g1(nx), dim3(256), 0, s, (const float*)y, target, nx, g);
        st_backward(c, d, P, S, objs, g, dobjs, G);
        c.hip(hipMemcpyAsync(dx, g, nx * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        c.hip(hipGetLastError(), "training slice kernel launch");
    } catch (const GlError& e) {
        return set_error(e.code, "%s", e.what());
    }
    return GL_OK;
}

// Forward + backward of one ResBlock (openaimodel.py:154-232, no up / down, no scale-shift norm) under mse_loss(y, target): every
// parameter of it is frozen in the reference's trainer (trainer.py:217-245), so what the training step needs from a ResBlock is the
// gradient w.r.t. its INPUT -- the path by which the loss reaches the fusers in front of it. Rows are pixels ([B][H*W][C], the
// layout of this library; the reference's NCHW is a permutation of it).
struct ResSaved {
    Ctx::GN n1, n2;
};
static void res_check(const TrainResDims& d, const float* const* P) {
    if (d.Cin % 64 || d.Cout % 64 || d.emb_dim % 64 || d.B < 1 || d.H < 1 || d.W < 1)
        throw GlError(GL_ERR_ARG, "resblock_train_step: Cin, Cout and emb_dim must be multiples of 64");
    const bool skip_conv = d.Cin != d.Cout;
    for (int i = 0; i < RP_COUNT; ++i)
        if (!P[i] && !(i >= RP_SKIP_W && !skip_conv)) throw GlError(GL_ERR_ARG, fmt("resblock_train_step: parameter slot %d is null", i));
    if (!skip_conv && (P[RP_SKIP_W] || P[RP_SKIP_B])) throw GlError(GL_ERR_ARG, "resblock_train_step: skip_connection is nn.Identity when Cin == Cout");
}
// silu_emb: SiLU(emb) [B][emb_dim] (emb_layers.0, shared by every ResBlock of a step).  h = conv(silu(gn(x))) + emb_layers(emb);
// y = skip(x) + conv(silu(gn(h)))
static ResSaved res_forward(const Ctx& c, const TrainResDims& d, const float* const* P, const float* x, const float* silu_emb, float* y) {
    const int B = d.B, HW = d.H * d.W, Cin = d.Cin, Cout = d.Cout, M = B * HW;
    const size_t ny = (size_t)M * Cout;
    ResSaved S;
    S.n1 = c.gn_silu_fwd(x, B, HW, Cin, P[RP_GN1_W], P[RP_GN1_B]);
    float* h1 = c.conv3(S.n1.a, B, d.H, d.W, P[RP_C1_W], P[RP_C1_B], Cin, Cout, false);
    float* eo = c.lin_fwd(silu_emb, B, d.emb_dim, P[RP_EMB_W], P[RP_EMB_B], Cout);
    float* h2 = c.f32(ny);
    hipLaunchKernelGGL(add_per_sample_kernel, Ctx::g1(ny), dim3(256), 0, c.s, (const float*)h1, (const float*)eo, HW, Cout, ny, h2);
    S.n2 = c.gn_silu_fwd(h2, B, HW, Cout, P[RP_GN2_W], P[RP_GN2_B]);
    float* h3 = c.conv3(S.n2.a, B, d.H, d.W, P[RP_C2_W], P[RP_C2_B], Cout, Cout, false);
    const float* sk = x;
    if (Cin != Cout) sk = c.lin_fwd(x, M, Cin, P[RP_SKIP_W], P[RP_SKIP_B], Cout);      // the 1 x 1 conv is a Linear over pixel rows
    hipLaunchKernelGGL(gated_add_kernel, Ctx::g1(ny), dim3(256), 0, c.s, sk, (const float*)h3, (const float*)nullptr, 1.f, ny, y);
    return S;
}
// g [.][Cout]: dL/dy (left untouched when Cin != Cout). Returns dL/dx [.][Cin] (g itself, updated in place, when Cin == Cout)
static float* res_backward(const Ctx& c, const TrainResDims& d, const float* const* P, const ResSaved& S, float* g) {
    const int B = d.B, HW = d.H * d.W, Cin = d.Cin, Cout = d.Cout, M = B * HW;
    float* g_a2 = c.conv3(g, B, d.H, d.W, P[RP_C2_W], nullptr, Cout, Cout, true);
    float* g_h2 = c.f32((size_t)M * Cout);
    c.gn_silu_bwd(g_a2, S.n2, P[RP_GN2_W], P[RP_GN2_B], B, HW, Cout, g_h2, false);    // = dL/dh1 (the emb path has nothing trainable upstream)
    float* g_a1 = c.conv3(g_h2, B, d.H, d.W, P[RP_C1_W], nullptr, Cin, Cout, true);
    float* g_x = Cin != Cout ? c.lin_dgrad(g, M, Cout, P[RP_SKIP_W], Cin) : g;         // through the skip connection
    c.gn_silu_bwd(g_a1, S.n1, P[RP_GN1_W], P[RP_GN1_B], B, HW, Cin, g_x, true);
    return g_x;
}

int resblock_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainResDims& d, const float* const* P, const float* x, const float* emb,
                        const float* target, float* y, float* loss, float* dx, hipStream_t s) {
    try {
        res_check(d, P);
        Ctx c{ar, ws, ws_bytes, s};
        const size_t ny = (size_t)d.B * d.H * d.W * d.Cout;
        float* se = c.f32((size_t)d.B * d.emb_dim);
        hipLaunchKernelGGL(silu_kernel, Ctx::g1((size_t)d.B * d.emb_dim), dim3(256), 0, s, emb, (size_t)d.B * d.emb_dim, se);
        const ResSaved S = res_forward(c, d, P, x, se, y);
        c.dot_reduce((const float*)y, target, ny, (const float*)nullptr, 1.f, 1, loss);
        float* g = c.f32(ny);
        hipLaunchKernelGGL(mse_grad_kernel, Ctx::g1(ny), dim3(256), 0, s, (const float*)y, target, ny, g);
        float* g_x = res_backward(c, d, P, S, g);
        c.hip(hipMemcpyAsync(dx, g_x, (size_t)d.B * d.H * d.W * d.Cin * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        c.hip(hipGetLastError(), "training slice kernel launch");
    } catch (const GlError& e) {
        return set_error(e.code, "%s", e.what());
    }
    return GL_OK;
}

// Downsample (mode 0: conv3x3 stride 2, openaimodel.py:99-124) / Upsample (mode 1: nearest 2x + conv3x3, openaimodel.py:64-96) of C
// channels; the conv is a frozen SD layer: forward, and the gradient w.r.t. the input.
static float* resample_forward(const Ctx& c, int mode, int B, int H, int W, int C, const float* w_oihw, const float* bias, const float* x) {
    return c.conv3(x, B, H, W, w_oihw, bias, C, C, false, mode ? 1 : 2, mode ? 1 : 0);
}
// g: dL/dy [B][Ho*Wo][C] -> dL/dx [B][H*W][C]
static float* resample_backward(const Ctx& c, int mode, int B, int H, int W, int C, const float* w_oihw, const float* g) {
    const size_t nx = (size_t)B * H * W * C;
    if (mode == 0) {
        float* z = c.f32(nx);
        hipLaunchKernelGGL(zero_insert2_kernel, Ctx::g1(nx), dim3(256), 0, c.s, g, H, W, C, nx, z);
        return c.conv3(z, B, H, W, w_oihw, nullptr, C, C, true);
    }
    float* gu = c.conv3(g, B, 2 * H, 2 * W, w_oihw, nullptr, C, C, true);
    float* dx = c.f32(nx);
    hipLaunchKernelGGL(sum2x2_kernel, Ctx::g1(nx), dim3(256), 0, c.s, (const float*)gu, H, W, C, nx, dx);
    return dx;
}

int resample_train_step(Arena& ar, float* ws, size_t ws_bytes, int mode, int B, int H, int W, int C, const float* w_oihw, const float* bias, const float* x,
                        const float* target, float* y, float* loss, float* dx, hipStream_t s) {
    try {
        if (C % 64 || (mode == 0 && ((H | W) & 1)) || B < 1 || mode < 0 || mode > 1) throw GlError(GL_ERR_ARG, "resample_train_step: C % 64, even H / W for mode 0");
        Ctx c{ar, ws, ws_bytes, s};
        const int Ho = mode ? 2 * H : H / 2, Wo = mode ? 2 * W : W / 2;
        const size_t ny = (size_t)B * Ho * Wo * C, nx = (size_t)B * H * W * C;
        float* yv = resample_forward(c, mode, B, H, W, C, w_oihw, bias, x);
        c.hip(hipMemcpyAsync(y, yv, ny * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        c.dot_reduce((const float*)y, target, ny, (const float*)nullptr, 1.f, 1, loss);
        float* g = c.f32(ny);
        hipLaunchKernelGGL(mse_grad_kernel, Ctx::g1(ny), dim3(256), 0, s, (const float*)y, target, ny, g);
        float* gx = resample_backward(c, mode, B, H, W, C, w_oihw, g);
        c.hip(hipMemcpyAsync(dx, gx, nx * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        c.hip(hipGetLastError(), "training slice kernel launch");
    } catch (const GlError& e) {
        return set_error(e.code, "%s", e.what());
    }
    return GL_OK;
}

// ======================================================================================================================
// The whole training iteration of the reference (trainer.py:353-392: model(input) -> mse_loss(model_output, noise) -> backward) for a
// UNetModel with the text grounding tokenizer and gatedSA fusers (openaimodel.py:237-464), composed of the slices above:
//   objs = position_net(boxes, masks, positive_embeddings)          text_grounding_net.py:30-52      TRAINABLE
//   emb  = time_embed(timestep_embedding(t))                         openaimodel.py:436-437           frozen, no backward needed
//   input_blocks / middle_block / output_blocks (skip concats) / out openaimodel.py:452-464
// Gradients for every fuser.* parameter of every SpatialTransformer and for position_net.*; nothing else is trainable in the
// reference (trainer.py:217-245), and no input gradient is needed in front of the first fuser. All activations of the forward stay
// in the arena (no recomputation at this size; DESIGN.md section 9 has the checkpointing plan for the full model).
namespace {

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int dim, float* __restrict__ out) {   // util.py:160-180
    const int b = blockIdx.x, half = dim / 2;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float a = t[b] * __expf(-9.210340371976184f * (float)i / (float)half);   // ln(10000)
        out[(size_t)b * dim + i] = cosf(a);
        out[(size_t)b * dim + half + i] = sinf(a);
    }
}
// PositionNet input rows (text_grounding_net.py:33-48): [pe * m + (1 - m) * null_positive | fourier(boxes) * m + (1 - m) * null_position],
// fourier = for k in 0..7: sin(f_k x) (4 values), cos(f_k x) (4 values), f_k = 100^(k / 8)   (util.py:12-26)
// (emb_masks: the mask of the embedding half -- `masks` itself for the text tokenizer, text_masks / image_masks for text+image,
// text_image_grounding_net.py:57-59). Keypoint tokenizer (keypoint_grounding_net.py:34-58): pe null, the embedding of token t of a sample
// is person_emb[t / 17] + keypoint_emb[t % 17], coords = 2 (x, y). Columns [D + 16 ncoord, W) are zero padding up to the GEMM's K step.
__global__ void posnet_input_kernel_f32(const float* __restrict__ coords, int ncoord, const float* __restrict__ masks, const float* __restrict__ emb_masks,
                                        const float* __restrict__ pe, const float* __restrict__ person_emb, const float* __restrict__ keypoint_emb, int tokens,
                                        const float* __restrict__ null_pos_feat, const float* __restrict__ null_xyxy, int D, int W, float* __restrict__ out) {
    const int row = blockIdx.x, PD = 16 * ncoord;
    const int t = row % tokens;
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        float v, nul;
        if (c >= D + PD) { out[(size_t)row * W + c] = 0.f; continue; }
        const float m = c < D ? emb_masks[row] : masks[row];
        if (c < D) {
            v = pe ? pe[(size_t)row * D + c] : person_emb[(size_t)(t / 17) * D + c] + keypoint_emb[(size_t)(t % 17) * D + c];
            nul = null_pos_feat[c];
        } else {
            const int j = c - D, k = j / (2 * ncoord), r = j % (2 * ncoord);
            const float a = powf(100.f, (float)k / 8.f) * coords[(size_t)row * ncoord + (r % ncoord)];
            v = r < ncoord ? sinf(a) : cosf(a);
            nul = null_xyxy[j];
        }
        out[(size_t)row * W + c] = v * m + (1.f - m) * nul;
    }
}
// dst [N][Kp] = src [N][K] zero-padded on the right
__global__ void pad_cols_kernel(const float* __restrict__ src, int K, int Kp, size_t n, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % Kp);
    dst[i] = c < K ? src[(i / Kp) * K + c] : 0.f;
}
// keypoint tokenizer tables: d person_emb[p][c] = sum_{b, k} m g[b][17 p + k][c] (by = 17, inner = 17), d keypoint_emb[k][c] = sum_{b, p} m g[b][17 p + k][c]
__global__ void table_grad_kernel(const float* __restrict__ g, const float* __restrict__ masks, int B, int tokens, int ld, int D, int person, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (c >= D) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < tokens; ++t)
            if ((person ? t / 17 : t % 17) == e) s += masks[b * tokens + t] * g[((size_t)b * tokens + t) * ld + c];
    out[(size_t)e * D + c] = s;
}
// out[c] = sum_rows (1 - m[row]) g[row][c0 + c]   (the gradient of a learnable null embedding)
__global__ void null_grad_kernel(const float* __restrict__ g, const float* __restrict__ masks, int R, int ld, int c0, int n, float* __restrict__ out,
                                 int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += (1.f - masks[r]) * g[(size_t)r * ld + c0 + c];
    out[c] = accumulate ? out[c] + s : s;
}
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, size_t n, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sg = 1.f / (1.f + __expf(-x[i]));
    dx[i] = dy[i] * sg * (1.f + x[i] * (1.f - sg));
}
// fp32 direct 3x3 conv, stride 1, pad 1, over pixel rows, for the two convs with 4 channels on one side (conv_in 4 -> C, out C -> 4):
// one thread per output element. w OIHW.
__global__ void conv3x3_direct_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int H, int W, int Cin,
                                      int Cout, size_t n, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % Cout), xw = (int)((i / Cout) % W), yh = (int)((i / ((size_t)Cout * W)) % H);
    const size_t b = i / ((size_t)Cout * W * H);
    float acc = bias ? bias[co] : 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = yh + ky - 1;
        if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = xw + kx - 1;
            if (xx < 0 || xx >= W) continue;
            const float* xp = x + ((b * H + yy) * W + xx) * Cin;
            const float* wp = w + (size_t)co * Cin * 9 + ky * 3 + kx;
            for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * 9], acc);
        }
    }
    y[i] = acc;
}
__global__ void concat_kernel(const float* __restrict__ a, int C0, const float* __restrict__ b, int C1, size_t rows, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = C0 + C1;
    if (i >= rows * C) return;
    const size_t r = i / C;
    const int c = (int)(i % C);
    out[i] = c < C0 ? a[r * C0 + c] : b[r * C1 + (c - C0)];
}
// dst[r][c] (+)= src[r][c0 + c]
__global__ void split_kernel(const float* __restrict__ src, int ld, int c0, int Cc, size_t rows, float* __restrict__ dst, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Cc) return;
    const size_t r = i / Cc;
    const int c = (int)(i % Cc);
    const float v = src[r * ld + c0 + c];
    dst[i] = accumulate ? dst[i] + v : v;
}

struct Names {
    std::unordered_map<std::string, int> idx;
    const float* const* params;
    float* const* grads;
    const float* w(const std::string& k) const {
        auto it = idx.find(k);
        if (it == idx.end() || !params[it->second]) throw GlError(GL_ERR_MISSING, "unet_train_step: missing parameter '" + k + "'");
        return params[it->second];
    }
    bool has(const std::string& k) const { auto it = idx.find(k); return it != idx.end() && params[it->second]; }
    float* g(const std::string& k) const {
        auto it = idx.find(k);
        return it == idx.end() ? nullptr : grads[it->second];
    }
};

}  // namespace

int unet_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainUNetCfg& cfg, const TrainUNetIn& in, int n_params, const char* const* names,
                    const float* const* params, float* const* grads, const char* const* block_names, float* eps_out, float* loss, hipStream_t s, hipEvent_t* grad_events, int n_grad_events,
                    TrainWeightCache* cache) {
    try {
        Names nm;
        nm.params = params;
        nm.grads = grads;
        for (int i = 0; i < n_params; ++i) nm.idx[names[i]] = i;
        for (int i = 0; i < n_params; ++i)
            if (grads[i] && !(strstr(names[i], ".fuser.") || !strncmp(names[i], "position_net.", 13)))
                throw GlError(GL_ERR_ARG, fmt("unet_train_step: a gradient was asked for '%s', which the reference keeps frozen", names[i]));
        Ctx c{ar, ws, ws_bytes, s};
        std::unordered_set<const void*> frozen;
        if (cache) {
            for (int i = 0; i < n_params; ++i)
                if (params[i] && !grads[i]) frozen.insert(params[i]);
            c.wc = cache;
            c.frozen = &frozen;
        }
        const int B = in.B, H0 = in.H, W0 = in.W, mc = cfg.model_channels, ED = 4 * mc, KD = cfg.context_dim, Ng = in.Ng;
        if (mc % 64 || KD % 64 || cfg.gr_dim % 64 || B < 1) throw GlError(GL_ERR_ARG, "unet_train_step: model_channels / context_dim / grounding dim must be multiples of 64");
        auto in_attn = [&](int ds) { for (int i = 0; i < cfg.n_attn; ++i) if (cfg.attention_resolutions[i] == ds) return true; return false; };

        // ---- grounding tokens (trainable): one MLP over [embedding | fourier(coords)] rows for the text tokenizer (boxes, 4 coords) and the
        // keypoint tokenizer (points, 2 coords; the embedding is person + keypoint table rows), two MLPs (text, image) whose tokens are
        // concatenated along the token axis for text+image (text_grounding_net.py:30-52, text_image_grounding_net.py:41-70,
        // keypoint_grounding_net.py:34-58)
        const int GK = cfg.grounding_kind, NB = in.Ng_boxes, MRB = B * NB, NC = GK == 2 ? 2 : 4, PWr = cfg.gr_dim + 16 * NC, PW = round_up(PWr, 64);
        const int NBR = GK == 1 ? 2 : 1;
        if (Ng != NB * NBR || (GK == 2 && NB % 17)) throw GlError(GL_ERR_ARG, "unet_train_step: Ng must be the box count (text), twice it (text+image), 17 per person (keypoint)");
        const int MR = B * Ng;
        const std::string null_pos = GK == 2 ? "position_net.null_xy_feature" : "position_net.null_position_feature";
        struct PosBranch { std::string lin, null_emb; const float* emb; const float* emb_mask; float *pcat, *l0, *a0, *l1, *a1, *out, *w0p; };
        PosBranch pb[2] = {{GK == 1 ? "position_net.linears_text" : "position_net.linears",
                            GK == 1 ? "position_net.null_text_feature" : GK == 2 ? "position_net.null_person_feature" : "position_net.null_positive_feature",
                            GK == 2 ? nullptr : in.positive_embeddings, GK == 1 ? in.text_masks : in.masks, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
                           {"position_net.linears_image", "position_net.null_image_feature", in.image_embeddings, in.image_masks, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr}};
        float* objs = NBR == 1 ? nullptr : c.f32((size_t)MR * KD);
        for (int r = 0; r < NBR; ++r) {
            PosBranch& p = pb[r];
            if ((!p.emb && GK != 2) || !p.emb_mask) throw GlError(GL_ERR_ARG, "unet_train_step: null grounding input");
            p.pcat = c.f32((size_t)MRB * PW);
            hipLaunchKernelGGL(posnet_input_kernel_f32, dim3(MRB), dim3(256), 0, s, in.boxes, NC, in.masks, p.emb_mask, p.emb,
                               GK == 2 ? nm.w("position_net.person_embeddings") : (const float*)nullptr,
                               GK == 2 ? nm.w("position_net.keypoint_embeddings") : (const float*)nullptr, NB, nm.w(p.null_emb), nm.w(null_pos), cfg.gr_dim, PW, p.pcat);
            const float* w0 = nm.w(p.lin + ".0.weight");
            if (PW != PWr) {        // the first Linear's K (800 for the keypoint tokenizer) padded to the GEMM's 64-step with zero columns
                p.w0p = c.f32((size_t)512 * PW);
                hipLaunchKernelGGL(pad_cols_kernel, Ctx::g1((size_t)512 * PW), dim3(256), 0, s, w0, PWr, PW, (size_t)512 * PW, p.w0p);
                w0 = p.w0p;
            }
            p.l0 = c.lin_fwd(p.pcat, MRB, PW, w0, nm.w(p.lin + ".0.bias"), 512);
            p.a0 = c.f32((size_t)MRB * 512);
            hipLaunchKernelGGL(silu_kernel, Ctx::g1((size_t)MRB * 512), dim3(256), 0, s, (const float*)p.l0, (size_t)MRB * 512, p.a0);
            p.l1 = c.lin_fwd(p.a0, MRB, 512, nm.w(p.lin + ".2.weight"), nm.w(p.lin + ".2.bias"), 512);
            p.a1 = c.f32((size_t)MRB * 512);
            hipLaunchKernelGGL(silu_kernel, Ctx::g1((size_t)MRB * 512), dim3(256), 0, s, (const float*)p.l1, (size_t)MRB * 512, p.a1);
            p.out = c.lin_fwd(p.a1, MRB, 512, nm.w(p.lin + ".4.weight"), nm.w(p.lin + ".4.bias"), KD);
            if (NBR == 1) objs = p.out;
            else c.put_rows(objs, B, Ng, r * NB, p.out, NB, KD);      // objs = cat([objs_text, objs_image], dim = 1)
        }
        // ---- time embedding (frozen): silu(emb) is what every ResBlock's emb_layers starts with
        float* te = c.f32((size_t)B * mc);
        hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(256), 0, s, in.timesteps, mc, te);
        float* e0 = c.lin_fwd(te, B, mc, nm.w("time_embed.0.weight"), nm.w("time_embed.0.bias"), ED);
        float* e0s = c.f32((size_t)B * ED);
        hipLaunchKernelGGL(silu_kernel, Ctx::g1((size_t)B * ED), dim3(256), 0, s, (const float*)e0, (size_t)B * ED, e0s);
        float* emb = c.lin_fwd(e0s, B, ED, nm.w("time_embed.2.weight"), nm.w("time_embed.2.bias"), ED);
        float* semb = c.f32((size_t)B * ED);
        hipLaunchKernelGGL(silu_kernel, Ctx::g1((size_t)B * ED), dim3(256), 0, s, (const float*)emb, (size_t)B * ED, semb);

        // ---- the layers, in forward order, each remembering what its backward needs
        struct Act { float* p; int C, H, W; };
        enum Kind { K_RES, K_ST, K_DOWN, K_UP, K_CAT };
        struct Layer {
            Kind kind;
            std::string prefix;
            int Cin, Cout, H, W, C0;                  // K_CAT: C0 = channels of h, Cin - C0 = channels of the skip; skip_idx = its producer
            int skip_idx;
            const float* x_in;                         // the layer's input (kept: with checkpointing the backward recomputes the forward from it)
            std::vector<const float*> P;
            std::vector<float*> G;
            ResSaved rs;
            STSaved ss;
        };
        std::vector<Layer> L;
        auto res_layer = [&](const std::string& p, int Cin, int Cout, int H, int W) {
            Layer l{K_RES, p, Cin, Cout, H, W, 0, -1, nullptr, {}, {}, {}, {}};
            static const char* k[RP_COUNT] = {"in_layers.0.weight", "in_layers.0.bias", "in_layers.2.weight", "in_layers.2.bias", "emb_layers.1.weight",
                                              "emb_layers.1.bias", "out_layers.0.weight", "out_layers.0.bias", "out_layers.3.weight", "out_layers.3.bias",
                                              "skip_connection.weight", "skip_connection.bias"};
            for (int i = 0; i < RP_COUNT; ++i) l.P.push_back((i >= RP_SKIP_W && Cin == Cout) ? nullptr : nm.w(p + "." + k[i]));
            return l;
        };
        auto st_layer = [&](const std::string& p, int C, int H, int W) {
            Layer l{K_ST, p, C, C, H, W, 0, -1, nullptr, {}, {}, {}, {}};
            l.P.resize(ST_COUNT);
            l.G.assign(ST_COUNT, nullptr);
            l.P[ST_NORM_W] = nm.w(p + ".norm.weight"); l.P[ST_NORM_B] = nm.w(p + ".norm.bias");
            l.P[ST_PIN_W] = nm.w(p + ".proj_in.weight"); l.P[ST_PIN_B] = nm.w(p + ".proj_in.bias");
            l.P[ST_POUT_W] = nm.w(p + ".proj_out.weight"); l.P[ST_POUT_B] = nm.w(p + ".proj_out.bias");
            for (int i = 0; i < TP_COUNT; ++i) {
                const std::string k = p + ".transformer_blocks.0." + block_names[i];
                l.P[ST_BLOCK0 + i] = nm.w(k);
                l.G[ST_BLOCK0 + i] = nm.g(k);
            }
            return l;
        };
        const int Cx = cfg.in_channels;
        const size_t M0 = (size_t)B * H0 * W0;
        // conv_in (frozen, in front of every trainable parameter: forward only)
        float* h0 = c.f32(M0 * mc);
        hipLaunchKernelGGL(conv3x3_direct_kernel, Ctx::g1(M0 * mc), dim3(256), 0, s, in.x, nm.w("input_blocks.0.0.weight"), nm.w("input_blocks.0.0.bias"), H0, W0,
                           Cx, mc, M0 * mc, h0);
        Act h{h0, mc, H0, W0};
        std::vector<Act> hs{h};
        std::vector<int> hs_layer{-1};          // which layer produced each skip (-1: conv_in)
        auto run = [&](Layer l) {
            const TrainBlockDims bd{B, l.H * l.W, Ng, l.Cout, cfg.num_heads, in.ctx_T, KD, in.fuser_scale};
            const size_t rows = (size_t)B * l.H * l.W;
            float* y = nullptr;
            l.x_in = h.p;
            // checkpointing: a block's output is allocated first, everything its forward keeps (activations, statistics, the bf16
            // operand copies) is given back to the arena behind it; the backward recomputes the forward from x_in
            if (l.kind == K_RES) {
                const TrainResDims rd{B, l.H, l.W, l.Cin, l.Cout, ED};
                res_check(rd, l.P.data());
                y = c.f32(rows * l.Cout);
                const size_t mk = ar.mark();
                l.rs = res_forward(c, rd, l.P.data(), h.p, semb, y);
                if (in.checkpoint) ar.release(mk);
            } else if (l.kind == K_ST) {
                st_check(bd, l.P.data());
                y = c.f32(rows * l.Cout);
                const size_t mk = ar.mark();
                l.ss = st_forward(c, bd, l.P.data(), h.p, objs, in.context, y);
                if (in.checkpoint) ar.release(mk);
            } else if (l.kind == K_DOWN) {
                y = resample_forward(c, 0, B, l.H, l.W, l.Cin, l.P[0], l.P[1], h.p);
                h.H = l.H / 2; h.W = l.W / 2;
            } else if (l.kind == K_UP) {
                y = resample_forward(c, 1, B, l.H, l.W, l.Cin, l.P[0], l.P[1], h.p);
                h.H = l.H * 2; h.W = l.W * 2;
            }
            h.p = y; h.C = l.Cout;
            L.push_back(std::move(l));
        };
        int ch = mc, ds = 1, n = 1;
        for (int level = 0; level < cfg.n_mult; ++level) {
            const int mult = cfg.channel_mult[level];
            for (int r = 0; r < cfg.num_res_blocks; ++r) {
                const std::string p = fmt("input_blocks.%d", n);
                run(res_layer(p + ".0", ch, mult * mc, h.H, h.W));
                ch = mult * mc;
                if (in_attn(ds)) run(st_layer(p + ".1", ch, h.H, h.W));
                hs.push_back(h); hs_layer.push_back((int)L.size() - 1);
                ++n;
            }
            if (level != cfg.n_mult - 1) {
                const std::string p = fmt("input_blocks.%d.0.op", n);
                Layer l{K_DOWN, p, ch, ch, h.H, h.W, 0, -1, nullptr, {nm.w(p + ".weight"), nm.w(p + ".bias")}, {}, {}, {}};
                run(l);
                hs.push_back(h); hs_layer.push_back((int)L.size() - 1);
                ds *= 2;
                ++n;
            }
        }
        run(res_layer("middle_block.0", ch, ch, h.H, h.W));
        run(st_layer("middle_block.1", ch, h.H, h.W));
        run(res_layer("middle_block.2", ch, ch, h.H, h.W));
        n = 0;
        for (int level = cfg.n_mult - 1; level >= 0; --level) {
            const int mult = cfg.channel_mult[level];
            for (int i = 0; i <= cfg.num_res_blocks; ++i) {
                const Act sk = hs.back();
                const int sk_layer = hs_layer.back();
                hs.pop_back(); hs_layer.pop_back();
                if (sk.H != h.H || sk.W != h.W) throw GlError(GL_ERR_STATE, "unet_train_step: skip / stream size mismatch");
                {   // h = cat([h, hs.pop()], dim = 1)
                    const size_t rows = (size_t)B * h.H * h.W;
                    float* cat = c.f32(rows * (h.C + sk.C));
                    hipLaunchKernelGGL(concat_kernel, Ctx::g1(rows * (h.C + sk.C)), dim3(256), 0, s, (const float*)h.p, h.C, (const float*)sk.p, sk.C, rows, cat);
                    Layer l{K_CAT, "", h.C + sk.C, h.C + sk.C, h.H, h.W, h.C, sk_layer, nullptr, {}, {}, {}, {}};
                    h.p = cat; h.C += sk.C;
                    L.push_back(l);
                }
                const std::string p = fmt("output_blocks.%d", n);
                run(res_layer(p + ".0", h.C, mc * mult, h.H, h.W));
                ch = mc * mult;
                int j = 1;
                if (in_attn(ds)) { run(st_layer(p + ".1", ch, h.H, h.W)); j = 2; }
                if (level && i == cfg.num_res_blocks) {
                    const std::string q = p + fmt(".%d.conv", j);
                    Layer l{K_UP, q, ch, ch, h.H, h.W, 0, -1, nullptr, {nm.w(q + ".weight"), nm.w(q + ".bias")}, {}, {}, {}};
                    run(l);
                    ds /= 2;
                }
                ++n;
            }
        }
        // out = conv(silu(gn(h)))  (openaimodel.py:389-393)
        const Ctx::GN on = c.gn_silu_fwd(h.p, B, H0 * W0, mc, nm.w("out.0.weight"), nm.w("out.0.bias"));
        const int Co = cfg.out_channels;
        const size_t ny = M0 * Co;
        float* y = c.f32(ny);
        hipLaunchKernelGGL(conv3x3_direct_kernel, Ctx::g1(ny), dim3(256), 0, s, (const float*)on.a, nm.w("out.2.weight"), nm.w("out.2.bias"), H0, W0, mc, Co, ny, y);
        if (eps_out) c.hip(hipMemcpyAsync(eps_out, y, ny * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        c.dot_reduce((const float*)y, in.target, ny, (const float*)nullptr, 1.f, 1, loss);

        // ---- backward
        float* gy = c.f32(ny);
        hipLaunchKernelGGL(mse_grad_kernel, Ctx::g1(ny), dim3(256), 0, s, (const float*)y, in.target, ny, gy);
        float* wt = c.f32((size_t)mc * Co * 9);       // out conv dgrad: the same direct conv on the flipped / transposed weight
        hipLaunchKernelGGL(conv_dgrad_weight_kernel, Ctx::g1((size_t)mc * Co * 9), dim3(256), 0, s, nm.w("out.2.weight"), Co, mc, wt);
        float* g_a = c.f32(M0 * mc);
        hipLaunchKernelGGL(conv3x3_direct_kernel, Ctx::g1(M0 * mc), dim3(256), 0, s, (const float*)gy, (const float*)wt, (const float*)nullptr, H0, W0, Co, mc,
                           M0 * mc, g_a);
        float* g = c.f32(M0 * mc);
        c.gn_silu_bwd(g_a, on, nm.w("out.0.weight"), nm.w("out.0.bias"), B, H0 * W0, mc, g, false);
        float* g_objs = c.f32((size_t)MR * KD);
        c.hip(hipMemsetAsync(g_objs, 0, (size_t)MR * KD * 4, s), "hipMemsetAsync");
        std::vector<float*> skip_grad(L.size(), nullptr);     // dL/d(output of layer i) arriving through a skip connection
        int first_st = -1;
        for (size_t i = 0; i < L.size(); ++i)
            if (L[i].kind == K_ST) { first_st = (int)i; break; }
        std::vector<int> st_ordinal(L.size(), -1);            // SpatialTransformer number in module order (input_blocks .. middle .. output_blocks)
        int n_st_layers = 0;
        for (size_t i = 0; i < L.size(); ++i)
            if (L[i].kind == K_ST) st_ordinal[i] = n_st_layers++;
        for (int i = (int)L.size() - 1; i >= 0 && i >= first_st; --i) {
            Layer& l = L[i];
            const size_t rows = (size_t)B * l.H * l.W;
            if (skip_grad[i]) {     // this layer's output also went into a skip connection (l.H, l.W are its INPUT size)
                const size_t out_rows = l.kind == K_DOWN ? rows / 4 : l.kind == K_UP ? rows * 4 : rows;
                c.add(g, skip_grad[i], out_rows * l.Cout);
            }
            if (l.kind == K_CAT) {
                float* gh = c.f32(rows * l.C0);
                hipLaunchKernelGGL(split_kernel, Ctx::g1(rows * l.C0), dim3(256), 0, s, (const float*)g, l.Cin, 0, l.C0, rows, gh, 0);
                const int C1 = l.Cin - l.C0;
                if (l.skip_idx >= first_st) {       // (a skip produced in front of the first fuser carries no gradient anybody needs)
                    float* gs = c.f32(rows * C1);
                    hipLaunchKernelGGL(split_kernel, Ctx::g1(rows * C1), dim3(256), 0, s, (const float*)g, l.Cin, l.C0, C1, rows, gs, 0);
                    skip_grad[l.skip_idx] = gs;
                }
                g = gh;
            } else if (l.kind == K_RES) {
                const TrainResDims rd{B, l.H, l.W, l.Cin, l.Cout, ED};
                float* keep = (in.checkpoint && l.Cin != l.Cout) ? c.f32(rows * l.Cin) : nullptr;    // (allocated in front of the scope below)
                const size_t mk = ar.mark();
                if (in.checkpoint) l.rs = res_forward(c, rd, l.P.data(), l.x_in, semb, c.f32(rows * l.Cout));
                float* gx = res_backward(c, rd, l.P.data(), l.rs, g);       // (g itself, updated in place, when Cin == Cout)
                if (keep) {
                    c.hip(hipMemcpyAsync(keep, gx, rows * l.Cin * 4, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
                    gx = keep;
                }
                if (in.checkpoint) ar.release(mk);
                g = gx;
            } else if (l.kind == K_ST) {
                const TrainBlockDims bd{B, l.H * l.W, Ng, l.Cout, cfg.num_heads, in.ctx_T, KD, in.fuser_scale};
                float* d_o = c.f32((size_t)MR * KD);
                const size_t mk = ar.mark();
                if (in.checkpoint) l.ss = st_forward(c, bd, l.P.data(), l.x_in, objs, in.context, c.f32(rows * l.Cout));
                st_backward(c, bd, l.P.data(), l.ss, objs, g, d_o, l.G.data());      // g in place
                c.add(g_objs, d_o, (size_t)MR * KD);
                if (in.checkpoint) ar.release(mk);
                // this block's fuser gradients are final: the caller's communication stream may pick them up (gl_train_wait_grads)
                // while the blocks in front of it are still in backward
                if (grad_events && st_ordinal[i] < n_grad_events) c.hip(hipEventRecord(grad_events[st_ordinal[i]], s), "hipEventRecord");
            } else if (l.kind == K_DOWN) {
                g = resample_backward(c, 0, B, l.H, l.W, l.Cin, l.P[0], g);
            } else if (l.kind == K_UP) {
                g = resample_backward(c, 1, B, l.H, l.W, l.Cin, l.P[0], g);
            }
        }
        // ---- position_net backward (Linear, SiLU, Linear, SiLU, Linear per branch; the learnable null embeddings: the position one is
        // shared by the branches; the keypoint tokenizer's person / keypoint embedding tables)
        for (int r = 0; r < NBR; ++r) {
            const PosBranch& p = pb[r];
            const float* go = NBR == 1 ? g_objs : c.slice_rows(g_objs, B, Ng, r * NB, NB, KD);
            c.lin_wgrad(go, p.a1, MRB, KD, 512, nm.g(p.lin + ".4.weight"), nm.g(p.lin + ".4.bias"));
            float* g_a1 = c.lin_dgrad(go, MRB, KD, nm.w(p.lin + ".4.weight"), 512);
            float* g_l1 = c.f32((size_t)MRB * 512);
            hipLaunchKernelGGL(silu_bwd_kernel, Ctx::g1((size_t)MRB * 512), dim3(256), 0, s, (const float*)g_a1, (const float*)p.l1, (size_t)MRB * 512, g_l1);
            c.lin_wgrad(g_l1, p.a0, MRB, 512, 512, nm.g(p.lin + ".2.weight"), nm.g(p.lin + ".2.bias"));
            float* g_a0 = c.lin_dgrad(g_l1, MRB, 512, nm.w(p.lin + ".2.weight"), 512);
            float* g_l0 = c.f32((size_t)MRB * 512);
            hipLaunchKernelGGL(silu_bwd_kernel, Ctx::g1((size_t)MRB * 512), dim3(256), 0, s, (const float*)g_a0, (const float*)p.l0, (size_t)MRB * 512, g_l0);
            float* gw0 = nm.g(p.lin + ".0.weight");
            if (PW == PWr) {
                c.lin_wgrad(g_l0, p.pcat, MRB, 512, PW, gw0, nm.g(p.lin + ".0.bias"));
            } else {
                float* gw0p = gw0 ? c.f32((size_t)512 * PW) : nullptr;
                c.lin_wgrad(g_l0, p.pcat, MRB, 512, PW, gw0p, nm.g(p.lin + ".0.bias"));
                if (gw0) hipLaunchKernelGGL(split_kernel, Ctx::g1((size_t)512 * PWr), dim3(256), 0, s, (const float*)gw0p, PW, 0, PWr, (size_t)512, gw0, 0);
            }
            float* g_cat = c.lin_dgrad(g_l0, MRB, 512, p.w0p ? p.w0p : nm.w(p.lin + ".0.weight"), PW);
            if (float* gp = nm.g(p.null_emb))
                hipLaunchKernelGGL(null_grad_kernel, Ctx::g1(cfg.gr_dim), dim3(256), 0, s, (const float*)g_cat, p.emb_mask, MRB, PW, 0, cfg.gr_dim, gp, 0);
            if (float* gp = nm.g(null_pos))
                hipLaunchKernelGGL(null_grad_kernel, Ctx::g1(16 * NC), dim3(256), 0, s, (const float*)g_cat, in.masks, MRB, PW, cfg.gr_dim, 16 * NC, gp, r);
            if (GK == 2) {
                if (float* gp = nm.g("position_net.person_embeddings"))
                    hipLaunchKernelGGL(table_grad_kernel, dim3(cdiv(cfg.gr_dim, 256), NB / 17), dim3(256), 0, s, (const float*)g_cat, in.masks, B, NB, PW, cfg.gr_dim, 1, gp);
                if (float* gp = nm.g("position_net.keypoint_embeddings"))
                    hipLaunchKernelGGL(table_grad_kernel, dim3(cdiv(cfg.gr_dim, 256), 17), dim3(256), 0, s, (const float*)g_cat, in.masks, B, NB, PW, cfg.gr_dim, 0, gp);
            }
        }
        // position_net's gradients -- the last ones of the step -- are final
        if (grad_events && n_st_layers < n_grad_events) c.hip(hipEventRecord(grad_events[n_st_layers], s), "hipEventRecord");
        c.hip(hipGetLastError(), "training step kernel launch");
    } catch (const GlError& e) {
        return set_error(e.code, "%s", e.what());
    }
    return GL_OK;
}

}  // namespace gl
