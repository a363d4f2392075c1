// Row-local feed-forward for gfx950: LayerNorm -> GEGLU projection -> erf-GEGLU -> FF-out -> (gated) residual -> row statistics
// in ONE launch, the hidden 4C activation never leaves the CU (ffn.hip). Replaces, for C = 320, the pair
// gemm_wide_kernel (GEGLU projection) + gemm_u_kernel (FF-out) of the reference's
//   FeedForward (ldm/modules/attention.py:37-64), as called from BasicTransformerBlock._forward (attention.py:333-338:
//   x = self.ff(self.norm3(x)) + x) and GatedSelfAttentionDense.forward (attention.py:236-244:
//   x = x + scale * tanh(alpha_dense) * self.ff(self.norm2(x))).
#pragma once
#include "common.h"

namespace gl {

struct FFRowsParams {
    const bf16* x;          // [M][ldx] input rows: the RAW residual stream when normalize = 1, LayerNorm output otherwise
    int ldx;
    int normalize;          // 1: (x - mean) * rstd per row in registers first (gamma / beta are folded into the packed W1 / b1)
    float eps;
    const void* stream;     // weights packed by ff_pack_launch
    const float* b2;        // [C] FF-out bias
    const bf16* res;        // residual rows [M][ldres] or null
    int ldres;
    const float* gate;      // optional device scalar: out = res + gate * ff
    bf16* out;              // [M][ldo]
    int ldo;
    float2* stats_out;      // optional: (sum, sum of squares) of each row's final bf16 outputs, one slot per row (Epilogue::ln_stats, nb = 1)
    int stats_ld;
    int M;
    // ---- chain (ff_chain_*): row-local C x C projections in front of / behind the feed-forward, in the same launch
    //  pre  = 1: x is the input of the LEADING projection (an attention output: attention.py:240 fuser.attn -> its to_out, :337 attn2
    //            -> its to_out); t = pre_res + pre_gate (x Wpre^T + pre_b) is written to mid_out and is what the feed-forward
    //            normalises, multiplies and adds back (res / ldres are ignored: the residual is t). Needs normalize = 1.
    //  post = 1: out = post_res + (y Wpost^T + post_b), y the feed-forward's result, which is not written (SpatialTransformer.proj_out
    //            + x_in, attention.py:374-376); stats_out then refers to that final output.
    int pre, post;
    const float* pre_b;
    const bf16* pre_res;
    int ld_pre_res;
    const float* pre_gate;
    bf16* mid_out;
    int ld_mid;
    const float* post_b;
    const bf16* post_res;
    int ld_post_res;
    //  post = 2: the trailing projection is the cross-attention's to_q behind the next LayerNorm (attention.py:336): `out` receives y
    //            (the residual stream, + stats_out), q = LN(y) Wq'^T + post_b goes to q [B*H][qTpad][qDP] (H = C / 40), the stream's
    //            trailing segment holds the LayerNorm-folded Wq', post_b the folded bias W beta; qT = rows per sample
    bf16* q;
    int qDP, qT, qTpad;
};

// is there a row-local feed-forward kernel for this problem? (C = 320, M a multiple of 128)
bool ff_rows_supported(int M, int C);
size_t ff_stream_bytes(int C);
// w1 [8C][C] fp32 (value rows, then gate rows: reference GEGLU.proj), b1 [8C], w2 [C][4C] fp32 -> the kernel's fragment stream
int ff_pack_launch(const float* w1, const float* b1, const float* w2, void* stream, int C, hipStream_t s);
int ff_rows_launch(const FFRowsParams& p, int C, hipStream_t s);
// The chained form: stream = [leading projection pre_w [C][C] (optional)][feed-forward][trailing projection post_w [C][C] (optional)].
// A projection behind another stage takes its input from accumulator registers, whose k-slot order differs from a row loaded from
// memory; the pack kernel orders those weights' columns to match, so a stream is packed FOR its chain shape.
size_t ff_chain_stream_bytes(int C, bool pre, bool post);
int ff_chain_pack_launch(const float* w1, const float* b1, const float* w2, const float* pre_w, const float* post_w, void* stream, int C, hipStream_t s);

// ---- Row-local q,k,v^T projection (ffn.hip qkv_rows_kernel): [leading C x C projection + residual ->] LayerNorm in registers ->
// to_q / to_k / to_v with the attention kernels' head layouts, ONE launch; rows stay in registers, weights stream through LDS.
// Replaces (C = 320, d = 40, the 64 x 64 level) SpatialTransformer.proj_in or attn1.to_out + residual (reference
// ldm/modules/attention.py:366-368, 183-186, 335) together with the fused q,k,v^T GEMM behind the next LayerNorm (:167-176).
struct QkvRowsParams {
    const bf16* x;          // [M][ldx]: input of the leading projection (pre = 1) or the raw residual rows (pre = 0)
    int ldx;
    int normalize;          // pre = 0: 1 = (x - mean) * rstd per row in registers first. pre = 1 always normalises t
    float eps;
    const void* stream;     // qkv_rows_pack_launch
    int M;
    int pre;                // 1: t = pre_res + (x Wpre^T + pre_b) -> mid_out; q,k,v are projections of LN(t)
    const float* pre_b;     // [C]
    const bf16* pre_res;    // [M][ld_pre_res] or null
    int ld_pre_res;
    bf16* mid_out;          // [M][ld_mid]
    int ld_mid;
    float2* stats_out;      // optional [M]: (sum, sum of squares) of each row of t (Epilogue::ln_stats with nb = 1)
    int np;                 // 3: q, k, v^T; 1: q only (cross-attention to_q)
    const float* bias;      // [np * C] (W beta of the folded LayerNorm)
    bf16* q;                // [B*H][Tpad_q][DP]
    bf16* k;                // key-tile layout (gemm.h ktile_off), [B*H][Tpad_k / 64][DP / 8][64][8]
    bf16* vt;               // [B*H][DPV][Tpad_k], tokens permuted within groups of 32 (vt_perm32 = 1)
    int H, d, DP, DPV, T, Tpad_q, Tpad_k, vt_perm32;   // T = rows per sample (a multiple of 128)
};
bool qkv_rows_supported(int M, int C, int d, int T);
size_t qkv_rows_stream_bytes(int C, bool pre, int np);
// pre_w [C][C] fp32 or null; w [np * C][C] bf16: the (LayerNorm-folded) to_q [; to_k ; to_v] rows
int qkv_rows_pack_launch(const float* pre_w, const bf16* w, int np, void* stream, int C, hipStream_t s);
int qkv_rows_launch(const QkvRowsParams& p, int C, hipStream_t s);

}  // namespace gl
