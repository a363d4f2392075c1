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

}  // namespace gl
