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
};

// is there a row-local feed-forward kernel for this problem? (C = 320, M a multiple of 128)
bool ff_rows_supported(int M, int C);
size_t ff_stream_bytes(int C);
// w1 [8C][C] fp32 (value rows, then gate rows: reference GEGLU.proj), b1 [8C], w2 [C][4C] fp32 -> the kernel's fragment stream
int ff_pack_launch(const float* w1, const float* b1, const float* w2, void* stream, int C, hipStream_t s);
int ff_rows_launch(const FFRowsParams& p, int C, hipStream_t s);

}  // namespace gl
