#pragma once
#include "common.h"

namespace gl {

// GroupNorm(32 groups) over an NHWC bf16 tensor that may be the channel-concat of two sources
// (the UNet decoder's skip concat, reference openaimodel.py:461), optional fused SiLU.
struct GNParams {
    const bf16* x0;
    const bf16* x1;
    int C0, C1;
    int B, HW;
    float eps;
    const float* gamma;
    const float* beta;
    bf16* y;       // [B][HW][C0+C1]
    int silu;
    float* partial;  // scratch [B][nsplit][32][2]
    float* coef;     // groupnorm_coef_launch only: [B][(C0+C1) / 8][16] = a of 8 channels, then c; y = silu(x * a + c) is the CONSUMER's job
};
int gn_nsplit(int HW);
size_t gn_partial_bytes(int B, int HW);
int groupnorm_launch(const GNParams& P, hipStream_t stream);
// Statistics only (round 6): per-(sample, channel) coefficients for a consumer that applies GroupNorm + SiLU while it stages its input
// (conv_halo_kernel, AOperand::gn). P.y is not written. 1 or 2 launches (groupnorm_coef_launches).
int groupnorm_coef_launch(const GNParams& P, hipStream_t stream);
int groupnorm_coef_launches(int HW, int C0, int C1);
inline size_t gn_coef_bytes(int B, int C) { return (size_t)B * C * 2 * sizeof(float); }

// LayerNorm over the last dim (C <= 1536, C % 8 == 0) of bf16 rows. Output rows are laid out
// [B][Tpad]: t < N1 from x, N1 <= t < N1+N2 from x2 (the GLIGEN fuser's [visual ; grounding]
// concat, reference attention.py:241), remaining pad rows are written as zeros.
struct LNParams {
    const bf16* x;
    const bf16* x2;
    int B, N1, N2, Tpad, C;
    float eps;
    const float* gamma;    // both null: plain (x - mean) * rstd (the affine part is folded into the consumer's weights)
    const float* beta;
    bf16* y;
    int ldx, ldy;  // row strides in elements of x / y; 0 = C. ldy > C: columns [C, ldy) of y are written as zeros (K padding
                   // of the GEMM that consumes y: the 96-channel stage of the ConvNeXt tokenizer)
};
int layernorm_launch(const LNParams& P, hipStream_t stream);

// row softmax: P[r][c] = softmax_c(S[r][c] * scale), fp32 in, bf16 out (VAE AttnBlock)
int softmax_rows_launch(const float* S, bf16* Pm, int rows, int cols, float scale, hipStream_t stream);

}  // namespace gl
