#pragma once
#include "common.h"

namespace gl {

// [cos | sin] sinusoidal timestep embedding (reference util.py:160-180), bf16 out [B][dim]
int timestep_embed_launch(const int64_t* t, bf16* out, int B, int dim, hipStream_t stream);

// Grounding-token MLP input (reference text_grounding_net.py:30-44, keypoint_grounding_net.py:34-56):
//   out[r] = [ feat[r]*m + (1-m)*null_feat | fourier(pos[r])*m + (1-m)*null_pos ]   (bf16)
// feat: [rows][F] fp32 (feat_mod > 0: row r uses feat[r % feat_mod], the keypoint person table),
// pos: [rows][P] fp32, fourier order [k][sin,cos][coord], f_k = 100^(k/8) (util.py:12-26).
struct PosNetIn {
    const float* feat;
    int feat_mod;
    const float* pos;
    const float* mask;       // [rows] mask for pos (and for feat when fmask is null)
    const float* fmask;      // optional separate feature mask (text/image masks)
    const float* null_feat;  // [F]
    const float* null_pos;   // [16*P]
    int rows, F, P;
    bf16* out;               // [rows][ld_out]
    int ld_out;              // >= F + 16*P, multiple of 64 (extra columns zero-filled)
};
int posnet_input_launch(const PosNetIn& p, hipStream_t stream);

// im2col (3x3, pad 1, stride 1) of a small-channel NCHW fp32 image into [B*H*W][Kpad] bf16,
// k = tap*Cin + c, zero for k >= 9*Cin. Two concatenated sources (latent + inpainting extra
// channels, reference openaimodel.py:443-447). Optional per-pixel affine pre-transform
// y = pre_scale * (Wp x) + bp on source 0 (VAE post_quant_conv, reference autoencoder.py:40-43).
struct Im2colParams {
    const float* x0; int C0;
    const float* x1; int C1;
    int B, H, W;
    const float* pre_w;  // [C0][C0] or null
    const float* pre_b;  // [C0]
    float pre_scale;
    bf16* out;
    int Kpad;
};
int im2col_small_launch(const Im2colParams& p, hipStream_t stream);

int cast_f32_bf16_launch(const float* src, bf16* dst, int64_t n, hipStream_t stream);
// rows x cols fp32 -> bf16 [B][rows_pad][cols], zero pad rows (CLIP context 77 -> 128)
int pad_rows_cast_launch(const float* src, bf16* dst, int B, int rows, int rows_pad, int cols, hipStream_t stream);
// [N][K] fp32 -> [N][Kpad] bf16, zero-filled columns K..Kpad
int cast_pad_cols_launch(const float* src, bf16* dst, int N, int K, int Kpad, hipStream_t stream);
// bf16 [B][rows][cols] -> [B][rows_pad][cols], zero pad rows
int pad_rows_bf16_launch(const bf16* src, bf16* dst, int B, int rows, int rows_pad, int cols, hipStream_t stream);
int set_f32_launch(float* dst, float v, hipStream_t stream);
// conv weight OIHW fp32 -> [O_pad][kh*kw][I] bf16 (K-major: k = tap*I + c), rows >= O zero
int pack_conv_weight_launch(const float* src, bf16* dst, int O, int I, int KH, int KW, int O_pad, hipStream_t stream);
// small-Cin conv weight OIHW fp32 -> [O][Kpad] bf16 with k = tap*I + c
// I_src (0 = I): input channels of the source weight when it has fewer than the packed layout (extra channels get zeros)
int pack_conv_small_launch(const float* src, bf16* dst, int O, int I, int Kpad, hipStream_t stream, int I_src = 0);
int bf16_rows_to_f32_launch(const bf16* src, float* dst, int B, int rows, int src_rows, int cols, hipStream_t stream);
// GroundingDownsampler pieces (spatial-map modalities): resize of the first n channels of img [B][Cimg][H][W] to [B][n][R][R]
// (torch bicubic align_corners=False, or legacy nearest), Conv2d(k 4, s 2, p 1) in fp32 NCHW with optional SiLU
int resize_f32_launch(const float* img, float* out, int B, int Cimg, int n, int H, int W, int R, int nearest, hipStream_t stream);
int conv4x4s2_f32_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Hin, int Win, int silu, hipStream_t stream);
// GEGLU projection [8C][K] fp32 (+bias [8C]) -> row-interleaved bf16 / fp32 (see gemm.hip epi_geglu4)
// layout: gemm_geglu_layout() of the GEMM variant that will consume the packed rows
int pack_geglu_launch(const float* w, const float* b, bf16* wp, float* bp, int C4, int K, int layout, hipStream_t stream);

// LayerNorm folded into the linear layer behind it (gemm.h Epilogue::ln_stats): wf = w * gamma (columns), bf = b + w beta (b may be null)
int ln_fold_launch(const float* w, const float* b, const float* gamma, const float* beta, float* wf, float* bf, int N, int K, hipStream_t stream);
// csum[n] = sum_k w[n][k] of packed bf16 rows
int rowsum_bf16_launch(const bf16* w, float* csum, int N, int K, hipStream_t stream);
int fill_f32_launch(float* dst, float v, int n, hipStream_t stream);
// gates[i] = scale[i / 2] * tanh(alpha[i])   (reference attention.py:241-242; one scale per fuser module)
int gates_launch(const float* const* alpha_ptrs, const float* scale, float* gates, int n, hipStream_t stream);

// e_t = e_u + s (e_c - e_u) ; e' = c0 e_t + c1 o1 + c2 o2 + c3 o3 ;
// pred_x0 = (x - sqrt(1-a_t) e') / sqrt(a_t) ; x_out = sqrt(a_prev) pred_x0 + sqrt(1-a_prev) e'
// (reference plms.py:116-158, sigma = 0). eps_pair holds [cond ; uncond] batches back to back.
struct PlmsParams {
    const float* eps_pair; int has_uncond; float guidance;
    float* e_t_out;
    const float* o1; const float* o2; const float* o3;
    float c0, c1, c2, c3;
    const float* x; float* x_out;
    float a_t, a_prev;
    int64_t n;  // elements per batch tensor (B*4*h*w)
};
int plms_update_launch(const PlmsParams& p, hipStream_t stream);

// img = (sqrt_ac x0 + sqrt_1mac noise) * mask + (1 - mask) * img   (reference plms.py:96-100)
int fuser_resize_launch(const bf16* tok, const bf16* x, const float* gate, bf16* y, int B, int row_stride, int grid_off, int sg, int sv,
                        int C, hipStream_t stream);
// x0B / noiseB / maskB: batch of that tensor, B or 1 (broadcast)
int inpaint_blend_launch(float* img, const float* x0, const float* noise, const float* mask,
                         float sqrt_ac, float sqrt_1mac, int B, int C, int HW, int x0B, int noiseB, int maskB, hipStream_t stream);

// NCHW fp32 in [-1,1] -> NHWC u8: trunc(255 * (clamp(x,-1,1)*0.5+0.5)) (reference gligen_inference.py:443-445)
int to_uint8_launch(const float* src, uint8_t* dst, int B, int C, int HW, hipStream_t stream);

// quant_conv 1x1 + posterior sample: h [B][2zc][HW] fp32 (encoder output), wq [2zc][2zc], bq [2zc], noise / z [B][zc][HW]
// (reference autoencoder.py:34-38, distributions.py:24-37)
int vae_posterior_launch(const float* h, const float* wq, const float* bq, const float* noise, float* z, int B, int zc, int HW,
                         float scale, hipStream_t stream);

int fill_i64_launch(int64_t* dst, int64_t v, int n, hipStream_t stream);
// box calibration (gl_box_calibrate): out3 = {float4-copy GB/s (read + write), LDS-DMA TB/s from L2 (chip), sustained bf16 MFMA TFLOP/s}
int box_calibrate_launch(void* scratch, size_t scratch_bytes, float* out3, hipStream_t stream);
// gl_mfma_calibrate: out4 = {TFLOP/s, shader clock in MHz measured inside the loop, ms, SIMD cycles per MFMA at that clock}; scratch >= 4 KiB
int mfma_calibrate_launch(int shape, int waves_per_simd, int n_acc, int zero_data, float target_ms, void* scratch, float* out4, hipStream_t stream);
int zero_launch(void* dst, size_t bytes, hipStream_t stream);

}  // namespace gl
