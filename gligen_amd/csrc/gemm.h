// Host-visible descriptors for the bf16 MFMA GEMM / implicit-GEMM conv kernel (gemm.hip).
#pragma once
#include "common.h"

namespace gl {

enum { A_ROWS = 0, A_CONV3 = 1 };

// The activation ("rows") operand: logical matrix [M][K], bf16.
//  A_ROWS : row m = concat(p0[m*ld0 .. +C0), p1[m*ld1 .. +C1)),  K = C0 + C1
//  A_CONV3: implicit im2col of an NHWC tensor (channel-concat of p0,p1), 3x3 taps,
//           k = tap*(C0+C1) + c; optional nearest-2x upsample of the source and stride 1|2.
struct AOperand {
    const bf16* p0;
    const bf16* p1;
    int C0, C1;
    int ld0, ld1;
    int mode;
    int Hin, Win, Ho, Wo;  // conv geometry (source dims are pre-upsample)
    int stride, ups, pad_lo;
    // GroupNorm + SiLU as the conv's prologue (round 6; reference openaimodel.py:212-232 in_layers / out_layers: GroupNorm32 -> SiLU ->
    // conv): coefficients [B][(C0+C1) / 8][16] from groupnorm_coef_launch (norm.h); the kernel stages the RAW tensor and applies
    // y = silu(x * a + c) to every real pixel inside LDS (padding stays zero). Only conv_halo_kernel implements it:
    // ask gemm_gn_prologue_supported() first.
    const float* gn = nullptr;
};

// EPI_QKV_HEADS: one GEMM over the concatenated [to_q ; to_k ; to_v] rows: columns [0,C) -> q, [C,2C) -> k (as EPI_QK_HEADS),
// [2C,3C) -> v^T. Work items whose columns lie in the V third run the MFMAs with the operand roles exchanged, so their
// accumulators hold 4 consecutive TOKENS of one feature per lane -- the v^T store pattern of EPI_VT_HEADS -- at no extra cost.
enum { EPI_ROWMAJOR = 0, EPI_QK_HEADS = 1, EPI_VT_HEADS = 2, EPI_NCHW_F32 = 3, EPI_QKV_HEADS = 4 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GEGLU = 2, ACT_GELU = 3 };   // ACT_GELU: erf GELU (nn.GELU(), the ConvNeXt tokenizer's MLPs)

struct Epilogue {
    int mode;
    int act;
    void* out;    // EPI_ROWMAJOR: [M][ldo] bf16|f32 ; EPI_VT_HEADS: vt ; EPI_NCHW_F32: [B][n_real][HW] f32
    int ldo;
    int out_f32;
    const float* bias;  // [N]
    const float* bias2; // [M / rows_per_b][bias2_ld] broadcast over the rows of one sample
    int bias2_ld;
    int rows_per_b;
    unsigned rpb_magic; // m / rows_per_b without a divide: (umulhi(m, rpb_magic) + m) >> rpb_shift for 0 <= m < 2^31 (gemm_launch fills both)
    int rpb_shift;
    const bf16* res;    // residual [M][ldres]
    int ldres;
    const float* gate;  // optional device scalar: out = res + gate * v
    // attention head layouts
    bf16* q;            // EPI_QK_HEADS: columns [0,C) -> q [B*H][Tpad_q][DP] (row-major), [C,2C) -> k in the key-tile layout (ktile_off)
    bf16* k;
    int q_tiled;        // the q target is a KEY buffer (cross-attention K of the context / grounding tokens): key-tile layout too
    int C, H, d, DP, T; // T = rows (tokens) per sample in M (multiple of 64)
    int Tpad_q, Tpad_k;
    bf16* vt;           // EPI_QKV_HEADS: v^T destination (EPI_VT_HEADS passes it in `out`)
    int DPV;            // EPI_VT_HEADS / EPI_QKV_HEADS: vt [B*H][DPV][Tpad_k], tokens permuted within groups of 16
    int tok_off;        // head-layout epilogues: token index = (row % T) + tok_off (rows of a SECOND token range of the same sequence: the
                        // grounding tokens behind the visual ones in the fuser's [x ; objs] attention, written once per prompt)
    int vt_perm32;      // 1: tokens permuted within groups of 32 instead (attention.h attn_vt_layout == 1: the 16x16x32 P V kernel)
    int n_real;         // EPI_NCHW_F32: number of real output channels (<= N)
    // EPI_ROWMAJOR optional row remap: out row = (m / remap_in) * remap_out + m % remap_in + remap_off
    int remap_in, remap_out, remap_off;
    int geglu16;        // ACT_GEGLU: weight rows packed for the 16x16-tile kernel (pack_geglu layout 1)
    // ---- LayerNorm folded into the GEMM that consumes it (reference attention.py:333-338: x + attn(norm(x)), ff(norm(x))):
    //   LN(x) W^T + b = rstd_m (x W'^T - mean_m csum) + b',  W' = W * gamma (bf16), csum[n] = sum_k W'[n][k], b' = b + W beta
    // so the consumer multiplies the RAW residual-stream rows and corrects in its epilogue, and the LayerNorm kernel (one
    // read + one write of the activation, one launch) disappears. The row statistics come from the epilogue of the GEMM that
    // PRODUCED x: it has every output value in registers anyway.
    // Producer (EPI_ROWMAJOR bf16, unsplit): stats_out[row * stats_ld + colblock] = (sum, sum of squares) of the FINAL bf16
    // outputs of that row over one wave's column block; gemm_last_stats_nb() tells how many column blocks a row has (0: this
    // launch took a path that does not produce them -- the caller then runs the plain LayerNorm kernel without affine).
    float2* stats_out;
    int stats_ld;
    // Consumer: W is the folded W', bias the folded b'; acc is corrected with the row's (mean, rstd) from ln_nb partials.
    const float2* ln_stats;
    int ln_nb, ln_ld;
    const float* ln_csum;   // [N], in the packed row order of W
    float ln_inv_c, ln_eps; // 1 / C of the normalised rows, LayerNorm eps
};

// Key-tile layout of every attention K buffer: [b*H + h][t / 64][DP / 8][64 keys][8]. A 64-key tile is one contiguous block of
// 64 * DP elements whose 16-byte chunks are chunk-major, so the attention kernels copy it to LDS verbatim by LDS-DMA (1 KiB
// contiguous per wave-instruction) and a ds_read_b128 of one chunk of 32 consecutive keys is bank-conflict free.
__host__ __device__ inline size_t ktile_off(size_t bh, int tpad, int t, int dd, int DP) {
    return (bh * (size_t)tpad + (size_t)(t & ~63)) * DP + (size_t)(dd >> 3) * 512 + (size_t)(t & 63) * 8 + (dd & 7);
}

// C[M][N] = A[M][K] * W[N][K]^T (+ epilogue). W is row-major bf16 with leading dim K.
// ws / ws_bytes: optional fp32 split-K workspace (device); may be null (no split-K then).
int gemm_launch(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E,
                float* ws, size_t ws_bytes, hipStream_t stream);

// Same kernel with operand roles exchanged: the "rows" operand is a plain row-major matrix
// Wrows[Mw][K] (e.g. a weight) and the other operand is the activation X[Nx][K]; used to emit
// V^T for attention (EPI_VT_HEADS).
int gemm_launch_t(const bf16* Wrows, int Mw, const bf16* X, int Nx, int K, const Epilogue& E,
                  hipStream_t stream);

void epilogue_defaults(Epilogue& E);
// true when the default main loop (v5) is active: EPI_QKV_HEADS exists only there
bool gemm_supports_qkv();
// developer switch (kbench A/B): 1 = one-tile-per-workgroup LDS-DMA kernel (v2), 2 = persistent 16x16-tile kernel (v3),
// 4 = v5: persistent, buffer-descriptor LDS-DMA, asm fragment reads, staged epilogue (default)
void gemm_set_variant(int v);
int gemm_geglu_layout();
void gemm_force_cfg(int tm, int tn, int splits);   // 0,0,0 = automatic
void gemm_force_grid(int blocks);                  // 0 = automatic (512)
void gemm_set_autotune(int on);                    // 1 (default): time candidates at the first eager launch of a problem
void gemm_last_cfg(int* tm, int* tn, int* splits);
const char* gemm_last_kernel_name();  // kernel symbol (template arguments included) of the most recent gemm_launch
bool gemm_gn_prologue_supported(const AOperand& A, int M, int N, int K, const Epilogue& E);   // may this launch take AOperand::gn?
bool gemm_ln_fold_supported(const AOperand& A, int M, int N, int K, const Epilogue& E);   // may this launch take Epilogue::ln_stats?
int gemm_last_stats_nb();             // column blocks per row written to Epilogue::stats_out by the most recent gemm_launch (0: none)
void aoperand_rows(AOperand& A, const bf16* p, int K, int ld);

}  // namespace gl
