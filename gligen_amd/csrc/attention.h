#pragma once
#include "common.h"

namespace gl {

struct AttnParams {
    const bf16* q;
    const bf16* k;       // key-tile layout
    const bf16* vt;
    bf16* o;
    int H, d;
    int Nq, Nk;          // real query / key counts per (b,h)
    int Tq_pad, Tk_pad;  // padded row counts of the q / k,vt buffers (128- / 64-multiples)
    int ldo;             // O row stride in elements
    int o_rows_per_b;    // rows per sample in O
    int nqb;             // query blocks (128 rows) per (b,h); filled by attn_launch
    float scale_log2e;   // d^-0.5 * log2(e)
    int vt_layout;       // attn_vt_layout() of the vt buffer: 0 = [DPV of attn_dims][Tk_pad], tokens permuted in 16s; 1 = DPV 48, permuted in 32s
};

// padded head dims used by the q/k (DP) and v^T (DPV) buffers for a real head dim d
int attn_dims(int d, int* DP, int* DPV);
// Layout of the V^T buffer the attention kernel for (d, Nk keys) reads: returns AttnParams::vt_layout / Epilogue::vt_perm32 and the
// buffer's row count *DPV. 1 (d = 40, more than two key tiles): attn3_kernel, P V on v_mfma_f32_16x16x32_bf16 -- 48 rows instead of
// 64, tokens permuted inside groups of 32 as [0-3, 8-11, 16-19, 24-27, 4-7, 12-15, 20-23, 28-31].
int attn_vt_layout(int d, int Nk, int* DPV);
int attn_launch(const AttnParams& P, int B, hipStream_t stream);
const char* attn_kernel_name(int d, int Nk, int vt_layout = -1);  // kernel symbol attn_launch uses for head dim d and Nk keys (buffers laid out per attn_vt_layout)
// one-time init of a V^T buffer [BH][DPV][Tk_pad]: padding row d := 1.0 (the softmax denominator row)
int attn_vt_ones_launch(bf16* vt, int BH, int d, int Tk_pad, hipStream_t stream, int DPV = 0);   // DPV 0: attn_dims'

// one-time init of a K buffer [BH][Tk_pad/64][DP/8][64][8] (key-tile layout, gemm.h ktile_off): d = 40: column 40 := 1.0
int attn_k_init_launch(bf16* k, int BH, int d, int Tk_pad, hipStream_t stream);

}  // namespace gl
