// Training slice (second half): forward + backward of ONE BasicTransformerBlock with a gatedSA fuser
// (reference ldm/modules/attention.py:333-338, 236-244) under the reference's loss, with the gradients the reference's trainer
// asks for (trainer.py:217-245: the fuser.* parameters; the block input and the grounding tokens so that the step chains into
// position_net and the blocks in front). See train.hip.
#pragma once
#include "common.h"

namespace gl {

struct TrainBlockDims {
    int B, N, Ng, C, heads, ctx_T, ctx_dim;
    float fuser_scale;      // GatedSelfAttentionDense.scale (attention.py:232)
};

// Parameter slots: the reference state_dict of BasicTransformerBlock(fuser_type="gatedSA"), fp32 device pointers
enum {
    TP_NORM1_W = 0, TP_NORM1_B, TP_A1_Q, TP_A1_K, TP_A1_V, TP_A1_O, TP_A1_OB,
    TP_F_LIN_W, TP_F_LIN_B, TP_F_N1_W, TP_F_N1_B, TP_F_Q, TP_F_K, TP_F_V, TP_F_O, TP_F_OB,
    TP_F_N2_W, TP_F_N2_B, TP_F_FF1_W, TP_F_FF1_B, TP_F_FF2_W, TP_F_FF2_B, TP_F_ALPHA_ATTN, TP_F_ALPHA_DENSE,
    TP_NORM2_W, TP_NORM2_B, TP_A2_Q, TP_A2_K, TP_A2_V, TP_A2_O, TP_A2_OB,
    TP_NORM3_W, TP_NORM3_B, TP_FF1_W, TP_FF1_B, TP_FF2_W, TP_FF2_B,
    TP_COUNT
};

class Arena;
// x [B][N][C], objs [B][Ng][ctx_dim], context [B][ctx_T][ctx_dim], target [B][N][C]: fp32 device. Outputs (fp32 device): y [B][N][C],
// loss[1] = mse_loss(y, target) (trainer.py:366), dx, dobjs, and grads[slot] for every non-null slot among the fuser's
// (TP_F_LIN_W .. TP_F_ALPHA_DENSE), each shaped like its parameter. bf16 GEMM operands, fp32 accumulation and fp32 elsewhere.
int block_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainBlockDims& d, const float* const* params, const float* x,
                     const float* objs, const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs,
                     float* const* grads, hipStream_t s);

// SpatialTransformer (attention.py:340-376) with one BasicTransformerBlock: slots = [norm.weight, norm.bias, proj_in.weight, proj_in.bias,
// the block's TP_* slots, proj_out.weight, proj_out.bias]; d.N = H * W pixels, d.C = in_channels = heads * d_head
enum { ST_NORM_W = 0, ST_NORM_B, ST_PIN_W, ST_PIN_B, ST_BLOCK0, ST_POUT_W = ST_BLOCK0 + TP_COUNT, ST_POUT_B, ST_COUNT };
int st_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainBlockDims& d, const float* const* params, const float* x, const float* objs,
                  const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* grads, hipStream_t s);

// ---- ResBlock (openaimodel.py:154-232): the second block type of the UNet. Frozen in the reference's trainer, so its backward is the
// input gradient only. Parameter slots = its state_dict (skip_connection.* null when Cin == Cout: nn.Identity), fp32 device pointers
struct TrainResDims {
    int B, H, W, Cin, Cout, emb_dim;
};
enum {
    RP_GN1_W = 0, RP_GN1_B, RP_C1_W, RP_C1_B, RP_EMB_W, RP_EMB_B, RP_GN2_W, RP_GN2_B, RP_C2_W, RP_C2_B, RP_SKIP_W, RP_SKIP_B,
    RP_COUNT
};
// x [B][H*W][Cin], emb [B][emb_dim], target [B][H*W][Cout] fp32 device (pixel rows) -> y [B][H*W][Cout], loss[1] = mse_loss(y, target), dx
int resblock_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainResDims& d, const float* const* params, const float* x, const float* emb,
                        const float* target, float* y, float* loss, float* dx, hipStream_t s);

// Downsample (mode 0: conv3x3 stride 2) / Upsample (mode 1: nearest 2x + conv3x3) of C channels (openaimodel.py:64-124): forward, loss, dx.
// x [B][H*W][C], y / target [B][Ho*Wo][C] pixel rows
int resample_train_step(Arena& ar, float* ws, size_t ws_bytes, int mode, int B, int H, int W, int C, const float* w_oihw, const float* bias, const float* x,
                        const float* target, float* y, float* loss, float* dx, hipStream_t s);

// ---- the whole training iteration for a UNetModel with the text grounding tokenizer and gatedSA fusers (openaimodel.py:237-464)
struct TrainUNetCfg {
    int in_channels, out_channels, model_channels, num_res_blocks, num_heads, context_dim, gr_dim;
    int grounding_kind;                 // 0 text, 1 text+image (two MLPs, tokens concatenated), 2 keypoint (points in `boxes`, 17 tokens per person)
    int n_mult, channel_mult[8], n_attn, attention_resolutions[8];
};
struct TrainUNetIn {
    int B, H, W, ctx_T, Ng;             // Ng: grounding tokens per sample = Ng_boxes (text) or 2 * Ng_boxes (text+image)
    int Ng_boxes;
    const float* x;                     // [B][H*W][in_channels] pixel rows: the noised latent
    const float* timesteps;             // [B]
    const float* context;               // [B][ctx_T][context_dim]
    const float* boxes;                 // [B][Ng_boxes][4]
    const float* masks;                 // [B][Ng_boxes]
    const float* positive_embeddings;   // [B][Ng_boxes][gr_dim]: the text embeddings
    const float* text_masks;            // text+image only: [B][Ng_boxes] each
    const float* image_masks;
    const float* image_embeddings;      // text+image only: [B][Ng_boxes][gr_dim]
    const float* target;                // [B][H*W][out_channels]: the noise
    float fuser_scale;
    int checkpoint;                     // 1: keep only every block's input and output; a block's backward recomputes its forward (same gradients, bit for bit)
};
// names / params / grads: the model's state_dict (fp32 device pointers; grads[i] non-null only for fuser.* and position_net.* entries:
// the reference's trainable set, trainer.py:217-245). block_names: the TP_* state_dict keys. eps_out (optional) [B][H*W][out_channels].
// cache (optional, round 6): the bf16 operand copies -- (hi | hi | lo) rows / transposes, packed conv weights and their dgrad forms -- of
// every parameter whose grads[i] is null are built once and reused by later calls: the caller promises that it only ever changes the
// parameters it asked gradients for (gligen_amd.train.TrainStep does; reference trainer.py:217-245 freezes everything else).
struct TrainWeightCache;
TrainWeightCache* train_cache_create();
void train_cache_destroy(TrainWeightCache* c);      // frees every cached copy
size_t train_cache_bytes(const TrainWeightCache* c);
int unet_train_step(Arena& ar, float* ws, size_t ws_bytes, const TrainUNetCfg& cfg, const TrainUNetIn& in, int n_params, const char* const* names,
                    const float* const* params, float* const* grads, const char* const* block_names, float* eps_out, float* loss, hipStream_t s,
                    hipEvent_t* grad_events = nullptr, int n_grad_events = 0, TrainWeightCache* cache = nullptr);
// grad_events (optional): event j is recorded on `s` when the backward of the j-th SpatialTransformer (module order) has written its
// fuser gradients -- the backward runs from the last block to the first, so high j come early --, event [number of SpatialTransformers]
// when position_net's (the last gradients of the step) are written

// One AdamW update of a flat fp32 parameter range, in place (torch.optim.AdamW semantics: trainer.py:245, :384 opt.step()); step = 1, 2, ...
int adamw_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, double wd, int step, hipStream_t s);

}  // namespace gl
