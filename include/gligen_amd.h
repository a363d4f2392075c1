/* gligen_amd.h — C ABI of libgligen_amd.so: the MI355X (gfx950) GLIGEN denoising engine.
 *
 * The reference (gligen/GLIGEN) has no FFI: its extension seam is Python classes instantiated
 * from dotted paths (reference ldm/util.py:71-86). This ABI is what the Python modules that
 * keep those dotted paths (ldm/…, grounding_input/… in this repo) bind through ctypes; every
 * entry point names the reference interface it stands behind.
 *
 * Conventions
 *  - every function returns 0 (GL_OK) or an error code; gl_last_error() has the message;
 *    nothing throws across the boundary;
 *  - pointers are raw device pointers unless a parameter says "host"; tensors are caller-owned;
 *  - forward calls do not allocate, do not synchronise the host and do not read device data on
 *    the host (hipGraph-capturable); all work is enqueued on the passed hipStream_t
 *    (pass torch.cuda.current_stream().cuda_stream);
 *  - one context per device/thread: thread-compatible, not thread-safe;
 *  - activations/latents at the boundary are fp32 NCHW exactly as the reference passes them;
 *    internally the engine computes in bf16 storage / fp32 accumulate.
 */
#ifndef GLIGEN_AMD_H
#define GLIGEN_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gl_ctx gl_ctx;
typedef void* gl_stream; /* hipStream_t */

/* UNetModel constructor kwargs (reference ldm/modules/diffusionmodules/openaimodel.py:238-259,
 * yaml configs/GoldG+SBU+CC3M+O365_box_text.yaml:9-29). */
typedef struct gl_unet_config {
    int in_channels, out_channels, model_channels, num_res_blocks, num_heads, context_dim;
    int n_mult;
    int channel_mult[8];
    int n_attn;
    int attention_resolutions[8];
    int inpaint_mode;     /* 9-channel first conv (openaimodel.py:299-302) */
    int grounding_kind;   /* 0 text (text_grounding_net.py), 1 text+image, 2 keypoint, 3 precomputed tokens (the spatial-map
                             tokenizers canny/hed/depth/normal/sem_grounding_net.py: ConvNeXt features computed by the caller) */
    int gr_in_dim;        /* PositionNet in_dim (768) */
    int gr_out_dim;       /* PositionNet out_dim (768) */
    int max_persons;      /* keypoint only */
    int fuser_kind;       /* BasicTransformerBlock fuser (attention.py:312-322): 0 gatedSA (every shipped discrete-token
                             config), 1 gatedSA2 (square grounding-token grid, bicubic resize), 2 gatedCA */
    int extra_channels;   /* additional_channel_from_downsampler (openaimodel.py:288-305): the first conv takes
                             in_channels + extra_channels inputs; the extra ones are the GroundingDownsampler output */
    int tok_resize;       /* grounding_kind 3: PositionNet resize_input (canny_grounding_net.py:13-22), a multiple of 32 */
    int tok_in_dim;       /* grounding_kind 3: channels of a semantic map (sem_grounding_net.py:13,21), 0 for 3-channel maps */
} gl_unet_config;

/* AutoencoderKL ddconfig (reference ldm/modules/diffusionmodules/model.py:462-533). */
typedef struct gl_vae_config {
    int ch, out_ch, z_channels, num_res_blocks, embed_dim;
    int n_mult;
    int ch_mult[8];
    float scale_factor;   /* 0.18215 (autoencoder.py:40-41) */
} gl_vae_config;

/* kwargs of PositionNet.forward for the three discrete tokenizers (reference
 * text_grounding_net.py:30, text_image_grounding_net.py:41, keypoint_grounding_net.py:34);
 * all fp32 device pointers, batch = the Beff passed alongside. */
typedef struct gl_grounding {
    int n;                       /* tokens per sample: max_objs (30) or persons*17 (136) */
    const float* boxes;          /* [Beff][n][4] xyxy in [0,1]          (kinds 0,1) */
    const float* masks;          /* [Beff][n] */
    const float* text_masks;     /* [Beff][n]                            (kind 1) */
    const float* image_masks;    /* [Beff][n]                            (kind 1) */
    const float* text_embeddings;  /* [Beff][n][in_dim] (kind 0: positive_embeddings) */
    const float* image_embeddings; /* [Beff][n][in_dim]                  (kind 1) */
    const float* points;         /* [Beff][n][2]                         (kind 2) */
    const float* tokens;         /* [Beff][n][out_dim] grounding tokens  (kind 3) */
} gl_grounding;

/* One PLMS run (reference ldm/models/diffusion/plms.py:65-162): schedule arrays are host
 * pointers of length n_steps, in sampling order (time descending). */
typedef struct gl_plms_args {
    unsigned struct_size;        /* = sizeof(gl_plms_args) of the header the caller was compiled against; anything else is
                                    rejected (GL_ERR_ARG), so a caller built against another layout fails loudly */
    int B, h, w;                 /* latent batch / size; cond must have been set with Beff = 2B
                                    ([cond ; uncond]) when guidance != 1, else Beff = B */
    int n_steps;
    const int64_t* timesteps;    /* host: flipped ddim_timesteps (plms.py:75) */
    const float* a_t;            /* host: ddim_alphas[index]      (plms.py:126) */
    const float* a_prev;         /* host: ddim_alphas_prev[index] (plms.py:127) */
    const float* fuser_scale;    /* host: alpha_generator schedule (gligen_inference.py:31-66) */
    float guidance_scale;
    float* x;                    /* device fp32 [B][C][h][w]: x_T in, x_0 out */
    const float* inpaint_extra;  /* device fp32 [B][C+1][h][w] or NULL (gligen_inference.py:406-407) */
    const float* mask;           /* device fp32 [mask_B][1][h][w] or NULL     (plms.py:96-100) */
    const float* x0;             /* device fp32 [x0_B][C][h][w] or NULL */
    const float* noise;          /* device fp32 [n_steps][noise_B][C][h][w] q_sample noise, required with mask */
    int mask_B, x0_B, noise_B;   /* batch of mask / x0 / noise: B, or 1 = broadcast over the latent batch, as the reference's
                                    q_sample(x0, ts) * mask does for the single encoded input image of run()
                                    (gligen_inference.py:396-407). 0 means B. Anything else is rejected. */
    const float* sqrt_ac;        /* host [n_steps]: sqrt_alphas_cumprod[t] (ldm.py:19-22) */
    const float* sqrt_1mac;      /* host [n_steps] */
    int use_graph;               /* capture one UNet evaluation in a hipGraph and replay it */
    const float* sd_conv_w;      /* device fp32 [mc][C][3][3] + [mc]: the SD first-conv weights, or NULL. With a fuser_scale */
    const float* sd_conv_b;      /*   schedule they are swapped in before the first step whose scale is 0 (plms.py:88-89),  */
    int sd_conv_step;            /*   whatever sd_conv_step says (a zero-initialised struct does the right thing; a value
                                      that names another step is rejected); never if no scale is 0. Without a schedule
                                      (per-fuser scales set by hand, gatedSA2) before step sd_conv_step, -1 = never. */
    int ddim;                    /* 0: PLMS (Adams-Bashforth multistep, plms.py:111-162). 1: DDIMSampler with eta = 0
                                    (reference ldm/models/diffusion/ddim.py:65-134): one evaluation per step,
                                    x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev) e_t */
} gl_plms_args;

const char* gl_last_error(void);

/* load_ckpt / instantiate_from_config(...).to(device) (reference gligen_inference.py:70-86) */
int gl_ctx_create(int device, size_t arena_bytes, gl_ctx** out);
/* A second execution context on the SAME packed weights (one more batch in flight on its own stream: bench.py --lanes,
 * gligen_inference.generate_lanes): shares everything gl_finalize packed with `parent` (which stays alive until its last fork is
 * destroyed) and owns its arena (arena_bytes; 0 = the parent's size), split-K slab, conditioning, gates, captured graphs and its
 * own copy of the restorable first conv. The parent must be finalized. The reference has one model object per process
 * (gligen_inference.py:343-446); this is the engine-side counterpart of running two of its batches at once. */
int gl_ctx_fork(gl_ctx* parent, size_t arena_bytes, gl_ctx** out);
/* device memory this context allocated itself outside the arena (packed weights + slabs; a fork: slabs only), its arena size, and
 * whether it is a fork */
int gl_ctx_memory(gl_ctx* ctx, size_t* own_bytes, size_t* arena_bytes, int* is_fork);
int gl_ctx_destroy(gl_ctx* ctx);
int gl_unet_configure(gl_ctx* ctx, const gl_unet_config* cfg);
int gl_vae_configure(gl_ctx* ctx, const gl_vae_config* cfg);
/* model.load_state_dict(...) (gligen_inference.py:81-84): key = "unet/" or "vae/" + the
 * reference state_dict key; data is fp32, host or device. */
int gl_weight_upload(gl_ctx* ctx, const char* key, const void* data, int ndim, const int64_t* shape, int is_device);
int gl_finalize(gl_ctx* ctx);

/* Step-invariant part of UNetModel.forward: position_net(**grounding_input) (openaimodel.py:433),
 * fuser.linear(objs) (attention.py:239) and attn2.to_k/to_v(context) (attention.py:130-131). */
int gl_unet_set_cond(gl_ctx* ctx, int Beff, const float* context, int n_ctx_tokens, const gl_grounding* g, gl_stream s);
/* objs = position_net(**grounding_input) (openaimodel.py:433) of the current conditioning: out fp32 [Beff][Ng][gr_out_dim]
 * (Ng = 2n for the text+image tokenizer). Parity aid: the reference exposes the same tensor as model.position_net(...). */
int gl_unet_grounding_tokens(gl_ctx* ctx, float* out, gl_stream s);
/* set_alpha_scale(model, alpha) (gligen_inference.py:24-28) */
int gl_unet_set_fuser_scale(gl_ctx* ctx, float scale, gl_stream s);
/* the fuser modules' individual `scale` attributes (attention.py:198,231,262), one host float per transformer block in module
 * order (input_blocks, middle_block, output_blocks), for callers that set them one by one instead of through set_alpha_scale */
int gl_unet_set_fuser_scales(gl_ctx* ctx, const float* scales_host, int n, gl_stream s);
/* UNetModel.restore_first_conv_from_SD (openaimodel.py:400-413): replace the first conv's weights
 * (OIHW fp32 [mc][in_channels][3][3], bias [mc]); stream-ordered, valid under hipGraph replay. */
int gl_unet_restore_first_conv(gl_ctx* ctx, const float* w, const float* b, gl_stream s);
/* UNetModel.forward (openaimodel.py:420-464): x [xB][C][h][w] (sample b reads x[b % xB]),
 * timesteps int64 [Beff], eps_out [Beff][out_ch][h][w]. inpaint_extra = the channels concatenated to x in front of the
 * first conv (openaimodel.py:442-447): inpainting_extra_input [extraB][C+1][h][w] for inpaint_mode, the GroundingDownsampler
 * output [extraB][extra_channels][h][w] (gl_op_grounding_downsample) for the spatial-map modalities, else NULL. */
int gl_unet_forward(gl_ctx* ctx, int Beff, int h, int w, const float* x, int xB, const int64_t* timesteps,
                    const float* inpaint_extra, int extraB, float* eps_out, gl_stream s);
/* AutoencoderKL.decode (autoencoder.py:40-44): z [B][zc][h][w] -> img [B][out_ch][8h][8w] */
int gl_vae_decode(gl_ctx* ctx, int B, int h, int w, const float* z, float* img, gl_stream s);
/* AutoencoderKL.encode (autoencoder.py:34-38; inpainting, gligen_inference.py:402-407): img [B][3][H][W] fp32 in [-1,1]
 * -> z [B][zc][H/8][W/8] = (mean + std * noise) * scale_factor. `noise` [B][zc][H/8][W/8] is the caller's draw for the
 * reference's torch.randn(mean.shape) (distributions.py:35). Needs the encoder.* and quant_conv.* weights uploaded. */
int gl_vae_encode(gl_ctx* ctx, int B, int H, int W, const float* img, const float* noise, float* z, gl_stream s);
/* PLMSSampler.plms_sampling (plms.py:65-108) / DDIMSampler.ddim_sampling (ddim.py:65-108, args->ddim) with
 * classifier-free guidance (plms.py:116-122, ddim.py:113-117) */
int gl_sample_plms(gl_ctx* ctx, const gl_plms_args* args, gl_stream s);
/* HIP-event timing of the UNet evaluations of the last gl_sample_plms call, measured on the stream
 * they ran on: mean over the graph-replayed evaluations, the first (eager) one, and their count. */
int gl_sampler_timing(gl_ctx* ctx, float* avg_unet_eval_ms, float* first_eval_ms, int* n_evals);
/* clamp(-1,1)*0.5+0.5 -> *255 -> uint8 HWC (gligen_inference.py:443-445) */
int gl_to_uint8(const float* img, uint8_t* out, int B, int C, int HW, gl_stream s);

/* Per-kernel profile of ONE eager UNetModel.forward (same arguments as gl_unet_forward): HIP events are recorded on
 * `s` around every GEMM / conv / attention / norm launch and aggregated by kernel symbol, longest total first. flops /
 * bytes are the ALGORITHMIC work of those launches (2 M N K; 4 B H Nq Nk d; read-once + write-once bytes for the
 * norms). Synchronises the stream; measurement aid for bench.py's roofline block, not part of the reference's surface. */
typedef struct gl_prof_rec {
    char name[96];
    int calls;
    double ms, flops, bytes;
} gl_prof_rec;
int gl_unet_profile(gl_ctx* ctx, int Beff, int h, int w, const float* x, int xB, const int64_t* timesteps,
                    const float* inpaint_extra, int extraB, float* eps_out, gl_prof_rec* recs, int max_recs, int* n_recs,
                    gl_stream s);

int gl_arena_high_water(gl_ctx* ctx, size_t* bytes);
int gl_launch_count(gl_ctx* ctx, int64_t* n);

/* FeedForward + the LayerNorm in front of it (+ the C x C projections either side: attention.py:37-64, 236-244, 333-338, 374-376) exist
 * in two forms at C = 320: ONE row-local launch (ffn.hip) or LayerNorm + GEGLU GEMM + FF-out GEMM (+ separate projections). Which is
 * faster depends on the row count and on the box (the row-local form ran 1.65x slower on a slow-fabric box), so the
 * engine TIMES both on the device at the first eager launch of each (block form, row count) and keeps the faster for the process.
 * mode: -1 = that (default), 0 = never the row-local kernel, 1 = the row-local kernel wherever it exists, 2 = a static rule (launches of
 * at least 7/8 of the chip's CUs take the row-local kernel): the two forms differ in their last bits, so under -1 identical inputs may
 * give bitwise different outputs on another box / process; 0, 1 and 2 select the same kernels everywhere. A shape first met inside a
 * stream capture is decided by the static rule and the decision is kept for the process. Process-wide; contexts drop
 * their captured graphs when it changes. gl_ff_rows_policy_report: "mode=..;<form> C M -> rows|gemm (rows us, gemm us);.." of what
 * has been timed so far. Measurement aids (bench.py's same-box A/B); the reference has no counterpart. */
int gl_set_ff_rows_policy(int mode);
int gl_ff_rows_policy_report(char* buf, size_t cap);

/* What this box delivers, measured in ~50 ms on ctx's device (uses 1 GiB of the arena): a float4 copy far beyond the Infinity Cache
 * (read + written GB/s), the global->LDS DMA path from an L2-resident window (TB/s over all CUs: the operand-delivery path of the
 * GEMM / conv / attention kernels) and the sustained dense bf16 MFMA rate with every SIMD issuing (TFLOP/s). bench.py prints them
 * beside images/s so that box-to-box spread of one commit is evidence, not a sentence. */
typedef struct gl_box_calibration {
    float hbm_copy_gbs;
    float lds_dma_tbs;
    float mfma_bf16_tflops;
} gl_box_calibration;
int gl_box_calibrate(gl_ctx* ctx, gl_box_calibration* out, gl_stream s);
/* The dense bf16 MFMA ceiling of THIS box with the clock that explains it: every SIMD issues independent MFMAs back to back for
 * target_ms (>= 20 ms: long enough for the power management to settle) -- shape 0 = v_mfma_f32_32x32x16_bf16, 1 = 16x16x32; 1 or 2 waves
 * per SIMD; 4 or 8 independent accumulators per wave; zero_data = 1 multiplies zeros (no operand toggling: the chip clocks higher).
 * The kernel reads its own shader clock over the loop (s_memtime against the constant 100 MHz s_memrealtime), so tflops / 2500 is
 * explained by sclk_mhz / 2400 and by cycles_per_mfma (32 for 32x32x16, 16 for 16x16x32 when the pipe is fully paced). A measurement
 * aid like gl_box_calibrate (bench.py prints the table as box_calibration.mfma); the reference has no counterpart. */
typedef struct gl_mfma_calibration {
    float tflops;
    float sclk_mhz;
    float ms;
    float cycles_per_mfma;
} gl_mfma_calibration;
int gl_mfma_calibrate(gl_ctx* ctx, int shape, int waves_per_simd, int n_acc, int zero_data, float target_ms, gl_mfma_calibration* out, gl_stream s);

/* PositionNet.forward of the spatial-map modalities (reference canny_/hed_/depth_/normal_/sem_grounding_net.py:38-62):
 * image fp32 [B][C][H][W] (the map as RGB in [-1,1], or in_dim one-hot planes), mask fp32 [B] -> nearest resize to
 * resize_input -> (sem: Conv2d(in_dim,3,3,1,1)) -> ConvNeXt-tiny (convnext.py) -> null-feature mixing, + pos_embedding,
 * 3-layer SiLU MLP -> tokens fp32 [B][(resize_input/32)^2][gr_out_dim], the gl_grounding.tokens of gl_unet_set_cond.
 * Needs the position_net.* weights uploaded; once per prompt. */
int gl_op_spatial_tokens(gl_ctx* ctx, const float* image, int B, int C, int H, int W, const float* mask, float* tokens, gl_stream s);
/* GroundingDownsampler.forward of the spatial-map modalities (reference canny_/depth_/normal_/sem_/hed_grounding_downsampler.py):
 * img fp32 [B][Cimg][H][W] -> first n_in channels resized to R x R (mode 0 bicubic, 1 nearest, as F.interpolate) ->
 * Conv2d(n_in, c_mid, 4, 2, 1) -> SiLU -> Conv2d(c_mid, c_out, 4, 2, 1) -> out fp32 [B][c_out][R/4][R/4].
 * w1 = NULL: no layers (hed: the resized map itself, out [B][n_in][R][R]). Step-invariant: once per prompt. */
int gl_op_grounding_downsample(gl_ctx* ctx, const float* img, int B, int Cimg, int H, int W, int n_in, int R, int mode,
                               const float* w1, const float* b1, int c_mid, const float* w2, const float* b2, int c_out,
                               float* out, gl_stream s);

/* ---- single-operator entry points (parity tests and per-kernel profiling) -------------- */
/* y = act(x W^T + b) [+ res]; x [M][K] bf16, W [N][K] bf16, b fp32|NULL, res bf16|NULL,
 * act: 0 none 1 SiLU; y bf16 (out_f32 = 0) or fp32. Replaces nn.Linear / 1x1 conv. */
int gl_op_linear(gl_ctx* ctx, const void* x, const void* w, const float* bias, const void* res, void* y,
                 int M, int N, int K, int act, int out_f32, gl_stream s);
/* GEGLU (reference attention.py:37-44): x [M][K] bf16, proj weight [2*inner][K] fp32 + bias ->
 * y [M][inner] bf16 = (x Wv^T + bv) * gelu(x Wg^T + bg) */
int gl_op_geglu(gl_ctx* ctx, const void* x, const float* w_f32, const float* b_f32, void* y, int M, int inner, int K, gl_stream s);
/* A LayerNorm folded into the projection behind it, as the engine runs the transformer blocks (reference attention.py:333-338,
 * x = x + f(norm(x))): first the PRODUCER x = a W0^T + b0 (+ res) [M][C] bf16, whose epilogue also emits the rows' partial
 * (sum, sum of squares); then the CONSUMER on the raw rows of x with weights W1 * gamma, bias b1 + W1 beta and the correction
 * rstd (acc - mean csum) in its epilogue:
 *   mode 0: y [M][inner] = GEGLU(LN(x) W1^T + b1), W1 [2*inner][C]   (feed-forward, attention.py:37-64)
 *   mode 1: y [M][C] = LN(x) W1^T (+ b1), W1 [C][C], written in the q head layout [M / T * heads][T][DP] (attn2.to_q)
 * a [M][K0] bf16, W0 [C][K0] / W1 fp32, b0 / b1 / res may be NULL. *used_fold = 1 if the statistics path ran, 0 if the launch
 * fell back to ln_kernel without affine + the same folded weights (both are product paths). eps 1e-5. */
int gl_op_ln_linear(gl_ctx* ctx, const void* a, int M, int K0, const float* w0, const float* b0, const void* res, int C,
                    const float* gamma, const float* beta, const float* w1, const float* b1, int mode, int inner_or_heads, int T,
                    void* x_out, void* y_out, int* used_fold, gl_stream s);
/* The whole feed-forward of a transformer block as the engine runs it (reference attention.py:37-64 FeedForward / GEGLU,
 * :333-338 x = ff(norm3(x)) + x, :236-244 x = x + scale*tanh(alpha_dense) * ff(norm2(x))):
 *   y = res + gate * ( GEGLU( LN(x) W1^T + b1 ) W2^T + b2 ),  LN over C with gamma / beta (NULL: x is multiplied as it is)
 * x / res / y [M][C] bf16 (res may be NULL), W1 [8C][C] fp32 (value rows, then gate rows), b1 [8C], W2 [C][4C] fp32, b2 [C],
 * gate: device scalar or NULL (1). stats: optional [M] float2 (sum, sum of squares) of each output row -- written by the row-local
 * kernel only: with stats != NULL a shape that kernel does not cover is refused (GL_ERR_UNSUPPORTED), not answered with zeros.
 * *used_rows = 1: the row-local kernel ran (one launch, hidden activation on chip: C = 320, M % 128 == 0); 0: LayerNorm kernel +
 * GEGLU GEMM + FF-out GEMM (both are product paths). */
int gl_op_feedforward(gl_ctx* ctx, const void* x, int M, int C, const float* gamma, const float* beta, const float* w1, const float* b1,
                      const float* w2, const float* b2, const void* res, const float* gate, void* y, void* stats, int* used_rows,
                      gl_stream s);
/* The chained row-local launch of a transformer block's tail (reference attention.py:236-244 fuser, :337-338 + :374-376):
 *   t = pre_res + pre_gate * (x Wpre^T + pre_b)         (an attention's to_out + (gated) residual)
 *   u = t + gate * ( GEGLU( LN(t) W1^T + b1 ) W2^T + b2 )
 *   y = u,  or with post_w:  post_res + (u Wpost^T + post_b)          (SpatialTransformer.proj_out + x_in)
 * x / pre_res / post_res / y [M][C] bf16, Wpre / Wpost [C][C] fp32, the feed-forward as in gl_op_feedforward (gamma / beta required),
 * pre_gate / gate device scalars or NULL (1). Needs the row-local kernel (C = 320, M % 128 == 0): GL_ERR_UNSUPPORTED otherwise. */
int gl_op_ff_chain(gl_ctx* ctx, const void* x, int M, int C, const float* pre_w, const float* pre_b, const void* pre_res, const float* pre_gate,
                   const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2, const float* b2, const float* gate,
                   const float* post_w, const float* post_b, const void* post_res, void* y, gl_stream s);

/* The fuser's chained launch with the cross-attention's query projection as its trailing projection (reference attention.py:236-244
 * GatedSelfAttentionDense, then :336 attn2(norm2(x)): to_q of the NEXT LayerNorm):
 *   t = pre_res + pre_gate * (x Wpre^T + pre_b);  y = t + gate * ff(LN(t; gamma, beta))  -> y [B][N][C] bf16 (the residual stream)
 *   q = LN(y; gamma_q, beta_q) Wq^T                                                       -> q [B * 8][N][48] bf16 (head layout of the
 *       attention kernels at d = 40: head-major, 48-column rows whose columns 40..47 are left untouched)
 * C = 320, N % 128 == 0; Wq [C][C] fp32 (no bias). GL_ERR_UNSUPPORTED otherwise. */
int gl_op_ff_chain_q(gl_ctx* ctx, const void* x, int B, int N, int C, const float* pre_w, const float* pre_b, const void* pre_res, const float* pre_gate,
                     const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2, const float* b2, const float* gate,
                     const float* gamma_q, const float* beta_q, const float* wq, void* y, void* q, gl_stream s);

/* ---- training slice: one BasicTransformerBlock, forward + backward ---------------------------------
 * Forward of the reference's BasicTransformerBlock with a gatedSA fuser (ldm/modules/attention.py:333-338, 236-244), the
 * reference's loss on its output (trainer.py:366: mse_loss(model_output, noise)) and the backward pass, with the gradients the
 * reference's optimizer holds for this block (trainer.py:217-245: every fuser.* parameter) plus d loss / d x and d loss / d objs
 * so that the step chains into position_net and the blocks in front. All tensors fp32 on the device:
 *   x [B][N][C], objs [B][Ng][ctx_dim], context [B][ctx_T][ctx_dim], target [B][N][C]  ->  y [B][N][C], loss[1], dx, dobjs.
 * params[GL_TRAIN_BLOCK_PARAMS]: the block's state_dict in the order of gl_train_block_param_names(); grads[]: same order, one
 * buffer shaped like the parameter for every fuser.* entry wanted (NULL: not computed; non-fuser entries must be NULL -- those
 * layers are frozen in the reference and their weight gradients are never formed). */
#define GL_TRAIN_BLOCK_PARAMS 37
typedef struct gl_train_block_dims {
    int B, N, Ng, C, heads, ctx_T, ctx_dim;
    float fuser_scale;               /* GatedSelfAttentionDense.scale */
} gl_train_block_dims;
const char* const* gl_train_block_param_names(void);   /* GL_TRAIN_BLOCK_PARAMS reference state_dict keys, e.g. "fuser.attn.to_q.weight" */
int gl_op_block_train(gl_ctx* ctx, const gl_train_block_dims* dims, const float* const* params, const float* x, const float* objs,
                      const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* grads,
                      gl_stream s);

/* SpatialTransformer.forward (reference ldm/modules/attention.py:340-376) around that block: x + proj_out(block(proj_in(norm(x)))),
 * same tensors and gradients as gl_op_block_train with x / target / y / dx as pixel rows [B][H*W][C] (dims->N = H*W, dims->C =
 * in_channels = heads * d_head). params / grads [GL_TRAIN_ST_PARAMS] in the order of gl_train_st_param_names(): norm.*, proj_in.*,
 * transformer_blocks.0.<the block's names>, proj_out.*; gradients only for transformer_blocks.0.fuser.* entries. */
#define GL_TRAIN_ST_PARAMS (GL_TRAIN_BLOCK_PARAMS + 6)
const char* const* gl_train_st_param_names(void);
int gl_op_st_train(gl_ctx* ctx, const gl_train_block_dims* dims, const float* const* params, const float* x, const float* objs,
                   const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* grads, gl_stream s);

/* The UNet's other block type: forward + backward of one ResBlock (reference ldm/modules/diffusionmodules/openaimodel.py:154-232,
 * no up / down, no scale-shift norm) under the same loss. Every ResBlock parameter is frozen in the reference's trainer
 * (trainer.py:217-245), so its backward is the gradient w.r.t. its input -- the path by which the loss reaches the fusers in front.
 * Tensors fp32 on the device, rows = pixels (this library's layout; the reference's NCHW is a permutation of it):
 *   x [B][H*W][Cin], emb [B][emb_dim], target [B][H*W][Cout]  ->  y [B][H*W][Cout], loss[1] = mse_loss(y, target), dx [B][H*W][Cin].
 * params[GL_TRAIN_RESBLOCK_PARAMS]: the block's state_dict in the order of gl_train_resblock_param_names(); the two
 * skip_connection.* entries are NULL when Cin == Cout (nn.Identity) and required otherwise (1 x 1 conv: weight [Cout][Cin][1][1]). */
#define GL_TRAIN_RESBLOCK_PARAMS 12
typedef struct gl_train_resblock_dims {
    int B, H, W, Cin, Cout, emb_dim;
} gl_train_resblock_dims;
const char* const* gl_train_resblock_param_names(void);
int gl_op_resblock_train(gl_ctx* ctx, const gl_train_resblock_dims* dims, const float* const* params, const float* x, const float* emb,
                         const float* target, float* y, float* loss, float* dx, gl_stream s);

/* Downsample (mode 0: conv3x3 stride 2, reference openaimodel.py:99-124) / Upsample (mode 1: nearest 2x + conv3x3, :64-96) of C channels:
 * forward, mse_loss(y, target) and the gradient w.r.t. the input (the conv is a frozen SD layer; w OIHW fp32 [C][C][3][3]).
 * x / dx [B][H*W][C], y / target [B][Ho*Wo][C] fp32 pixel rows, Ho = H/2 (mode 0) or 2H (mode 1). */
int gl_op_resample_train(gl_ctx* ctx, int mode, int B, int H, int W, int C, const float* w_oihw, const float* bias, const float* x,
                         const float* target, float* y, float* loss, float* dx, gl_stream s);

/* ---- one whole training iteration (reference trainer.py:353-392: model(input), mse_loss(model_output, noise), loss.backward()) for a
 * UNetModel with the text, the text+image or the keypoint grounding tokenizer and gatedSA fusers (openaimodel.py:237-464; gl_unet_config with
 * grounding_kind 0 / 1 / 2, fuser_kind 0, no inpainting / extra channels). The model's parameters come as its state_dict: n_params names (the reference's
 * keys, e.g. "input_blocks.1.1.transformer_blocks.0.fuser.linear.weight") with fp32 device pointers; grads[i] is a buffer shaped
 * like parameter i for every trainable parameter wanted -- the reference's trainable set is every "*.fuser.*" key and "position_net.*"
 * (trainer.py:217-245); any other non-NULL entry is rejected -- and NULL elsewhere. Tensors fp32 on the device, x / target / eps_out
 * as pixel rows [B][H*W][channels] (eps_out optional). loss[1] = mse_loss(eps, target). */
typedef struct gl_train_unet_in {
    int B, H, W, ctx_T, Ng;
    const float* x;                     /* [B][H*W][in_channels]: the noised latent */
    const float* timesteps;             /* [B], as float */
    const float* context;               /* [B][ctx_T][context_dim] */
    const float* boxes;                 /* [B][Ng][4]; keypoint tokenizer (grounding_kind 2, keypoint_grounding_net.py:34): points [B][Ng][2], Ng = 17 per person */
    const float* masks;                 /* [B][Ng] */
    const float* positive_embeddings;   /* [B][Ng][gr_in_dim]: positive_embeddings (text tokenizer) / text_embeddings (text+image); NULL for keypoints */
    const float* target;                /* [B][H*W][out_channels]: the noise */
    float fuser_scale;
    /* text+image tokenizer (grounding_kind 1, text_image_grounding_net.py:41): NULL for the text tokenizer. The model then sees 2 * Ng
     * grounding tokens per sample (text tokens, then image tokens) */
    const float* text_masks;            /* [B][Ng] */
    const float* image_masks;           /* [B][Ng] */
    const float* image_embeddings;      /* [B][Ng][gr_in_dim] */
    int checkpoint;                     /* activation checkpointing per ResBlock / SpatialTransformer: 1 = keep block inputs and outputs only and
                                           recompute a block's forward in its backward (same gradients bit for bit, a fraction of the memory) */
    int use_weight_cache;               /* 1: take / leave the frozen parameters' operand copies in the context's cache (gl_train_weight_cache) */
} gl_train_unet_in;
int gl_unet_train_step(gl_ctx* ctx, const gl_unet_config* cfg, const gl_train_unet_in* in, int n_params, const char* const* names,
                       const float* const* params, float* const* grads, float* eps_out, float* loss, gl_stream s);

/* Operand copies of the FROZEN parameters across training steps (reference trainer.py:217-245: only fuser.* / position_net.* are ever
 * updated). gl_unet_train_step multiplies every fp32 weight as (hi | hi | lo) bf16 operands -- rows for the forward, transposes for the
 * data gradient, packed / flipped filters for the convs --, built per product. With the cache enabled those copies are built once for every
 * parameter whose grads[i] is NULL and reused by later steps of this context that set gl_train_unet_in.use_weight_cache: the caller
 * promises to change only the parameters it asks gradients for, and that a frozen tensor's device address means the same values for as
 * long as the cache lives (enable = 0 frees the copies: after loading other weights into the same buffers, or before buffers are
 * re-allocated). *bytes (optional): device memory the cache holds now. */
int gl_train_weight_cache(gl_ctx* ctx, int enable, size_t* bytes);

/* Gradient milestones of the last gl_unet_train_step on this context: make stream `s` wait until the fuser gradients of the
 * index-th SpatialTransformer (module order: input_blocks .., middle_block, output_blocks ..) are written -- the backward runs from the
 * last block to the first, so the last blocks' gradients are final long before the step ends -- or, with index = the number of
 * SpatialTransformers, until position_net's (the step's last) are. The training step itself is enqueued on its own stream and
 * returns at once; a communication stream that waits here can start a bucket's reduce-scatter while the rest of the backward still
 * runs, which is what DistributedDataParallel's bucket hooks do under loss.backward() (reference trainer.py:321-322, 366-384). */
int gl_train_wait_grads(gl_ctx* ctx, int index, gl_stream s);

/* One AdamW step over a flat fp32 range, in place: p, exp_avg m, exp_avg_sq v [n]; g the (all-reduced) gradient; step counts from 1.
 * torch.optim.AdamW semantics -- the reference's optimizer over the trainable set (trainer.py:245, opt.step() at :384) -- down to where
 * the scalars are formed: lr, the betas, eps and weight_decay arrive as doubles (Python floats), 1 - beta^step, lr / (1 - beta1^step),
 * sqrt(1 - beta2^step) and 1 - lr weight_decay are computed in double and rounded to fp32 once, as torch does. */
int gl_op_adamw_step(gl_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int step, gl_stream s);

/* 3x3 conv over NHWC bf16 (channel-concat of x0,x1), weight OIHW fp32, stride 1|2, optional
 * nearest 2x upsample of the input, pad_lo 1 (symmetric) or 0 (VAE-encoder style). y NHWC bf16. */
int gl_op_conv3x3(gl_ctx* ctx, const void* x0, int C0, const void* x1, int C1, int B, int H, int W,
                  const float* w_oihw, const float* bias, int Cout, int stride, int ups, int pad_lo,
                  const void* res, void* y, gl_stream s);
/* GroupNorm32 -> SiLU -> conv3x3 (stride 1, pad 1) over NHWC bf16 (channel-concat of x0,x1): the in_layers / out_layers of the
 * reference's ResBlock (ldm/modules/diffusionmodules/openaimodel.py:212-232; VAE ResnetBlock model.py:118-141) as one operator.
 * bias2: optional [B][Cout] per-sample bias (the time-embedding term), res: optional residual [B*H*W][Cout] bf16 (never both).
 * Two forms. Separate pass: GroupNorm + SiLU write a normalised copy that the conv reads (two kernels + conv). Prologue: where the
 * conv runs on conv_halo_kernel (M >= 2048, power-of-two images up to 64 wide with H*W % 256 == 0) GroupNorm keeps only its
 * statistics pass and the conv normalises + activates its input tile while staging it in LDS -- no normalised copy in HBM.
 * mode -1: the engine's choice (the separate pass: on MI355X it is the faster form at every UNet shape);
 * mode 0: the separate pass; mode 1: the prologue form or GL_ERR_UNSUPPORTED. *used_prologue tells which form ran. Same
 * coefficients, same arithmetic: the two outputs are bit-identical for H*W > 256. */
int gl_op_gn_silu_conv3x3(gl_ctx* ctx, const void* x0, int C0, const void* x1, int C1, int B, int H, int W, const float* gamma, const float* beta,
                          float eps, const float* w_oihw, const float* bias, int Cout, const float* bias2, const void* res, void* y, int mode,
                          int* used_prologue, gl_stream s);
/* GroupNorm(32)+optional SiLU over NHWC bf16 (channel-concat of x0,x1) */
int gl_op_groupnorm(gl_ctx* ctx, const void* x0, int C0, const void* x1, int C1, int B, int HW,
                    const float* gamma, const float* beta, float eps, int silu, void* y, gl_stream s);
int gl_op_layernorm(gl_ctx* ctx, const void* x, const void* x2, int B, int N1, int N2, int Tpad, int C,
                    const float* gamma, const float* beta, float eps, void* y, gl_stream s);
/* softmax(q k^T d^-0.5) v with q,k,v projections applied to token rows:
 * xq [B][Nq][C], xkv [B][Nk][Ck] bf16, Wq [C][C], Wk/Wv [C][Ck] fp32 (no bias), heads H -> o [B][Nq][C] bf16 */
int gl_op_attention(gl_ctx* ctx, const void* xq, const void* xkv, int B, int Nq, int Nk, int C, int Ck, int H,
                    const float* wq, const float* wk, const float* wv, void* o, gl_stream s);

/* A projection, the LayerNorm behind it and the self-attention behind that, as the engine runs them at the 64 x 64 level (reference
 * ldm/modules/attention.py:366-368 SpatialTransformer.proj_in -> :335 norm1 -> attn1, and :183-186 attn1.to_out + residual ->
 * :240 fuser.norm1 -> fuser.attn):
 *   mid = pre_res + (x Wpre^T + pre_b)              (pre_res may be NULL)
 *   o   = softmax(q k^T d^-0.5) v,  q / k / v = LN(mid; gamma, beta) Wq^T / Wk^T / Wv^T     (no to_out)
 * x / pre_res / mid / o [B][N][C] bf16, Wpre / Wq / Wk / Wv [C][C] fp32, N % 64 == 0, Nkv_extra must be 0.
 * *used_rows in: 1 = take the row-local projection kernel (qkv_rows_kernel: one launch for the projection, the LayerNorm and q,k,v^T;
 * C = 320, H = 8, B N % 128 == 0, N % 128 == 0, N > 128) where it exists, 0 = the three-launch form (GEMM, LayerNorm kernel, fused q,k,v^T
 * GEMM); out: which one ran. Both are product paths. */
int gl_op_proj_attention(gl_ctx* ctx, const void* x, int B, int N, int C, int H, const float* pre_w, const float* pre_b, const void* pre_res,
                         const float* gamma, const float* beta, const float* wq, const float* wk, const float* wv, int Nkv_extra,
                         void* mid, void* o, int* used_rows, gl_stream s);

#ifdef __cplusplus
}
#endif
#endif
