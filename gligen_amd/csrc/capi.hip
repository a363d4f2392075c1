// extern "C" surface of libgligen_amd.so (see include/gligen_amd.h). Nothing throws across it.
#include "engine.h"
#include "train.h"

#include <cstdlib>

#include <cmath>

using namespace gl;

struct gl_ctx {
    std::shared_ptr<Engine> holder;   // forks (gl_ctx_fork) keep the context whose weights they share alive
    Engine* eng;
};

#define GL_API_BEGIN try {
#define GL_API_END                                       \
    }                                                    \
    catch (const gl::GlError& e) {                       \
        return gl::set_error(e.code, "%s", e.what());    \
    }                                                    \
    catch (const std::exception& e) {                    \
        return gl::set_error(GL_ERR_STATE, "%s", e.what()); \
    }                                                    \
    return GL_OK;

// every entry point runs on the context's own device, whatever the caller's current device is, and hands the caller's
// current device back on return (a process may hold engines on several GPUs next to torch's own notion of "current")
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
        else prev = -1;   // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define NEED(ctx)                                                         \
    if (!(ctx) || !(ctx)->eng) return gl::set_error(GL_ERR_ARG, "null context"); \
    DeviceGuard device_guard_((ctx)->eng->device());                      \
    if (!device_guard_.ok) return gl::set_error(GL_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->eng->device());

static inline hipStream_t S(gl_stream s) { return reinterpret_cast<hipStream_t>(s); }
#define HIPCK_API(expr)                                                                           \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) throw GlError(GL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

extern "C" {

const char* gl_last_error(void) { return gl::last_error(); }

int gl_ctx_create(int device, size_t arena_bytes, gl_ctx** out) {
    if (!out) return gl::set_error(GL_ERR_ARG, "null out pointer");
    GL_API_BEGIN
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        throw GlError(GL_ERR_HIP, "no HIP device available: libgligen_amd has no CPU path");
    if (device < 0 || device >= ndev) throw GlError(GL_ERR_ARG, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) throw GlError(GL_ERR_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) throw GlError(GL_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw GlError(GL_ERR_UNSUPPORTED, std::string("libgligen_amd is built for gfx950 (MI355X); device is ") + prop.gcnArchName);
    std::shared_ptr<Engine> eng(new Engine(device));
    eng->arena().init(arena_bytes ? arena_bytes : (size_t(8) << 30));
    eng->init_workspace();
    *out = new gl_ctx{eng, eng.get()};
    GL_API_END
}

int gl_ctx_fork(gl_ctx* parent, size_t arena_bytes, gl_ctx** out) {
    NEED(parent);
    if (!out) return gl::set_error(GL_ERR_ARG, "null out pointer");
    GL_API_BEGIN
    std::shared_ptr<Engine> e = Engine::fork(parent->holder, arena_bytes);
    *out = new gl_ctx{e, e.get()};
    GL_API_END
}

int gl_ctx_memory(gl_ctx* ctx, size_t* own_bytes, size_t* arena_bytes, int* is_fork) {
    NEED(ctx);
    GL_API_BEGIN
    if (own_bytes) *own_bytes = ctx->eng->weight_bytes();
    if (arena_bytes) *arena_bytes = ctx->eng->arena_bytes();
    if (is_fork) *is_fork = ctx->eng->is_fork() ? 1 : 0;
    GL_API_END
}

int gl_ctx_destroy(gl_ctx* ctx) {
    if (!ctx) return GL_OK;
    GL_API_BEGIN
    (void)hipDeviceSynchronize();
    delete ctx;      // the engine goes with its last holder: a parent outlives its forks
    GL_API_END
}

int gl_unet_configure(gl_ctx* ctx, const gl_unet_config* cfg) {
    NEED(ctx);
    if (!cfg) return gl::set_error(GL_ERR_ARG, "null config");
    GL_API_BEGIN
    ctx->eng->configure_unet(*cfg);
    GL_API_END
}

int gl_vae_configure(gl_ctx* ctx, const gl_vae_config* cfg) {
    NEED(ctx);
    if (!cfg) return gl::set_error(GL_ERR_ARG, "null config");
    GL_API_BEGIN
    ctx->eng->configure_vae(*cfg);
    GL_API_END
}

int gl_weight_upload(gl_ctx* ctx, const char* key, const void* data, int ndim, const int64_t* shape, int is_device) {
    NEED(ctx);
    if (!key || !data || ndim < 0 || (ndim > 0 && !shape)) return gl::set_error(GL_ERR_ARG, "bad weight upload arguments");
    GL_API_BEGIN
    ctx->eng->upload(key, data, ndim, shape, is_device != 0);
    GL_API_END
}

int gl_finalize(gl_ctx* ctx) {
    NEED(ctx);
    GL_API_BEGIN
    ctx->eng->finalize();
    GL_API_END
}

int gl_unet_set_cond(gl_ctx* ctx, int Beff, const float* context, int n_ctx_tokens, const gl_grounding* g, gl_stream s) {
    NEED(ctx);
    if (!context || !g) return gl::set_error(GL_ERR_ARG, "null context/grounding");
    GL_API_BEGIN
    ctx->eng->set_cond(Beff, context, n_ctx_tokens, *g, S(s));
    GL_API_END
}

int gl_unet_set_fuser_scales(gl_ctx* ctx, const float* scales_host, int n, gl_stream s) {
    NEED(ctx);
    if (!scales_host) return gl::set_error(GL_ERR_ARG, "null scales");
    GL_API_BEGIN
    ctx->eng->set_fuser_scales(scales_host, n, S(s));
    GL_API_END
}

int gl_unet_grounding_tokens(gl_ctx* ctx, float* out, gl_stream s) {
    NEED(ctx);
    if (!out) return gl::set_error(GL_ERR_ARG, "null out pointer");
    GL_API_BEGIN
    ctx->eng->grounding_tokens(out, S(s));
    GL_API_END
}

int gl_op_spatial_tokens(gl_ctx* ctx, const float* image, int B, int C, int H, int W, const float* mask, float* tokens, gl_stream s) {
    NEED(ctx);
    if (!image || !mask || !tokens) return gl::set_error(GL_ERR_ARG, "null pointer");
    GL_API_BEGIN
    ctx->eng->spatial_tokens(B, image, C, H, W, mask, tokens, S(s));
    GL_API_END
}

int gl_op_grounding_downsample(gl_ctx* ctx, const float* img, int B, int Cimg, int H, int W, int n_in, int R, int mode,
                               const float* w1, const float* b1, int c_mid, const float* w2, const float* b2, int c_out,
                               float* out, gl_stream s) {
    NEED(ctx);
    if (!img || !out || B <= 0 || n_in <= 0 || n_in > Cimg || R <= 0) return gl::set_error(GL_ERR_ARG, "grounding_downsample: bad arguments");
    if (w1 && (!b1 || !w2 || !b2 || c_mid <= 0 || c_out <= 0 || R % 4)) return gl::set_error(GL_ERR_ARG, "grounding_downsample: bad conv arguments");
    GL_API_BEGIN
    if (!w1) {
        GL_TRY(gl::resize_f32_launch(img, out, B, Cimg, n_in, H, W, R, mode, S(s)));
    } else {
        gl::Arena& ar = ctx->eng->arena();
        const size_t mk = ar.mark();
        float* r = ar.get<float>((size_t)B * n_in * R * R);
        float* h1 = ar.get<float>((size_t)B * c_mid * (R / 2) * (R / 2));
        int rc = gl::resize_f32_launch(img, r, B, Cimg, n_in, H, W, R, mode, S(s));
        if (rc == GL_OK) rc = gl::conv4x4s2_f32_launch(r, w1, b1, h1, B, n_in, c_mid, R, R, 1, S(s));
        if (rc == GL_OK) rc = gl::conv4x4s2_f32_launch(h1, w2, b2, out, B, c_mid, c_out, R / 2, R / 2, 0, S(s));
        ar.release(mk);
        if (rc != GL_OK) return rc;
    }
    GL_API_END
}

int gl_unet_set_fuser_scale(gl_ctx* ctx, float scale, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    ctx->eng->set_fuser_scale(scale, S(s));
    GL_API_END
}

int gl_unet_restore_first_conv(gl_ctx* ctx, const float* w, const float* b, gl_stream s) {
    NEED(ctx);
    if (!w || !b) return gl::set_error(GL_ERR_ARG, "null weights");
    GL_API_BEGIN
    ctx->eng->restore_first_conv(w, b, S(s));
    GL_API_END
}

int gl_unet_forward(gl_ctx* ctx, int Beff, int h, int w, const float* x, int xB, const int64_t* timesteps,
                    const float* inpaint_extra, int extraB, float* eps_out, gl_stream s) {
    NEED(ctx);
    if (!x || !timesteps || !eps_out || Beff <= 0 || h <= 0 || w <= 0) return gl::set_error(GL_ERR_ARG, "bad unet_forward arguments");
    GL_API_BEGIN
    ctx->eng->unet_forward(Beff, h, w, x, xB, timesteps, inpaint_extra, extraB, eps_out, S(s));
    GL_API_END
}

int gl_vae_decode(gl_ctx* ctx, int B, int h, int w, const float* z, float* img, gl_stream s) {
    NEED(ctx);
    if (!z || !img || B <= 0 || h <= 0 || w <= 0) return gl::set_error(GL_ERR_ARG, "bad vae_decode arguments");
    GL_API_BEGIN
    ctx->eng->vae_decode(B, h, w, z, img, S(s));
    GL_API_END
}

int gl_sample_plms(gl_ctx* ctx, const gl_plms_args* args, gl_stream s) {
    NEED(ctx);
    if (!args) return gl::set_error(GL_ERR_ARG, "null args");
    if (args->struct_size != sizeof(gl_plms_args))
        return gl::set_error(GL_ERR_ARG, "gl_sample_plms: args->struct_size is %u, this library's gl_plms_args has %zu bytes (set it to sizeof(gl_plms_args))",
                             args->struct_size, sizeof(gl_plms_args));
    GL_API_BEGIN
    ctx->eng->sample_plms(*args, S(s));
    GL_API_END
}

int gl_sampler_timing(gl_ctx* ctx, float* avg_unet_eval_ms, float* first_eval_ms, int* n_evals) {
    NEED(ctx);
    if (!avg_unet_eval_ms || !first_eval_ms || !n_evals) return gl::set_error(GL_ERR_ARG, "null pointer");
    GL_API_BEGIN
    ctx->eng->sampler_timing(avg_unet_eval_ms, first_eval_ms, n_evals);
    GL_API_END
}

int gl_vae_encode(gl_ctx* ctx, int B, int H, int W, const float* img, const float* noise, float* z, gl_stream s) {
    NEED(ctx);
    if (!img || !noise || !z) return gl::set_error(GL_ERR_ARG, "null pointer");
    GL_API_BEGIN
    ctx->eng->vae_encode(B, H, W, img, noise, z, S(s));
    GL_API_END
}

int gl_unet_profile(gl_ctx* ctx, int Beff, int h, int w, const float* x, int xB, const int64_t* timesteps,
                    const float* inpaint_extra, int extraB, float* eps_out, gl_prof_rec* recs, int max_recs, int* n_recs,
                    gl_stream s) {
    NEED(ctx);
    if (!x || !timesteps || !eps_out || !recs || !n_recs || max_recs <= 0) return gl::set_error(GL_ERR_ARG, "null pointer");
    GL_API_BEGIN
    ctx->eng->profile_begin();
    std::vector<Engine::ProfRec> out;
    try {
        ctx->eng->unet_forward(Beff, h, w, x, xB, timesteps, inpaint_extra, extraB, eps_out, S(s));
    } catch (...) {
        (void)ctx->eng->profile_end(S(s));
        throw;
    }
    out = ctx->eng->profile_end(S(s));
    int n = 0;
    for (const auto& r : out) {
        if (n >= max_recs) break;
        gl_prof_rec& d = recs[n++];
        snprintf(d.name, sizeof d.name, "%s", r.name.c_str());
        d.calls = r.calls; d.ms = r.ms; d.flops = r.flops; d.bytes = r.bytes;
    }
    *n_recs = n;
    GL_API_END
}

int gl_to_uint8(const float* img, uint8_t* out, int B, int C, int HW, gl_stream s) {
    if (!img || !out) return gl::set_error(GL_ERR_ARG, "null pointer");
    // no context: the launch has to go to the device that holds the image, whatever the caller's current device is
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, img) != hipSuccess) {
        (void)hipGetLastError();
        return gl::set_error(GL_ERR_ARG, "gl_to_uint8: img is not a device pointer");
    }
    DeviceGuard guard(at.device);
    if (!guard.ok) return gl::set_error(GL_ERR_HIP, "hipSetDevice(%d) failed", at.device);
    return to_uint8_launch(img, out, B, C, HW, S(s));
}

int gl_arena_high_water(gl_ctx* ctx, size_t* bytes) {
    NEED(ctx);
    *bytes = ctx->eng->arena().high_water();
    return GL_OK;
}

int gl_launch_count(gl_ctx* ctx, int64_t* n) {
    NEED(ctx);
    *n = ctx->eng->n_launches;
    return GL_OK;
}

int gl_set_ff_rows_policy(int mode) { return gl::ff_rows_policy_set(mode); }

int gl_ff_rows_policy_report(char* buf, size_t cap) { return gl::ff_rows_policy_report(buf, cap); }

int gl_box_calibrate(gl_ctx* ctx, gl_box_calibration* out, gl_stream s) {
    NEED(ctx);
    if (!out) return gl::set_error(GL_ERR_ARG, "gl_box_calibrate: null out pointer");
    GL_API_BEGIN
    Arena& ar = ctx->eng->arena();
    const size_t mk = ar.mark();
    const size_t bytes = size_t(1) << 30;   // 512 MiB -> 512 MiB: twice the Infinity Cache in each direction
    void* scratch = ar.alloc(bytes);
    HIPCK_API(hipMemsetAsync(scratch, 1, bytes, S(s)));
    float v[3] = {0.f, 0.f, 0.f};
    int rc = box_calibrate_launch(scratch, bytes, v, S(s));
    ar.release(mk);
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    out->hbm_copy_gbs = v[0];
    out->lds_dma_tbs = v[1];
    out->mfma_bf16_tflops = v[2];
    GL_API_END
}

int gl_mfma_calibrate(gl_ctx* ctx, int shape, int waves_per_simd, int n_acc, int zero_data, float target_ms, gl_mfma_calibration* out, gl_stream s) {
    NEED(ctx);
    if (!out) return gl::set_error(GL_ERR_ARG, "gl_mfma_calibrate: null out pointer");
    GL_API_BEGIN
    Arena& ar = ctx->eng->arena();
    const size_t mk = ar.mark();
    void* scratch = ar.alloc(4096);
    HIPCK_API(hipMemsetAsync(scratch, 0, 4096, S(s)));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int rc = mfma_calibrate_launch(shape, waves_per_simd, n_acc, zero_data, target_ms, scratch, v, S(s));
    ar.release(mk);
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    out->tflops = v[0];
    out->sclk_mhz = v[1];
    out->ms = v[2];
    out->cycles_per_mfma = v[3];
    GL_API_END
}

// ------------------------------------------------------------------ single operators
int gl_op_linear(gl_ctx* ctx, const void* x, const void* w, const float* bias, const void* res, void* y,
                 int M, int N, int K, int act, int out_f32, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    AOperand A;
    aoperand_rows(A, (const bf16*)x, K, K);
    Epilogue E;
    epilogue_defaults(E);
    E.out = y; E.ldo = N; E.out_f32 = out_f32; E.bias = bias; E.act = act; E.res = (const bf16*)res; E.ldres = N;
    int r = gemm_launch(A, (const bf16*)w, M, N, K, E, ctx->eng->splitk_ws(), ctx->eng->splitk_ws_bytes(), S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    GL_API_END
}

int gl_op_geglu(gl_ctx* ctx, const void* x, const float* w_f32, const float* b_f32, void* y, int M, int inner, int K, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    Arena& ar = ctx->eng->arena();
    ar.reset();
    bf16* wp = ar.get<bf16>((size_t)2 * inner * K);
    float* bp = ar.get<float>((size_t)2 * inner);
    const int layout = gl::gemm_geglu_layout();
    int r = pack_geglu_launch(w_f32, b_f32, wp, bp, inner, K, layout, S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    AOperand A;
    aoperand_rows(A, (const bf16*)x, K, K);
    Epilogue E;
    epilogue_defaults(E);
    E.act = ACT_GEGLU; E.geglu16 = layout; E.out = y; E.ldo = inner; E.bias = bp;
    r = gemm_launch(A, wp, M, 2 * inner, K, E, ctx->eng->splitk_ws(), ctx->eng->splitk_ws_bytes(), S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    GL_API_END
}

int gl_op_ln_linear(gl_ctx* ctx, const void* a, int M, int K0, const float* w0, const float* b0, const void* res, int C,
                    const float* gamma, const float* beta, const float* w1, const float* b1, int mode, int inner_or_heads, int T,
                    void* x_out, void* y_out, int* used_fold, gl_stream s) {
    NEED(ctx);
    if (!a || !w0 || !gamma || !beta || !w1 || !x_out || !y_out || !used_fold) return gl::set_error(GL_ERR_ARG, "gl_op_ln_linear: null pointer");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    auto ck = [&](int rc) { if (rc != GL_OK) throw GlError(rc, gl::last_error()); };
    if (C % 64 || K0 % 64 || M <= 0) throw GlError(GL_ERR_ARG, "gl_op_ln_linear: C and K0 must be multiples of 64");
    const int N1 = mode == 0 ? 2 * inner_or_heads : C;
    // ---- producer: x = a W0^T + b0 (+ res), with the row statistics of x
    bf16* w0b = ar.get<bf16>((size_t)C * K0);
    ck(cast_f32_bf16_launch(w0, w0b, (int64_t)C * K0, S(s)));
    const int ld = C / 32;
    float2* stats = ar.get<float2>((size_t)M * ld);
    {
        AOperand A;
        aoperand_rows(A, (const bf16*)a, K0, K0);
        Epilogue E;
        epilogue_defaults(E);
        E.out = x_out; E.ldo = C; E.bias = b0; E.res = (const bf16*)res; E.ldres = C;
        E.stats_out = stats; E.stats_ld = ld;
        ck(gemm_launch(A, w0b, M, C, K0, E, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
    }
    const int nb = gemm_last_stats_nb();
    // ---- the consumer's folded weights: W1 * gamma, b1 + W1 beta, csum of the packed rows
    float* wf = ar.get<float>((size_t)N1 * C);
    float* bf = ar.get<float>((size_t)N1);
    ck(ln_fold_launch(w1, b1, gamma, beta, wf, bf, N1, C, S(s)));
    bf16* wp = ar.get<bf16>((size_t)N1 * C);
    float* bp = ar.get<float>((size_t)N1);
    float* cs = ar.get<float>((size_t)N1);
    AOperand A;
    Epilogue E;
    epilogue_defaults(E);
    if (mode == 0) {
        const int layout = gl::gemm_geglu_layout();
        ck(pack_geglu_launch(wf, bf, wp, bp, inner_or_heads, C, layout, S(s)));
        E.act = ACT_GEGLU; E.geglu16 = layout; E.out = y_out; E.ldo = inner_or_heads; E.bias = bp;
    } else {
        const int H = inner_or_heads, d = C / H;
        int dp, dpv;
        ck(attn_dims(d, &dp, &dpv));
        if (T <= 0 || M % T || T % 64) throw GlError(GL_ERR_ARG, "gl_op_ln_linear: mode 1 needs T (tokens per sample, a multiple of 64) dividing M");
        ck(cast_f32_bf16_launch(wf, wp, (int64_t)N1 * C, S(s)));
        HIPCK_API(hipMemcpyAsync(bp, bf, (size_t)N1 * sizeof(float), hipMemcpyDeviceToDevice, S(s)));
        E.mode = EPI_QK_HEADS; E.q = (bf16*)y_out; E.k = nullptr; E.C = C; E.H = H; E.d = d; E.DP = dp; E.T = T;
        E.Tpad_q = round_up(T, 128); E.bias = bp;
    }
    ck(rowsum_bf16_launch(wp, cs, N1, C, S(s)));
    aoperand_rows(A, (const bf16*)x_out, C, C);
    const bool fold = nb > 0 && gemm_ln_fold_supported(A, M, N1, C, E);
    *used_fold = fold ? 1 : 0;
    if (fold) {
        E.ln_stats = stats; E.ln_nb = nb; E.ln_ld = ld; E.ln_csum = cs; E.ln_inv_c = 1.f / (float)C; E.ln_eps = 1e-5f;
    } else {   // the fallback the engine takes where no statistics exist: ln_kernel without affine, same folded weights
        bf16* xn = ar.get<bf16>((size_t)M * C);
        LNParams P{};
        P.x = (const bf16*)x_out; P.B = 1; P.N1 = M; P.N2 = 0; P.Tpad = M; P.C = C; P.eps = 1e-5f; P.y = xn;
        ck(layernorm_launch(P, S(s)));
        aoperand_rows(A, xn, C, C);
    }
    ck(gemm_launch(A, wp, M, N1, C, E, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
    GL_API_END
}

int gl_op_feedforward(gl_ctx* ctx, const void* x, int M, int C, const float* gamma, const float* beta, const float* w1, const float* b1,
                      const float* w2, const float* b2, const void* res, const float* gate, void* y, void* stats, int* used_rows,
                      gl_stream s) {
    NEED(ctx);
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || !used_rows) return gl::set_error(GL_ERR_ARG, "gl_op_feedforward: null pointer");
    if ((gamma == nullptr) != (beta == nullptr)) return gl::set_error(GL_ERR_ARG, "gl_op_feedforward: gamma and beta come together");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    auto ck = [&](int rc) { if (rc != GL_OK) throw GlError(rc, gl::last_error()); };
    if (C % 64 || M <= 0) throw GlError(GL_ERR_ARG, "gl_op_feedforward: C must be a multiple of 64");
    const float* w1e = w1;
    const float* b1e = b1;
    if (gamma) {   // LayerNorm folded into the projection: W1 * gamma, b1 + W1 beta
        float* wf = ar.get<float>((size_t)8 * C * C);
        float* bf = ar.get<float>((size_t)8 * C);
        ck(ln_fold_launch(w1, b1, gamma, beta, wf, bf, 8 * C, C, S(s)));
        w1e = wf; b1e = bf;
    }
    const bool rows = ff_rows_supported(M, C) && !(gl::dev_env("GL_FF_ROWS") && atoi(gl::dev_env("GL_FF_ROWS")) == 0);
    *used_rows = rows ? 1 : 0;
    // the per-row (sum, sum of squares) come out of the row-local kernel's epilogue; the two-GEMM form writes per-column-block partials
    // in the consumer's own layout (gemm.h Epilogue::stats_out), not this one: refuse rather than hand back zeros
    if (stats && !rows) throw GlError(GL_ERR_UNSUPPORTED, "gl_op_feedforward: row statistics are produced only by the row-local kernel (C = 320, M % 128 == 0)");
    if (rows) {
        void* st = ar.alloc(ff_stream_bytes(C));
        ck(ff_pack_launch(w1e, b1e, w2, st, C, S(s)));
        FFRowsParams P{};
        P.x = (const bf16*)x; P.ldx = C; P.normalize = gamma ? 1 : 0; P.eps = 1e-5f; P.stream = st; P.b2 = b2;
        P.res = (const bf16*)res; P.ldres = C; P.gate = gate; P.out = (bf16*)y; P.ldo = C; P.M = M;
        P.stats_out = (float2*)stats; P.stats_ld = 1;
        ck(ff_rows_launch(P, C, S(s)));
    } else {
        const bf16* in = (const bf16*)x;
        if (gamma) {
            bf16* xn = ar.get<bf16>((size_t)M * C);
            LNParams L{};
            L.x = in; L.B = 1; L.N1 = M; L.N2 = 0; L.Tpad = M; L.C = C; L.eps = 1e-5f; L.y = xn;
            ck(layernorm_launch(L, S(s)));
            in = xn;
        }
        const int layout = gl::gemm_geglu_layout();
        bf16* wp = ar.get<bf16>((size_t)8 * C * C);
        float* bp = ar.get<float>((size_t)8 * C);
        ck(pack_geglu_launch(w1e, b1e, wp, bp, 4 * C, C, layout, S(s)));
        bf16* w2b = ar.get<bf16>((size_t)C * 4 * C);
        ck(cast_f32_bf16_launch(w2, w2b, (int64_t)C * 4 * C, S(s)));
        bf16* hid = ar.get<bf16>((size_t)M * 4 * C);
        AOperand A;
        aoperand_rows(A, in, C, C);
        Epilogue E;
        epilogue_defaults(E);
        E.act = ACT_GEGLU; E.geglu16 = layout; E.out = hid; E.ldo = 4 * C; E.bias = bp;
        ck(gemm_launch(A, wp, M, 8 * C, C, E, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
        AOperand A2;
        aoperand_rows(A2, hid, 4 * C, 4 * C);
        Epilogue E2;
        epilogue_defaults(E2);
        E2.out = y; E2.ldo = C; E2.bias = b2; E2.res = (const bf16*)res; E2.ldres = C; E2.gate = gate;
        ck(gemm_launch(A2, w2b, M, C, 4 * C, E2, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
    }
    GL_API_END
}

int gl_op_ff_chain(gl_ctx* ctx, const void* x, int M, int C, const float* pre_w, const float* pre_b, const void* pre_res, const float* pre_gate,
                   const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2, const float* b2, const float* gate,
                   const float* post_w, const float* post_b, const void* post_res, void* y, gl_stream s) {
    NEED(ctx);
    if (!x || !pre_w || !pre_b || !pre_res || !gamma || !beta || !w1 || !b1 || !w2 || !b2 || !y) return gl::set_error(GL_ERR_ARG, "gl_op_ff_chain: null pointer");
    if (post_w && (!post_b || !post_res)) return gl::set_error(GL_ERR_ARG, "gl_op_ff_chain: the trailing projection needs its bias and residual");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    auto ck = [&](int rc) { if (rc != GL_OK) throw GlError(rc, gl::last_error()); };
    if (!ff_rows_supported(M, C)) throw GlError(GL_ERR_UNSUPPORTED, "gl_op_ff_chain: no row-local kernel for this shape (C = 320, M % 128 == 0)");
    float* wf = ar.get<float>((size_t)8 * C * C);
    float* bf = ar.get<float>((size_t)8 * C);
    ck(ln_fold_launch(w1, b1, gamma, beta, wf, bf, 8 * C, C, S(s)));
    void* st = ar.alloc(ff_chain_stream_bytes(C, true, post_w != nullptr));
    ck(ff_chain_pack_launch(wf, bf, w2, pre_w, post_w, st, C, S(s)));
    FFRowsParams P{};
    P.x = (const bf16*)x; P.ldx = C; P.normalize = 1; P.eps = 1e-5f; P.stream = st; P.b2 = b2; P.gate = gate; P.out = (bf16*)y; P.ldo = C; P.M = M;
    P.pre = 1; P.pre_b = pre_b; P.pre_res = (const bf16*)pre_res; P.ld_pre_res = C; P.pre_gate = pre_gate;
    P.mid_out = ar.get<bf16>((size_t)M * C); P.ld_mid = C;
    if (post_w) { P.post = 1; P.post_b = post_b; P.post_res = (const bf16*)post_res; P.ld_post_res = C; }
    ck(ff_rows_launch(P, C, S(s)));
    GL_API_END
}

int gl_op_ff_chain_q(gl_ctx* ctx, const void* x, int B, int N, int C, const float* pre_w, const float* pre_b, const void* pre_res, const float* pre_gate,
                     const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2, const float* b2, const float* gate,
                     const float* gamma_q, const float* beta_q, const float* wq, void* y, void* q, gl_stream s) {
    NEED(ctx);
    if (!x || !pre_w || !pre_b || !pre_res || !gamma || !beta || !w1 || !b1 || !w2 || !b2 || !gamma_q || !beta_q || !wq || !y || !q)
        return gl::set_error(GL_ERR_ARG, "gl_op_ff_chain_q: null pointer");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    auto ck = [&](int rc) { if (rc != GL_OK) throw GlError(rc, gl::last_error()); };
    const int M = B * N;
    if (!ff_rows_supported(M, C) || N % 128) throw GlError(GL_ERR_UNSUPPORTED, "gl_op_ff_chain_q: no row-local kernel for this shape (C = 320, N % 128 == 0)");
    float* wf = ar.get<float>((size_t)8 * C * C);
    float* bf = ar.get<float>((size_t)8 * C);
    ck(ln_fold_launch(w1, b1, gamma, beta, wf, bf, 8 * C, C, S(s)));
    float* wqf = ar.get<float>((size_t)C * C);
    float* bqf = ar.get<float>((size_t)C);
    ck(ln_fold_launch(wq, nullptr, gamma_q, beta_q, wqf, bqf, C, C, S(s)));
    void* st = ar.alloc(ff_chain_stream_bytes(C, true, true));
    ck(ff_chain_pack_launch(wf, bf, w2, pre_w, wqf, st, C, S(s)));
    FFRowsParams P{};
    P.x = (const bf16*)x; P.ldx = C; P.normalize = 1; P.eps = 1e-5f; P.stream = st; P.b2 = b2; P.gate = gate; P.out = (bf16*)y; P.ldo = C; P.M = M;
    P.pre = 1; P.pre_b = pre_b; P.pre_res = (const bf16*)pre_res; P.ld_pre_res = C; P.pre_gate = pre_gate;
    P.mid_out = ar.get<bf16>((size_t)M * C); P.ld_mid = C;
    P.post = 2; P.post_b = bqf; P.q = (bf16*)q; P.qDP = 48; P.qT = N; P.qTpad = N;
    ck(ff_rows_launch(P, C, S(s)));
    GL_API_END
}

int gl_op_adamw_step(gl_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int step, gl_stream s) {
    NEED(ctx);
    if (!p || !g || !m || !v || n < 0) return gl::set_error(GL_ERR_ARG, "gl_op_adamw_step: null pointer");
    return gl::adamw_step(p, g, m, v, (size_t)n, lr, beta1, beta2, eps, weight_decay, step, S(s));
}

static const char* const k_train_block_names[GL_TRAIN_BLOCK_PARAMS] = {
    "norm1.weight", "norm1.bias", "attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight", "attn1.to_out.0.weight", "attn1.to_out.0.bias",
    "fuser.linear.weight", "fuser.linear.bias", "fuser.norm1.weight", "fuser.norm1.bias", "fuser.attn.to_q.weight", "fuser.attn.to_k.weight",
    "fuser.attn.to_v.weight", "fuser.attn.to_out.0.weight", "fuser.attn.to_out.0.bias", "fuser.norm2.weight", "fuser.norm2.bias",
    "fuser.ff.net.0.proj.weight", "fuser.ff.net.0.proj.bias", "fuser.ff.net.2.weight", "fuser.ff.net.2.bias", "fuser.alpha_attn", "fuser.alpha_dense",
    "norm2.weight", "norm2.bias", "attn2.to_q.weight", "attn2.to_k.weight", "attn2.to_v.weight", "attn2.to_out.0.weight", "attn2.to_out.0.bias",
    "norm3.weight", "norm3.bias", "ff.net.0.proj.weight", "ff.net.0.proj.bias", "ff.net.2.weight", "ff.net.2.bias"};
static_assert(GL_TRAIN_BLOCK_PARAMS == gl::TP_COUNT, "parameter table out of step with train.h");

const char* const* gl_train_block_param_names(void) { return k_train_block_names; }

int gl_op_block_train(gl_ctx* ctx, const gl_train_block_dims* dims, const float* const* params, const float* x, const float* objs,
                      const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* grads,
                      gl_stream s) {
    NEED(ctx);
    if (!dims || !params || !x || !objs || !context || !target || !y || !loss || !dx || !dobjs || !grads)
        return gl::set_error(GL_ERR_ARG, "gl_op_block_train: null pointer");
    for (int i = 0; i < GL_TRAIN_BLOCK_PARAMS; ++i)
        if (grads[i] && !(i >= gl::TP_F_LIN_W && i <= gl::TP_F_ALPHA_DENSE))
            return gl::set_error(GL_ERR_ARG, "gl_op_block_train: a gradient was asked for '%s', which the reference keeps frozen", k_train_block_names[i]);
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    eng.arena().reset();
    gl::TrainBlockDims d{dims->B, dims->N, dims->Ng, dims->C, dims->heads, dims->ctx_T, dims->ctx_dim, dims->fuser_scale};
    int rc = gl::block_train_step(eng.arena(), eng.splitk_ws(), eng.splitk_ws_bytes(), d, params, x, objs, context, target, y, loss, dx, dobjs, grads, S(s));
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    GL_API_END
}

static_assert(GL_TRAIN_ST_PARAMS == gl::ST_COUNT, "parameter table out of step with train.h");
const char* const* gl_train_st_param_names(void) {
    static std::string store[GL_TRAIN_ST_PARAMS];
    static const char* names[GL_TRAIN_ST_PARAMS];
    static const bool once = [] {
        store[gl::ST_NORM_W] = "norm.weight"; store[gl::ST_NORM_B] = "norm.bias";
        store[gl::ST_PIN_W] = "proj_in.weight"; store[gl::ST_PIN_B] = "proj_in.bias";
        for (int i = 0; i < GL_TRAIN_BLOCK_PARAMS; ++i) store[gl::ST_BLOCK0 + i] = std::string("transformer_blocks.0.") + k_train_block_names[i];
        store[gl::ST_POUT_W] = "proj_out.weight"; store[gl::ST_POUT_B] = "proj_out.bias";
        for (int i = 0; i < GL_TRAIN_ST_PARAMS; ++i) names[i] = store[i].c_str();
        return true;
    }();
    (void)once;
    return names;
}

int gl_op_st_train(gl_ctx* ctx, const gl_train_block_dims* dims, const float* const* params, const float* x, const float* objs,
                   const float* context, const float* target, float* y, float* loss, float* dx, float* dobjs, float* const* grads, gl_stream s) {
    NEED(ctx);
    if (!dims || !params || !x || !objs || !context || !target || !y || !loss || !dx || !dobjs || !grads)
        return gl::set_error(GL_ERR_ARG, "gl_op_st_train: null pointer");
    for (int i = 0; i < GL_TRAIN_ST_PARAMS; ++i) {
        const int bi = i - gl::ST_BLOCK0;
        if (grads[i] && !(bi >= gl::TP_F_LIN_W && bi <= gl::TP_F_ALPHA_DENSE))
            return gl::set_error(GL_ERR_ARG, "gl_op_st_train: a gradient was asked for '%s', which the reference keeps frozen", gl_train_st_param_names()[i]);
    }
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    eng.arena().reset();
    gl::TrainBlockDims d{dims->B, dims->N, dims->Ng, dims->C, dims->heads, dims->ctx_T, dims->ctx_dim, dims->fuser_scale};
    int rc = gl::st_train_step(eng.arena(), eng.splitk_ws(), eng.splitk_ws_bytes(), d, params, x, objs, context, target, y, loss, dx, dobjs, grads, S(s));
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    GL_API_END
}

static const char* const k_train_resblock_names[GL_TRAIN_RESBLOCK_PARAMS] = {
    "in_layers.0.weight", "in_layers.0.bias", "in_layers.2.weight", "in_layers.2.bias", "emb_layers.1.weight", "emb_layers.1.bias",
    "out_layers.0.weight", "out_layers.0.bias", "out_layers.3.weight", "out_layers.3.bias", "skip_connection.weight", "skip_connection.bias"};
static_assert(GL_TRAIN_RESBLOCK_PARAMS == gl::RP_COUNT, "parameter table out of step with train.h");

const char* const* gl_train_resblock_param_names(void) { return k_train_resblock_names; }

int gl_op_resblock_train(gl_ctx* ctx, const gl_train_resblock_dims* dims, const float* const* params, const float* x, const float* emb,
                         const float* target, float* y, float* loss, float* dx, gl_stream s) {
    NEED(ctx);
    if (!dims || !params || !x || !emb || !target || !y || !loss || !dx) return gl::set_error(GL_ERR_ARG, "gl_op_resblock_train: null pointer");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    eng.arena().reset();
    gl::TrainResDims d{dims->B, dims->H, dims->W, dims->Cin, dims->Cout, dims->emb_dim};
    int rc = gl::resblock_train_step(eng.arena(), eng.splitk_ws(), eng.splitk_ws_bytes(), d, params, x, emb, target, y, loss, dx, S(s));
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    GL_API_END
}

int gl_op_resample_train(gl_ctx* ctx, int mode, int B, int H, int W, int C, const float* w_oihw, const float* bias, const float* x,
                         const float* target, float* y, float* loss, float* dx, gl_stream s) {
    NEED(ctx);
    if (!w_oihw || !x || !target || !y || !loss || !dx) return gl::set_error(GL_ERR_ARG, "gl_op_resample_train: null pointer");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    eng.arena().reset();
    int rc = gl::resample_train_step(eng.arena(), eng.splitk_ws(), eng.splitk_ws_bytes(), mode, B, H, W, C, w_oihw, bias, x, target, y, loss, dx, S(s));
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    GL_API_END
}

int gl_unet_train_step(gl_ctx* ctx, const gl_unet_config* cfg, const gl_train_unet_in* in, int n_params, const char* const* names,
                       const float* const* params, float* const* grads, float* eps_out, float* loss, gl_stream s) {
    NEED(ctx);
    if (!cfg || !in || !names || !params || !grads || !loss || n_params <= 0) return gl::set_error(GL_ERR_ARG, "gl_unet_train_step: null pointer");
    if (cfg->grounding_kind < 0 || cfg->grounding_kind > 2 || cfg->fuser_kind != 0 || cfg->inpaint_mode || cfg->extra_channels)
        return gl::set_error(GL_ERR_UNSUPPORTED, "gl_unet_train_step: the training step is built for the text, text+image and keypoint tokenizers with gatedSA fusers (no inpainting / downsampler channels)");
    if (cfg->grounding_kind == 1 && (!in->text_masks || !in->image_masks || !in->image_embeddings))
        return gl::set_error(GL_ERR_ARG, "gl_unet_train_step: the text+image tokenizer needs text_masks, image_masks and image_embeddings");
    if (!in->x || !in->timesteps || !in->context || !in->boxes || !in->masks || (!in->positive_embeddings && cfg->grounding_kind != 2) || !in->target)
        return gl::set_error(GL_ERR_ARG, "gl_unet_train_step: null input");
    if (cfg->gr_in_dim != cfg->gr_out_dim || cfg->gr_out_dim != cfg->context_dim)
        return gl::set_error(GL_ERR_UNSUPPORTED, "gl_unet_train_step: grounding in / out dim and context_dim are expected to be equal (768 in every shipped config)");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    eng.arena().reset();
    gl::TrainUNetCfg c{};
    c.in_channels = cfg->in_channels; c.out_channels = cfg->out_channels; c.model_channels = cfg->model_channels; c.num_res_blocks = cfg->num_res_blocks;
    c.num_heads = cfg->num_heads; c.context_dim = cfg->context_dim; c.gr_dim = cfg->gr_in_dim; c.grounding_kind = cfg->grounding_kind; c.n_mult = cfg->n_mult; c.n_attn = cfg->n_attn;
    for (int i = 0; i < 8; ++i) { c.channel_mult[i] = cfg->channel_mult[i]; c.attention_resolutions[i] = cfg->attention_resolutions[i]; }
    gl::TrainUNetIn u{in->B, in->H, in->W, in->ctx_T, cfg->grounding_kind == 1 ? 2 * in->Ng : in->Ng, in->Ng, in->x, in->timesteps, in->context, in->boxes,
                      in->masks, in->positive_embeddings, in->text_masks, in->image_masks, in->image_embeddings, in->target, in->fuser_scale, in->checkpoint};
    int rc = gl::unet_train_step(eng.arena(), eng.splitk_ws(), eng.splitk_ws_bytes(), c, u, n_params, names, params, grads, k_train_block_names, eps_out, loss, S(s),
                                 eng.train_events(), Engine::kTrainEvents, in->use_weight_cache ? eng.train_cache : nullptr);
    if (rc != GL_OK) throw GlError(rc, gl::last_error());
    eng.train_events_recorded = true;
    GL_API_END
}

int gl_train_weight_cache(gl_ctx* ctx, int enable, size_t* bytes) {
    NEED(ctx);
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    if (enable && !eng.train_cache) eng.train_cache = gl::train_cache_create();
    if (!enable && eng.train_cache) {
        HIPCK_API(hipDeviceSynchronize());    // (a step that reads the copies may still be in flight)
        gl::train_cache_destroy(eng.train_cache);
        eng.train_cache = nullptr;
    }
    if (bytes) *bytes = gl::train_cache_bytes(eng.train_cache);
    GL_API_END
}

int gl_train_wait_grads(gl_ctx* ctx, int index, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    if (index < 0 || index >= Engine::kTrainEvents) throw GlError(GL_ERR_ARG, "gl_train_wait_grads: milestone index out of range");
    if (!eng.train_events_recorded) throw GlError(GL_ERR_STATE, "gl_train_wait_grads: no gl_unet_train_step has run on this context");
    HIPCK_API(hipStreamWaitEvent(S(s), eng.train_events()[index], 0));
    GL_API_END
}

int gl_op_conv3x3(gl_ctx* ctx, const void* x0, int C0, const void* x1, int C1, int B, int H, int W,
                  const float* w_oihw, const float* bias, int Cout, int stride, int ups, int pad_lo,
                  const void* res, void* y, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    Arena& ar = ctx->eng->arena();
    ar.reset();
    const int Cin = C0 + C1;
    bf16* wp = ar.get<bf16>((size_t)Cout * 9 * Cin);
    int r = pack_conv_weight_launch(w_oihw, wp, Cout, Cin, 3, 3, Cout, S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    const int Hup = H << ups, Wup = W << ups;
    const int Ho = stride == 1 ? Hup : (pad_lo ? (Hup + 2 - 3) / 2 + 1 : (Hup + 1 - 3) / 2 + 1);
    const int Wo = stride == 1 ? Wup : (pad_lo ? (Wup + 2 - 3) / 2 + 1 : (Wup + 1 - 3) / 2 + 1);
    AOperand A{};
    A.p0 = (const bf16*)x0; A.C0 = C0; A.ld0 = C0; A.p1 = (const bf16*)x1; A.C1 = C1; A.ld1 = C1;
    A.mode = A_CONV3; A.Hin = H; A.Win = W; A.Ho = Ho; A.Wo = Wo; A.stride = stride; A.ups = ups; A.pad_lo = pad_lo;
    Epilogue E;
    epilogue_defaults(E);
    E.out = y; E.ldo = Cout; E.bias = bias; E.res = (const bf16*)res; E.ldres = Cout; E.rows_per_b = Ho * Wo;
    r = gemm_launch(A, wp, B * Ho * Wo, Cout, 9 * Cin, E, ctx->eng->splitk_ws(), ctx->eng->splitk_ws_bytes(), S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    GL_API_END
}

int gl_op_gn_silu_conv3x3(gl_ctx* ctx, const void* x0, int C0, const void* x1, int C1, int B, int H, int W, const float* gamma, const float* beta,
                          float eps, const float* w_oihw, const float* bias, int Cout, const float* bias2, const void* res, void* y, int mode,
                          int* used_prologue, gl_stream s) {
    NEED(ctx);
    if (!x0 || !gamma || !beta || !w_oihw || !y || !used_prologue) return gl::set_error(GL_ERR_ARG, "gl_op_gn_silu_conv3x3: null pointer");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    const int Cin = C0 + C1;
    bf16* wp = ar.get<bf16>((size_t)Cout * 9 * Cin);
    int r = pack_conv_weight_launch(w_oihw, wp, Cout, Cin, 3, 3, Cout, S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    NormW n;
    n.g = gamma; n.b = beta; n.C = Cin;
    ConvW c;
    c.w = wp; c.b = bias; c.Cin = Cin; c.Cout = Cout;
    const bool saved = eng.gn_prologue_;
    if (mode >= 0) eng.gn_prologue_ = mode != 0;
    const int64_t n0 = eng.n_prologue_convs;
    try {
        eng.gn_silu_conv3x3(TRef{(const bf16*)x0, C0, (const bf16*)x1, C1}, B, H, W, n, eps, c, bias2, Cout, (const bf16*)res, (bf16*)y, S(s));
    } catch (...) {
        eng.gn_prologue_ = saved;
        throw;
    }
    eng.gn_prologue_ = saved;
    *used_prologue = eng.n_prologue_convs > n0 ? 1 : 0;
    if (mode == 1 && !*used_prologue) throw GlError(GL_ERR_UNSUPPORTED, "gl_op_gn_silu_conv3x3: this shape has no GroupNorm prologue (conv_halo_kernel, H W % 256 == 0)");
    GL_API_END
}

int gl_op_groupnorm(gl_ctx* ctx, const void* x0, int C0, const void* x1, int C1, int B, int HW,
                    const float* gamma, const float* beta, float eps, int silu, void* y, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    Arena& ar = ctx->eng->arena();
    ar.reset();
    GNParams P{};
    P.x0 = (const bf16*)x0; P.C0 = C0; P.x1 = (const bf16*)x1; P.C1 = C1; P.B = B; P.HW = HW; P.eps = eps;
    P.gamma = gamma; P.beta = beta; P.y = (bf16*)y; P.silu = silu;
    P.partial = reinterpret_cast<float*>(ar.alloc(gn_partial_bytes(B, HW)));
    int r = groupnorm_launch(P, S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    GL_API_END
}

int gl_op_layernorm(gl_ctx* ctx, const void* x, const void* x2, int B, int N1, int N2, int Tpad, int C,
                    const float* gamma, const float* beta, float eps, void* y, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    LNParams P{};
    P.x = (const bf16*)x; P.x2 = (const bf16*)x2; P.B = B; P.N1 = N1; P.N2 = N2; P.Tpad = Tpad; P.C = C; P.eps = eps;
    P.gamma = gamma; P.beta = beta; P.y = (bf16*)y;
    int r = layernorm_launch(P, S(s));
    if (r != GL_OK) throw GlError(r, gl::last_error());
    GL_API_END
}

int gl_op_attention(gl_ctx* ctx, const void* xq, const void* xkv, int B, int Nq, int Nk, int C, int Ck, int H,
                    const float* wq, const float* wk, const float* wv, void* o, gl_stream s) {
    NEED(ctx);
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    if (C % H != 0) throw GlError(GL_ERR_ARG, "C must be divisible by H");
    const int d = C / H;
    int dp, dpv;
    int r = attn_dims(d, &dp, &dpv);
    if (r != GL_OK) throw GlError(r, gl::last_error());
    const int Tq = round_up(Nq, 64), Tk = round_up(Nk, 64);
    const int vt_layout = attn_vt_layout(d, Nk, &dpv);
    AttnBufs& bufs = eng.attn_bufs(B, H, d, Tq, Tk, dpv);
    bf16* wqb = ar.get<bf16>((size_t)C * C);
    bf16* wkb = ar.get<bf16>((size_t)C * Ck);
    bf16* wvb = ar.get<bf16>((size_t)C * Ck);
    bf16* xqp = ar.get<bf16>((size_t)B * Tq * C);
    bf16* xkp = ar.get<bf16>((size_t)B * Tk * Ck);
    auto ck = [&](int rc) { if (rc != GL_OK) throw GlError(rc, gl::last_error()); };
    ck(cast_f32_bf16_launch(wq, wqb, (int64_t)C * C, S(s)));
    ck(cast_f32_bf16_launch(wk, wkb, (int64_t)C * Ck, S(s)));
    ck(cast_f32_bf16_launch(wv, wvb, (int64_t)C * Ck, S(s)));
    ck(pad_rows_bf16_launch((const bf16*)xq, xqp, B, Nq, Tq, C, S(s)));
    ck(pad_rows_bf16_launch((const bf16*)xkv, xkp, B, Nk, Tk, Ck, S(s)));
    // self-attention (same rows for q, k, v): the engine's fused projection, one EPI_QKV_HEADS GEMM (GL_QKV_FUSED=0: off)
    const bool fused = xq == xkv && C == Ck && Nq == Nk && gemm_supports_qkv() && (2 * C) % 128 == 0 &&
                       !(dev_env("GL_QKV_FUSED") && atoi(dev_env("GL_QKV_FUSED")) == 0);
    if (fused) {
        bf16* wqkv = ar.get<bf16>((size_t)3 * C * C);
        ck(cast_f32_bf16_launch(wq, wqkv, (int64_t)C * C, S(s)));
        ck(cast_f32_bf16_launch(wk, wqkv + (size_t)C * C, (int64_t)C * C, S(s)));
        ck(cast_f32_bf16_launch(wv, wqkv + (size_t)2 * C * C, (int64_t)C * C, S(s)));
        AOperand A;
        aoperand_rows(A, xqp, C, C);
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_QKV_HEADS; E.q = bufs.q; E.k = bufs.k; E.vt = bufs.vt; E.C = C; E.H = H; E.d = d; E.DP = dp; E.DPV = dpv; E.T = Tq; E.vt_perm32 = vt_layout;
        E.Tpad_q = bufs.Tq_pad; E.Tpad_k = bufs.Tk_pad;
        ck(gemm_launch(A, wqkv, B * Tq, 3 * C, C, E, nullptr, 0, S(s)));
    } else {
        {
            AOperand A;
            aoperand_rows(A, xqp, C, C);
            Epilogue E;
            epilogue_defaults(E);
            E.mode = EPI_QK_HEADS; E.q = bufs.q; E.C = C; E.H = H; E.d = d; E.DP = dp; E.T = Tq; E.Tpad_q = bufs.Tq_pad;
            ck(gemm_launch(A, wqb, B * Tq, C, C, E, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
        }
        {
            AOperand A;
            aoperand_rows(A, xkp, Ck, Ck);
            Epilogue E;
            epilogue_defaults(E);
            E.mode = EPI_QK_HEADS; E.q = bufs.k; E.q_tiled = 1; E.C = C; E.H = H; E.d = d; E.DP = dp; E.T = Tk; E.Tpad_q = bufs.Tk_pad;
            ck(gemm_launch(A, wkb, B * Tk, C, Ck, E, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
        }
        {
            Epilogue E;
            epilogue_defaults(E);
            E.mode = EPI_VT_HEADS; E.out = bufs.vt; E.H = H; E.d = d; E.DPV = dpv; E.T = Tk; E.Tpad_k = bufs.Tk_pad; E.vt_perm32 = vt_layout;
            ck(gemm_launch_t(wvb, C, xkp, B * Tk, Ck, E, S(s)));
        }
    }
    AttnParams P{};
    P.q = bufs.q; P.k = bufs.k; P.vt = bufs.vt; P.o = (bf16*)o; P.H = H; P.d = d; P.Nq = Nq; P.Nk = Nk;
    P.Tq_pad = bufs.Tq_pad; P.Tk_pad = bufs.Tk_pad; P.ldo = C; P.o_rows_per_b = Nq; P.vt_layout = vt_layout;
    P.scale_log2e = (float)(1.4426950408889634 / std::sqrt((double)d));
    ck(attn_launch(P, B, S(s)));
    GL_API_END
}


int gl_op_proj_attention(gl_ctx* ctx, const void* x, int B, int N, int C, int H, const float* pre_w, const float* pre_b, const void* pre_res,
                         const float* gamma, const float* beta, const float* wq, const float* wk, const float* wv, int Nkv_extra,
                         void* mid, void* o, int* used_rows, gl_stream s) {
    NEED(ctx);
    if (!x || !pre_w || !pre_b || !gamma || !beta || !wq || !wk || !wv || !mid || !o || !used_rows)
        return gl::set_error(GL_ERR_ARG, "gl_op_proj_attention: null pointer");
    GL_API_BEGIN
    Engine& eng = *ctx->eng;
    Arena& ar = eng.arena();
    ar.reset();
    auto ck = [&](int rc) { if (rc != GL_OK) throw GlError(rc, gl::last_error()); };
    if (C % H != 0 || N % 64 != 0 || Nkv_extra != 0) throw GlError(GL_ERR_ARG, "gl_op_proj_attention: C % H == 0, N % 64 == 0");
    const int d = C / H, M = B * N;
    int dp, dpv;
    ck(attn_dims(d, &dp, &dpv));
    const int vt_layout = attn_vt_layout(d, N, &dpv);
    AttnBufs& bufs = eng.attn_bufs(B, H, d, N, N, dpv);
    // LayerNorm folded into the three projections: W gamma (bf16), bias W beta
    bf16* wqkv = ar.get<bf16>((size_t)3 * C * C);
    float* bias = ar.get<float>((size_t)3 * C);
    {
        float* wf = ar.get<float>((size_t)C * C);
        const float* ws[3] = {wq, wk, wv};
        for (int i = 0; i < 3; ++i) {
            ck(ln_fold_launch(ws[i], nullptr, gamma, beta, wf, bias + (size_t)i * C, C, C, S(s)));
            ck(cast_f32_bf16_launch(wf, wqkv + (size_t)i * C * C, (int64_t)C * C, S(s)));
        }
    }
    const bool want = *used_rows != 0;
    const bool rows = want && qkv_rows_supported(M, C, d, N) && vt_layout == 1;
    *used_rows = rows ? 1 : 0;
    if (rows) {
        void* st = ar.alloc(qkv_rows_stream_bytes(C, true, 3));
        ck(qkv_rows_pack_launch(pre_w, wqkv, 3, st, C, S(s)));
        QkvRowsParams P{};
        P.x = (const bf16*)x; P.ldx = C; P.eps = 1e-5f; P.stream = st; P.M = M;
        P.pre = 1; P.pre_b = pre_b; P.pre_res = (const bf16*)pre_res; P.ld_pre_res = C; P.mid_out = (bf16*)mid; P.ld_mid = C;
        P.np = 3; P.bias = bias; P.q = bufs.q; P.k = bufs.k; P.vt = bufs.vt;
        P.H = H; P.d = d; P.DP = dp; P.DPV = dpv; P.T = N; P.Tpad_q = bufs.Tq_pad; P.Tpad_k = bufs.Tk_pad; P.vt_perm32 = vt_layout;
        ck(qkv_rows_launch(P, C, S(s)));
    } else {
        // the form it replaces: projection GEMM (+ residual), LayerNorm kernel without affine, fused q,k,v^T GEMM
        bf16* pw = ar.get<bf16>((size_t)C * C);
        ck(cast_f32_bf16_launch(pre_w, pw, (int64_t)C * C, S(s)));
        {
            AOperand A;
            aoperand_rows(A, (const bf16*)x, C, C);
            Epilogue E;
            epilogue_defaults(E);
            E.out = mid; E.ldo = C; E.bias = pre_b; E.res = (const bf16*)pre_res; E.ldres = C;
            ck(gemm_launch(A, pw, M, C, C, E, eng.splitk_ws(), eng.splitk_ws_bytes(), S(s)));
        }
        bf16* xn = ar.get<bf16>((size_t)M * C);
        LNParams L{};
        L.x = (const bf16*)mid; L.B = 1; L.N1 = M; L.N2 = 0; L.Tpad = M; L.C = C; L.eps = 1e-5f; L.y = xn;
        ck(layernorm_launch(L, S(s)));
        AOperand A;
        aoperand_rows(A, xn, C, C);
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_QKV_HEADS; E.q = bufs.q; E.k = bufs.k; E.vt = bufs.vt; E.C = C; E.H = H; E.d = d; E.DP = dp; E.DPV = dpv; E.T = N; E.vt_perm32 = vt_layout;
        E.Tpad_q = bufs.Tq_pad; E.Tpad_k = bufs.Tk_pad; E.bias = bias;
        ck(gemm_launch(A, wqkv, M, 3 * C, C, E, nullptr, 0, S(s)));
    }
    AttnParams P{};
    P.q = bufs.q; P.k = bufs.k; P.vt = bufs.vt; P.o = (bf16*)o; P.H = H; P.d = d; P.Nq = N; P.Nk = N;
    P.Tq_pad = bufs.Tq_pad; P.Tk_pad = bufs.Tk_pad; P.ldo = C; P.o_rows_per_b = N; P.vt_layout = vt_layout;
    P.scale_log2e = (float)(1.4426950408889634 / std::sqrt((double)d));
    ck(attn_launch(P, B, S(s)));
    GL_API_END
}

}  // extern "C"
