// gligen_amd engine: owns packed weights + workspace for one device and runs the GLIGEN
// denoising path (UNet forward, CFG + PLMS loop, VAE decode) as sequences of HIP kernels.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "attention.h"
#include "common.h"
#include "convnext.h"
#include "ffn.h"
#include "gemm.h"
#include "misc.h"
#include "norm.h"
#include "train.h"
#include "../../include/gligen_amd.h"

namespace gl {

struct GlError : std::runtime_error {
    int code;
    GlError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct RawTensor {
    float* p = nullptr;  // device fp32
    std::vector<int64_t> shape;
    int64_t numel = 0;
};

// Bump allocator over one hipMalloc'd slab with stack-style scopes. Addresses are a pure
// function of the allocation sequence, so a captured hipGraph replays against the same buffers.
class Arena {
   public:
    void init(size_t bytes);
    void destroy();
    void* alloc(size_t bytes);
    template <class T> T* get(size_t n) { return reinterpret_cast<T*>(alloc(n * sizeof(T))); }
    size_t mark() const { return off_; }
    void release(size_t m) { off_ = m; }
    void reset() { off_ = 0; }
    size_t capacity() const { return cap_; }
    size_t high_water() const { return hw_; }
    size_t committed() const { return vmm_ ? mapped_ : cap_; }   // device memory actually behind the reservation

   private:
    void grow(size_t need);
    char* base_ = nullptr;
    size_t cap_ = 0, off_ = 0, hw_ = 0;
    // virtual-memory form: `cap_` bytes of ADDRESSES are reserved (the bump sequence, and with it every address a captured graph
    // holds, is a pure function of the allocation sequence), physical memory is mapped behind them in chunks as the high-water
    // mark rises -- a context costs what it uses, not what it was told it might
    bool vmm_ = false;
    size_t mapped_ = 0, gran_ = 0;
    std::vector<void*> handles_;   // hipMemGenericAllocationHandle_t, one per mapped chunk
    int dev_ = 0;
};

struct NormW { const float* g = nullptr; const float* b = nullptr; int C = 0; };
struct LinW { const bf16* w = nullptr; const float* b = nullptr; int K = 0, N = 0; };
struct ConvW { const bf16* w = nullptr; const float* b = nullptr; int Cin = 0, Cout = 0, Npad = 0; };

struct ResW {
    int Cin = 0, Cout = 0;
    NormW n1, n2;
    ConvW c1, c2;
    int emb_off = -1;  // column offset into the concatenated emb_layers output (UNet only)
    bool has_skip = false;
    LinW skip;
};
// `folded`: the LayerNorm in front of the projection lives in its weights (W * gamma, bias = b + W beta, csum = row sums of the
// packed bf16 rows; gemm.h Epilogue::ln_stats): the GEMM takes the raw residual-stream rows + row statistics, or rows normalised
// WITHOUT affine by ln_kernel where no statistics exist
struct SelfAttnW {
    const bf16* wqk = nullptr;   // [to_q ; to_k] rows (wqkv when the fused projection is active: [to_q ; to_k ; to_v])
    const bf16* wv = nullptr;
    bool fused = false;          // wqk holds all three: one EPI_QKV_HEADS GEMM instead of q,k + v^T launches
    bool folded = false;
    const float* b = nullptr;    // folded: [3C] bias, [3C] csum
    const float* csum = nullptr;
    LinW out;
    // the row-local form (ffn.h qkv_rows_kernel, C = 320 / d = 40): [the C x C projection in front of this attention's LayerNorm]
    // [to_q][to_k][to_v] as one fragment stream -- attn1: SpatialTransformer.proj_in, fuser.attn: attn1.to_out
    const void* rows_stream = nullptr;
};
struct CrossAttnW { LinW q; const bf16* wk = nullptr; const bf16* wv = nullptr; LinW out; int ctx_dim = 0; bool folded = false; const float* q_csum = nullptr; };
struct FFW {
    const bf16* w1 = nullptr; const float* b1 = nullptr; LinW w2; int C = 0; int geglu16 = 0; bool folded = false; const float* csum1 = nullptr;
    const void* rows_stream = nullptr;   // the same weights as the fragment stream of the row-local kernel (ffn.hip), where one exists for C
    // the chained form (ffn.h FFRowsParams::pre / post): [projection in front][this feed-forward][projection behind] as ONE stream
    const void* chain_stream = nullptr;
    bool chain_post = false;
    // fuser.ff only: [fuser.attn.to_out][fuser.ff][attn2.to_q behind norm2] -- the cross-attention's query projection as the trailing
    // projection (FFRowsParams::post = 2); qb = its folded bias W beta
    const void* chain_q_stream = nullptr;
    const float* chain_q_bias = nullptr;
};
// partial row statistics written by the GEMM that produced a residual-stream tensor (nb = 0: none)
struct RowStats { float2* p = nullptr; int nb = 0, ld = 0; };
struct STW {
    int C = 0, d = 0, idx = 0;
    NormW gn;
    LinW proj_in, proj_out;
    NormW ln1, ln2, ln3, fn1, fn2;
    SelfAttnW a1, fa;
    CrossAttnW a2;
    FFW ff, fff;
    LinW flin;  // fuser.linear (context_dim -> C), gatedSA
    CrossAttnW fca;  // fuser.attn of gatedCA (keys / values = grounding tokens)
};

enum LayerKind { L_CONV_IN, L_RES, L_ST, L_DOWN, L_UP };
struct Layer {
    LayerKind kind;
    int idx;  // index into res_/st_/updown_ vectors
};
struct UNetBlock { std::vector<Layer> layers; };

struct VaeAttnW { NormW gn; LinW q, k, v, proj; };

// ConvNeXt-tiny tokenizer of the spatial-map modalities (reference convnext.py:52-118, canny_grounding_net.py:12-62)
struct CnxBlock { const float* dw_w = nullptr; const float* dw_b = nullptr; NormW ln; LinW pw1, pw2; };
struct CnxNet {
    bool present = false;
    int resize = 0, in_dim = 0, tokens = 0;
    const float* inconv_w = nullptr; const float* inconv_b = nullptr;   // sem: Conv2d(in_dim, 3, 3, 1, 1)
    LinW stem; NormW stem_ln;
    NormW ds_ln[3]; LinW ds[3];
    std::vector<CnxBlock> blocks[4];
    int dims[4] = {0, 0, 0, 0};
    const float* pos = nullptr; const float* null_feat = nullptr;
    LinW mlp[3];
};

struct AttnBufs {  // persistent, zero-initialised head-layout buffers for one attention shape
    bf16* q = nullptr;
    bf16* k = nullptr;
    bf16* vt = nullptr;
    int Tq_pad = 0, Tk_pad = 0;
};

// channel-concat view of up to two NHWC bf16 tensors
struct TRef { const bf16* p0 = nullptr; int C0 = 0; const bf16* p1 = nullptr; int C1 = 0; int C() const { return C0 + C1; } };

// row-local feed-forward kernel or two GEMMs (engine.hip ff_policy): -1 = by on-device timing (default), 0 = never, 1 = wherever it exists
int ff_rows_policy_set(int mode);
int ff_rows_policy_report(char* buf, size_t cap);

class Engine {
   public:
    explicit Engine(int device);
    ~Engine();
    // A second EXECUTION CONTEXT over the same packed weights (gl_ctx_fork): everything finalize() packed is shared with -- and
    // kept alive by -- `parent`; the fork owns what a forward pass writes: its arena, split-K slab, conditioning cache, attention
    // buffers, gate / fuser-scale vectors, sampler state + captured graphs, and its own copy of the restorable first conv.
    static std::shared_ptr<Engine> fork(const std::shared_ptr<Engine>& parent, size_t arena_bytes);
    size_t weight_bytes() const { return persist_bytes_; }       // device bytes this context allocated outside the arena
    size_t arena_bytes() const { return arena_.committed(); }       // device memory behind the arena (reservation: arena().capacity())
    bool is_fork() const { return (bool)parent_; }

    void upload(const std::string& key, const void* src, int ndim, const int64_t* shape, bool is_device);
    void configure_unet(const gl_unet_config& c);
    void configure_vae(const gl_vae_config& c);
    void finalize();
    bool finalized() const { return finalized_; }

    void set_cond(int Beff, const float* context, int n_ctx, const gl_grounding& g, hipStream_t s);
    void set_fuser_scale(float v, hipStream_t s);
    void set_fuser_scales(const float* scales_host, int n, hipStream_t s);
    int n_fusers() const { return (int)st_.size(); }
    void grounding_tokens(float* out, hipStream_t s);
    // PositionNet.forward of the spatial-map tokenizers: image [B][C][H][W] fp32, mask [B] -> out fp32 [B][tokens][gr_out_dim]
    void spatial_tokens(int B, const float* image, int C, int H, int W, const float* mask, float* out, hipStream_t s);
    int spatial_token_count() const { return cnx_.tokens; }  // [Beff][Ng][gr_out_dim] fp32: objs of openaimodel.py:433 for the current conditioning
    void restore_first_conv(const float* w, const float* b, hipStream_t s);
    // emb_row (the sampler): the step's emb_layers outputs [sum Cout] fp32, the same for every sample -- the time-embedding MLP and
    // the concatenated emb_layers GEMM are then not launched (emb_table_build computed them for the whole schedule)
    void unet_forward(int Beff, int h, int w, const float* x, int xB, const int64_t* t, const float* extra,
                      int extraB, float* eps, hipStream_t s, const float* emb_row = nullptr);
    void vae_decode(int B, int h, int w, const float* z, float* out, hipStream_t s);
    void vae_encode(int B, int H, int W, const float* img, const float* noise, float* z, hipStream_t s);
    bool has_vae_encoder() const { return has_venc_; }
    void sample_plms(const gl_plms_args& a, hipStream_t s);
    void sampler_timing(float* avg_ms, float* first_ms, int* n);

    // single-op entry points (tests / profiling)
    Arena& arena() { return arena_; }
    float* splitk_ws() { return ws_; }
    size_t splitk_ws_bytes() const { return ws_bytes_; }
    void* persist(size_t bytes, bool zero);
    void init_workspace();
    AttnBufs& attn_bufs(int B, int H, int d, int Tq, int Tk, int dpv_layout = 0, int slot = 0);   // dpv_layout: V^T rows when not attn_dims' (attn_vt_layout)

    // GroupNorm32 -> SiLU -> conv3x3 (stride 1, pad 1); out = nullptr: from the arena. GroupNorm-apply + SiLU run inside the conv's
    // loader where conv_halo_kernel takes the problem (gn_prologue_), as a separate pass elsewhere
    bf16* gn_silu_conv3x3(const TRef& x, int B, int H, int W, const NormW& n, float eps, const ConvW& c, const float* bias2, int bias2_ld,
                          const bf16* res, bf16* out, hipStream_t s);
    // Off by default: measured on MI355X (profiles/r6/gnconv_kbench.txt) the prologue costs conv_halo_kernel more (LDS port + VALU slots,
    // the two things that kernel is short of) than the apply pass it removes; GL_GN_PROLOGUE=1 (developer switch) turns it on in the
    // engine, gl_op_gn_silu_conv3x3(mode = 1) runs it for one operator
    bool gn_prologue_ = dev_env("GL_GN_PROLOGUE") && atoi(dev_env("GL_GN_PROLOGUE")) == 1;
    int64_t n_prologue_convs = 0;   // convs launched with the GroupNorm prologue so far (gl_op_gn_silu_conv3x3 reports which form ran)
    int device() const { return device_; }
    int64_t n_launches = 0;
    // events of the training step's gradient milestones (train.h unet_train_step: grad_events), created on first use
    static constexpr int kTrainEvents = 64;
    hipEvent_t* train_events();
    bool train_events_recorded = false;
    TrainWeightCache* train_cache = nullptr;   // gl_train_weight_cache: operand copies of the frozen parameters kept across training steps

    // ---- per-kernel profile of one eager UNetModel.forward: HIP events around every GEMM / conv / attention /
    // norm launch on the stream it is launched on, aggregated by kernel symbol (bench.py's roofline block)
    struct ProfRec { std::string name; int calls = 0; double ms = 0, flops = 0, bytes = 0; };
    void profile_begin();
    std::vector<ProfRec> profile_end(hipStream_t s);

   private:
    struct ProfEvt { std::string name; double flops, bytes; hipEvent_t e0, e1; };
    bool profiling_ = false;
    std::vector<ProfEvt> prof_;
    std::vector<hipEvent_t> prof_pool_;
    // RAII: records an event pair around the launches issued in its scope (no-op unless profiling)
    struct ProfScope {
        Engine* e; hipStream_t s; size_t idx;
        ProfScope(Engine* eng, hipStream_t st, const std::string& name, double flops, double bytes);
        ~ProfScope();
        void rename(const std::string& n) { if (e->profiling_) e->prof_[idx].name = n; }
    };
    friend struct ProfScope;
    const RawTensor& raw(const std::string& key) const;
    bool has(const std::string& key) const { return raw_.count(key) != 0; }
    const float* F(const std::string& key) const { return raw(key).p; }
    // a >= 2-D tensor that is used in fp32 as uploaded: finalize() must not drop it with the packed matrices' staging copies
    const float* FK(const std::string& key) { keep_raw_.insert(key); return raw(key).p; }
    std::unordered_set<std::string> keep_raw_;
    NormW norm(const std::string& prefix);
    LinW linear(const std::string& prefix, bool bias = true);
    ConvW conv3(const std::string& prefix, int Npad = 0);
    LinW conv1(const std::string& prefix);
    const bf16* cast_rows(const std::vector<std::string>& weight_keys);
    // pre_key / post_key: weights of the C x C projections chained in front of / behind this feed-forward ("" = none)
    // post_q_w / post_q_b: LayerNorm-folded fp32 to_q weight [C][C] and bias of the cross-attention behind this feed-forward (chain_q_stream)
    FFW ffw(const std::string& prefix, int C, const NormW* fold = nullptr, const std::string& pre_key = "", const std::string& post_key = "",
            const float* post_q_w = nullptr, const float* post_q_b = nullptr);
    // fp32 temporaries W * gamma, b + W beta of one linear layer (freed at the end of build_unet)
    struct FoldTmp { float* w = nullptr; float* b = nullptr; int N = 0, K = 0; };
    FoldTmp fold_ln(const std::string& weight_key, const float* bias, const NormW& n);
    std::vector<void*> fold_tmps_;
    bool ln_fold_ = false;       // LayerNorms folded into their consumers (GL_LN_FOLD=0: off)
    bool ff_rows_ = true;        // row-local feed-forward kernel where it exists (GL_FF_ROWS=0: off)
    int ff_chain_ = 2;           // its chained forms: 1 = attn2.to_out -> ff -> proj_out, 2 = also fuser.attn.to_out -> fuser.ff (GL_FF_CHAIN)
    ResW resw(const std::string& prefix, int Cin, int Cout, bool unet);
    void build_unet();
    void build_vae();

    // execution helpers
    void gemm(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, hipStream_t s);
    bf16* linear_rows(const bf16* x, int M, const LinW& L, int act, const bf16* res, const float* gate, hipStream_t s, RowStats* stats = nullptr);
    bf16* layernorm_plain(const bf16* x, int B, int N, int C, bool pad64, hipStream_t s);   // (x - mean) * rstd, no affine
    bf16* groupnorm(const TRef& x, int B, int HW, const NormW& n, float eps, bool silu, hipStream_t s);
    bf16* layernorm(const bf16* x, int B, int N, int C, const NormW& n, bool pad64, hipStream_t s);
    bf16* conv3x3(const TRef& x, int B, int Hin, int Win, const ConvW& c, int stride, int ups, int pad_lo,
                  const float* bias2, int bias2_ld, const bf16* res, hipStream_t s);
    bf16* resblock(const ResW& r, const TRef& x, int B, int H, int W, const float* embout, int emb_ld, float eps, hipStream_t s);
    bf16* transformer(const STW& t, const bf16* x, int B, int H, int W, hipStream_t s);
    // in_stats: `ln` holds RAW rows whose statistics are in_stats (folded LayerNorm applied by the GEMM); out_stats: statistics of the result
    // raw_rows: with a folded LayerNorm, `ln` may be the RAW rows (no in_stats needed) when ff_rows(f, M) says the row-local kernel runs
    // use_rows: run the row-local kernel (the caller asked ff_rows_for)
    bf16* feedforward(const FFW& f, const bf16* ln, int M, const bf16* res, const float* gate, hipStream_t s, const RowStats* in_stats = nullptr,
                      RowStats* out_stats = nullptr, bool raw_rows = false, bool use_rows = false);
    // Row-local kernel or LayerNorm + two GEMMs (+ the separate projections) for the feed-forward `which` (1 = fuser.ff, 2 = ff) of
    // block t at B x HW rows: decided once per (shape, chain form) by TIMING both forms on this device at the first eager launch
    // (ff_policy.* below, gl_set_ff_rows_policy); under a stream capture an undecided shape takes the static rule
    bool ff_rows_for(const STW& t, int which, int B, int HW, hipStream_t s);
    // fuser.attn.to_out (+ gated residual) -> LayerNorm -> fuser.ff (+ gated residual)  [gatedSA, attention.py:236-244]
    // q_done (optional, out): the launch also projected attn2.to_q(norm2(.)) into the cross-attention's q buffer (chain_q_stream)
    bf16* fuser_ff_tail(const STW& t, const bf16* o, const bf16* t1, int B, int HW, bool rows, hipStream_t s, RowStats* st3, bool* q_done = nullptr);
    // attn2.to_q(norm2(rows)) by GEMM into the cross-attention's q buffer (st: the rows' statistics, if their producer wrote any)
    void cross_q_gemm(const STW& t, const bf16* rows, const RowStats& st, int B, int HW, hipStream_t s);
    // attn2.to_out (+ residual) -> LayerNorm -> ff (+ residual) -> proj_out + x_in  [attention.py:337-338, 374-376]
    void block_ff_tail(const STW& t, const bf16* o, const bf16* t3, const bf16* x, bf16* out, int B, int HW, bool rows, hipStream_t s);
    bf16* ff_behind(const FFW& f, const NormW& nw, const bf16* rows_in, RowStats& st_in, int B, int HW, const float* gate, bool rows, hipStream_t s,
                    RowStats* out_stats);
    bool can_fold(const RowStats& st, int M, int C, int Nc, int mode, int act, bool aligned);
    // the chained launch: t = pre_res + pre_gate (x Wpre^T + pre_b); y = t + gate ff(LN(t)); out = y, or post_res + y Wpost^T + post_b
    struct ChainQ { bf16* q; int DP, T, Tpad; };    // post = 2: where the trailing to_q projection writes
    bf16* feedforward_chain(const FFW& f, const bf16* x, int M, const LinW& pre, const bf16* pre_res, const float* pre_gate, const float* gate,
                            const LinW* post, const bf16* post_res, bf16* out, hipStream_t s, RowStats* out_stats, const ChainQ* cq = nullptr);
    // Tbuf / slot: the head-layout buffers are sized for Tbuf tokens per sample (0: T) and private to attention number `slot`
    // (0: shared by every attention of this shape) -- the fuser's attention with hoisted grounding-token keys (fuser_kv_fill)
    // projected: q / k / v^T of this attention are already in its head-layout buffers (qkv_rows_project): launch the attention only
    void self_attention(const SelfAttnW& a, const bf16* ln, int B, int T, int Nq, int Nk, int C, int d, bf16* o, hipStream_t s,
                        const RowStats* in_stats = nullptr, int Tbuf = 0, int slot = 0, bool projected = false);
    // The row-local projection launch (ffn.h QkvRowsParams): mid = pre_res + (x Wpre^T + pre_b) -> LayerNorm in registers -> q, k, v^T
    // of attention `a` into its head-layout buffers; T tokens per sample, Nk keys decide the V^T form (attn_vt_layout)
    bool qkv_rows_ok(const SelfAttnW& a, int B, int T, int Nk, int C, int d) const;
    bool qkv_rows_for(const STW& t, int which, int B, int HW, int Nk, hipStream_t s);   // the timed choice (ff_policy), which: 0 = attn1, 1 = fuser.attn
    void qkv_project_gemm(const SelfAttnW& a, const bf16* ln, int B, int T, int Nk, int C, int d, hipStream_t s, const RowStats* in_stats, int Tbuf, int slot);
    void qkv_rows_project(const SelfAttnW& a, const bf16* x, int B, int T, int Nk, int C, int d, const LinW& pre, const bf16* pre_res, bf16* mid,
                          RowStats* mid_stats, int Tbuf, int slot, hipStream_t s);
    int qkv_rows_ = 1;           // GL_QKV_ROWS (developer A/B): 0 = off, 1 = by the timed policy (default), 2 = wherever the kernel exists
    // GatedSelfAttentionDense (attention.py:236-244): the K / V rows of the grounding tokens linear(objs) do not depend on the step.
    // They are projected ONCE per prompt into the tail (tokens HW .. HW + Ng - 1) of this block's own K / V^T buffers; per step only
    // the HW visual rows go through the (LayerNorm-folded) q,k,v^T projection -- no [x ; objs] concat LayerNorm pass.
    void fuser_kv_fill(const STW& t, int B, int HW, hipStream_t s);
    struct FuserKV { uint64_t epoch = 0; int B = 0, HW = 0; };
    std::vector<FuserKV> fuser_kv_;
    uint64_t cond_epoch_ = 0;
    bool fuser_hoist_ = true;    // GL_FUSER_KV_HOIST=0 (developer A/B): the [x ; objs] concat LayerNorm + projection over all rows every step
    bf16* vae_attn(const VaeAttnW& a, const bf16* x, int B, int HW, hipStream_t s);

    Engine(const Engine&) = default;       // (fork() copies the weight descriptors member by member, then resets the per-context state)
    Engine& operator=(const Engine&) = delete;
    std::shared_ptr<Engine> parent_;       // fork: the context that owns the shared weights
    size_t persist_bytes_ = 0;
    int device_;
    bool finalized_ = false;
    std::unordered_map<std::string, RawTensor> raw_;   // (a fork sees the parent's tensors, it does not own them)
    std::vector<void*> owned_;  // persistent device allocations
    Arena arena_;
    float* ws_ = nullptr;
    size_t ws_bytes_ = 0;

    // ---- UNet
    bool has_unet_ = false;
    gl_unet_config ucfg_{};
    LinW te0_, te2_;
    LinW embcat_;  // all ResBlock emb_layers concatenated: [sumCout][4*mc]
    ConvW conv_in_small_;  // [mc][Kpad] via small im2col
    int conv_in_kpad_ = 0;
    std::vector<ResW> res_;
    std::vector<STW> st_;
    std::vector<ConvW> updown_;
    std::vector<UNetBlock> in_blocks_, out_blocks_;
    UNetBlock mid_block_;
    NormW out_norm_;
    ConvW out_conv_;
    const float* const* alpha_ptrs_ = nullptr;  // device array [2*n_st]
    float* gates_ = nullptr;                    // device [2*n_st]: (attn, dense) per transformer
    float* fuser_scale_ = nullptr;              // device [n_st]: the fusers' external gate multipliers
    bool fuser_off_ = false;                    // host mirror of "scale == 0": the fuser branches are then skipped outright
    // grounding tokenizer
    int gkind_ = 0;
    LinW pn_[2][3];
    const float* pn_null_feat_[2] = {nullptr, nullptr};
    const float* pn_null_pos_ = nullptr;
    const float* kp_table_ = nullptr;  // keypoint: person+keypoint embedding table [P*17][out]
    CnxNet cnx_;
    void build_convnext(const std::string& PN);

    // ---- conditioning cache
    struct Cond {
        int Beff = 0, Ng = 0, ctx_T = 0, ctx_Tpad = 0, obj_Tpad = 0;
        bf16* tokens = nullptr;     // position_net output [Beff][obj_stride][gr_out_dim] (kept for gl_unet_grounding_tokens)
        int obj_stride = 0;
        std::vector<bf16*> objs;    // per transformer: [Beff*Ng][C]
        std::vector<bf16*> obj_k;   // gatedCA, per transformer: [Beff*H][obj_Tpad][DP]   (K of the grounding tokens)
        std::vector<bf16*> obj_vt;  // gatedCA, per transformer: [Beff*H][DPV][obj_Tpad]
        std::vector<bf16*> ctx_k;   // per transformer: [Beff*H][ctx_Tpad][DP]
        std::vector<bf16*> ctx_vt;  // per transformer: [Beff*H][DPV][ctx_Tpad]
        std::vector<void*> allocs;
    } cond_;

    typedef std::array<int, 7> AttnKey;            // B, H, d, Tq_pad, Tk_pad, dpv, slot
    std::map<AttnKey, AttnBufs> attn_bufs_;
    std::vector<hipEvent_t> train_events_;

    // ---- VAE decoder
    bool has_vae_ = false;
    gl_vae_config vcfg_{};
    const float* pq_w_ = nullptr;
    const float* pq_b_ = nullptr;
    ConvW vae_in_small_;
    int vae_in_kpad_ = 0;
    ResW vmid1_, vmid2_;
    VaeAttnW vattn_;
    struct VaeUp { std::vector<ResW> blocks; bool has_up = false; ConvW up; };
    std::vector<VaeUp> vup_;  // index = level (0 = full resolution)
    NormW vnorm_out_;
    ConvW vconv_out_;
    // ---- VAE encoder (inpainting configuration; built when the encoder.* / quant_conv.* weights were uploaded)
    bool has_venc_ = false;
    ConvW venc_in_small_;
    int venc_in_kpad_ = 0;
    struct VaeDown { std::vector<ResW> blocks; bool has_down = false; ConvW down; };
    std::vector<VaeDown> vdown_;
    ResW vemid1_, vemid2_;
    VaeAttnW veattn_;
    NormW venorm_out_;
    ConvW veconv_out_;
    const float* qc_w_ = nullptr;
    const float* qc_b_ = nullptr;
    void build_vae_encoder();

    // ---- sampler state
    struct Sampler {
        int B = 0, h = 0, w = 0;
        float* x2 = nullptr;
        float* eps_pair = nullptr;
        float* hist[4] = {nullptr, nullptr, nullptr, nullptr};
        float* x_tmp = nullptr;
        int64_t* t_dev = nullptr;
        // one captured [cond ; uncond] evaluation per fuser state: [0] fusers on, [1] fusers skipped (gate scale 0)
        hipGraph_t graph[2] = {nullptr, nullptr};
        hipGraphExec_t exec[2] = {nullptr, nullptr};
        hipStream_t stream = nullptr;  // engine-owned capture/replay stream (used when the caller hands the legacy default stream)
        hipStream_t run_stream = nullptr;   // the stream the last sampling run was issued on
        bool ran = false;
        hipEvent_t ev_in = nullptr, ev_out = nullptr;
        hipEvent_t ev_done = nullptr;       // engine-owned: recorded behind the last launch of every sampling run (what later calls wait for)
        std::vector<hipEvent_t> tev;  // (start, stop) per UNet evaluation of the last run
        int n_evals = 0;
        bool has_extra = false;
        const float* extra = nullptr;
        bool warm[2] = {false, false};   // this variant has run eagerly on this context (arena mapped, shapes tuned): it may be captured
        unsigned policy_epoch = 0;       // ff_policy epoch the graphs were captured under
    } smp_;
    void sampler_release_graph();
    void sampler_wait_idle();
    // time_embed + every ResBlock's emb_layers (openaimodel.py:436-437, 220-221) depend on the timestep alone, and a sampling run knows
    // its timesteps: one batched pass over the whole schedule at the start of gl_sample_plms instead of 4 tiny GEMM chains per evaluation
    void emb_table_build(const int64_t* t_host, int R, hipStream_t s);
    float* emb_table_ = nullptr;     // [rows][embcat_.N]
    float* emb_cur_ = nullptr;       // [embcat_.N]: the row of the evaluation in flight (what the captured graph reads)
    int64_t* emb_t_dev_ = nullptr;
    int emb_table_cap_ = 0;
    std::vector<int64_t> emb_t_cache_;
};

}  // namespace gl
