// gligen_amd engine — weight packing and the kernel schedules of UNetModel.forward,
// AutoencoderKL.decode and the CFG + PLMS sampling loop (see engine.h, include/gligen_amd.h).
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>

namespace gl {

static thread_local char g_err[2048] = "";
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_error() { return g_err; }

#define CK(expr)                                              \
    do {                                                      \
        int _r = (expr);                                      \
        if (_r != GL_OK) throw GlError(_r, gl::last_error()); \
    } while (0)
#define HIPCK(expr)                                                                             \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            throw GlError(GL_ERR_HIP, std::string(#expr) + " -> " + hipGetErrorString(_e));     \
    } while (0)

static std::string fmt(const char* f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}

// ---------------------------------------------------------------- Arena
void Arena::init(size_t bytes) {
    off_ = hw_ = 0;
    HIPCK(hipGetDevice(&dev_));
    int vmm = 0;
    const char* sw = dev_env("GL_ARENA_VMM");
    if (!(sw && atoi(sw) == 0) && hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev_) == hipSuccess && vmm) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev_;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
            const size_t want = (bytes + gran - 1) / gran * gran;
            void* p = nullptr;
            if (hipMemAddressReserve(&p, want, gran, nullptr, 0) == hipSuccess && p) {
                base_ = reinterpret_cast<char*>(p);
                cap_ = want;
                gran_ = gran;
                vmm_ = true;
                mapped_ = 0;
                return;
            }
        }
        (void)hipGetLastError();
    }
    HIPCK(hipMalloc(reinterpret_cast<void**>(&base_), bytes));
    cap_ = bytes;
}
void Arena::grow(size_t need) {
    // 64 MiB steps (a multiple of the granularity): a few dozen mappings for a 1 GB high-water mark, none once it is reached
    const size_t step = std::max(gran_, ((size_t(64) << 20) + gran_ - 1) / gran_ * gran_);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev_;
    while (mapped_ < need) {
        const size_t chunk = std::min(step, cap_ - mapped_);
        hipMemGenericAllocationHandle_t h;
        HIPCK(hipMemCreate(&h, chunk, &prop, 0));
        hipError_t e = hipMemMap(base_ + mapped_, chunk, 0, h, 0);
        if (e != hipSuccess) {
            (void)hipMemRelease(h);
            throw GlError(GL_ERR_HIP, fmt("workspace arena: hipMemMap of %zu bytes failed: %s", chunk, hipGetErrorString(e)));
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(base_ + mapped_, chunk, &acc, 1);
        if (e != hipSuccess) {
            (void)hipMemUnmap(base_ + mapped_, chunk);
            (void)hipMemRelease(h);
            throw GlError(GL_ERR_HIP, fmt("workspace arena: hipMemSetAccess failed: %s", hipGetErrorString(e)));
        }
        handles_.push_back(reinterpret_cast<void*>(h));
        mapped_ += chunk;
    }
}
void Arena::destroy() {
    if (vmm_) {
        const size_t step = handles_.empty() ? 0 : std::max(gran_, ((size_t(64) << 20) + gran_ - 1) / gran_ * gran_);
        size_t at = 0;
        for (void* h : handles_) {
            const size_t chunk = std::min(step, cap_ - at);
            (void)hipMemUnmap(base_ + at, chunk);
            (void)hipMemRelease(reinterpret_cast<hipMemGenericAllocationHandle_t>(h));
            at += chunk;
        }
        handles_.clear();
        if (base_) (void)hipMemAddressFree(base_, cap_);
        vmm_ = false;
        mapped_ = 0;
    } else if (base_) {
        (void)hipFree(base_);
    }
    base_ = nullptr;
}
void* Arena::alloc(size_t bytes) {
    size_t a = (off_ + 255) & ~size_t(255);
    if (a + bytes > cap_)
        throw GlError(GL_ERR_STATE, fmt("workspace arena exhausted: need %zu more bytes (capacity %zu); "
                                        "create the context with a larger arena", bytes, cap_));
    if (vmm_ && a + bytes > mapped_) grow(a + bytes);
    off_ = a + bytes;
    hw_ = std::max(hw_, off_);
    return base_ + a;
}

// ---------------------------------------------------------------- Engine basics
Engine::Engine(int device) : device_(device) {}

Engine::~Engine() {
    sampler_release_graph();
    for (hipEvent_t e : smp_.tev) (void)hipEventDestroy(e);
    if (smp_.ev_in) (void)hipEventDestroy(smp_.ev_in);
    if (smp_.ev_out) (void)hipEventDestroy(smp_.ev_out);
    if (smp_.ev_done) (void)hipEventDestroy(smp_.ev_done);
    if (smp_.stream) (void)hipStreamDestroy(smp_.stream);
    for (hipEvent_t e : train_events_)
        if (e) (void)hipEventDestroy(e);
    if (!parent_)
        for (auto& kv : raw_)
            if (kv.second.p) (void)hipFree(kv.second.p);
    train_cache_destroy(train_cache);
    for (void* p : owned_) (void)hipFree(p);
    for (void* p : cond_.allocs) (void)hipFree(p);
    arena_.destroy();
}

// ---------------------------------------------------------------- per-kernel profile
Engine::ProfScope::ProfScope(Engine* eng, hipStream_t st, const std::string& name, double flops, double bytes) : e(eng), s(st), idx(0) {
    if (!e->profiling_) return;
    auto get = [&]() {
        hipEvent_t ev;
        if (!e->prof_pool_.empty()) { ev = e->prof_pool_.back(); e->prof_pool_.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) throw GlError(GL_ERR_HIP, "hipEventCreate failed");
        return ev;
    };
    ProfEvt pe{name, flops, bytes, get(), get()};
    idx = e->prof_.size();
    e->prof_.push_back(pe);
    (void)hipEventRecord(pe.e0, s);
}
Engine::ProfScope::~ProfScope() {
    if (e->profiling_) (void)hipEventRecord(e->prof_[idx].e1, s);
}
hipEvent_t* Engine::train_events() {
    if (train_events_.empty()) {
        train_events_.resize(kTrainEvents, nullptr);
        for (auto& e : train_events_) HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return train_events_.data();
}

void Engine::profile_begin() {
    prof_.clear();
    profiling_ = true;
}
std::vector<Engine::ProfRec> Engine::profile_end(hipStream_t s) {
    profiling_ = false;
    HIPCK(hipStreamSynchronize(s));
    std::vector<ProfRec> out;
    for (auto& pe : prof_) {
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, pe.e0, pe.e1));
        prof_pool_.push_back(pe.e0);
        prof_pool_.push_back(pe.e1);
        auto it = std::find_if(out.begin(), out.end(), [&](const ProfRec& r) { return r.name == pe.name; });
        if (it == out.end()) { out.push_back(ProfRec{pe.name}); it = out.end() - 1; }
        it->calls += 1; it->ms += ms; it->flops += pe.flops; it->bytes += pe.bytes;
    }
    prof_.clear();
    std::sort(out.begin(), out.end(), [](const ProfRec& a, const ProfRec& b) { return a.ms > b.ms; });
    return out;
}

void Engine::init_workspace() {
    if (ws_) return;
    ws_bytes_ = size_t(256) << 20;  // fp32 split-K slabs
    ws_ = reinterpret_cast<float*>(persist(ws_bytes_, false));
}

void* Engine::persist(size_t bytes, bool zero) {
    void* p = nullptr;
    HIPCK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
    if (zero) HIPCK(hipMemset(p, 0, std::max<size_t>(bytes, 256)));
    owned_.push_back(p);
    persist_bytes_ += std::max<size_t>(bytes, 256);
    return p;
}

std::shared_ptr<Engine> Engine::fork(const std::shared_ptr<Engine>& parent, size_t arena_bytes) {
    if (!parent || !parent->finalized_) throw GlError(GL_ERR_STATE, "gl_ctx_fork: the parent context is not finalized");
    HIPCK(hipDeviceSynchronize());                     // (the first-conv copy below reads what the parent's streams may still be writing)
    std::shared_ptr<Engine> c(new Engine(*parent));    // every weight descriptor: plain pointers into the parent's allocations
    Engine& e = *c;
    e.parent_ = parent->parent_ ? parent->parent_ : parent;
    // ---- what a context owns itself
    e.owned_.clear();
    e.persist_bytes_ = 0;
    e.fold_tmps_.clear();
    e.arena_ = Arena{};
    e.ws_ = nullptr;
    e.ws_bytes_ = 0;
    e.cond_ = Cond{};
    e.attn_bufs_.clear();
    e.smp_ = Sampler{};
    e.train_events_.clear();
    e.fuser_kv_.clear();
    e.emb_table_ = nullptr; e.emb_cur_ = nullptr; e.emb_t_dev_ = nullptr; e.emb_table_cap_ = 0; e.emb_t_cache_.clear();
    e.train_events_recorded = false;
    e.profiling_ = false;
    e.prof_.clear();
    e.prof_pool_.clear();
    e.n_launches = 0;
    try {
        e.arena_.init(arena_bytes ? arena_bytes : parent->arena_.capacity());
        e.init_workspace();
        if (e.has_unet_) {
            const size_t ng = 2 * e.st_.size();
            e.gates_ = reinterpret_cast<float*>(e.persist(std::max<size_t>(ng, 1) * sizeof(float), true));
            e.fuser_scale_ = reinterpret_cast<float*>(e.persist(std::max<size_t>(e.st_.size(), 1) * sizeof(float), true));
            if (ng) HIPCK(hipMemcpy(e.gates_, parent->gates_, ng * sizeof(float), hipMemcpyDeviceToDevice));
            if (!e.st_.empty()) HIPCK(hipMemcpy(e.fuser_scale_, parent->fuser_scale_, e.st_.size() * sizeof(float), hipMemcpyDeviceToDevice));
            // the first conv is rewritten in place by gl_unet_restore_first_conv: a copy per context, in the parent's current state
            const int mc = e.ucfg_.model_channels;
            const size_t wb = (size_t)mc * e.conv_in_kpad_ * sizeof(bf16);
            bf16* w = reinterpret_cast<bf16*>(e.persist(wb, false));
            float* b = reinterpret_cast<float*>(e.persist(mc * sizeof(float), false));
            HIPCK(hipMemcpy(w, parent->conv_in_small_.w, wb, hipMemcpyDeviceToDevice));
            HIPCK(hipMemcpy(b, parent->conv_in_small_.b, mc * sizeof(float), hipMemcpyDeviceToDevice));
            e.conv_in_small_.w = w;
            e.conv_in_small_.b = b;
        }
    } catch (...) {
        c.reset();
        throw;
    }
    return c;
}

void Engine::upload(const std::string& key, const void* src, int ndim, const int64_t* shape, bool is_device) {
    if (finalized_) throw GlError(GL_ERR_STATE, "weights cannot be uploaded after gl_finalize");
    RawTensor t;
    t.numel = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        t.numel *= shape[i];
    }
    HIPCK(hipMalloc(reinterpret_cast<void**>(&t.p), std::max<int64_t>(t.numel, 4) * sizeof(float)));
    HIPCK(hipMemcpy(t.p, src, t.numel * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    auto it = raw_.find(key);
    if (it != raw_.end()) (void)hipFree(it->second.p);
    raw_[key] = t;
}

const RawTensor& Engine::raw(const std::string& key) const {
    auto it = raw_.find(key);
    if (it == raw_.end()) throw GlError(GL_ERR_MISSING, "missing weight '" + key + "'");
    return it->second;
}

void Engine::configure_unet(const gl_unet_config& c) {
    if (c.n_mult < 1 || c.n_mult > 8 || c.n_attn < 0 || c.n_attn > 8) throw GlError(GL_ERR_ARG, "bad unet config");
    if (c.model_channels % 64 != 0) throw GlError(GL_ERR_UNSUPPORTED, "model_channels must be a multiple of 64");
    if (c.fuser_kind < 0 || c.fuser_kind > 2) throw GlError(GL_ERR_ARG, "fuser_kind: 0 gatedSA, 1 gatedSA2, 2 gatedCA");
    if (c.extra_channels < 0 || c.extra_channels > 64) throw GlError(GL_ERR_ARG, "extra_channels out of range");
    // openaimodel.py:446-447 is a breakpoint() in the reference: no shipped model combines the two
    if (c.extra_channels && c.inpaint_mode) throw GlError(GL_ERR_UNSUPPORTED, "inpaint_mode with a grounding downsampler is undefined in the reference");
    ucfg_ = c;
    has_unet_ = true;
}
void Engine::configure_vae(const gl_vae_config& c) {
    if (c.n_mult < 1 || c.n_mult > 8) throw GlError(GL_ERR_ARG, "bad vae config");
    vcfg_ = c;
    has_vae_ = true;
}

// ---------------------------------------------------------------- weight packing
NormW Engine::norm(const std::string& p) {
    NormW n;
    n.g = F(p + ".weight");
    n.b = F(p + ".bias");
    n.C = (int)raw(p + ".weight").numel;
    return n;
}

const bf16* Engine::cast_rows(const std::vector<std::string>& keys) {
    int64_t total = 0;
    for (auto& k : keys) total += raw(k).numel;
    bf16* dst = reinterpret_cast<bf16*>(persist(total * sizeof(bf16), false));
    int64_t off = 0;
    for (auto& k : keys) {
        const RawTensor& t = raw(k);
        CK(cast_f32_bf16_launch(t.p, dst + off, t.numel, 0));
        off += t.numel;
    }
    return dst;
}

LinW Engine::linear(const std::string& p, bool bias) {
    LinW l;
    const RawTensor& w = raw(p + ".weight");
    l.N = (int)w.shape[0];
    l.K = (int)(w.numel / w.shape[0]);
    if (l.K % 64 != 0) {  // zero-pad K (keypoint PositionNet: 768 + 32 -> 832)
        int Kp = round_up(l.K, 64);
        bf16* dst = reinterpret_cast<bf16*>(persist((size_t)l.N * Kp * sizeof(bf16), false));
        CK(cast_pad_cols_launch(w.p, dst, l.N, l.K, Kp, 0));
        l.w = dst;
        l.K = Kp;
    } else {
        l.w = cast_rows({p + ".weight"});
    }
    l.b = bias ? F(p + ".bias") : nullptr;
    return l;
}

LinW Engine::conv1(const std::string& p) { return linear(p, true); }

ConvW Engine::conv3(const std::string& p, int Npad) {
    ConvW c;
    const RawTensor& w = raw(p + ".weight");
    if (w.shape.size() != 4 || w.shape[2] != 3 || w.shape[3] != 3) throw GlError(GL_ERR_ARG, "'" + p + ".weight' is not a 3x3 conv");
    c.Cout = (int)w.shape[0];
    c.Cin = (int)w.shape[1];
    c.Npad = Npad ? Npad : c.Cout;
    bf16* dst = reinterpret_cast<bf16*>(persist((size_t)c.Npad * 9 * c.Cin * sizeof(bf16), false));
    CK(pack_conv_weight_launch(w.p, dst, c.Cout, c.Cin, 3, 3, c.Npad, 0));
    c.w = dst;
    if (c.Npad != c.Cout) {
        float* b = reinterpret_cast<float*>(persist(c.Npad * sizeof(float), true));
        HIPCK(hipMemcpy(b, F(p + ".bias"), c.Cout * sizeof(float), hipMemcpyDeviceToDevice));
        c.b = b;
    } else {
        c.b = F(p + ".bias");
    }
    return c;
}

Engine::FoldTmp Engine::fold_ln(const std::string& wkey, const float* bias, const NormW& n) {
    const RawTensor& w = raw(wkey);
    FoldTmp t;
    t.N = (int)w.shape[0];
    t.K = (int)(w.numel / w.shape[0]);
    if (n.C != t.K) throw GlError(GL_ERR_ARG, "'" + wkey + "': LayerNorm width does not match the projection's input width");
    HIPCK(hipMalloc(reinterpret_cast<void**>(&t.w), (size_t)t.N * t.K * sizeof(float)));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&t.b), (size_t)t.N * sizeof(float)));
    fold_tmps_.push_back(t.w);
    fold_tmps_.push_back(t.b);
    CK(ln_fold_launch(w.p, bias, n.g, n.b, t.w, t.b, t.N, t.K, 0));
    return t;
}

FFW Engine::ffw(const std::string& p, int C, const NormW* fold, const std::string& pre_key, const std::string& post_key, const float* post_q_w,
                const float* post_q_b) {
    FFW f;
    f.C = C;
    const RawTensor& w = raw(p + ".net.0.proj.weight");
    const int C8 = (int)w.shape[0], K = (int)w.shape[1];
    if (C8 != 8 * C || K != C) throw GlError(GL_ERR_ARG, "'" + p + "' GEGLU projection has unexpected shape");
    bf16* wp = reinterpret_cast<bf16*>(persist((size_t)C8 * K * sizeof(bf16), false));
    float* bp = reinterpret_cast<float*>(persist(C8 * sizeof(float), false));
    f.geglu16 = gemm_geglu_layout();
    const float* wsrc = w.p;
    const float* bsrc = F(p + ".net.0.proj.bias");
    if (fold) {   // the LayerNorm in front of this feed-forward, folded into its GEGLU projection
        const FoldTmp t = fold_ln(p + ".net.0.proj.weight", bsrc, *fold);
        wsrc = t.w;
        bsrc = t.b;
    }
    CK(pack_geglu_launch(wsrc, bsrc, wp, bp, 4 * C, K, f.geglu16, 0));
    if (fold) {
        float* cs = reinterpret_cast<float*>(persist(C8 * sizeof(float), false));
        CK(rowsum_bf16_launch(wp, cs, C8, K, 0));
        f.csum1 = cs;
        f.folded = true;
    }
    f.w1 = wp;
    f.b1 = bp;
    f.w2 = linear(p + ".net.2");
    if (ff_rows_ && ff_stream_bytes(C)) {   // the row-local kernel's fragment stream of the same (folded) weights
        void* st = persist(ff_stream_bytes(C), false);
        CK(ff_pack_launch(wsrc, bsrc, raw(p + ".net.2.weight").p, st, C, 0));
        f.rows_stream = st;
        if (fold && !pre_key.empty()) {      // chained form: the projections around this feed-forward ride in the same stream
            const bool post = !post_key.empty();
            void* cs = persist(ff_chain_stream_bytes(C, true, post), false);
            CK(ff_chain_pack_launch(wsrc, bsrc, raw(p + ".net.2.weight").p, raw(pre_key).p, post ? raw(post_key).p : nullptr, cs, C, 0));
            f.chain_stream = cs;
            f.chain_post = post;
            if (!post && post_q_w && post_q_b && C == 320) {     // + the cross-attention's to_q behind the next LayerNorm (d = 40: C / 8 heads)
                void* cq = persist(ff_chain_stream_bytes(C, true, true), false);
                CK(ff_chain_pack_launch(wsrc, bsrc, raw(p + ".net.2.weight").p, raw(pre_key).p, post_q_w, cq, C, 0));
                float* qb = reinterpret_cast<float*>(persist((size_t)C * sizeof(float), false));
                HIPCK(hipMemcpy(qb, post_q_b, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice));
                f.chain_q_stream = cq;
                f.chain_q_bias = qb;
            }
        }
    }
    return f;
}

ResW Engine::resw(const std::string& p, int Cin, int Cout, bool unet) {
    ResW r;
    r.Cin = Cin;
    r.Cout = Cout;
    if (unet) {
        r.n1 = norm(p + ".in_layers.0");
        r.c1 = conv3(p + ".in_layers.2");
        r.n2 = norm(p + ".out_layers.0");
        r.c2 = conv3(p + ".out_layers.3");
        r.has_skip = has(p + ".skip_connection.weight");
        if (r.has_skip) r.skip = conv1(p + ".skip_connection");
    } else {
        r.n1 = norm(p + ".norm1");
        r.c1 = conv3(p + ".conv1");
        r.n2 = norm(p + ".norm2");
        r.c2 = conv3(p + ".conv2");
        r.has_skip = has(p + ".nin_shortcut.weight");
        if (r.has_skip) r.skip = conv1(p + ".nin_shortcut");
        if (has(p + ".conv_shortcut.weight")) throw GlError(GL_ERR_UNSUPPORTED, "'" + p + "': 3x3 conv_shortcut is not supported");
    }
    if (r.c1.Cin != Cin || r.c1.Cout != Cout || (Cin != Cout) != r.has_skip)
        throw GlError(GL_ERR_ARG, fmt("'%s': weights do not match the configured topology (%d -> %d)", p.c_str(), Cin, Cout));
    return r;
}

void Engine::build_unet() {
    const gl_unet_config& c = ucfg_;
    const int mc = c.model_channels;
    const std::string U = "unet/";
    auto in_attn = [&](int ds) {
        for (int i = 0; i < c.n_attn; ++i)
            if (c.attention_resolutions[i] == ds) return true;
        return false;
    };
    te0_ = linear(U + "time_embed.0");
    te2_ = linear(U + "time_embed.2");

    // first conv through the small-channel im2col path (K = 9*in_c padded to 64)
    {
        const RawTensor& w = raw(U + "input_blocks.0.0.weight");
        const int in_c = (int)w.shape[1];
        const int expect = c.inpaint_mode ? 2 * c.in_channels + 1 : c.in_channels + c.extra_channels;
        if (in_c != expect) throw GlError(GL_ERR_ARG, fmt("first conv has %d input channels, config implies %d", in_c, expect));
        conv_in_kpad_ = round_up(9 * in_c, 64);
        bf16* dst = reinterpret_cast<bf16*>(persist((size_t)mc * conv_in_kpad_ * sizeof(bf16), false));
        CK(pack_conv_small_launch(w.p, dst, mc, in_c, conv_in_kpad_, 0));
        conv_in_small_.w = dst;
        float* bcopy = reinterpret_cast<float*>(persist(mc * sizeof(float), false));  // engine-owned: restorable
        HIPCK(hipMemcpy(bcopy, F(U + "input_blocks.0.0.bias"), mc * sizeof(float), hipMemcpyDeviceToDevice));
        conv_in_small_.b = bcopy;
        conv_in_small_.Cin = in_c;
        conv_in_small_.Cout = mc;
    }

    std::vector<std::string> emb_keys;
    std::vector<const float*> emb_bias;
    std::vector<int> emb_n;
    int emb_total = 0;
    auto add_res = [&](const std::string& p, int Cin, int Cout) {
        ResW r = resw(p, Cin, Cout, true);
        r.emb_off = emb_total;
        emb_keys.push_back(p + ".emb_layers.1.weight");
        emb_bias.push_back(F(p + ".emb_layers.1.bias"));
        emb_n.push_back(Cout);
        emb_total += Cout;
        res_.push_back(r);
        return Layer{L_RES, (int)res_.size() - 1};
    };
    auto add_st = [&](const std::string& p, int C) {
        STW t;
        t.C = C;
        t.d = C / c.num_heads;
        t.idx = (int)st_.size();
        int dp, dpv;
        CK(attn_dims(t.d, &dp, &dpv));
        const std::string tb = p + ".transformer_blocks.0";
        t.gn = norm(p + ".norm");
        t.proj_in = conv1(p + ".proj_in");
        t.proj_out = conv1(p + ".proj_out");
        t.ln1 = norm(tb + ".norm1");
        t.ln2 = norm(tb + ".norm2");
        t.ln3 = norm(tb + ".norm3");
        // q, k and v^T of a self-attention come out of ONE GEMM over the LayerNorm'ed rows (EPI_QKV_HEADS) when the head
        // count / width allow it; GL_QKV_FUSED=0 keeps the two-launch form (q,k GEMM + operand-swapped v^T GEMM) for A/B runs
        const bool want_fused = !(dev_env("GL_QKV_FUSED") && atoi(dev_env("GL_QKV_FUSED")) == 0);
        const bool fuse_qkv = want_fused && gemm_supports_qkv() && (2 * C) % 128 == 0;
        // LayerNorms folded into the projections behind them (gemm.h Epilogue::ln_stats; GL_LN_FOLD=0: the LayerNorm kernels of
        // rounds 1-2): norm1 -> attn1 q,k,v; fuser.norm1 -> fuser q,k,v; fuser.norm2 -> fuser.ff; norm2 -> attn2.to_q; norm3 -> ff
        const bool fold = fuse_qkv && !(dev_env("GL_LN_FOLD") && atoi(dev_env("GL_LN_FOLD")) == 0);
        ln_fold_ = fold;
        ff_rows_ = !(dev_env("GL_FF_ROWS") && atoi(dev_env("GL_FF_ROWS")) == 0);   // row-local feed-forward kernel (ffn.hip) where it exists
        ff_chain_ = dev_env("GL_FF_CHAIN") ? atoi(dev_env("GL_FF_CHAIN")) : 3;
        fuser_hoist_ = !(dev_env("GL_FUSER_KV_HOIST") && atoi(dev_env("GL_FUSER_KV_HOIST")) == 0);
        qkv_rows_ = dev_env("GL_QKV_ROWS") ? atoi(dev_env("GL_QKV_ROWS")) : 1;
        // pre_key: weight of the C x C projection in front of this attention's LayerNorm (row-local form, ffn.h qkv_rows_kernel)
        auto self_attn_w = [&](const std::string& a, const NormW* ln, const std::string& pre_key) {
            SelfAttnW w;
            w.fused = fuse_qkv;
            if (fuse_qkv && ln) {
                const char* names[3] = {".to_q.weight", ".to_k.weight", ".to_v.weight"};
                bf16* dst = reinterpret_cast<bf16*>(persist((size_t)3 * C * C * sizeof(bf16), false));
                float* bias = reinterpret_cast<float*>(persist((size_t)3 * C * sizeof(float), false));
                float* cs = reinterpret_cast<float*>(persist((size_t)3 * C * sizeof(float), false));
                for (int i = 0; i < 3; ++i) {
                    const FoldTmp t = fold_ln(a + names[i], nullptr, *ln);
                    if (t.N != C || t.K != C) throw GlError(GL_ERR_ARG, "'" + a + "': q / k / v projections must be C x C");
                    CK(cast_f32_bf16_launch(t.w, dst + (size_t)i * C * C, (int64_t)C * C, 0));
                    HIPCK(hipMemcpy(bias + (size_t)i * C, t.b, C * sizeof(float), hipMemcpyDeviceToDevice));
                }
                CK(rowsum_bf16_launch(dst, cs, 3 * C, C, 0));
                w.wqk = dst; w.b = bias; w.csum = cs; w.folded = true;
                if (qkv_rows_ && !pre_key.empty() && t.d == 40 && qkv_rows_stream_bytes(C, true, 3)) {
                    void* st = persist(qkv_rows_stream_bytes(C, true, 3), false);
                    CK(qkv_rows_pack_launch(raw(pre_key).p, dst, 3, st, C, 0));
                    w.rows_stream = st;
                }
            } else if (fuse_qkv) {
                w.wqk = cast_rows({a + ".to_q.weight", a + ".to_k.weight", a + ".to_v.weight"});
            } else {
                w.wqk = cast_rows({a + ".to_q.weight", a + ".to_k.weight"});
                w.wv = cast_rows({a + ".to_v.weight"});
            }
            return w;
        };
        t.a1 = self_attn_w(tb + ".attn1", fold ? &t.ln1 : nullptr, p + ".proj_in.weight");
        t.a1.out = linear(tb + ".attn1.to_out.0");
        FoldTmp qfold;
        if (fold) {
            const FoldTmp q = fold_ln(tb + ".attn2.to_q.weight", nullptr, t.ln2);
            qfold = q;
            bf16* dst = reinterpret_cast<bf16*>(persist((size_t)q.N * q.K * sizeof(bf16), false));
            float* bias = reinterpret_cast<float*>(persist((size_t)q.N * sizeof(float), false));
            float* cs = reinterpret_cast<float*>(persist((size_t)q.N * sizeof(float), false));
            CK(cast_f32_bf16_launch(q.w, dst, (int64_t)q.N * q.K, 0));
            HIPCK(hipMemcpy(bias, q.b, q.N * sizeof(float), hipMemcpyDeviceToDevice));
            CK(rowsum_bf16_launch(dst, cs, q.N, q.K, 0));
            t.a2.q.w = dst; t.a2.q.b = bias; t.a2.q.N = q.N; t.a2.q.K = q.K;
            t.a2.q_csum = cs; t.a2.folded = true;
        } else {
            t.a2.q = linear(tb + ".attn2.to_q", false);
        }
        t.a2.wk = cast_rows({tb + ".attn2.to_k.weight"});
        t.a2.wv = cast_rows({tb + ".attn2.to_v.weight"});
        t.a2.ctx_dim = (int)raw(tb + ".attn2.to_k.weight").shape[1];
        t.a2.out = linear(tb + ".attn2.to_out.0");
        t.ff = ffw(tb + ".ff", C, fold ? &t.ln3 : nullptr, ff_chain_ >= 1 ? tb + ".attn2.to_out.0.weight" : "", ff_chain_ >= 1 ? p + ".proj_out.weight" : "");
        if (has(tb + ".fuser.linear.weight") != (c.fuser_kind != 2))
            throw GlError(GL_ERR_ARG, "fuser weights do not match fuser_kind (gatedSA has fuser.linear, gatedCA does not)");
        t.fn1 = norm(tb + ".fuser.norm1");
        t.fn2 = norm(tb + ".fuser.norm2");
        if (c.fuser_kind != 2) {  // gatedSA and gatedSA2 hold the same parameters
            t.flin = linear(tb + ".fuser.linear");
            t.fa = self_attn_w(tb + ".fuser.attn", fold ? &t.fn1 : nullptr, tb + ".attn1.to_out.0.weight");
            t.fa.out = linear(tb + ".fuser.attn.to_out.0");
        } else {  // gatedCA: CrossAttention(query_dim, key_dim = value_dim = grounding-token dim) -- attention.py:194
            t.fca.q = linear(tb + ".fuser.attn.to_q", false);
            t.fca.wk = cast_rows({tb + ".fuser.attn.to_k.weight"});
            t.fca.wv = cast_rows({tb + ".fuser.attn.to_v.weight"});
            t.fca.ctx_dim = (int)raw(tb + ".fuser.attn.to_k.weight").shape[1];
            if (t.fca.ctx_dim != c.gr_out_dim || (int)raw(tb + ".fuser.attn.to_v.weight").shape[1] != c.gr_out_dim)
                throw GlError(GL_ERR_ARG, "gatedCA: fuser.attn key / value dim must equal the grounding-token dim");
            t.fca.out = linear(tb + ".fuser.attn.to_out.0");
        }
        const bool chain_q = ff_chain_ >= 3 && c.fuser_kind == 0 && fold && t.d == 40 && qfold.N == C && qfold.K == C;
        t.fff = ffw(tb + ".fuser.ff", C, fold ? &t.fn2 : nullptr, (ff_chain_ >= 2 && c.fuser_kind == 0) ? tb + ".fuser.attn.to_out.0.weight" : "", "",
                    chain_q ? qfold.w : nullptr, chain_q ? qfold.b : nullptr);
        raw(tb + ".fuser.alpha_attn");
        raw(tb + ".fuser.alpha_dense");
        st_.push_back(t);
        return Layer{L_ST, t.idx};
    };

    std::vector<std::string> st_prefix;  // for alpha pointer table
    in_blocks_.clear();
    in_blocks_.push_back(UNetBlock{{Layer{L_CONV_IN, 0}}});
    std::vector<int> chans{mc};
    int ch = mc, ds = 1, n = 1;
    for (int level = 0; level < c.n_mult; ++level) {
        const int mult = c.channel_mult[level];
        for (int r = 0; r < c.num_res_blocks; ++r) {
            UNetBlock b;
            const std::string p = U + fmt("input_blocks.%d", n);
            b.layers.push_back(add_res(p + ".0", ch, mult * mc));
            ch = mult * mc;
            if (in_attn(ds)) {
                b.layers.push_back(add_st(p + ".1", ch));
                st_prefix.push_back(p + ".1");
            }
            in_blocks_.push_back(b);
            chans.push_back(ch);
            ++n;
        }
        if (level != c.n_mult - 1) {
            updown_.push_back(conv3(U + fmt("input_blocks.%d.0.op", n)));
            in_blocks_.push_back(UNetBlock{{Layer{L_DOWN, (int)updown_.size() - 1}}});
            chans.push_back(ch);
            ds *= 2;
            ++n;
        }
    }
    mid_block_.layers.clear();
    mid_block_.layers.push_back(add_res(U + "middle_block.0", ch, ch));
    mid_block_.layers.push_back(add_st(U + "middle_block.1", ch));
    st_prefix.push_back(U + "middle_block.1");
    mid_block_.layers.push_back(add_res(U + "middle_block.2", ch, ch));

    out_blocks_.clear();
    n = 0;
    for (int level = c.n_mult - 1; level >= 0; --level) {
        const int mult = c.channel_mult[level];
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            const int ich = chans.back();
            chans.pop_back();
            UNetBlock b;
            const std::string p = U + fmt("output_blocks.%d", n);
            b.layers.push_back(add_res(p + ".0", ch + ich, mc * mult));
            ch = mc * mult;
            int j = 1;
            if (in_attn(ds)) {
                b.layers.push_back(add_st(p + ".1", ch));
                st_prefix.push_back(p + ".1");
                j = 2;
            }
            if (level && i == c.num_res_blocks) {
                updown_.push_back(conv3(p + fmt(".%d.conv", j)));
                b.layers.push_back(Layer{L_UP, (int)updown_.size() - 1});
                ds /= 2;
            }
            out_blocks_.push_back(b);
            ++n;
        }
    }
    out_norm_ = norm(U + "out.0");
    out_conv_ = conv3(U + "out.2", 32);
    if (out_conv_.Cout != c.out_channels) throw GlError(GL_ERR_ARG, "out conv channels do not match out_channels");

    // all emb_layers in one GEMM
    embcat_.w = cast_rows(emb_keys);
    embcat_.K = 4 * mc;
    embcat_.N = emb_total;
    {
        float* b = reinterpret_cast<float*>(persist(emb_total * sizeof(float), false));
        int off = 0;
        for (size_t i = 0; i < emb_bias.size(); ++i) {
            HIPCK(hipMemcpy(b + off, emb_bias[i], emb_n[i] * sizeof(float), hipMemcpyDeviceToDevice));
            off += emb_n[i];
        }
        embcat_.b = b;
    }

    // fuser gates
    {
        std::vector<const float*> ptrs;
        for (auto& p : st_prefix) {
            ptrs.push_back(F(p + ".transformer_blocks.0.fuser.alpha_attn"));
            ptrs.push_back(F(p + ".transformer_blocks.0.fuser.alpha_dense"));
        }
        void* d = persist(ptrs.size() * sizeof(float*), false);
        HIPCK(hipMemcpy(d, ptrs.data(), ptrs.size() * sizeof(float*), hipMemcpyHostToDevice));
        alpha_ptrs_ = reinterpret_cast<const float* const*>(d);
        gates_ = reinterpret_cast<float*>(persist(ptrs.size() * sizeof(float), true));
        fuser_scale_ = reinterpret_cast<float*>(persist(st_.size() * sizeof(float), true));   // one scale per fuser
        CK(fill_f32_launch(fuser_scale_, 1.f, (int)st_.size(), 0));
    }

    // grounding tokenizer (position_net)
    gkind_ = c.grounding_kind;
    const std::string PN = U + "position_net.";
    if (gkind_ == 0) {
        for (int i = 0; i < 3; ++i) pn_[0][i] = linear(PN + fmt("linears.%d", 2 * i));
        pn_null_feat_[0] = F(PN + "null_positive_feature");
        pn_null_pos_ = F(PN + "null_position_feature");
    } else if (gkind_ == 1) {
        for (int i = 0; i < 3; ++i) {
            pn_[0][i] = linear(PN + fmt("linears_text.%d", 2 * i));
            pn_[1][i] = linear(PN + fmt("linears_image.%d", 2 * i));
        }
        pn_null_feat_[0] = F(PN + "null_text_feature");
        pn_null_feat_[1] = F(PN + "null_image_feature");
        pn_null_pos_ = F(PN + "null_position_feature");
    } else if (gkind_ == 2) {
        for (int i = 0; i < 3; ++i) pn_[0][i] = linear(PN + fmt("linears.%d", 2 * i));
        pn_null_feat_[0] = F(PN + "null_person_feature");
        pn_null_pos_ = F(PN + "null_xy_feature");
        // person_embeddings[p] + keypoint_embeddings[j] (keypoint_grounding_net.py:39-42)
        const RawTensor& pe = raw(PN + "person_embeddings");
        const RawTensor& ke = raw(PN + "keypoint_embeddings");
        const int P = (int)pe.shape[0], D = (int)pe.shape[1];
        std::vector<float> hp(pe.numel), hk(ke.numel), tab((size_t)P * 17 * D);
        HIPCK(hipMemcpy(hp.data(), pe.p, pe.numel * sizeof(float), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(hk.data(), ke.p, ke.numel * sizeof(float), hipMemcpyDeviceToHost));
        for (int p = 0; p < P; ++p)
            for (int j = 0; j < 17; ++j)
                for (int k = 0; k < D; ++k) tab[((size_t)p * 17 + j) * D + k] = hp[(size_t)p * D + k] + hk[(size_t)j * D + k];
        float* d = reinterpret_cast<float*>(persist(tab.size() * sizeof(float), false));
        HIPCK(hipMemcpy(d, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        kp_table_ = d;
    } else if (gkind_ == 3) {
        // spatial-map modalities (canny / hed / depth / normal / sem _grounding_net.py): tokens arrive through
        // gl_grounding.tokens; they are the output of gl_op_spatial_tokens (the ConvNeXt tokenizer below, built when its
        // weights were uploaded) or were computed elsewhere
        if (has(PN + "convnext_tiny_backbone.downsample_layers.0.0.weight")) build_convnext(PN);
    } else {
        throw GlError(GL_ERR_UNSUPPORTED, "grounding_kind must be 0 (text), 1 (text+image), 2 (keypoint) or 3 (precomputed tokens)");
    }
}

// PositionNet of the spatial-map modalities: ConvNeXt-tiny (depths 3,3,9,3; dims 96..768) + position embedding + 3-layer MLP
void Engine::build_convnext(const std::string& PN) {
    CnxNet& n = cnx_;
    const gl_unet_config& c = ucfg_;
    if (c.tok_resize <= 0 || c.tok_resize % 32) throw GlError(GL_ERR_ARG, "spatial tokenizer: resize_input must be a positive multiple of 32");
    n.resize = c.tok_resize;
    n.in_dim = c.tok_in_dim;
    n.tokens = (c.tok_resize / 32) * (c.tok_resize / 32);
    const std::string BB = PN + "convnext_tiny_backbone.";
    if (n.in_dim) {
        n.inconv_w = FK(PN + "in_conv.weight");
        n.inconv_b = F(PN + "in_conv.bias");
    }
    auto patch_conv = [&](const std::string& p, int k) {
        const RawTensor& w = raw(p + ".weight");
        if (w.shape.size() != 4 || w.shape[2] != k || w.shape[3] != k) throw GlError(GL_ERR_ARG, "'" + p + "' is not a " + std::to_string(k) + "x" + std::to_string(k) + " conv");
        LinW l;
        l.N = (int)w.shape[0];
        l.K = round_up(k * k * (int)w.shape[1], 64);
        bf16* dst = reinterpret_cast<bf16*>(persist((size_t)l.N * l.K * sizeof(bf16), false));
        CK(pack_patch_weight_launch(w.p, dst, l.N, (int)w.shape[1], k, l.K, 0));
        l.w = dst;
        l.b = F(p + ".bias");
        return l;
    };
    n.stem = patch_conv(BB + "downsample_layers.0.0", 4);
    n.stem_ln = norm(BB + "downsample_layers.0.1");
    n.dims[0] = n.stem.N;
    for (int i = 0; i < 3; ++i) {
        n.ds_ln[i] = norm(BB + fmt("downsample_layers.%d.0", i + 1));
        n.ds[i] = patch_conv(BB + fmt("downsample_layers.%d.1", i + 1), 2);
        n.dims[i + 1] = n.ds[i].N;
    }
    for (int st = 0; st < 4; ++st) {
        for (int j = 0;; ++j) {
            const std::string p = BB + fmt("stages.%d.%d", st, j);
            if (!has(p + ".dwconv.weight")) break;
            CnxBlock b;
            const int C = n.dims[st];
            if (raw(p + ".dwconv.weight").numel != (int64_t)C * 49) throw GlError(GL_ERR_ARG, "'" + p + ".dwconv' is not a depthwise 7x7 conv");
            b.dw_w = FK(p + ".dwconv.weight");
            b.dw_b = F(p + ".dwconv.bias");
            b.ln = norm(p + ".norm");
            b.pw1 = linear(p + ".pwconv1");
            if (has(p + ".gamma"))  // layer scale folded into pwconv2: gamma * (W h + b) = (gamma W) h + gamma b
                CK(scale_rows_launch(raw(p + ".pwconv2.weight").p, raw(p + ".pwconv2.bias").p, F(p + ".gamma"), C, 4 * C, 0));
            b.pw2 = linear(p + ".pwconv2");
            n.blocks[st].push_back(b);
        }
        if (n.blocks[st].empty()) throw GlError(GL_ERR_MISSING, fmt("ConvNeXt stage %d has no blocks", st));
    }
    n.pos = FK(PN + "pos_embedding");
    if (raw(PN + "pos_embedding").numel != (int64_t)n.tokens * n.dims[3])
        throw GlError(GL_ERR_ARG, "pos_embedding does not match (resize_input / 32)^2 tokens");
    n.null_feat = F(PN + "null_feature");
    for (int i = 0; i < 3; ++i) n.mlp[i] = linear(PN + fmt("linears.%d", 2 * i));
    if (n.mlp[2].N != c.gr_out_dim) throw GlError(GL_ERR_ARG, "spatial tokenizer out_dim does not match gr_out_dim");
    n.present = true;
}

void Engine::spatial_tokens(int B, const float* image, int Cimg, int H, int W, const float* mask, float* out, hipStream_t s) {
    if (!has_unet_ || !finalized_ || !cnx_.present) throw GlError(GL_ERR_STATE, "no spatial-map tokenizer (ConvNeXt weights) in this engine");
    const CnxNet& n = cnx_;
    const int Cuse = n.in_dim ? n.in_dim : 3;
    if (B <= 0 || Cimg < Cuse) throw GlError(GL_ERR_ARG, fmt("spatial_tokens: image has %d channels, the tokenizer reads %d", Cimg, Cuse));
    const size_t mk = arena_.mark();
    const int R = n.resize;
    // F.interpolate(x, resize_input): default mode 'nearest' (canny_grounding_net.py:42, sem: explicit nearest)
    float* img = arena_.get<float>((size_t)B * Cuse * R * R);
    CK(resize_f32_launch(image, img, B, Cimg, Cuse, H, W, R, 1, s));
    if (n.in_dim) {
        float* img3 = arena_.get<float>((size_t)B * 3 * R * R);
        CK(conv3x3_f32_launch(img, n.inconv_w, n.inconv_b, img3, B, n.in_dim, 3, R, R, s));
        img = img3;
    }
    auto ln_rows = [&](const bf16* x, int M, int C, int ld, const NormW& w) {
        bf16* y = arena_.get<bf16>((size_t)M * ld);
        LNParams P{};
        P.x = x; P.B = 1; P.N1 = M; P.N2 = 0; P.Tpad = M; P.C = C; P.eps = 1e-6f; P.gamma = w.g; P.beta = w.b; P.y = y; P.ldx = ld; P.ldy = ld;
        CK(layernorm_launch(P, s));
        return y;
    };
    auto mm = [&](const bf16* a, int M, int K, const LinW& L, int ldo, int act, const bf16* res) {
        if (L.K != K) throw GlError(GL_ERR_STATE, fmt("spatial_tokens: GEMM K %d against packed weight K %d", K, L.K));
        bf16* y = arena_.get<bf16>((size_t)M * ldo);
        AOperand A;
        aoperand_rows(A, a, K, K);
        Epilogue E;
        epilogue_defaults(E);
        E.out = y; E.ldo = ldo; E.bias = L.b; E.act = act; E.res = res; E.ldres = ldo;
        gemm(A, L.w, M, L.N, K, E, s);
        return y;
    };
    // stem: Conv2d(3, 96, 4, 4) as patchify + GEMM, LayerNorm over channels (convnext.py:71-74)
    int Hs = R / 4;
    int M = B * Hs * Hs;
    int C = n.dims[0], ld = round_up(C, 64);
    bf16* col = arena_.get<bf16>((size_t)M * n.stem.K);
    CK(patchify_f32_launch(img, col, B, 3, R, R, 4, n.stem.K, s));
    bf16* x = mm(col, M, n.stem.K, n.stem, ld, ACT_NONE, nullptr);
    x = ln_rows(x, M, C, ld, n.stem_ln);
    for (int st = 0; st < 4; ++st) {
        if (st > 0) {  // LayerNorm + Conv2d(C, C', 2, 2) (convnext.py:76-81)
            bf16* l = ln_rows(x, M, C, ld, n.ds_ln[st - 1]);
            const int Cn = n.dims[st];
            bf16* pc = arena_.get<bf16>((size_t)(M / 4) * 4 * C);
            CK(patchify_bf16_launch(l, pc, B, Hs, Hs, C, ld, 2, s));
            Hs /= 2; M /= 4;
            const int ldn = round_up(Cn, 64);
            if (n.ds[st - 1].K != 4 * C) throw GlError(GL_ERR_STATE, "ConvNeXt downsample weight shape");
            x = mm(pc, M, 4 * C, n.ds[st - 1], ldn, ACT_NONE, nullptr);
            C = Cn; ld = ldn;
        }
        for (const CnxBlock& b : n.blocks[st]) {  // Block.forward (convnext.py:36-50)
            bf16* y = arena_.get<bf16>((size_t)M * ld);   // dwconv output; dead after the LayerNorm, then reused for the block output
            const size_t after_y = arena_.mark();
            CK(dwconv7_launch(x, b.dw_w, b.dw_b, y, B, Hs, Hs, C, ld, s));
            bf16* l = ln_rows(y, M, C, ld, b.ln);
            bf16* h = mm(l, M, ld, b.pw1, 4 * C, ACT_GELU, nullptr);
            AOperand A;
            aoperand_rows(A, h, 4 * C, 4 * C);
            Epilogue E;
            epilogue_defaults(E);
            E.out = y; E.ldo = ld; E.bias = b.pw2.b; E.res = x; E.ldres = ld;   // x + gamma * pwconv2(...) (gamma folded at build)
            gemm(A, b.pw2.w, M, b.pw2.N, 4 * C, E, s);
            arena_.release(after_y);   // l, h
            x = y;
        }
    }
    if (Hs * Hs != n.tokens) throw GlError(GL_ERR_STATE, "ConvNeXt output grid does not match the token count");
    // objs = feat * mask + null * (1 - mask) + pos_embedding -> MLP (canny_grounding_net.py:48-59)
    bf16* mix = arena_.get<bf16>((size_t)M * C);
    CK(token_mix_launch(x, ld, mask, n.null_feat, n.pos, mix, B, n.tokens, C, s));
    bf16* h1 = linear_rows(mix, M, n.mlp[0], ACT_SILU, nullptr, nullptr, s);
    bf16* h2 = linear_rows(h1, M, n.mlp[1], ACT_SILU, nullptr, nullptr, s);
    bf16* objs = linear_rows(h2, M, n.mlp[2], ACT_NONE, nullptr, nullptr, s);
    CK(bf16_rows_to_f32_launch(objs, out, B, n.tokens, n.tokens, n.mlp[2].N, s));
    arena_.release(mk);
}

void Engine::build_vae() {
    const gl_vae_config& c = vcfg_;
    const std::string V = "vae/";
    pq_w_ = F(V + "post_quant_conv.weight");
    pq_b_ = F(V + "post_quant_conv.bias");
    const int block_in0 = c.ch * c.ch_mult[c.n_mult - 1];
    {
        const RawTensor& w = raw(V + "decoder.conv_in.weight");
        const int in_c = (int)w.shape[1];
        vae_in_kpad_ = round_up(9 * in_c, 64);
        bf16* dst = reinterpret_cast<bf16*>(persist((size_t)block_in0 * vae_in_kpad_ * sizeof(bf16), false));
        CK(pack_conv_small_launch(w.p, dst, block_in0, in_c, vae_in_kpad_, 0));
        vae_in_small_.w = dst;
        vae_in_small_.b = F(V + "decoder.conv_in.bias");
        vae_in_small_.Cin = in_c;
        vae_in_small_.Cout = block_in0;
    }
    vmid1_ = resw(V + "decoder.mid.block_1", block_in0, block_in0, false);
    vmid2_ = resw(V + "decoder.mid.block_2", block_in0, block_in0, false);
    vattn_.gn = norm(V + "decoder.mid.attn_1.norm");
    vattn_.q = conv1(V + "decoder.mid.attn_1.q");
    vattn_.k = conv1(V + "decoder.mid.attn_1.k");
    vattn_.v = conv1(V + "decoder.mid.attn_1.v");
    vattn_.proj = conv1(V + "decoder.mid.attn_1.proj_out");
    vup_.assign(c.n_mult, VaeUp{});
    int block_in = block_in0;
    for (int level = c.n_mult - 1; level >= 0; --level) {
        const int block_out = c.ch * c.ch_mult[level];
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            vup_[level].blocks.push_back(resw(V + fmt("decoder.up.%d.block.%d", level, i), block_in, block_out, false));
            block_in = block_out;
        }
        if (level != 0) {
            vup_[level].has_up = true;
            vup_[level].up = conv3(V + fmt("decoder.up.%d.upsample.conv", level));
        }
    }
    vnorm_out_ = norm(V + "decoder.norm_out");
    vconv_out_ = conv3(V + "decoder.conv_out", 32);
}

// Encoder of AutoencoderKL (reference model.py:368-459)
void Engine::build_vae_encoder() {
    const gl_vae_config& c = vcfg_;
    const std::string V = "vae/";
    {
        const RawTensor& w = raw(V + "encoder.conv_in.weight");
        const int in_c = (int)w.shape[1];
        venc_in_kpad_ = round_up(9 * in_c, 64);
        bf16* dst = reinterpret_cast<bf16*>(persist((size_t)c.ch * venc_in_kpad_ * sizeof(bf16), false));
        CK(pack_conv_small_launch(w.p, dst, c.ch, in_c, venc_in_kpad_, 0));
        venc_in_small_.w = dst;
        venc_in_small_.b = F(V + "encoder.conv_in.bias");
        venc_in_small_.Cin = in_c;
        venc_in_small_.Cout = c.ch;
    }
    vdown_.assign(c.n_mult, VaeDown{});
    int block_in = c.ch;
    for (int level = 0; level < c.n_mult; ++level) {
        const int block_out = c.ch * c.ch_mult[level];
        for (int i = 0; i < c.num_res_blocks; ++i) {
            vdown_[level].blocks.push_back(resw(V + fmt("encoder.down.%d.block.%d", level, i), block_in, block_out, false));
            block_in = block_out;
        }
        if (level != c.n_mult - 1) {
            vdown_[level].has_down = true;
            vdown_[level].down = conv3(V + fmt("encoder.down.%d.downsample.conv", level));
        }
    }
    vemid1_ = resw(V + "encoder.mid.block_1", block_in, block_in, false);
    vemid2_ = resw(V + "encoder.mid.block_2", block_in, block_in, false);
    veattn_.gn = norm(V + "encoder.mid.attn_1.norm");
    veattn_.q = conv1(V + "encoder.mid.attn_1.q");
    veattn_.k = conv1(V + "encoder.mid.attn_1.k");
    veattn_.v = conv1(V + "encoder.mid.attn_1.v");
    veattn_.proj = conv1(V + "encoder.mid.attn_1.proj_out");
    venorm_out_ = norm(V + "encoder.norm_out");
    veconv_out_ = conv3(V + "encoder.conv_out", 32);
    if (veconv_out_.Cout != 2 * c.z_channels) throw GlError(GL_ERR_ARG, "encoder.conv_out must produce 2 * z_channels moments");
    const RawTensor& q = raw(V + "quant_conv.weight");
    if (q.shape[0] != 2 * c.z_channels || q.shape[1] != 2 * c.z_channels)
        throw GlError(GL_ERR_UNSUPPORTED, "quant_conv must map 2*z_channels -> 2*z_channels (embed_dim == z_channels)");
    qc_w_ = F(V + "quant_conv.weight");
    qc_b_ = F(V + "quant_conv.bias");
    has_venc_ = true;
}

void Engine::finalize() {
    if (finalized_) throw GlError(GL_ERR_STATE, "gl_finalize called twice");
    HIPCK(hipSetDevice(device_));
    if (has_unet_) build_unet();
    if (has_vae_) build_vae();
    if (has_vae_ && has("vae/encoder.conv_in.weight")) build_vae_encoder();
    HIPCK(hipDeviceSynchronize());
    for (void* p : fold_tmps_) (void)hipFree(p);   // fp32 W * gamma / b + W beta temporaries of the folded LayerNorms
    fold_tmps_.clear();
    // matrices now live packed in bf16: drop their fp32 staging copies (vectors stay, they are used as is)
    for (auto it = raw_.begin(); it != raw_.end();) {
        if (it->second.shape.size() >= 2 && it->first.find("quant_conv") == std::string::npos && !keep_raw_.count(it->first)) {
            (void)hipFree(it->second.p);
            it = raw_.erase(it);
        } else {
            ++it;
        }
    }
    finalized_ = true;
}

// ---------------------------------------------------------------- execution helpers
// developer aid (GL_LAUNCH_LOG=file, tools/gpu_traffic.sh): one line per GEMM / conv / attention launch, in launch order
static FILE* launch_log_file() {
    static FILE* f = dev_env("GL_LAUNCH_LOG") ? fopen(dev_env("GL_LAUNCH_LOG"), "w") : nullptr;
    return f;
}
// attention launches in the same log: symbol | Nq | Nk | d | mode 2 | algorithmic bytes (q, k, v read once, o written once, unpadded)
static void log_attention(const char* sym, int B, int H, int Nq, int Nk, int d) {
    if (FILE* f = launch_log_file()) {
        fprintf(f, "%s|%d|%d|%d|2|%.0f\n", sym, Nq, Nk, d, 2.0 * B * H * d * (2.0 * Nq + 2.0 * Nk));
        fflush(f);
    }
}

void Engine::gemm(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, hipStream_t s) {
    ProfScope ps(this, s, "gemm", 2.0 * M * N * K, 0.0);
    CK(gemm_launch(A, W, M, N, K, E, ws_, ws_bytes_, s));
    if (profiling_) {
        static const bool by_shape = dev_env("GL_PROF_SHAPES") != nullptr;  // developer aid: one record per problem, not per symbol
        std::string nm = gemm_last_kernel_name();  // the symbol the tile selection actually launched
        if (by_shape) nm += fmt(" M%d N%d K%d", M, N, K);
        ps.rename(nm);
    }
    // developer aid (tools/gpu_traffic.sh): one line per GEMM / conv launch, in launch order, to join rocprofv3's per-dispatch
    // counter rows (which carry the kernel symbol but not the problem) with their shapes
    FILE* launch_log = launch_log_file();
    if (launch_log) {
        const double a_rows = A.mode == A_CONV3 ? (double)(M / (A.Ho * A.Wo)) * A.Hin * A.Win : (double)M;
        const double out_b = E.mode == EPI_NCHW_F32 ? 4.0 * M * E.n_real : (E.act == ACT_GEGLU ? 1.0 : 2.0) * M * (double)N * (E.out_f32 ? 2 : 1);
        const double bytes = a_rows * (A.C0 + A.C1) * 2 + (double)N * K * 2 + out_b + (E.res ? 2.0 * M * N : 0.0);
        fprintf(launch_log, "%s|%d|%d|%d|%d|%.0f\n", gemm_last_kernel_name(), M, N, K, A.mode, bytes);
        fflush(launch_log);
    }
    {
        int tm, tn, sp;
        gemm_last_cfg(&tm, &tn, &sp);
        n_launches += sp > 1 ? 2 : 1;   // a split-K problem is followed by splitk_reduce_kernel
    }
}

bf16* Engine::linear_rows(const bf16* x, int M, const LinW& L, int act, const bf16* res, const float* gate, hipStream_t s, RowStats* stats) {
    bf16* out = arena_.get<bf16>((size_t)M * L.N);
    AOperand A;
    aoperand_rows(A, x, L.K, L.K);
    Epilogue E;
    epilogue_defaults(E);
    E.out = out;
    E.ldo = L.N;
    E.bias = L.b;
    E.act = act;
    E.res = res;
    E.ldres = L.N;
    E.gate = gate;
    if (stats) {   // row statistics of the result, for the folded LayerNorm of the GEMM that reads it next
        *stats = RowStats{};
        if (ln_fold_ && L.N % 64 == 0) {
            stats->ld = L.N / 32;     // one slot per wave column block of the narrowest tile (64 x 64: 32 columns per wave)
            stats->p = arena_.get<float2>((size_t)M * stats->ld);
            E.stats_out = stats->p;
            E.stats_ld = stats->ld;
        }
    }
    gemm(A, L.w, M, L.N, L.K, E, s);
    if (stats && stats->p) stats->nb = gemm_last_stats_nb();
    return out;
}

bf16* Engine::groupnorm(const TRef& x, int B, int HW, const NormW& n, float eps, bool silu, hipStream_t s) {
    const int C = x.C();
    if (n.C != C) throw GlError(GL_ERR_ARG, fmt("groupnorm: %d channels given to a norm with %d", C, n.C));
    bf16* y = arena_.get<bf16>((size_t)B * HW * C);
    GNParams P{};
    P.x0 = x.p0; P.C0 = x.C0; P.x1 = x.p1; P.C1 = x.C1;
    P.B = B; P.HW = HW; P.eps = eps; P.gamma = n.g; P.beta = n.b; P.y = y; P.silu = silu ? 1 : 0;
    P.partial = reinterpret_cast<float*>(arena_.alloc(gn_partial_bytes(B, HW)));
    ProfScope ps(this, s, HW <= 256 ? "gn_small_kernel" : "gn_stats_kernel + gn_apply_kernel", 0.0, 2.0 * B * HW * (double)C * 2);
    CK(groupnorm_launch(P, s));
    n_launches += HW <= 256 ? 1 : 2;   // gn_small_kernel, or gn_stats_kernel + gn_apply_kernel
    return y;
}

// GroupNorm32 -> SiLU -> conv3x3 (reference openaimodel.py:212-232 in_layers / out_layers; VAE model.py:118-141). Where the conv runs
// on conv_halo_kernel the GroupNorm keeps only its statistics pass (groupnorm_coef_launch) and the conv normalises + activates
// its input while staging it (AOperand::gn): the normalised copy is never written to HBM. Elsewhere: the two passes of rounds 1-5.
bf16* Engine::gn_silu_conv3x3(const TRef& x, int B, int H, int W, const NormW& n, float eps, const ConvW& c, const float* bias2, int bias2_ld,
                              const bf16* res, bf16* out, hipStream_t s) {
    if (x.C() != c.Cin || n.C != c.Cin) throw GlError(GL_ERR_ARG, fmt("gn_silu_conv3x3: %d channels into a norm of %d and a conv of %d", x.C(), n.C, c.Cin));
    const int HW = H * W, M = B * HW;
    if (!out) out = arena_.get<bf16>((size_t)M * c.Cout);
    AOperand A{};
    A.p0 = x.p0; A.C0 = x.C0; A.ld0 = x.C0;
    A.p1 = x.p1; A.C1 = x.C1; A.ld1 = x.C1;
    A.mode = A_CONV3;
    A.Hin = H; A.Win = W; A.Ho = H; A.Wo = W; A.stride = 1; A.ups = 0; A.pad_lo = 1;
    Epilogue E;
    epilogue_defaults(E);
    E.out = out; E.ldo = c.Cout; E.bias = c.b;
    E.bias2 = bias2; E.bias2_ld = bias2_ld; E.rows_per_b = HW;
    E.res = res; E.ldres = c.Cout;
    if (gn_prologue_ && gemm_gn_prologue_supported(A, M, c.Cout, 9 * c.Cin, E)) {
        GNParams P{};
        P.x0 = x.p0; P.C0 = x.C0; P.x1 = x.p1; P.C1 = x.C1;
        P.B = B; P.HW = HW; P.eps = eps; P.gamma = n.g; P.beta = n.b; P.silu = 1;
        P.partial = reinterpret_cast<float*>(arena_.alloc(gn_partial_bytes(B, HW)));
        P.coef = reinterpret_cast<float*>(arena_.alloc(gn_coef_bytes(B, c.Cin)));
        {
            const int nl = groupnorm_coef_launches(HW, x.C0, x.C1);
            ProfScope ps(this, s, nl == 1 ? "gn_small_coef_kernel" : "gn_stats_kernel + gn_coef_kernel", 0.0, 1.0 * B * HW * (double)c.Cin * 2);
            CK(groupnorm_coef_launch(P, s));
            n_launches += nl;
        }
        A.gn = P.coef;
        gemm(A, c.w, M, c.Cout, 9 * c.Cin, E, s);
        ++n_prologue_convs;
        return out;
    }
    bf16* a = groupnorm(x, B, HW, n, eps, true, s);
    A.p0 = a; A.C0 = c.Cin; A.ld0 = c.Cin; A.p1 = nullptr; A.C1 = 0; A.ld1 = 0;
    gemm(A, c.w, M, c.Cout, 9 * c.Cin, E, s);
    return out;
}

bf16* Engine::layernorm(const bf16* x, int B, int N, int C, const NormW& n, bool pad64, hipStream_t s) {
    const int Tp = pad64 ? round_up(N, 64) : N;
    bf16* y = arena_.get<bf16>((size_t)B * Tp * C);
    LNParams P{};
    P.x = x; P.x2 = nullptr; P.B = B; P.N1 = N; P.N2 = 0; P.Tpad = Tp; P.C = C; P.eps = 1e-5f;
    P.gamma = n.g; P.beta = n.b; P.y = y;
    ProfScope ps(this, s, "ln_kernel", 0.0, 2.0 * B * N * (double)C * 2);
    CK(layernorm_launch(P, s));
    ++n_launches;
    return y;
}

bf16* Engine::layernorm_plain(const bf16* x, int B, int N, int C, bool pad64, hipStream_t s) {
    NormW none;
    none.C = C;
    return layernorm(x, B, N, C, none, pad64, s);
}

bf16* Engine::conv3x3(const TRef& x, int B, int Hin, int Win, const ConvW& c, int stride, int ups, int pad_lo,
                      const float* bias2, int bias2_ld, const bf16* res, hipStream_t s) {
    if (x.C() != c.Cin) throw GlError(GL_ERR_ARG, fmt("conv3x3: input has %d channels, weight expects %d", x.C(), c.Cin));
    const int Hup = Hin << ups, Wup = Win << ups;
    const int Ho = stride == 1 ? Hup : (pad_lo ? (Hup + 2 - 3) / 2 + 1 : (Hup + 1 - 3) / 2 + 1);
    const int Wo = stride == 1 ? Wup : (pad_lo ? (Wup + 2 - 3) / 2 + 1 : (Wup + 1 - 3) / 2 + 1);
    const int M = B * Ho * Wo;
    bf16* out = arena_.get<bf16>((size_t)M * c.Cout);
    AOperand A{};
    A.p0 = x.p0; A.C0 = x.C0; A.ld0 = x.C0;
    A.p1 = x.p1; A.C1 = x.C1; A.ld1 = x.C1;
    A.mode = A_CONV3;
    A.Hin = Hin; A.Win = Win; A.Ho = Ho; A.Wo = Wo; A.stride = stride; A.ups = ups; A.pad_lo = pad_lo;
    Epilogue E;
    epilogue_defaults(E);
    E.out = out; E.ldo = c.Cout; E.bias = c.b;
    E.bias2 = bias2; E.bias2_ld = bias2_ld; E.rows_per_b = Ho * Wo;
    E.res = res; E.ldres = c.Cout;
    gemm(A, c.w, M, c.Cout, 9 * c.Cin, E, s);
    return out;
}

// ResBlock._forward (openaimodel.py:212-232) / VAE ResnetBlock.forward (model.py:118-141)
bf16* Engine::resblock(const ResW& r, const TRef& x, int B, int H, int W, const float* embout, int emb_ld, float eps, hipStream_t s) {
    const int HW = H * W, M = B * HW;
    bf16* out = arena_.get<bf16>((size_t)M * r.Cout);
    const size_t mk = arena_.mark();
    bf16* h = gn_silu_conv3x3(x, B, H, W, r.n1, eps, r.c1, embout ? embout + r.emb_off : nullptr, emb_ld, nullptr, nullptr, s);
    const bf16* sk;
    if (r.has_skip) {
        bf16* skb = arena_.get<bf16>((size_t)M * r.Cout);
        AOperand A{};
        A.p0 = x.p0; A.C0 = x.C0; A.ld0 = x.C0; A.p1 = x.p1; A.C1 = x.C1; A.ld1 = x.C1; A.mode = A_ROWS;
        Epilogue E;
        epilogue_defaults(E);
        E.out = skb; E.ldo = r.Cout; E.bias = r.skip.b;
        gemm(A, r.skip.w, M, r.Cout, r.Cin, E, s);
        sk = skb;
    } else {
        if (x.p1) throw GlError(GL_ERR_STATE, "identity skip over a concatenated input");
        sk = x.p0;
    }
    gn_silu_conv3x3(TRef{h, r.Cout, nullptr, 0}, B, H, W, r.n2, eps, r.c2, nullptr, 0, sk, out, s);
    arena_.release(mk);
    return out;
}

AttnBufs& Engine::attn_bufs(int B, int H, int d, int Tq, int Tk, int dpv_layout, int slot) {
    const int Tq_pad = round_up(Tq, 128), Tk_pad = round_up(Tk, 64);
    int dp, dpv;
    CK(attn_dims(d, &dp, &dpv));
    if (dpv_layout) dpv = dpv_layout;      // (attn_vt_layout: the 16x16x32 P V kernel reads a 48-row V^T at d = 40)
    const AttnKey key{{B, H, d, Tq_pad, Tk_pad, dpv, slot}};    // every field in full (XOR-ed shifted fields let dpv = 64 vanish and 96 / 160 collide)
    auto it = attn_bufs_.find(key);
    if (it != attn_bufs_.end()) return it->second;
    AttnBufs b;
    b.Tq_pad = Tq_pad;
    b.Tk_pad = Tk_pad;
    b.q = reinterpret_cast<bf16*>(persist((size_t)B * H * Tq_pad * dp * sizeof(bf16), true));
    b.k = reinterpret_cast<bf16*>(persist((size_t)B * H * Tk_pad * dp * sizeof(bf16), true));
    b.vt = reinterpret_cast<bf16*>(persist((size_t)B * H * dpv * Tk_pad * sizeof(bf16), true));
    CK(attn_vt_ones_launch(b.vt, B * H, d, Tk_pad, 0, dpv));  // denominator row (attention.hip), once per buffer
    CK(attn_k_init_launch(b.k, B * H, d, Tk_pad, 0));     // d = 40: the stabiliser's multiplier column
    HIPCK(hipStreamSynchronize(0));
    return attn_bufs_.emplace(key, b).first->second;
}

// SelfAttention.forward (attention.py:167-186) on LayerNorm'ed rows ln [B][T][C] (T % 64 == 0,
// rows >= Nk are zero), queries = first Nq rows, keys/values = first Nk rows.
void Engine::fuser_kv_fill(const STW& t, int B, int HW, hipStream_t s) {
    const int Ng = cond_.Ng, C = t.C, d = t.d, H = C / d;
    const int Ng64 = round_up(Ng, 64), Tf = HW + Ng64;
    int dp, dpv;
    CK(attn_dims(d, &dp, &dpv));
    const int vt_layout = attn_vt_layout(d, HW + Ng, &dpv);
    AttnBufs& bufs = attn_bufs(B, H, d, Tf, Tf, dpv, t.idx + 1);
    const size_t mk = arena_.mark();
    // LayerNorm of the grounding-token rows alone (per row: what the [x ; objs] pass computed for them), no affine: gamma / beta live in
    // the folded q,k,v weights; rows Ng .. Ng64 - 1 of every sample are zero and land behind the last key
    bf16* lno = arena_.get<bf16>((size_t)B * Ng64 * C);
    {
        LNParams P{};
        P.x = cond_.objs[t.idx]; P.B = B; P.N1 = Ng; P.N2 = 0; P.Tpad = Ng64; P.C = C; P.eps = 1e-5f; P.y = lno;
        CK(layernorm_launch(P, s));
    }
    AOperand A;
    aoperand_rows(A, lno, C, C);
    Epilogue E;
    epilogue_defaults(E);
    E.mode = EPI_QKV_HEADS;
    E.q = bufs.q; E.k = bufs.k; E.vt = bufs.vt; E.C = C; E.H = H; E.d = d; E.DP = dp; E.DPV = dpv; E.T = Ng64; E.tok_off = HW; E.vt_perm32 = vt_layout;
    E.Tpad_q = bufs.Tq_pad; E.Tpad_k = bufs.Tk_pad;
    E.bias = t.fa.b;          // W beta of the folded LayerNorm
    CK(gemm_launch(A, t.fa.wqk, B * Ng64, 3 * C, C, E, ws_, ws_bytes_, s));   // (its q rows land behind the last query: never read)
    arena_.release(mk);
    if (fuser_kv_.size() < st_.size()) fuser_kv_.resize(st_.size());
    fuser_kv_[t.idx] = FuserKV{cond_epoch_, B, HW};
}

bool Engine::qkv_rows_ok(const SelfAttnW& a, int B, int T, int Nk, int C, int d) const {
    if (!qkv_rows_ || !a.rows_stream || !a.fused || !a.folded || !qkv_rows_supported(B * T, C, d, T)) return false;
    int dpv = 0;
    return attn_vt_layout(d, Nk, &dpv) == 1;                     // the kernel writes V^T in the 32-token form of attn3_kernel only
}

void Engine::qkv_rows_project(const SelfAttnW& a, const bf16* x, int B, int T, int Nk, int C, int d, const LinW& pre, const bf16* pre_res, bf16* mid,
                              RowStats* mid_stats, int Tbuf, int slot, hipStream_t s) {
    const int H = C / d, M = B * T;
    int dp, dpv;
    CK(attn_dims(d, &dp, &dpv));
    const int vt_layout = attn_vt_layout(d, Nk, &dpv);
    AttnBufs& bufs = attn_bufs(B, H, d, Tbuf ? Tbuf : T, Tbuf ? Tbuf : T, dpv, slot);
    QkvRowsParams P{};
    P.x = x; P.ldx = C; P.eps = 1e-5f; P.stream = a.rows_stream; P.M = M;
    P.pre = 1; P.pre_b = pre.b; P.pre_res = pre_res; P.ld_pre_res = C; P.mid_out = mid; P.ld_mid = C;
    if (mid_stats) {
        *mid_stats = RowStats{};
        if (ln_fold_) {
            mid_stats->ld = 1; mid_stats->nb = 1;
            mid_stats->p = arena_.get<float2>((size_t)M);
            P.stats_out = mid_stats->p;
        }
    }
    P.np = 3; P.bias = a.b; P.q = bufs.q; P.k = bufs.k; P.vt = bufs.vt;
    P.H = H; P.d = d; P.DP = dp; P.DPV = dpv; P.T = T; P.Tpad_q = bufs.Tq_pad; P.Tpad_k = bufs.Tk_pad; P.vt_perm32 = vt_layout;
    ProfScope ps(this, s, "qkv_rows_kernel<pre, 3>", 8.0 * M * (double)C * C, 0.0);
    CK(qkv_rows_launch(P, C, s));
    FILE* launch_log = launch_log_file();
    if (launch_log) {
        fprintf(launch_log, "qkv_rows_kernel<320, 40, true, 3>|%d|%d|%d|0|%.0f\n", M, 4 * C, C,
                (double)qkv_rows_stream_bytes(C, true, 3) + (pre_res ? 6.0 : 4.0) * M * C + 2.0 * M * H * (2.0 * dp + dpv));
        fflush(launch_log);
    }
    ++n_launches;
}

void Engine::self_attention(const SelfAttnW& a, const bf16* ln, int B, int T, int Nq, int Nk, int C, int d, bf16* o, hipStream_t s,
                            const RowStats* in_stats, int Tbuf, int slot, bool projected) {
    const int H = C / d;
    int dp, dpv;
    CK(attn_dims(d, &dp, &dpv));
    const int vt_layout = attn_vt_layout(d, Nk, &dpv);    // which V^T form the attention kernel for this (d, Nk) reads
    AttnBufs& bufs = attn_bufs(B, H, d, Tbuf ? Tbuf : T, Tbuf ? Tbuf : T, dpv, slot);
    if (in_stats && !(a.fused && a.folded)) throw GlError(GL_ERR_STATE, "self_attention: row statistics given to an unfolded projection");
    if (!projected) qkv_project_gemm(a, ln, B, T, Nk, C, d, s, in_stats, Tbuf, slot);
    AttnParams P{};
    P.q = bufs.q; P.k = bufs.k; P.vt = bufs.vt; P.o = o;
    P.H = H; P.d = d; P.Nq = Nq; P.Nk = Nk; P.Tq_pad = bufs.Tq_pad; P.Tk_pad = bufs.Tk_pad;
    P.ldo = C; P.o_rows_per_b = Nq; P.vt_layout = vt_layout;
    P.scale_log2e = (float)(1.4426950408889634 / std::sqrt((double)d));
    {
        ProfScope ps(this, s, attn_kernel_name(d, Nk, vt_layout), 4.0 * B * H * (double)Nq * Nk * d, 0.0);
        CK(attn_launch(P, B, s));
        log_attention(attn_kernel_name(d, Nk, vt_layout), B, H, Nq, Nk, d);
    }
    ++n_launches;
}

// q, k, v^T of a self-attention by GEMM: the fused EPI_QKV_HEADS launch (or the q,k GEMM + operand-swapped v^T GEMM)
void Engine::qkv_project_gemm(const SelfAttnW& a, const bf16* ln, int B, int T, int Nk, int C, int d, hipStream_t s, const RowStats* in_stats, int Tbuf, int slot) {
    const int H = C / d;
    int dp, dpv;
    CK(attn_dims(d, &dp, &dpv));
    const int vt_layout = attn_vt_layout(d, Nk, &dpv);
    AttnBufs& bufs = attn_bufs(B, H, d, Tbuf ? Tbuf : T, Tbuf ? Tbuf : T, dpv, slot);
    if (a.fused) {
        AOperand A;
        aoperand_rows(A, ln, C, C);
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_QKV_HEADS;
        E.q = bufs.q; E.k = bufs.k; E.vt = bufs.vt; E.C = C; E.H = H; E.d = d; E.DP = dp; E.DPV = dpv; E.T = T; E.vt_perm32 = vt_layout;
        E.Tpad_q = bufs.Tq_pad; E.Tpad_k = bufs.Tk_pad;
        if (a.folded) E.bias = a.b;      // W beta of the folded LayerNorm (to_q / to_k / to_v have no bias of their own)
        if (in_stats) {                  // `ln` holds the raw rows: (x - mean) * rstd happens in the epilogue
            E.ln_stats = in_stats->p; E.ln_nb = in_stats->nb; E.ln_ld = in_stats->ld; E.ln_csum = a.csum;
            E.ln_inv_c = 1.f / (float)C; E.ln_eps = 1e-5f;
        }
        gemm(A, a.wqk, B * T, 3 * C, C, E, s);
    } else {
        AOperand A;
        aoperand_rows(A, ln, C, C);
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_QK_HEADS;
        E.q = bufs.q; E.k = bufs.k; E.C = C; E.H = H; E.d = d; E.DP = dp; E.T = T;
        E.Tpad_q = bufs.Tq_pad; E.Tpad_k = bufs.Tk_pad;
        gemm(A, a.wqk, B * T, 2 * C, C, E, s);
    }
    if (!a.fused) {
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_VT_HEADS;
        E.out = bufs.vt; E.H = H; E.d = d; E.DPV = dpv; E.T = T; E.Tpad_k = bufs.Tk_pad; E.vt_perm32 = vt_layout;
        ProfScope ps(this, s, "gemm", 2.0 * C * (double)B * T * C, 0.0);
        CK(gemm_launch_t(a.wv, C, ln, B * T, C, E, s));
        if (profiling_) ps.rename(gemm_last_kernel_name());
        ++n_launches;
    }
}

bf16* Engine::feedforward_chain(const FFW& f, const bf16* x, int M, const LinW& pre, const bf16* pre_res, const float* pre_gate, const float* gate,
                                const LinW* post, const bf16* post_res, bf16* out, hipStream_t s, RowStats* out_stats, const ChainQ* cq) {
    const int C = f.C;
    if (!f.chain_stream || !ff_rows_supported(M, C) || (post != nullptr) != f.chain_post || (cq && (post || !f.chain_q_stream)))
        throw GlError(GL_ERR_STATE, "feedforward_chain: no chained stream of this shape");
    if (!out) out = arena_.get<bf16>((size_t)M * C);
    FFRowsParams P{};
    P.x = x; P.ldx = C; P.normalize = 1; P.eps = 1e-5f; P.stream = f.chain_stream; P.b2 = f.w2.b; P.gate = gate; P.out = out; P.ldo = C; P.M = M;
    P.pre = 1; P.pre_b = pre.b; P.pre_res = pre_res; P.ld_pre_res = C; P.pre_gate = pre_gate;
    P.mid_out = arena_.get<bf16>((size_t)M * C);     // (only written when the gate is too small for the residual to ride in the accumulator)
    P.ld_mid = C;
    if (post) { P.post = 1; P.post_b = post->b; P.post_res = post_res; P.ld_post_res = C; }
    if (cq) { P.post = 2; P.stream = f.chain_q_stream; P.post_b = f.chain_q_bias; P.q = cq->q; P.qDP = cq->DP; P.qT = cq->T; P.qTpad = cq->Tpad; }
    if (out_stats) {
        *out_stats = RowStats{};
        if (ln_fold_) {
            out_stats->ld = 1; out_stats->nb = 1;
            out_stats->p = arena_.get<float2>((size_t)M);
            P.stats_out = out_stats->p; P.stats_ld = 1;
        }
    }
    ProfScope ps(this, s, post ? "ff_rows_kernel<pre, post>" : cq ? "ff_rows_kernel<pre, to_q>" : "ff_rows_kernel<pre>", (24.0 + ((post || cq) ? 4.0 : 2.0)) * M * (double)C * C, 0.0);
    CK(ff_rows_launch(P, C, s));
    FILE* launch_log = launch_log_file();
    if (launch_log) {
        fprintf(launch_log, "ff_rows_kernel<320, 0, true, %d>|%d|%d|%d|0|%.0f\n", post ? 1 : cq ? 2 : 0, M, C, 4 * C,
                (double)ff_chain_stream_bytes(C, true, post != nullptr || cq != nullptr) + (post ? 8.0 : 6.0) * M * C + (cq ? 2.0 * M * (C / 40) * cq->DP : 0.0));   // (the symbol as rocprofv3 prints it: pmc_summarize.py joins on it)
        fflush(launch_log);
    }
    ++n_launches;
    return out;
}

// ---------------------------------------------------------------- row-local kernel or two GEMMs: decided by timing
// The row-local kernel (ffn.hip) wins where a launch fills the chip (one 128-row workgroup per CU: M = 32768) on a box whose
// fabric keeps up with its exposed row loads / stores, and loses elsewhere (M = 8192: 68 against 42 us for LayerNorm + two GEMMs;
// 1.65x slower on a slow-fabric box). So the engine does not dispatch on shape: at the first EAGER launch of a
// (feed-forward, chain form, M) it runs both forms on scratch rows with the block's real weights, times them with HIP events on the
// launch stream and keeps the faster for the life of the process (all contexts of a device share the table, so forks agree).
namespace ff_policy {
static std::mutex mu;
struct Entry { int rows; float us_rows, us_gemm; };
static std::map<uint64_t, Entry> table;
static std::atomic<int> mode{-1};          // -1 timed (default), 0 never the row-local kernel, 1 wherever it exists
static std::atomic<unsigned> epoch{1};     // bumped when the mode changes: captured graphs of another epoch are dropped
static uint64_t key(int dev, int which, int form, int C, int M) {
    return ((uint64_t)dev << 56) | ((uint64_t)which << 52) | ((uint64_t)form << 48) | ((uint64_t)C << 32) | (uint64_t)(unsigned)M;
}
// the rule for shapes that cannot be timed (first seen inside a stream capture): a launch of at least 7/8 of the chip's CUs
static bool static_rule(int M) { return M / 128 >= 224; }
}  // namespace ff_policy

int ff_rows_policy_set(int mode) {
    if (mode < -1 || mode > 2) return set_error(GL_ERR_ARG, "ff_rows policy %d (expected -1 timed, 0 off, 1 wherever supported, 2 static rule)", mode);
    if (ff_policy::mode.exchange(mode) != mode) ff_policy::epoch.fetch_add(1);
    return GL_OK;
}
int ff_rows_policy_report(char* buf, size_t cap) {
    if (!buf || !cap) return set_error(GL_ERR_ARG, "ff_rows_policy_report: no buffer");
    std::lock_guard<std::mutex> lk(ff_policy::mu);
    size_t n = (size_t)snprintf(buf, cap, "mode=%d", ff_policy::mode.load());
    for (const auto& kv : ff_policy::table) {
        if (n + 96 >= cap) break;
        const uint64_t k = kv.first;
        n += (size_t)snprintf(buf + n, cap - n, ";dev%d %s form%d C%d M%d -> %s (rows %.1f us, gemm %.1f us)", (int)(k >> 56),
                              ((k >> 52) & 15) == 1 ? "fuser.ff" : ((k >> 52) & 15) == 2 ? "ff" : ((k >> 52) & 15) == 3 ? "proj_in+attn1.qkv" : "attn1.to_out+fuser.qkv", (int)((k >> 48) & 15), (int)((k >> 32) & 0xffff), (int)(k & 0xffffffffu),
                              kv.second.rows ? "rows" : "gemm", kv.second.us_rows, kv.second.us_gemm);
    }
    return GL_OK;
}

bool Engine::can_fold(const RowStats& st, int M, int C, int Nc, int mode, int act, bool aligned) {
    if (!st.nb || !aligned) return false;
    AOperand A;
    aoperand_rows(A, nullptr, C, C);
    Epilogue E;
    epilogue_defaults(E);
    E.mode = mode; E.act = act; E.geglu16 = gemm_geglu_layout();
    return gemm_ln_fold_supported(A, M, Nc, C, E);
}

// LayerNorm + feed-forward (+ gated residual) behind a projection that produced rows_in (statistics st_in, if it wrote any)
bf16* Engine::ff_behind(const FFW& f, const NormW& nw, const bf16* rows_in, RowStats& st_in, int B, int HW, const float* gate, bool rows, hipStream_t s,
                        RowStats* out_stats) {
    const int M = B * HW, C = f.C;
    const bool fold = rows || (f.folded && can_fold(st_in, M, C, 8 * C, EPI_ROWMAJOR, ACT_GEGLU, round_up(HW, 64) == HW));
    const bf16* ln = fold ? rows_in : (f.folded ? layernorm_plain(rows_in, B, HW, C, false, s) : layernorm(rows_in, B, HW, C, nw, false, s));
    return feedforward(f, ln, M, rows_in, gate, s, (fold && !rows) ? &st_in : nullptr, out_stats, rows, rows);
}

bf16* Engine::fuser_ff_tail(const STW& t, const bf16* o, const bf16* t1, int B, int HW, bool rows, hipStream_t s, RowStats* st3, bool* q_done) {
    const int M = B * HW;
    const float* g_attn = gates_ + 2 * t.idx;
    if (q_done) *q_done = false;
    if (rows && t.fff.chain_stream && !t.fff.chain_post) {   // one row-local launch for the three
        if (q_done && t.fff.chain_q_stream && t.a2.folded && HW % 128 == 0) {
            // ... and attn2.to_q(norm2(.)) behind them: the cross-attention's q buffer is filled by the same launch
            int dp, dpv;
            CK(attn_dims(t.d, &dp, &dpv));
            AttnBufs& bufs = attn_bufs(B, t.C / t.d, t.d, HW, cond_.ctx_Tpad);
            const ChainQ cq{bufs.q, dp, HW, bufs.Tq_pad};
            *q_done = true;
            return feedforward_chain(t.fff, o, M, t.fa.out, t1, g_attn, g_attn + 1, nullptr, nullptr, nullptr, s, st3, &cq);
        }
        return feedforward_chain(t.fff, o, M, t.fa.out, t1, g_attn, g_attn + 1, nullptr, nullptr, nullptr, s, st3);
    }
    RowStats st2;
    bf16* t2 = linear_rows(o, M, t.fa.out, ACT_NONE, t1, g_attn, s, rows ? nullptr : &st2);
    return ff_behind(t.fff, t.fn2, t2, st2, B, HW, g_attn + 1, rows, s, st3);
}

void Engine::block_ff_tail(const STW& t, const bf16* o, const bf16* t3, const bf16* x, bf16* out, int B, int HW, bool rows, hipStream_t s) {
    const int M = B * HW, C = t.C;
    if (rows && t.ff.chain_stream && t.ff.chain_post) {   // attn2.to_out + residual, LayerNorm, ff + residual, proj_out + x_in: one row-local launch
        feedforward_chain(t.ff, o, M, t.a2.out, t3, nullptr, nullptr, &t.proj_out, x, out, s, nullptr);
        return;
    }
    RowStats st4;
    bf16* t4 = linear_rows(o, M, t.a2.out, ACT_NONE, t3, nullptr, s, rows ? nullptr : &st4);
    bf16* t5 = ff_behind(t.ff, t.ln3, t4, st4, B, HW, nullptr, rows, s, nullptr);
    AOperand A;
    aoperand_rows(A, t5, C, C);
    Epilogue E;
    epilogue_defaults(E);
    E.out = out; E.ldo = C; E.bias = t.proj_out.b; E.res = x; E.ldres = C;
    gemm(A, t.proj_out.w, M, C, C, E, s);
}

bool Engine::ff_rows_for(const STW& t, int which, int B, int HW, hipStream_t s) {
    if (which == 1 && ucfg_.fuser_kind != 0) which = 2;   // (gatedSA2 / gatedCA produce the fuser's rows differently: they follow the block's ff)
    const FFW& f = which == 1 ? t.fff : t.ff;
    const int M = B * HW, C = t.C;
    if (!f.folded || !f.rows_stream || !ff_rows_supported(M, C)) return false;
    static const bool env_once = [] {      // developer switch: GL_FF_POLICY=-1|0|1 presets the mode (PMC passes: no timing launches)
        if (const char* e = dev_env("GL_FF_POLICY")) ff_policy::mode.store(atoi(e));
        return true;
    }();
    (void)env_once;
    const int mode = ff_policy::mode.load();
    if (mode == 0) return false;
    if (mode == 1) return true;
    if (mode == 2) return ff_policy::static_rule(M);     // deterministic: the same kernel form -- hence the same output bits -- on every box, rank and run
    const int form = f.chain_stream ? (f.chain_post ? 2 : 1) : 0;
    const uint64_t key = ff_policy::key(device_, which, form, C, M);
    std::unique_lock<std::mutex> lk(ff_policy::mu);   // held across the timing pass: two contexts timing at once would time each other
    auto it = ff_policy::table.find(key);
    if (it != ff_policy::table.end()) return it->second.rows != 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone) {
        // a shape first met inside a capture cannot be timed: the static rule decides, and the decision is KEPT -- a later eager pass
        // must take the form the captured graph replays (the two forms differ in their last bits)
        const bool rows = ff_policy::static_rule(M);
        ff_policy::table[key] = ff_policy::Entry{rows ? 1 : 0, 0.f, 0.f};
        return rows;
    }

    // ---- time both forms of this tail on scratch rows (zeros: no kernel on the path is data dependent), real weights. The device is
    // drained first: another lane's batch in flight on its own stream (gligen_inference.generate_stream starts lanes back to back) would
    // otherwise share the chip with the timed launches and decide the table
    HIPCK(hipDeviceSynchronize());
    // whatever the timed launches throw, the context is left as it was found: arena mark, launch counter, profiling switch
    struct Restore {
        Engine* e; size_t mk; int64_t launches; bool prof;
        ~Restore() { e->arena_.release(mk); e->n_launches = launches; e->profiling_ = prof; }
    } restore{this, arena_.mark(), n_launches, profiling_};
    profiling_ = false;
    bf16* zo = arena_.get<bf16>((size_t)M * C);
    bf16* zr = arena_.get<bf16>((size_t)M * C);
    bf16* zx = arena_.get<bf16>((size_t)M * C);
    bf16* zout = arena_.get<bf16>((size_t)M * C);
    HIPCK(hipMemsetAsync(zo, 0, (size_t)M * C * sizeof(bf16), s));
    HIPCK(hipMemsetAsync(zr, 0, (size_t)M * C * sizeof(bf16), s));
    HIPCK(hipMemsetAsync(zx, 0, (size_t)M * C * sizeof(bf16), s));
    auto run = [&](bool rows) {
        const size_t m2 = arena_.mark();
        RowStats st;
        if (which == 1) {
            bool qd = false;
            bf16* t3z = fuser_ff_tail(t, zo, zr, B, HW, rows, s, &st, &qd);
            if (!qd && ucfg_.fuser_kind == 0 && t.fff.chain_q_stream) cross_q_gemm(t, t3z, st, B, HW, s);   // (the form that folds to_q in is timed against the form that launches it)
        } else block_ff_tail(t, zo, zr, zx, zout, B, HW, rows, s);
        arena_.release(m2);
    };
    hipEvent_t ev[2];
    HIPCK(hipEventCreate(&ev[0]));
    HIPCK(hipEventCreate(&ev[1]));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); } } guard{ev};
    float best[2] = {1e30f, 1e30f};
    constexpr int REPS = 4, ROUNDS = 3;
    for (int form_i = 0; form_i < 2; ++form_i) run(form_i == 1);    // warm-up: GEMM tile tuning, kernel attributes
    for (int r = 0; r < ROUNDS; ++r)
        for (int form_i = 0; form_i < 2; ++form_i) {                // alternate the two forms: a clock ramp hits both
            HIPCK(hipEventRecord(ev[0], s));
            for (int i = 0; i < REPS; ++i) run(form_i == 1);
            HIPCK(hipEventRecord(ev[1], s));
            HIPCK(hipEventSynchronize(ev[1]));
            float ms = 0.f;
            HIPCK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            best[form_i] = std::min(best[form_i], ms * 1e3f / REPS);
        }
    const ff_policy::Entry e{best[1] < best[0] ? 1 : 0, best[1], best[0]};
    ff_policy::table[key] = e;
    static const bool log = dev_env("GL_FF_POLICY_LOG") != nullptr;
    if (log) fprintf(stderr, "[ff policy] %s form %d C %d M %d: rows %.1f us, gemm %.1f us -> %s\n", which == 1 ? "fuser.ff" : "ff", form, C, M, e.us_rows, e.us_gemm, e.rows ? "rows" : "gemm");
    return e.rows != 0;
}

// Row-local projection launch or GEMM + LayerNorm-folded q,k,v^T GEMM for attention `which` (0 = attn1 behind proj_in, 1 = fuser.attn
// behind attn1.to_out) of block t: the same policy as the row-local feed-forward (ff_policy: timed at the first eager launch per
// device and shape, gl_set_ff_rows_policy forces it) -- on a box whose fabric is slow the kernel's exposed row loads / stores lose
// what its fused launch wins (profiles/r6: 54 against 72 us on one box, 57 against 58 in the graph on another).
bool Engine::qkv_rows_for(const STW& t, int which, int B, int HW, int Nk, hipStream_t s) {
    const SelfAttnW& a = which ? t.fa : t.a1;
    const int M = B * HW, C = t.C, d = t.d;
    if (!qkv_rows_ok(a, B, HW, Nk, C, d)) return false;
    if (qkv_rows_ >= 2) return true;
    const int mode = ff_policy::mode.load();
    if (mode == 0) return false;
    if (mode == 1) return true;
    if (mode == 2) return ff_policy::static_rule(M);
    const uint64_t key = ff_policy::key(device_, 3 + which, 0, C, M);
    std::unique_lock<std::mutex> lk(ff_policy::mu);
    auto it = ff_policy::table.find(key);
    if (it != ff_policy::table.end()) return it->second.rows != 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone) {
        const bool rows = ff_policy::static_rule(M);
        ff_policy::table[key] = ff_policy::Entry{rows ? 1 : 0, 0.f, 0.f};
        return rows;
    }
    HIPCK(hipDeviceSynchronize());
    struct Restore {
        Engine* e; size_t mk; int64_t launches; bool prof;
        ~Restore() { e->arena_.release(mk); e->n_launches = launches; e->profiling_ = prof; }
    } restore{this, arena_.mark(), n_launches, profiling_};
    profiling_ = false;
    bf16* zx = arena_.get<bf16>((size_t)M * C);
    bf16* zr = arena_.get<bf16>((size_t)M * C);
    bf16* zm = arena_.get<bf16>((size_t)M * C);
    bf16* zm2 = arena_.get<bf16>((size_t)M * C);
    HIPCK(hipMemsetAsync(zx, 0, (size_t)M * C * sizeof(bf16), s));
    HIPCK(hipMemsetAsync(zr, 0, (size_t)M * C * sizeof(bf16), s));
    const LinW& pre = which ? t.a1.out : t.proj_in;
    const int Tf = which ? round_up(Nk, 64) : 0, slot = which ? t.idx + 1 : 0;
    auto run = [&](bool rows) {
        const size_t m2 = arena_.mark();
        if (rows) {
            qkv_rows_project(a, zx, B, HW, Nk, C, d, pre, which ? zr : nullptr, zm, nullptr, Tf, slot, s);
        } else {
            RowStats st;
            bf16* mid = linear_rows(zx, M, pre, ACT_NONE, which ? zr : nullptr, nullptr, s, &st);
            const bool f = can_fold(st, M, C, 3 * C, EPI_QKV_HEADS, ACT_NONE, true);
            const bf16* ln = f ? mid : layernorm_plain(mid, B, HW, C, !which, s);
            qkv_project_gemm(a, ln, B, HW, Nk, C, d, s, f ? &st : nullptr, Tf, slot);
        }
        // ... and the attention that reads what was projected: the row-local launch leaves q / k / v^T (and the clocks) in another
        // state than the GEMM does, and the attention behind it was measured 4 us slower in situ -- time the pair, not the projection
        self_attention(a, nullptr, B, HW, HW, Nk, C, d, zm2, s, nullptr, Tf, slot, true);
        arena_.release(m2);
    };
    hipEvent_t ev[2];
    HIPCK(hipEventCreate(&ev[0]));
    HIPCK(hipEventCreate(&ev[1]));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); } } guard{ev};
    float best[2] = {1e30f, 1e30f};
    constexpr int REPS = 4, ROUNDS = 3;
    for (int form_i = 0; form_i < 2; ++form_i) run(form_i == 1);
    for (int r = 0; r < ROUNDS; ++r)
        for (int form_i = 0; form_i < 2; ++form_i) {
            HIPCK(hipEventRecord(ev[0], s));
            for (int i = 0; i < REPS; ++i) run(form_i == 1);
            HIPCK(hipEventRecord(ev[1], s));
            HIPCK(hipEventSynchronize(ev[1]));
            float ms = 0.f;
            HIPCK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            best[form_i] = std::min(best[form_i], ms * 1e3f / REPS);
        }
    const ff_policy::Entry e{best[1] < best[0] ? 1 : 0, best[1], best[0]};
    ff_policy::table[key] = e;
    static const bool log = dev_env("GL_FF_POLICY_LOG") != nullptr;
    if (log) fprintf(stderr, "[ff policy] %s C %d M %d: rows %.1f us, gemm %.1f us -> %s\n", which ? "attn1.to_out+fuser.qkv" : "proj_in+attn1.qkv", C, M, e.us_rows, e.us_gemm, e.rows ? "rows" : "gemm");
    return e.rows != 0;
}

bf16* Engine::feedforward(const FFW& f, const bf16* ln, int M, const bf16* res, const float* gate, hipStream_t s, const RowStats* in_stats,
                          RowStats* out_stats, bool raw_rows, bool use_rows) {
    const int C = f.C;
    if (use_rows) {
        if (!f.rows_stream || !ff_rows_supported(M, C)) throw GlError(GL_ERR_STATE, "feedforward: no row-local stream of this shape");
        // one launch: LayerNorm (where folded and the rows are raw) + GEGLU projection + FF-out + (gated) residual + row statistics
        bf16* out = arena_.get<bf16>((size_t)M * C);
        FFRowsParams P{};
        P.x = ln; P.ldx = C; P.normalize = (raw_rows || in_stats) ? 1 : 0; P.eps = 1e-5f;
        if (P.normalize && !f.folded) throw GlError(GL_ERR_STATE, "feedforward: raw rows given to an unfolded projection");
        P.stream = f.rows_stream; P.b2 = f.w2.b; P.res = res; P.ldres = C; P.gate = gate; P.out = out; P.ldo = C; P.M = M;
        if (out_stats) {
            *out_stats = RowStats{};
            if (ln_fold_) {
                out_stats->ld = 1; out_stats->nb = 1;
                out_stats->p = arena_.get<float2>((size_t)M);
                P.stats_out = out_stats->p; P.stats_ld = 1;
            }
        }
        ProfScope ps(this, s, "ff_rows_kernel", 24.0 * M * (double)C * C, 0.0);
        CK(ff_rows_launch(P, C, s));
        FILE* launch_log = launch_log_file();
        if (launch_log) {
            fprintf(launch_log, "ff_rows_kernel<320, 0, false, false>|%d|%d|%d|0|%.0f\n", M, C, 4 * C, (double)ff_stream_bytes(C) + (res ? 6.0 : 4.0) * M * C);
            fflush(launch_log);
        }
        ++n_launches;
        return out;
    }
    if (raw_rows && !in_stats) throw GlError(GL_ERR_STATE, "feedforward: raw rows without statistics outside the row-local kernel");
    bf16* hbuf = arena_.get<bf16>((size_t)M * 4 * C);
    AOperand A;
    aoperand_rows(A, ln, C, C);
    Epilogue E;
    epilogue_defaults(E);
    E.act = ACT_GEGLU; E.geglu16 = f.geglu16; E.out = hbuf; E.ldo = 4 * C; E.bias = f.b1;
    if (in_stats) {
        if (!f.folded) throw GlError(GL_ERR_STATE, "feedforward: row statistics given to an unfolded projection");
        E.ln_stats = in_stats->p; E.ln_nb = in_stats->nb; E.ln_ld = in_stats->ld; E.ln_csum = f.csum1;
        E.ln_inv_c = 1.f / (float)C; E.ln_eps = 1e-5f;
    }
    gemm(A, f.w1, M, 8 * C, C, E, s);
    return linear_rows(hbuf, M, f.w2, ACT_NONE, res, gate, s, out_stats);
}

// attn2.to_q(norm2(rows)) (attention.py:336, 136): LayerNorm folded into the GEMM where the rows came with statistics, else ln_kernel
void Engine::cross_q_gemm(const STW& t, const bf16* rows, const RowStats& st, int B, int HW, hipStream_t s) {
    const int C = t.C, d = t.d, heads = C / d, M = B * HW;
    const int Tp = round_up(HW, 64);
    const bool aligned = Tp == HW;
    const bool f3 = t.a2.folded && can_fold(st, M, C, C, EPI_QK_HEADS, ACT_NONE, aligned);
    const bf16* ln = f3 ? rows : (t.a2.folded ? layernorm_plain(rows, B, HW, C, true, s) : layernorm(rows, B, HW, C, t.ln2, true, s));
    int dp, dpv;
    CK(attn_dims(d, &dp, &dpv));
    AttnBufs& bufs = attn_bufs(B, heads, d, Tp, cond_.ctx_Tpad);
    AOperand A;
    aoperand_rows(A, ln, C, C);
    Epilogue E;
    epilogue_defaults(E);
    E.mode = EPI_QK_HEADS;
    E.q = bufs.q; E.k = nullptr; E.C = C; E.H = heads; E.d = d; E.DP = dp; E.T = Tp; E.Tpad_q = bufs.Tq_pad; E.Tpad_k = 0;
    E.bias = t.a2.q.b;               // null unless the LayerNorm is folded (to_q has no bias of its own)
    if (f3) {
        E.ln_stats = st.p; E.ln_nb = st.nb; E.ln_ld = st.ld; E.ln_csum = t.a2.q_csum;
        E.ln_inv_c = 1.f / (float)C; E.ln_eps = 1e-5f;
    }
    gemm(A, t.a2.q.w, B * Tp, C, C, E, s);
}

// SpatialTransformer.forward + BasicTransformerBlock._forward + GatedSelfAttentionDense.forward
// (attention.py:366-376, 333-338, 236-244)
bf16* Engine::transformer(const STW& t, const bf16* x, int B, int H, int W, hipStream_t s) {
    const int HW = H * W, M = B * HW, C = t.C, d = t.d, heads = C / d;
    bf16* out = arena_.get<bf16>((size_t)M * C);
    const size_t mk = arena_.mark();
    if (cond_.Beff != B) throw GlError(GL_ERR_STATE, fmt("unet_forward batch %d != batch %d of the conditioning set by gl_unet_set_cond", B, cond_.Beff));

    bf16* n = groupnorm(TRef{x, C, nullptr, 0}, B, HW, t.gn, 1e-6f, false, s);
    // Folded LayerNorms (gemm.h Epilogue::ln_stats): every GEMM that writes the residual stream also writes its rows' partial
    // (sum, sum of squares); the projection behind the next LayerNorm then reads the RAW rows and normalises in its epilogue.
    // Where no statistics exist (split-K producer, the [x ; objs] concatenation, an epilogue without the fold) the rows go
    // through ln_kernel without affine -- gamma / beta live in the folded weights either way.
    const int Tp = round_up(HW, 64);
    const bool aligned = Tp == HW;
    auto normed = [&](const bf16* rows, const NormW& nw, bool folded, bool fuse, bool pad64) -> const bf16* {
        if (fuse) return rows;
        return folded ? layernorm_plain(rows, B, HW, C, pad64, s) : layernorm(rows, B, HW, C, nw, pad64, s);
    };
    RowStats st0, st1, st2, st3;
    // the row-local feed-forward kernel normalises its raw input rows itself: their producers need not write statistics
    const bool r2 = !fuser_off_ && ff_rows_for(t, 1, B, HW, s), r4 = ff_rows_for(t, 2, B, HW, s);
    // proj_in -> norm1 -> attn1's q,k,v^T: one row-local launch where that kernel exists and fills the chip (ffn.h qkv_rows_kernel)
    const bool rq1 = aligned && qkv_rows_for(t, 0, B, HW, HW, s);
    bf16* t0;
    const bf16* ln = nullptr;
    bf16* o = nullptr;
    if (rq1) {
        t0 = arena_.get<bf16>((size_t)M * C);
        o = arena_.get<bf16>((size_t)M * C);
        qkv_rows_project(t.a1, n, B, HW, HW, C, d, t.proj_in, nullptr, t0, nullptr, 0, 0, s);
        self_attention(t.a1, nullptr, B, Tp, HW, HW, C, d, o, s, nullptr, 0, 0, true);
    } else {
        t0 = linear_rows(n, M, t.proj_in, ACT_NONE, nullptr, nullptr, s, &st0);
        // x = attn1(norm1(x)) + x
        const bool f1 = t.a1.folded && can_fold(st0, M, C, 3 * C, EPI_QKV_HEADS, ACT_NONE, aligned);
        ln = normed(t0, t.ln1, t.a1.folded, f1, true);
        o = arena_.get<bf16>((size_t)M * C);
        self_attention(t.a1, ln, B, Tp, HW, HW, C, d, o, s, f1 ? &st0 : nullptr);
    }
    const int Ng = cond_.Ng;
    // attn1.to_out + residual -> fuser.norm1 -> the fuser's q,k,v^T over the visual rows: the same launch shape
    const bool rq2 = !fuser_off_ && ucfg_.fuser_kind == 0 && fuser_hoist_ && aligned && cond_.Ng > 0 && qkv_rows_for(t, 1, B, HW, HW + Ng, s);
    bf16* t1 = rq2 ? arena_.get<bf16>((size_t)M * C) : linear_rows(o, M, t.a1.out, ACT_NONE, t0, nullptr, s, &st1);

    bf16* t3;
    bool q_done = false;
    if (fuser_off_) {
        t3 = t1;
        st3 = st1;
    } else if (ucfg_.fuser_kind == 0) {
        // fuser (gatedSA): x = x + scale*tanh(alpha_attn) * attn(norm1([x ; linear(objs)]))[:, :N]
        const int Tf = round_up(HW + Ng, 64);
        if (rq2) {
            if (fuser_kv_.size() < st_.size()) fuser_kv_.resize(st_.size());
            const FuserKV& kv = fuser_kv_[t.idx];
            if (kv.epoch != cond_epoch_ || kv.B != B || kv.HW != HW) {
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(s, &cap);
                if (cap != hipStreamCaptureStatusNone) throw GlError(GL_ERR_STATE, "fuser keys of this prompt / shape were not projected before the graph capture");
                fuser_kv_fill(t, B, HW, s);
                n_launches += 2;
            }
            qkv_rows_project(t.fa, o, B, HW, HW + Ng, C, d, t.a1.out, t0, t1, nullptr, Tf, t.idx + 1, s);
            self_attention(t.fa, nullptr, B, HW, HW, HW + Ng, C, d, o, s, nullptr, Tf, t.idx + 1, true);
        } else if (fuser_hoist_ && t.fa.fused && t.fa.folded && aligned) {
            // the grounding tokens' keys / values are in this block's buffers since the prompt was set (fuser_kv_fill); only the visual
            // rows are projected, raw, with the statistics attn1.to_out wrote (or through the plain LayerNorm where it wrote none)
            if (fuser_kv_.size() < st_.size()) fuser_kv_.resize(st_.size());
            const FuserKV& kv = fuser_kv_[t.idx];
            if (kv.epoch != cond_epoch_ || kv.B != B || kv.HW != HW) {
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(s, &cap);
                if (cap != hipStreamCaptureStatusNone) throw GlError(GL_ERR_STATE, "fuser keys of this prompt / shape were not projected before the graph capture");
                fuser_kv_fill(t, B, HW, s);
                n_launches += 2;
            }
            const bool ff = can_fold(st1, M, C, 3 * C, EPI_QKV_HEADS, ACT_NONE, aligned);
            const bf16* rows = ff ? t1 : layernorm_plain(t1, B, HW, C, false, s);
            self_attention(t.fa, rows, B, HW, HW, HW + Ng, C, d, o, s, ff ? &st1 : nullptr, Tf, t.idx + 1);
        } else {
        bf16* lnc = arena_.get<bf16>((size_t)B * Tf * C);
        {
            LNParams P{};
            P.x = t1; P.x2 = cond_.objs[t.idx]; P.B = B; P.N1 = HW; P.N2 = Ng; P.Tpad = Tf; P.C = C; P.eps = 1e-5f;
            P.gamma = t.fa.folded ? nullptr : t.fn1.g; P.beta = t.fa.folded ? nullptr : t.fn1.b; P.y = lnc;
            ProfScope ps(this, s, "ln_kernel", 0.0, 2.0 * B * (HW + Ng) * (double)C * 2);
            CK(layernorm_launch(P, s));
            ++n_launches;
        }
        self_attention(t.fa, lnc, B, Tf, HW, HW + Ng, C, d, o, s);
        }
        //    x = x + scale*tanh(alpha_dense) * ff(norm2(x))
        t3 = fuser_ff_tail(t, o, t1, B, HW, r2, s, &st3, &q_done);
    } else {
        bf16* t2;
        if (ucfg_.fuser_kind == 1) {
            // fuser (gatedSA2, attention.py:271-297): the attention outputs AT the grounding tokens (an sg x sg grid) are
            // projected, resized bicubically to the visual grid and added as the gated residual
            int sg = 0;
            while (sg * sg < Ng) ++sg;
            if (sg * sg != Ng || H != W) throw GlError(GL_ERR_ARG, fmt("gatedSA2 needs square token grids (visual %dx%d, %d grounding tokens)", H, W, Ng));
            const int Ta = HW + Ng;
            const int Tf = round_up(Ta, 64);
            bf16* lnc = arena_.get<bf16>((size_t)B * Tf * C);
            {
                LNParams P{};
                P.x = t1; P.x2 = cond_.objs[t.idx]; P.B = B; P.N1 = HW; P.N2 = Ng; P.Tpad = Tf; P.C = C; P.eps = 1e-5f;
                P.gamma = t.fa.folded ? nullptr : t.fn1.g; P.beta = t.fa.folded ? nullptr : t.fn1.b; P.y = lnc;
                ProfScope ps(this, s, "ln_kernel", 0.0, 2.0 * B * Ta * (double)C * 2);
                CK(layernorm_launch(P, s));
                ++n_launches;
            }
            bf16* oa = arena_.get<bf16>((size_t)B * Ta * C);
            self_attention(t.fa, lnc, B, Tf, Ta, Ta, C, d, oa, s);                 // every token is a query here
            bf16* pr = linear_rows(oa, B * Ta, t.fa.out, ACT_NONE, nullptr, nullptr, s);  // [B][HW + Ng][C]
            t2 = arena_.get<bf16>((size_t)M * C);
            CK(fuser_resize_launch(pr, t1, gates_ + 2 * t.idx, t2, B, Ta, HW, sg, H, C, s));
            ++n_launches;
        } else {
            // fuser (gatedCA, attention.py:207-212): x = x + scale*tanh(alpha_attn) * attn(norm1(x), objs, objs)
            ln = layernorm(t1, B, HW, C, t.fn1, true, s);
            int dp, dpv;
            CK(attn_dims(d, &dp, &dpv));
            AttnBufs& bufs = attn_bufs(B, heads, d, Tp, cond_.obj_Tpad);
            AOperand A;
            aoperand_rows(A, ln, C, C);
            Epilogue E;
            epilogue_defaults(E);
            E.mode = EPI_QK_HEADS;
            E.q = bufs.q; E.k = nullptr; E.C = C; E.H = heads; E.d = d; E.DP = dp; E.T = Tp; E.Tpad_q = bufs.Tq_pad; E.Tpad_k = 0;
            gemm(A, t.fca.q.w, B * Tp, C, C, E, s);
            AttnParams P{};
            P.q = bufs.q; P.k = cond_.obj_k[t.idx]; P.vt = cond_.obj_vt[t.idx]; P.o = o;
            P.H = heads; P.d = d; P.Nq = HW; P.Nk = Ng; P.Tq_pad = bufs.Tq_pad; P.Tk_pad = cond_.obj_Tpad;
            P.ldo = C; P.o_rows_per_b = HW;
            P.scale_log2e = (float)(1.4426950408889634 / std::sqrt((double)d));
            {
                ProfScope ps(this, s, attn_kernel_name(d, Ng, 0), 4.0 * B * heads * (double)HW * Ng * d, 0.0);
                CK(attn_launch(P, B, s));
                log_attention(attn_kernel_name(d, Ng, 0), B, heads, HW, Ng, d);
            }
            ++n_launches;
            t2 = linear_rows(o, M, t.fca.out, ACT_NONE, t1, gates_ + 2 * t.idx, s, r2 ? nullptr : &st2);
        }
        //        x = x + scale*tanh(alpha_dense) * ff(norm2(x))
        t3 = ff_behind(t.fff, t.fn2, t2, st2, B, HW, gates_ + 2 * t.idx + 1, r2, s, &st3);
    }

    // x = attn2(norm2(x), context) + x
    if (!q_done) cross_q_gemm(t, t3, st3, B, HW, s);
    {
        int dp, dpv;
        CK(attn_dims(d, &dp, &dpv));
        AttnBufs& bufs = attn_bufs(B, heads, d, Tp, cond_.ctx_Tpad);
        AttnParams P{};
        P.q = bufs.q; P.k = cond_.ctx_k[t.idx]; P.vt = cond_.ctx_vt[t.idx]; P.o = o;
        P.H = heads; P.d = d; P.Nq = HW; P.Nk = cond_.ctx_T; P.Tq_pad = bufs.Tq_pad; P.Tk_pad = cond_.ctx_Tpad;
        P.ldo = C; P.o_rows_per_b = HW;
        P.scale_log2e = (float)(1.4426950408889634 / std::sqrt((double)d));
        {
            ProfScope ps(this, s, attn_kernel_name(d, cond_.ctx_T, 0), 4.0 * B * heads * (double)HW * cond_.ctx_T * d, 0.0);
            CK(attn_launch(P, B, s));
            log_attention(attn_kernel_name(d, cond_.ctx_T, 0), B, heads, HW, cond_.ctx_T, d);
        }
        ++n_launches;
    }
    // x = attn2.to_out(.) + x;  x = ff(norm3(x)) + x;  proj_out + x_in
    block_ff_tail(t, o, t3, x, out, B, HW, r4, s);
    arena_.release(mk);
    return out;
}

// ---------------------------------------------------------------- conditioning
void Engine::set_fuser_scale(float v, hipStream_t s) {
    if (!has_unet_ || !finalized_) throw GlError(GL_ERR_STATE, "unet not finalized");
    CK(fill_f32_launch(fuser_scale_, v, (int)st_.size(), s));
    // set_alpha_scale(model, 0) (the tail of the reference's alpha schedules, gligen_inference.py:31-66): every gated residual is
    // x + 0 * f(x) = x, so the fuser's attention and feed-forward are not launched at all (gatedSA2 never gets here with 0: the
    // reference's set_alpha_scale does not reach it)
    fuser_off_ = v == 0.f;
}

// UNetModel.restore_first_conv_from_SD (openaimodel.py:400-413): overwrite the packed first-conv
// buffers in place, so captured graphs (which hold these addresses) pick the new weights up.
void Engine::restore_first_conv(const float* w, const float* b, hipStream_t s) {
    if (!has_unet_ || !finalized_) throw GlError(GL_ERR_STATE, "unet not finalized");
    if (ucfg_.inpaint_mode) throw GlError(GL_ERR_STATE, "first conv of an inpainting model is not restorable");
    const int mc = ucfg_.model_channels;
    // a 4 + k channel GLIGEN first conv (grounding downsampler) becomes the 4-channel SD conv: the k extra input
    // channels get zero weights, which is what dropping the concat (openaimodel.py:442-444, first_conv_type "SD") computes
    CK(pack_conv_small_launch(w, const_cast<bf16*>(conv_in_small_.w), mc, conv_in_small_.Cin, conv_in_kpad_, s, ucfg_.in_channels));
    HIPCK(hipMemcpyAsync(const_cast<float*>(conv_in_small_.b), b, mc * sizeof(float), hipMemcpyDeviceToDevice, s));
}

void Engine::set_cond(int Beff, const float* context, int n_ctx, const gl_grounding& g, hipStream_t s) {
    if (!has_unet_ || !finalized_) throw GlError(GL_ERR_STATE, "unet not finalized");
    if (Beff <= 0 || n_ctx <= 0 || g.n <= 0) throw GlError(GL_ERR_ARG, "set_cond: empty batch / context / grounding");
    if (ucfg_.fuser_kind == 1) {
        int sg = 0;
        while (sg * sg < (gkind_ == 1 ? 2 * g.n : g.n)) ++sg;
        if (sg * sg != (gkind_ == 1 ? 2 * g.n : g.n)) throw GlError(GL_ERR_ARG, "gatedSA2 needs a square number of grounding tokens");
    }
    const gl_unet_config& c = ucfg_;
    const int Ng = gkind_ == 1 ? 2 * g.n : g.n;
    const int ctx_Tpad = round_up(n_ctx, 64);
    const int heads = c.num_heads;
    const bool ca = c.fuser_kind == 2;
    const int obj_Tpad = round_up(Ng, 64);
    const int obj_stride = ca ? obj_Tpad : Ng;   // rows per sample of the grounding-token matrix (gatedCA pads to the key tile)
    if (cond_.Beff != Beff || cond_.Ng != Ng || cond_.ctx_Tpad != ctx_Tpad) {
        // captured graphs bake Nk (= HW + Ng), the conditioning buffers and the batch: a last run -- on whatever stream it was issued --
        // is over before they and the buffers go
        sampler_wait_idle();
        HIPCK(hipStreamSynchronize(s));
        sampler_release_graph();
        for (void* p : cond_.allocs) (void)hipFree(p);
        cond_ = Cond{};
        auto palloc = [&](size_t bytes) {
            void* p = nullptr;
            HIPCK(hipMalloc(&p, bytes));
            HIPCK(hipMemset(p, 0, bytes));
            cond_.allocs.push_back(p);
            return p;
        };
        for (const STW& t : st_) {
            int dp, dpv;
            CK(attn_dims(t.d, &dp, &dpv));
            if (ca) {
                cond_.objs.push_back(nullptr);
                cond_.obj_k.push_back(reinterpret_cast<bf16*>(palloc((size_t)Beff * heads * obj_Tpad * dp * sizeof(bf16))));
                cond_.obj_vt.push_back(reinterpret_cast<bf16*>(palloc((size_t)Beff * heads * dpv * obj_Tpad * sizeof(bf16))));
                CK(attn_vt_ones_launch(cond_.obj_vt.back(), Beff * heads, t.d, obj_Tpad, 0));
                CK(attn_k_init_launch(cond_.obj_k.back(), Beff * heads, t.d, obj_Tpad, 0));
            } else {
                cond_.objs.push_back(reinterpret_cast<bf16*>(palloc((size_t)Beff * Ng * t.C * sizeof(bf16))));
            }
            cond_.ctx_k.push_back(reinterpret_cast<bf16*>(palloc((size_t)Beff * heads * ctx_Tpad * dp * sizeof(bf16))));
            cond_.ctx_vt.push_back(reinterpret_cast<bf16*>(palloc((size_t)Beff * heads * dpv * ctx_Tpad * sizeof(bf16))));
            CK(attn_vt_ones_launch(cond_.ctx_vt.back(), Beff * heads, t.d, ctx_Tpad, 0));
            CK(attn_k_init_launch(cond_.ctx_k.back(), Beff * heads, t.d, ctx_Tpad, 0));
        }
        cond_.tokens = reinterpret_cast<bf16*>(palloc((size_t)Beff * obj_stride * c.gr_out_dim * sizeof(bf16)));
        HIPCK(hipStreamSynchronize(0));
        cond_.Beff = Beff;
        cond_.Ng = Ng;
        cond_.ctx_Tpad = ctx_Tpad;
        cond_.obj_Tpad = obj_Tpad;
    }
    if (cond_.ctx_T != n_ctx && (smp_.exec[0] || smp_.exec[1])) {  // captured cross-attention launches bake Nk = ctx_T
        sampler_wait_idle();
        sampler_release_graph();
    }
    cond_.ctx_T = n_ctx;
    arena_.reset();

    // ---- grounding tokens: objs = position_net(**grounding_input)  -> [Beff][Ng][out_dim]
    const int out_dim = c.gr_out_dim;
    bf16* objs = arena_.get<bf16>((size_t)Beff * obj_stride * out_dim);
    if (ca) HIPCK(hipMemsetAsync(objs, 0, (size_t)Beff * obj_stride * out_dim * sizeof(bf16), s));  // key-tile padding rows
    const int rows = Beff * g.n;
    auto mlp = [&](int which, const PosNetIn& pin_in, int remap_off) {
        PosNetIn pin = pin_in;
        const int Kp = pn_[which][0].K;
        pin.out = arena_.get<bf16>((size_t)rows * Kp);
        pin.ld_out = Kp;
        pin.rows = rows;
        CK(posnet_input_launch(pin, s));
        bf16* h1 = linear_rows(pin.out, rows, pn_[which][0], ACT_SILU, nullptr, nullptr, s);
        bf16* h2 = linear_rows(h1, rows, pn_[which][1], ACT_SILU, nullptr, nullptr, s);
        AOperand A;
        aoperand_rows(A, h2, pn_[which][2].K, pn_[which][2].K);
        Epilogue E;
        epilogue_defaults(E);
        E.out = objs; E.ldo = out_dim; E.bias = pn_[which][2].b;
        E.remap_in = g.n; E.remap_out = obj_stride; E.remap_off = remap_off;
        gemm(A, pn_[which][2].w, rows, out_dim, pn_[which][2].K, E, s);
    };
    PosNetIn pin{};
    pin.null_pos = pn_null_pos_;
    pin.mask = g.masks;
    if (gkind_ == 0) {
        if (!g.boxes || !g.masks || !g.text_embeddings) throw GlError(GL_ERR_ARG, "text grounding needs boxes, masks, text_embeddings");
        pin.feat = g.text_embeddings; pin.pos = g.boxes; pin.F = c.gr_in_dim; pin.P = 4; pin.null_feat = pn_null_feat_[0];
        mlp(0, pin, 0);
    } else if (gkind_ == 1) {
        if (!g.boxes || !g.masks || !g.text_masks || !g.image_masks || !g.text_embeddings || !g.image_embeddings)
            throw GlError(GL_ERR_ARG, "text+image grounding needs boxes, masks, text/image masks and embeddings");
        pin.pos = g.boxes; pin.F = c.gr_in_dim; pin.P = 4;
        pin.feat = g.text_embeddings; pin.fmask = g.text_masks; pin.null_feat = pn_null_feat_[0];
        mlp(0, pin, 0);
        pin.feat = g.image_embeddings; pin.fmask = g.image_masks; pin.null_feat = pn_null_feat_[1];
        mlp(1, pin, g.n);
    } else if (gkind_ == 3) {
        if (!g.tokens) throw GlError(GL_ERR_ARG, "grounding_kind 3 needs gl_grounding.tokens");
        CK(pad_rows_cast_launch(g.tokens, objs, Beff, g.n, obj_stride, out_dim, s));
    } else {
        if (!g.points || !g.masks) throw GlError(GL_ERR_ARG, "keypoint grounding needs points and masks");
        if (g.n != c.max_persons * 17) throw GlError(GL_ERR_ARG, "keypoint grounding: n must be max_persons*17");
        pin.feat = kp_table_; pin.feat_mod = g.n; pin.pos = g.points; pin.F = out_dim; pin.P = 2; pin.null_feat = pn_null_feat_[0];
        mlp(0, pin, 0);
    }

    ++cond_epoch_;
    cond_.obj_stride = obj_stride;
    HIPCK(hipMemcpyAsync(cond_.tokens, objs, (size_t)Beff * obj_stride * out_dim * sizeof(bf16), hipMemcpyDeviceToDevice, s));  // gl_unet_grounding_tokens

    // ---- per transformer: fuser.linear(objs), attn2.to_k / to_v (context)
    bf16* ctxb = arena_.get<bf16>((size_t)Beff * ctx_Tpad * c.context_dim);
    CK(pad_rows_cast_launch(context, ctxb, Beff, n_ctx, ctx_Tpad, c.context_dim, s));
    for (const STW& t : st_) {
        int dp, dpv;
        CK(attn_dims(t.d, &dp, &dpv));
        if (!ca) {
            AOperand A;
            aoperand_rows(A, objs, t.flin.K, t.flin.K);
            Epilogue E;
            epilogue_defaults(E);
            E.out = cond_.objs[t.idx]; E.ldo = t.C; E.bias = t.flin.b;
            gemm(A, t.flin.w, Beff * Ng, t.C, t.flin.K, E, s);
        } else {  // gatedCA: fuser.attn.to_k / to_v of the grounding tokens, head layouts of the attention kernel
            {
                AOperand A;
                aoperand_rows(A, objs, t.fca.ctx_dim, t.fca.ctx_dim);
                Epilogue E;
                epilogue_defaults(E);
                E.mode = EPI_QK_HEADS;
                E.q = cond_.obj_k[t.idx]; E.q_tiled = 1; E.C = t.C; E.H = heads; E.d = t.d; E.DP = dp; E.T = obj_Tpad; E.Tpad_q = obj_Tpad;
                gemm(A, t.fca.wk, Beff * obj_Tpad, t.C, t.fca.ctx_dim, E, s);
            }
            {
                Epilogue E;
                epilogue_defaults(E);
                E.mode = EPI_VT_HEADS;
                E.out = cond_.obj_vt[t.idx]; E.H = heads; E.d = t.d; E.DPV = dpv; E.T = obj_Tpad; E.Tpad_k = obj_Tpad;
                CK(gemm_launch_t(t.fca.wv, t.C, objs, Beff * obj_Tpad, t.fca.ctx_dim, E, s));
            }
        }
        {
            AOperand A;
            aoperand_rows(A, ctxb, t.a2.ctx_dim, t.a2.ctx_dim);
            Epilogue E;
            epilogue_defaults(E);
            E.mode = EPI_QK_HEADS;
            E.q = cond_.ctx_k[t.idx]; E.q_tiled = 1; E.C = t.C; E.H = heads; E.d = t.d; E.DP = dp; E.T = ctx_Tpad; E.Tpad_q = ctx_Tpad;
            gemm(A, t.a2.wk, Beff * ctx_Tpad, t.C, t.a2.ctx_dim, E, s);
        }
        {
            Epilogue E;
            epilogue_defaults(E);
            E.mode = EPI_VT_HEADS;
            E.out = cond_.ctx_vt[t.idx]; E.H = heads; E.d = t.d; E.DPV = dpv; E.T = ctx_Tpad; E.Tpad_k = ctx_Tpad;
            CK(gemm_launch_t(t.a2.wv, t.C, ctxb, Beff * ctx_Tpad, t.a2.ctx_dim, E, s));
        }
    }
    // A captured graph of these shapes may be replayed for this prompt without another eager pass: the blocks whose shape is known
    // from the previous prompt get the new grounding-token keys / values now
    if (!ca && fuser_hoist_ && c.fuser_kind == 0)
        for (const STW& t : st_)
            if ((size_t)t.idx < fuser_kv_.size() && fuser_kv_[t.idx].HW && fuser_kv_[t.idx].B == Beff && t.fa.fused && t.fa.folded) fuser_kv_fill(t, Beff, fuser_kv_[t.idx].HW, s);
}

// the `scale` attributes of the fuser modules, one per transformer block in module order (they are plain Python attributes in
// the reference: set_alpha_scale writes the same value into all of them, anything else may write them individually)
void Engine::set_fuser_scales(const float* scales, int n, hipStream_t s) {
    if (!has_unet_ || !finalized_) throw GlError(GL_ERR_STATE, "unet not finalized");
    if (n != (int)st_.size()) throw GlError(GL_ERR_ARG, fmt("set_fuser_scales: %d values for %d fusers", n, (int)st_.size()));
    HIPCK(hipMemcpyAsync(fuser_scale_, scales, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));   // `scales` is the caller's host memory
    bool all_zero = true;
    for (int i = 0; i < n; ++i) all_zero = all_zero && scales[i] == 0.f;
    fuser_off_ = all_zero;
}

void Engine::grounding_tokens(float* out, hipStream_t s) {
    if (!cond_.tokens) throw GlError(GL_ERR_STATE, "no conditioning set");
    CK(bf16_rows_to_f32_launch(cond_.tokens, out, cond_.Beff, cond_.Ng, cond_.obj_stride, ucfg_.gr_out_dim, s));
}

// ---------------------------------------------------------------- UNetModel.forward (openaimodel.py:420-464)
void Engine::emb_table_build(const int64_t* t_host, int R, hipStream_t s) {
    const int mc = ucfg_.model_channels;
    if ((int)emb_t_cache_.size() == R && std::equal(t_host, t_host + R, emb_t_cache_.begin())) return;   // the schedule of the last run
    if (R > emb_table_cap_) {
        const int cap = std::max(R, 64);
        emb_table_ = reinterpret_cast<float*>(persist((size_t)cap * embcat_.N * sizeof(float), false));
        emb_t_dev_ = reinterpret_cast<int64_t*>(persist((size_t)cap * sizeof(int64_t), false));
        if (!emb_cur_) emb_cur_ = reinterpret_cast<float*>(persist((size_t)embcat_.N * sizeof(float), true));
        emb_table_cap_ = cap;
    }
    HIPCK(hipMemcpyAsync(emb_t_dev_, t_host, (size_t)R * sizeof(int64_t), hipMemcpyHostToDevice, s));
    const size_t mk = arena_.mark();
    bf16* temb = arena_.get<bf16>((size_t)R * mc);
    CK(timestep_embed_launch(emb_t_dev_, temb, R, mc, s));
    bf16* e1 = linear_rows(temb, R, te0_, ACT_SILU, nullptr, nullptr, s);
    bf16* semb = linear_rows(e1, R, te2_, ACT_SILU, nullptr, nullptr, s);
    AOperand A;
    aoperand_rows(A, semb, embcat_.K, embcat_.K);
    Epilogue E;
    epilogue_defaults(E);
    E.out = emb_table_; E.ldo = embcat_.N; E.out_f32 = 1; E.bias = embcat_.b;
    gemm(A, embcat_.w, R, embcat_.N, embcat_.K, E, s);
    arena_.release(mk);
    emb_t_cache_.assign(t_host, t_host + R);
}

void Engine::unet_forward(int Beff, int h, int w, const float* x, int xB, const int64_t* t, const float* extra,
                          int extraB, float* eps, hipStream_t s, const float* emb_row) {
    if (!has_unet_ || !finalized_) throw GlError(GL_ERR_STATE, "unet not finalized");
    if (cond_.Beff != Beff) throw GlError(GL_ERR_STATE, fmt("unet_forward batch %d but conditioning was set for %d", Beff, cond_.Beff));
    const gl_unet_config& c = ucfg_;
    const int extra_C = c.inpaint_mode ? c.in_channels + 1 : c.extra_channels;
    if ((extra_C != 0) != (extra != nullptr))
        throw GlError(GL_ERR_ARG, "the extra first-conv input (inpainting_extra_input / downsampled grounding_extra_input) must be given iff the model has those channels");
    if (xB <= 0 || Beff % xB != 0) throw GlError(GL_ERR_ARG, "x batch must divide the effective batch");
    const int mc = c.model_channels;
    arena_.reset();

    // time embedding: emb = time_embed(timestep_embedding(t)); every ResBlock consumes SiLU(emb)
    const float* embout = emb_row;
    int emb_ld = 0;                   // (a precomputed row is every sample's)
    if (!emb_row) {
        bf16* temb = arena_.get<bf16>((size_t)Beff * mc);
        CK(timestep_embed_launch(t, temb, Beff, mc, s));
        bf16* e1 = linear_rows(temb, Beff, te0_, ACT_SILU, nullptr, nullptr, s);
        bf16* semb = linear_rows(e1, Beff, te2_, ACT_SILU, nullptr, nullptr, s);
        float* eo = arena_.get<float>((size_t)Beff * embcat_.N);
        AOperand A;
        aoperand_rows(A, semb, embcat_.K, embcat_.K);
        Epilogue E;
        epilogue_defaults(E);
        E.out = eo; E.ldo = embcat_.N; E.out_f32 = 1; E.bias = embcat_.b;
        gemm(A, embcat_.w, Beff, embcat_.N, embcat_.K, E, s);
        embout = eo;
        emb_ld = embcat_.N;
        ++n_launches;
    }
    CK(gates_launch(alpha_ptrs_, fuser_scale_, gates_, 2 * (int)st_.size(), s));
    ++n_launches;

    struct Act { bf16* p; int C, H, W; };
    std::vector<Act> hs;
    Act cur{nullptr, 0, h, w};

    auto run_layer = [&](const Layer& L, const TRef& in) {
        switch (L.kind) {
            case L_CONV_IN: {
                const int HW = cur.H * cur.W;
                bf16* col = arena_.get<bf16>((size_t)Beff * HW * conv_in_kpad_);
                bf16* out = arena_.get<bf16>((size_t)Beff * HW * mc);
                // sample b reads x[b % xB]: launch per replica group
                const int reps = Beff / xB;
                for (int r = 0; r < reps; ++r) {
                    Im2colParams P{};
                    P.x0 = x; P.C0 = c.in_channels;
                    P.x1 = extra; P.C1 = extra_C;
                    P.B = xB; P.H = cur.H; P.W = cur.W; P.Kpad = conv_in_kpad_;
                    P.out = col + (size_t)r * xB * HW * conv_in_kpad_;
                    if (extra && extraB != xB) throw GlError(GL_ERR_ARG, "extra first-conv input batch must equal x batch");
                    CK(im2col_small_launch(P, s));
                    ++n_launches;
                }
                AOperand A;
                aoperand_rows(A, col, conv_in_kpad_, conv_in_kpad_);
                Epilogue E;
                epilogue_defaults(E);
                E.out = out; E.ldo = mc; E.bias = conv_in_small_.b;
                gemm(A, conv_in_small_.w, Beff * HW, mc, conv_in_kpad_, E, s);
                cur.p = out; cur.C = mc;
                break;
            }
            case L_RES: {
                const ResW& r = res_[L.idx];
                cur.p = resblock(r, in, Beff, cur.H, cur.W, embout, emb_ld, 1e-5f, s);
                cur.C = r.Cout;
                break;
            }
            case L_ST: {
                if (in.p1) throw GlError(GL_ERR_STATE, "transformer over concatenated input");
                cur.p = transformer(st_[L.idx], in.p0, Beff, cur.H, cur.W, s);
                break;
            }
            case L_DOWN: {
                cur.p = conv3x3(in, Beff, cur.H, cur.W, updown_[L.idx], 2, 0, 1, nullptr, 0, nullptr, s);
                cur.H = (cur.H + 2 - 3) / 2 + 1;
                cur.W = (cur.W + 2 - 3) / 2 + 1;
                break;
            }
            case L_UP: {
                cur.p = conv3x3(in, Beff, cur.H, cur.W, updown_[L.idx], 1, 1, 1, nullptr, 0, nullptr, s);
                cur.H *= 2;
                cur.W *= 2;
                break;
            }
        }
    };

    for (const UNetBlock& b : in_blocks_) {
        for (const Layer& L : b.layers) run_layer(L, TRef{cur.p, cur.C, nullptr, 0});
        hs.push_back(cur);
    }
    for (const Layer& L : mid_block_.layers) run_layer(L, TRef{cur.p, cur.C, nullptr, 0});
    for (const UNetBlock& b : out_blocks_) {
        Act sk = hs.back();
        hs.pop_back();
        if (sk.H != cur.H || sk.W != cur.W) throw GlError(GL_ERR_ARG, "latent size must be divisible by the UNet's total stride");
        bool first = true;
        for (const Layer& L : b.layers) {
            if (first) run_layer(L, TRef{cur.p, cur.C, sk.p, sk.C});  // th.cat([h, hs.pop()], dim=1)
            else run_layer(L, TRef{cur.p, cur.C, nullptr, 0});
            first = false;
        }
    }
    // out: GroupNorm32 -> SiLU -> conv3x3 -> NCHW fp32
    {
        const int HW = cur.H * cur.W;
        bf16* a = groupnorm(TRef{cur.p, cur.C, nullptr, 0}, Beff, HW, out_norm_, 1e-5f, true, s);
        AOperand A{};
        A.p0 = a; A.C0 = cur.C; A.ld0 = cur.C; A.mode = A_CONV3;
        A.Hin = cur.H; A.Win = cur.W; A.Ho = cur.H; A.Wo = cur.W; A.stride = 1; A.ups = 0; A.pad_lo = 1;
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_NCHW_F32; E.out = eps; E.bias = out_conv_.b; E.rows_per_b = HW; E.n_real = c.out_channels;
        gemm(A, out_conv_.w, Beff * HW, out_conv_.Npad, 9 * cur.C, E, s);
    }
}

// ---------------------------------------------------------------- VAE
// AttnBlock.forward (model.py:177-202): single head over HW tokens, scale C^-0.5
bf16* Engine::vae_attn(const VaeAttnW& a, const bf16* x, int B, int HW, hipStream_t s) {
    const int C = a.gn.C, M = B * HW;
    if (HW % 64 != 0) throw GlError(GL_ERR_UNSUPPORTED, "VAE attention needs h*w to be a multiple of 64");
    bf16* out = arena_.get<bf16>((size_t)M * C);
    const size_t mk = arena_.mark();
    bf16* n = groupnorm(TRef{x, C, nullptr, 0}, B, HW, a.gn, 1e-6f, false, s);
    bf16* q = linear_rows(n, M, a.q, ACT_NONE, nullptr, nullptr, s);
    bf16* k = linear_rows(n, M, a.k, ACT_NONE, nullptr, nullptr, s);
    bf16* o = arena_.get<bf16>((size_t)M * C);
    for (int b = 0; b < B; ++b) {
        const size_t mb = arena_.mark();
        // v^T [C][HW] = Wv n_b^T  (bias folded into the P v product: rows of P sum to 1)
        bf16* vT = arena_.get<bf16>((size_t)C * HW);
        {
            AOperand A;
            aoperand_rows(A, a.v.w, C, C);
            Epilogue E;
            epilogue_defaults(E);
            E.out = vT; E.ldo = HW;
            gemm(A, n + (size_t)b * HW * C, C, HW, C, E, s);
        }
        float* S = arena_.get<float>((size_t)HW * HW);
        {
            AOperand A;
            aoperand_rows(A, q + (size_t)b * HW * C, C, C);
            Epilogue E;
            epilogue_defaults(E);
            E.out = S; E.ldo = HW; E.out_f32 = 1;
            gemm(A, k + (size_t)b * HW * C, HW, HW, C, E, s);
        }
        bf16* Pm = arena_.get<bf16>((size_t)HW * HW);
        CK(softmax_rows_launch(S, Pm, HW, HW, 1.f / std::sqrt((float)C), s));
        ++n_launches;
        {
            AOperand A;
            aoperand_rows(A, Pm, HW, HW);
            Epilogue E;
            epilogue_defaults(E);
            E.out = o + (size_t)b * HW * C; E.ldo = C; E.bias = a.v.b;
            gemm(A, vT, HW, C, HW, E, s);
        }
        arena_.release(mb);
    }
    {
        AOperand A;
        aoperand_rows(A, o, C, C);
        Epilogue E;
        epilogue_defaults(E);
        E.out = out; E.ldo = C; E.bias = a.proj.b; E.res = x; E.ldres = C;
        gemm(A, a.proj.w, M, C, C, E, s);
    }
    arena_.release(mk);
    return out;
}

// AutoencoderKL.decode (autoencoder.py:40-44) -> Decoder.forward (model.py:535-568)
void Engine::vae_decode(int B, int h, int w, const float* z, float* out, hipStream_t s) {
    if (!has_vae_ || !finalized_) throw GlError(GL_ERR_STATE, "vae not finalized");
    const gl_vae_config& c = vcfg_;
    arena_.reset();
    int H = h, W = w;
    bf16* cur;
    int C = vae_in_small_.Cout;
    {
        const int HW = H * W;
        bf16* col = arena_.get<bf16>((size_t)B * HW * vae_in_kpad_);
        Im2colParams P{};
        P.x0 = z; P.C0 = c.z_channels; P.B = B; P.H = H; P.W = W; P.Kpad = vae_in_kpad_; P.out = col;
        P.pre_w = pq_w_; P.pre_b = pq_b_; P.pre_scale = 1.f / c.scale_factor;
        CK(im2col_small_launch(P, s));
        ++n_launches;
        cur = arena_.get<bf16>((size_t)B * HW * C);
        AOperand A;
        aoperand_rows(A, col, vae_in_kpad_, vae_in_kpad_);
        Epilogue E;
        epilogue_defaults(E);
        E.out = cur; E.ldo = C; E.bias = vae_in_small_.b;
        gemm(A, vae_in_small_.w, B * HW, C, vae_in_kpad_, E, s);
    }
    const float eps = 1e-6f;
    cur = resblock(vmid1_, TRef{cur, C, nullptr, 0}, B, H, W, nullptr, 0, eps, s);
    cur = vae_attn(vattn_, cur, B, H * W, s);
    cur = resblock(vmid2_, TRef{cur, C, nullptr, 0}, B, H, W, nullptr, 0, eps, s);
    for (int level = c.n_mult - 1; level >= 0; --level) {
        for (const ResW& r : vup_[level].blocks) {
            cur = resblock(r, TRef{cur, C, nullptr, 0}, B, H, W, nullptr, 0, eps, s);
            C = r.Cout;
        }
        if (vup_[level].has_up) {
            cur = conv3x3(TRef{cur, C, nullptr, 0}, B, H, W, vup_[level].up, 1, 1, 1, nullptr, 0, nullptr, s);
            H *= 2;
            W *= 2;
        }
    }
    {
        const int HW = H * W;
        bf16* a = groupnorm(TRef{cur, C, nullptr, 0}, B, HW, vnorm_out_, eps, true, s);
        AOperand A{};
        A.p0 = a; A.C0 = C; A.ld0 = C; A.mode = A_CONV3;
        A.Hin = H; A.Win = W; A.Ho = H; A.Wo = W; A.stride = 1; A.ups = 0; A.pad_lo = 1;
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_NCHW_F32; E.out = out; E.bias = vconv_out_.b; E.rows_per_b = HW; E.n_real = c.out_ch;
        gemm(A, vconv_out_.w, B * HW, vconv_out_.Npad, 9 * C, E, s);
    }
}

// AutoencoderKL.encode (autoencoder.py:34-38) -> Encoder.forward (model.py:434-459) -> quant_conv -> posterior sample
void Engine::vae_encode(int B, int H, int W, const float* img, const float* noise, float* z, hipStream_t s) {
    if (!has_venc_ || !finalized_) throw GlError(GL_ERR_STATE, "vae encoder weights were not uploaded / not finalized");
    const gl_vae_config& c = vcfg_;
    const int total_stride = 1 << (c.n_mult - 1);
    if (H % total_stride || W % total_stride) throw GlError(GL_ERR_ARG, "vae_encode: image size must be divisible by the encoder stride");
    arena_.reset();
    int C = venc_in_small_.Cout;
    bf16* cur;
    {
        const int HW = H * W;
        bf16* col = arena_.get<bf16>((size_t)B * HW * venc_in_kpad_);
        Im2colParams P{};
        P.x0 = img; P.C0 = venc_in_small_.Cin; P.B = B; P.H = H; P.W = W; P.Kpad = venc_in_kpad_; P.out = col;
        P.pre_scale = 1.f;
        CK(im2col_small_launch(P, s));
        ++n_launches;
        cur = arena_.get<bf16>((size_t)B * HW * C);
        AOperand A;
        aoperand_rows(A, col, venc_in_kpad_, venc_in_kpad_);
        Epilogue E;
        epilogue_defaults(E);
        E.out = cur; E.ldo = C; E.bias = venc_in_small_.b;
        gemm(A, venc_in_small_.w, B * HW, C, venc_in_kpad_, E, s);
    }
    const float eps = 1e-6f;
    for (int level = 0; level < c.n_mult; ++level) {
        for (const ResW& r : vdown_[level].blocks) {
            cur = resblock(r, TRef{cur, C, nullptr, 0}, B, H, W, nullptr, 0, eps, s);
            C = r.Cout;
        }
        if (vdown_[level].has_down) {
            // Downsample: F.pad(x, (0,1,0,1)) + conv3x3 stride 2 padding 0 (model.py:72-76) = pad_lo 0 in the gather
            cur = conv3x3(TRef{cur, C, nullptr, 0}, B, H, W, vdown_[level].down, 2, 0, 0, nullptr, 0, nullptr, s);
            H /= 2;
            W /= 2;
        }
    }
    cur = resblock(vemid1_, TRef{cur, C, nullptr, 0}, B, H, W, nullptr, 0, eps, s);
    cur = vae_attn(veattn_, cur, B, H * W, s);
    cur = resblock(vemid2_, TRef{cur, C, nullptr, 0}, B, H, W, nullptr, 0, eps, s);
    const int HW = H * W;
    float* moments = arena_.get<float>((size_t)B * 2 * c.z_channels * HW);
    {
        bf16* a = groupnorm(TRef{cur, C, nullptr, 0}, B, HW, venorm_out_, eps, true, s);
        AOperand A{};
        A.p0 = a; A.C0 = C; A.ld0 = C; A.mode = A_CONV3;
        A.Hin = H; A.Win = W; A.Ho = H; A.Wo = W; A.stride = 1; A.ups = 0; A.pad_lo = 1;
        Epilogue E;
        epilogue_defaults(E);
        E.mode = EPI_NCHW_F32; E.out = moments; E.bias = veconv_out_.b; E.rows_per_b = HW; E.n_real = 2 * c.z_channels;
        gemm(A, veconv_out_.w, B * HW, veconv_out_.Npad, 9 * C, E, s);
    }
    CK(vae_posterior_launch(moments, qc_w_, qc_b_, noise, z, B, c.z_channels, HW, c.scale_factor, s));
    ++n_launches;
}

// ---------------------------------------------------------------- PLMS sampler (plms.py:65-162)
// HIP-event time of the UNet evaluations of the last sample_plms call (on the engine's stream).
void Engine::sampler_timing(float* avg_ms, float* first_ms, int* n) {
    if (!smp_.ran || smp_.n_evals == 0) throw GlError(GL_ERR_STATE, "no sampling run to report");
    sampler_wait_idle();
    double sum = 0;
    int cnt = 0;
    float first = 0.f;
    for (int i = 0; i < smp_.n_evals; ++i) {
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, smp_.tev[2 * i], smp_.tev[2 * i + 1]));
        if (i == 0) first = ms;
        if (i >= 2 || smp_.n_evals <= 2) { sum += ms; ++cnt; }  // skip the eager warm-up eval and the capture
    }
    *avg_ms = cnt ? (float)(sum / cnt) : 0.f;
    *first_ms = first;
    *n = smp_.n_evals;
}

// The last sampling run has finished. The engine keeps no caller stream handle across calls (a C-API user may have destroyed the
// stream since): every run ends in an engine-owned event, and that is what later calls wait for.
void Engine::sampler_wait_idle() {
    if (smp_.ran && smp_.ev_done) HIPCK(hipEventSynchronize(smp_.ev_done));
}

void Engine::sampler_release_graph() {
    for (int i = 0; i < 2; ++i) {
        if (smp_.exec[i]) (void)hipGraphExecDestroy(smp_.exec[i]);
        if (smp_.graph[i]) (void)hipGraphDestroy(smp_.graph[i]);
        smp_.exec[i] = nullptr;
        smp_.graph[i] = nullptr;
        smp_.warm[i] = false;
    }
}

void Engine::sample_plms(const gl_plms_args& a, hipStream_t caller) {
    if (!has_unet_ || !finalized_) throw GlError(GL_ERR_STATE, "unet not finalized");
    // The loop runs on the CALLER's stream -- one stream per execution context: with a second, engine-owned stream per context the
    // lanes of a process are four streams on the chip's few hardware queues, and which lanes overlap depends on the order the streams
    // were first used (gligen_inference.py --repeat got none, bench.py 13 %, same library; tools/dbg_cli2.py). Only the legacy default
    // stream, which cannot be captured into a hipGraph, is replaced by an engine-owned non-blocking stream ordered after / before it
    // with events. GL_SAMPLER_OWN_STREAM=1 (developer A/B): always the engine-owned stream, the round 1-4 behaviour.
    static const bool own_env = dev_env("GL_SAMPLER_OWN_STREAM") && atoi(dev_env("GL_SAMPLER_OWN_STREAM")) != 0;
    const bool own = own_env || caller == nullptr;
    if (own && !smp_.stream) {
        HIPCK(hipStreamCreateWithFlags(&smp_.stream, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&smp_.ev_in, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&smp_.ev_out, hipEventDisableTiming));
    }
    hipStream_t s = own ? smp_.stream : caller;
    if (own) {
        HIPCK(hipEventRecord(smp_.ev_in, caller));
        HIPCK(hipStreamWaitEvent(s, smp_.ev_in, 0));
    }
    if (!smp_.ev_done) HIPCK(hipEventCreateWithFlags(&smp_.ev_done, hipEventDisableTiming));
    // the sampler's buffers (x2, eps_pair, the eps history, the time-embedding row) are per context, not per stream: a run on another
    // caller stream than the last one is ordered behind that run's end
    if (smp_.ran && smp_.run_stream != s) HIPCK(hipStreamWaitEvent(s, smp_.ev_done, 0));
    struct DoneGuard {       // the run's end -- also when it ends in an exception: whatever was issued is what later calls wait for
        Engine* e; hipStream_t s;
        ~DoneGuard() { (void)hipEventRecord(e->smp_.ev_done, s); e->smp_.run_stream = s; e->smp_.ran = true; }
    } done_guard{this, s};
    const gl_unet_config& c = ucfg_;
    if (a.n_steps < 1 || !a.timesteps || !a.a_t || !a.a_prev || !a.x) throw GlError(GL_ERR_ARG, "sample_plms: missing schedule or latent");
    if (a.mask && (!a.x0 || !a.noise || !a.sqrt_ac || !a.sqrt_1mac)) throw GlError(GL_ERR_ARG, "sample_plms: mask needs x0, noise and q_sample coefficients");
    const int maskB = a.mask_B ? a.mask_B : a.B, x0B = a.x0_B ? a.x0_B : a.B, noiseB = a.noise_B ? a.noise_B : a.B;
    if (a.mask && ((maskB != 1 && maskB != a.B) || (x0B != 1 && x0B != a.B) || (noiseB != 1 && noiseB != a.B)))
        throw GlError(GL_ERR_ARG, fmt("sample_plms: mask / x0 / noise batch (%d, %d, %d) must be 1 or the latent batch %d", maskB, x0B, noiseB, a.B));
    const bool cfg = a.guidance_scale != 1.f;
    const int Beff = cfg ? 2 * a.B : a.B;
    if (cond_.Beff != Beff) throw GlError(GL_ERR_STATE, fmt("sample_plms: conditioning batch is %d, need %d", cond_.Beff, Beff));
    const int Cl = c.in_channels;
    const int64_t n = (int64_t)a.B * Cl * a.h * a.w;
    if (smp_.B != a.B || smp_.h != a.h || smp_.w != a.w || smp_.extra != a.inpaint_extra || smp_.policy_epoch != ff_policy::epoch.load()) {
        sampler_wait_idle();               // a graph exec still in flight on the PREVIOUS run's stream must not be destroyed
        HIPCK(hipStreamSynchronize(s));
        sampler_release_graph();
        smp_.policy_epoch = ff_policy::epoch.load();
        if (smp_.B != a.B || smp_.h != a.h || smp_.w != a.w) {
            smp_.x2 = reinterpret_cast<float*>(persist(n * sizeof(float), false));
            smp_.eps_pair = reinterpret_cast<float*>(persist(2 * n * sizeof(float), false));
            for (int i = 0; i < 4; ++i) smp_.hist[i] = reinterpret_cast<float*>(persist(n * sizeof(float), false));
            smp_.x_tmp = reinterpret_cast<float*>(persist(n * sizeof(float), false));
            smp_.t_dev = reinterpret_cast<int64_t*>(persist(2 * a.B * sizeof(int64_t), false));
        }
        smp_.B = a.B; smp_.h = a.h; smp_.w = a.w; smp_.extra = a.inpaint_extra;
    }

    // every evaluation of the run shares one timestep over its samples, and the schedule is known: the time-embedding MLP and the
    // emb_layers GEMM of all steps in one batched pass; an evaluation copies its row (80 KB) instead of launching four tiny GEMM chains
    static const bool emb_table_on = !(dev_env("GL_EMB_TABLE") && atoi(dev_env("GL_EMB_TABLE")) == 0);    // developer A/B: 0 = per-evaluation time MLP
    if (emb_table_on) emb_table_build(a.timesteps, a.n_steps, s);
    int evals = 0;
    auto eval = [&](const float* xin, int64_t t, int row) {
        HIPCK(hipMemcpyAsync(smp_.x2, xin, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (emb_table_on) HIPCK(hipMemcpyAsync(emb_cur_, emb_table_ + (size_t)row * embcat_.N, (size_t)embcat_.N * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (!emb_table_on) CK(fill_i64_launch(smp_.t_dev, t, Beff, s));      // (with the table nothing on the device reads the timestep)
        while ((int)smp_.tev.size() < 2 * (evals + 1)) {
            hipEvent_t e;
            HIPCK(hipEventCreate(&e));
            smp_.tev.push_back(e);
        }
        HIPCK(hipEventRecord(smp_.tev[2 * evals], s));
        const int gi = fuser_off_ ? 1 : 0;
        // A variant is captured only after it has run EAGERLY once on this context with these shapes: that pass commits the arena up to
        // the variant's high-water mark (no hipMemMap inside a capture), tunes GEMM tiles and times the row-local / two-GEMM choice
        if (a.use_graph && evals >= 1 && smp_.warm[gi]) {
            if (!smp_.exec[gi]) {
                HIPCK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                try {
                    unet_forward(Beff, a.h, a.w, smp_.x2, a.B, smp_.t_dev, a.inpaint_extra, a.B, smp_.eps_pair, s, emb_table_on ? emb_cur_ : nullptr);
                } catch (...) {
                    hipGraph_t g = nullptr;
                    (void)hipStreamEndCapture(s, &g);
                    if (g) (void)hipGraphDestroy(g);
                    throw;
                }
                HIPCK(hipStreamEndCapture(s, &smp_.graph[gi]));
                HIPCK(hipGraphInstantiate(&smp_.exec[gi], smp_.graph[gi], nullptr, nullptr, 0));
            }
            HIPCK(hipGraphLaunch(smp_.exec[gi], s));
        } else if (a.use_graph && smp_.exec[gi]) {
            HIPCK(hipGraphLaunch(smp_.exec[gi], s));
        } else {
            unet_forward(Beff, a.h, a.w, smp_.x2, a.B, smp_.t_dev, a.inpaint_extra, a.B, smp_.eps_pair, s, emb_table_on ? emb_cur_ : nullptr);
            smp_.warm[gi] = true;
        }
        HIPCK(hipEventRecord(smp_.tev[2 * evals + 1], s));
        ++evals;
        smp_.n_evals = evals;
    };

    // restore_first_conv_from_SD (plms.py:88-89): at the first step whose gate scale is 0. With a schedule that step is
    // derived here (a caller-supplied sd_conv_step must agree or be 0); without one the caller names it.
    int sd_step = -1;
    if (a.sd_conv_w && a.sd_conv_b) {
        if (a.fuser_scale) {
            for (int i = 0; i < a.n_steps && sd_step < 0; ++i)
                if (a.fuser_scale[i] == 0.f) sd_step = i;
            if (a.sd_conv_step > 0 && a.sd_conv_step != sd_step)
                throw GlError(GL_ERR_ARG, fmt("sample_plms: sd_conv_step %d is not the first step with fuser_scale 0 (%d)", a.sd_conv_step, sd_step));
        } else {
            sd_step = a.sd_conv_step;
            if (sd_step >= a.n_steps) throw GlError(GL_ERR_ARG, "sample_plms: sd_conv_step beyond the last step");
        }
    }
    bool restored = false;
    for (int i = 0; i < a.n_steps; ++i) {
        if (a.fuser_scale) set_fuser_scale(a.fuser_scale[i], s);
        if (a.sd_conv_w && a.sd_conv_b && !restored && i == sd_step) {
            restore_first_conv(a.sd_conv_w, a.sd_conv_b, s);
            restored = true;
        }
        if (a.mask)
            CK(inpaint_blend_launch(a.x, a.x0, a.noise + (size_t)i * (n / a.B) * noiseB, a.mask, a.sqrt_ac[i], a.sqrt_1mac[i], a.B, Cl, a.h * a.w,
                                    x0B, noiseB, maskB, s));
        eval(a.x, a.timesteps[i], i);
        PlmsParams P{};
        P.eps_pair = smp_.eps_pair; P.has_uncond = cfg ? 1 : 0; P.guidance = a.guidance_scale;
        P.a_t = a.a_t[i]; P.a_prev = a.a_prev[i]; P.n = n;
        float* slot = smp_.hist[i & 3];
        if (a.ddim) {
            // DDIMSampler.p_sample_ddim, eta = 0 (ddim.py:111-134): same x_prev formula driven by e_t itself
            P.e_t_out = slot; P.c0 = 1.f; P.x = a.x; P.x_out = a.x;
            CK(plms_update_launch(P, s));
        } else if (i == 0) {
            // pseudo improved Euler (plms.py:143-149): x_prev from e_t, evaluate at t_next, average
            P.e_t_out = slot; P.c0 = 1.f; P.x = a.x; P.x_out = smp_.x_tmp;
            CK(plms_update_launch(P, s));
            eval(smp_.x_tmp, a.timesteps[std::min(1, a.n_steps - 1)], std::min(1, a.n_steps - 1));
            P.e_t_out = smp_.hist[1]; P.c0 = 0.5f; P.o1 = slot; P.c1 = 0.5f; P.x = a.x; P.x_out = a.x;
            CK(plms_update_launch(P, s));
        } else {
            P.e_t_out = slot; P.x = a.x; P.x_out = a.x;
            P.o1 = smp_.hist[(i - 1) & 3];
            if (i == 1) { P.c0 = 1.5f; P.c1 = -0.5f; }
            else if (i == 2) { P.o2 = smp_.hist[(i - 2) & 3]; P.c0 = 23.f / 12.f; P.c1 = -16.f / 12.f; P.c2 = 5.f / 12.f; }
            else {
                P.o2 = smp_.hist[(i - 2) & 3]; P.o3 = smp_.hist[(i - 3) & 3];
                P.c0 = 55.f / 24.f; P.c1 = -59.f / 24.f; P.c2 = 37.f / 24.f; P.c3 = -9.f / 24.f;
            }
            CK(plms_update_launch(P, s));
        }
    }
    if (own) {
        HIPCK(hipEventRecord(smp_.ev_out, s));
        HIPCK(hipStreamWaitEvent(caller, smp_.ev_out, 0));
    }
}

}  // namespace gl
