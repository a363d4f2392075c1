// Row-local feed-forward for gfx950 (CDNA4, wave64): one launch for
//     y = res + gate * ( GEGLU( LN(x) W1^T + b1 ) W2^T + b2 )            (+ row statistics of y)
// i.e. the reference's FeedForward with the LayerNorm in front of it and the residual behind it
// (ldm/modules/attention.py:37-64 GEGLU / FeedForward, :333-338 BasicTransformerBlock._forward, :236-244 GatedSelfAttentionDense).
//
// Why: as two GEMMs the pair is bound by operand delivery, not by the matrix cores -- the 256 x 128 tiles of gemm_wide_kernel stage
// one byte through the LDS-DMA per 85 FLOP, the hidden [M][4C] activation goes out to HBM and comes back (2 x 84 MB at the 64 x 64
// level), and the erf-GEGLU epilogue of a K = 320 projection is as long as its K loop. Here a WAVE owns 32 complete rows (one per
// lane pair) from the first instruction to the last:
//   * its x rows live in the AGPR half as v_mfma_f32_32x32x16_bf16 B operands (80 registers at C = 320), normalised in registers;
//   * per chunk of 32 hidden features:  S^T[64 x 32 rows] = W1c X^T (40 MFMAs + 2 for the bias, which rides as a 21st k-step against
//     a constant one-hot operand), erf-GEGLU on the accumulators in registers, and the packed result IS the B operand of
//     Y^T[C x 32 rows] += W2c P^T (20 MFMAs): with the weights as the A operand a lane's accumulator holds features 8 j + 4 h + e of
//     ONE row, value rows and their gate rows of a block sit in the same lane, and the pack kernel orders W2's k-slots to match
//     (the S^T -> P^T trick of attention.hip);
//   * the FF-out accumulator Y^T (160 registers) stays in the AGPR half until the epilogue adds bias, gate and residual.
// So nothing but weights goes through LDS: one byte per 128 FLOP (4 waves x 32 rows share every weight fragment), no activation
// staging, no LDS hop for the hidden tile, no cross-wave data at all. The price is ONE wave per SIMD (up to 512 registers): nothing
// hides a wave's own fragment reads, DMA issue and GEGLU arithmetic except its own MFMAs, and a first version that left the order to
// hipcc ran the three back to back (profiles/r4/ffn_v2_ablation.txt: skeleton 34 us + MFMAs 47 us = 81 us per workgroup). The main
// loop is therefore one fixed instruction order: every MFMA, fragment read, wait and DMA is an asm statement (volatile asm keeps its
// order), and the GEGLU is cut into 64 micro-steps of 2-3 VALU instructions, each pinned between two empty asm statements into the
// gap behind one particular MFMA -- ~5 issue slots per 32-cycle MFMA, the budget MI355X_MICROARCH.md gives a lone wave.
//
// The weights are a STREAM of 1 KiB fragment blocks in exactly the order the waves consume them (ff_pack_kernel): a block is one
// MFMA A operand in lane order, so an LDS-DMA instruction copies it verbatim and a fragment read is ds_read_b128 at lane * 16 --
// conflict free without any swizzle. Stage c = [W1 + bias of chunk c : 42 blocks][W2 of chunk c-1 : 20 blocks][2 blocks of padding];
// two stages in LDS, one barrier per chunk: vmcnt(0) -> barrier -> compute stage c with the DMA of stage c+1 spread over it.
#include "ffn.h"

#include <type_traits>
#include <utility>

namespace gl {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int C>
struct FFGeom {
    static_assert(C % 32 == 0, "C must be a multiple of 32");
    static constexpr int KS = C / 16;            // k-steps (16 wide) of the GEGLU projection (K = C)
    static constexpr int NKP = KS + 1;           // + the bias step
    static constexpr int NCB = C / 32;           // output-feature blocks (32 rows) of FF-out
    static constexpr int NCH = 4 * C / 32;       // hidden chunks of 32 features
    static constexpr int NPB = 2 * NKP;          // projection blocks per chunk, order (k-step, block nb)
    static constexpr int NOB = 2 * NCB;          // FF-out blocks per chunk, order (k-step nb, feature block cb)
    static constexpr int NBLK = NPB + NOB;       // = MFMAs per chunk
    static constexpr int NBS = (NBLK + 3) / 4 * 4;   // blocks per stage incl. padding: every wave copies NBS / 4
    static constexpr int STAGE = NBS * 1024;
    static_assert((NBLK - 1) * 1024 < 65536, "fragment reads address a stage through the 16-bit offset field");
    static constexpr int NSTAGE = NCH + 1;       // stage c = [W1(c) + b1(c)][W2(c-1)], c = 0 .. NCH
    static constexpr size_t STREAM_BYTES = (size_t)NSTAGE * STAGE;
    static constexpr int LDS_BYTES = 2 * STAGE + 3 * C * 4;      // two stages + [b2][pre_b][post_b] as floats (the epilogues read their biases from LDS)
    static_assert(NCH % 2 == 0, "chunk parity of the last chunk");
    // a chained C x C projection: KS k-steps x NCB feature blocks = KS * NCB blocks, in PJ_ST stages of PJ_BLK (k-step major inside a stage)
    static constexpr int PJ_KS = 4;                   // k-steps per stage
    static constexpr int PJ_BLK = PJ_KS * NCB;        // 40 blocks per stage
    static constexpr int PJ_ST = KS / PJ_KS;          // 5 stages
    static_assert(KS % PJ_KS == 0 && PJ_BLK % 4 == 0 && PJ_ST % 2 == 1, "projection stage geometry (an odd stage count flips the LDS buffer parity)");
    static constexpr size_t PJ_BYTES = (size_t)PJ_ST * PJ_BLK * 1024;
    static constexpr size_t chain_bytes(bool pre, bool post) { return STREAM_BYTES + (pre ? PJ_BYTES : 0) + (post ? PJ_BYTES : 0); }
};

// Fragment layouts (v_mfma_f32_32x32x16_bf16, weights as the A operand): A lane L holds row L & 31, k-slots 8 (L >> 5) .. + 7;
// B lane L holds column (= activation row) L & 31, the same k-slots; D lane L holds column L & 31, register 4 j + e = row 8 j + 4 (L >> 5) + e.
//  * projection block (chunk c, nb): rows 0..15 = VALUE features c 32 + nb 16 + r, rows 16..31 = their GATE features (reference GEGLU:
//    x, gate = proj(x).chunk(2)), so a lane's registers 0..7 are 8 values and 8..15 the gates of the same 8 hidden features
//    f = 8 jj + 4 h + e (jj = 0, 1);
//  * P^T operand of FF-out k-step nb: k-slot 8 h + 4 jj + e  <->  hidden feature c 32 + nb 16 + 8 jj + 4 h + e.
// KPERM: the B operand of this product comes out of accumulator registers (a projection or the feed-forward behind another stage):
// its k-slot 8 h + 4 jj + e holds feature 16 ks + 8 jj + 4 h + e instead of 16 ks + 8 h + (4 jj + e) -- the weights' columns follow
__device__ __forceinline__ int ff_kidx(int ks, int h, int e, bool kperm) { return kperm ? 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3) : 16 * ks + 8 * h + e; }

template <int C>
__global__ void __launch_bounds__(256) ff_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ pre_w, const float* __restrict__ post_w,
                                                      unsigned char* __restrict__ stream) {
    using G = FFGeom<C>;
    const size_t pre_bytes = pre_w ? G::PJ_BYTES : 0, ff_end = pre_bytes + G::STREAM_BYTES;
    const size_t n_pieces = (ff_end + (post_w ? G::PJ_BYTES : 0)) / 16;
    for (size_t pc = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pc < n_pieces; pc += (size_t)gridDim.x * blockDim.x) {
        U4BF8 o;
        o.u = make_uint4(0, 0, 0, 0);
        const size_t byte = pc * 16;
        const int L = (int)(byte >> 4) & 63, r = L & 31, h = L >> 5;
        if (byte < pre_bytes || byte >= ff_end) {
            // projection segment: block b = stage * PJ_BLK + (k-step inside the stage) * NCB + feature block
            const bool post = byte >= ff_end;
            const float* w = post ? post_w : pre_w;
            const int b = (int)((byte - (post ? ff_end : 0)) >> 10);
            const int st = b / G::PJ_BLK, i = b % G::PJ_BLK, ks = st * G::PJ_KS + i / G::NCB, cb = i % G::NCB;
            const float* src = w + (size_t)(cb * 32 + r) * C;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = f2bf(src[ff_kidx(ks, h, e, post)]);
            *reinterpret_cast<uint4*>(stream + byte) = o.u;
            continue;
        }
        const size_t fb = byte - pre_bytes;
        const int st = (int)(fb / G::STAGE);
        const int in = (int)(fb % G::STAGE);
        const int blk = in >> 10;
        if (blk < G::NPB) {
            const int ks = blk >> 1, nb = blk & 1;
            if (st < G::NCH) {
                const int row = (r >> 4) * 4 * C + st * 32 + nb * 16 + (r & 15);
                if (ks < G::KS) {
                    const float* src = w1 + (size_t)row * C;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.e[e] = f2bf(src[ff_kidx(ks, h, e, pre_w != nullptr)]);
                } else if (h == 0) {
                    o.e[0] = f2bf(b1[row]);      // the bias step multiplies k-slot 0 by one (bf16 bias: the accumulator is rounded to bf16 anyway)
                }
            }
        } else if (blk < G::NBLK) {
            const int ob = blk - G::NPB, nb = ob / G::NCB, cb = ob % G::NCB, c = st - 1;
            if (c >= 0) {
                const float* src = w2 + (size_t)(cb * 32 + r) * 4 * C + c * 32 + nb * 16 + 4 * h;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = f2bf(src[8 * (e >> 2) + (e & 3)]);
            }
        }
        *reinterpret_cast<uint4*>(stream + byte) = o.u;
    }
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- the asm vocabulary of the main loop (all volatile: they keep their source order; none clobbers "memory")
template <int OFF>
__device__ __forceinline__ void ffr_rd16(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void ffr_wait_lgkm() {
#ifdef GL_FFN_LGKM0
    asm volatile("s_waitcnt lgkmcnt(0)");
#else
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
#endif
}
// LDS-DMA of one 1 KiB block (gfx950: buffer_load_dwordx4 ... lds): M0 = LDS byte address of the block, lane * 16 rides in the VGPR
// offset, the stream offset in the SGPR offset. Both cursors step to the wave's next block (4 KiB on) inside the statement; the first
// s_add is also the wait state between the write of M0 and its use.
// PAD: a VMEM instruction must not read an SGPR (descriptor, offset) within 5 wait states of the SALU write -- hipcc does not pad
// an asm statement. In the main loop the statements are four MFMAs apart; the prologue's sit back to back and open with s_nop 4.
template <bool PAD = false>
__device__ __forceinline__ void ffr_dma(const u32x4& rsrc, unsigned lane16, unsigned& lds_dst, unsigned& soff) {
#ifdef GL_FFN_BUILTIN_DMA
    {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)rsrc.y << 32) | rsrc.x), 0, (int)rsrc.z, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(uintptr_t)lds_dst, 16, (int)lane16, (int)soff, 0, 0);
        lds_dst += 0x1000;
        soff += 0x1000;
        return;
    }
#endif
    if constexpr (PAD)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_add_u32 %0, %0, 0x1000\n\tbuffer_load_dwordx4 %2, %3, %1 offen lds\n\ts_add_u32 %1, %1, 0x1000"
                     : "+s"(lds_dst), "+s"(soff)
                     : "v"(lane16), "s"(rsrc)
                     : "scc");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_add_u32 %0, %0, 0x1000\n\tbuffer_load_dwordx4 %2, %3, %1 offen lds\n\ts_add_u32 %1, %1, 0x1000"
                     : "+s"(lds_dst), "+s"(soff)
                     : "v"(lane16), "s"(rsrc)
                     : "scc");
}
// The GEGLU state rides through every MFMA statement as in/out operands (no instruction reads them): the VALU micro-step that
// follows an MFMA in the source consumes what that statement "defined" and feeds the next one, so hipcc can place it nowhere but
// in the gap between the two.
struct GegluState { float l0, l1, m0, m1; };
#define FFR_GS(s) "+v"(s.l0), "+v"(s.l1), "+v"(s.m0), "+v"(s.m1)
// D = A B (+ D): A = weight fragment (VGPR), B = activation operand
#ifdef GL_FFN_MFMA_PAD
#define FFR_PAD "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
#else
#define FFR_PAD
#endif
__device__ __forceinline__ void mfma_first_ba(f32x16& d, const bf16x8& a, const bf16x8& b, GegluState& s) {   // B in the AGPR half, C = 0
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, 0" FFR_PAD : "=&v"(d), FFR_GS(s) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_acc_ba(f32x16& d, const bf16x8& a, const bf16x8& b, GegluState& s) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0" FFR_PAD : "+v"(d), FFR_GS(s) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_acc_bv(f32x16& d, const bf16x8& a, const bf16x8& b, GegluState& s) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0" FFR_PAD : "+v"(d), FFR_GS(s) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_aa(f32x16& d, const bf16x8& a, const u32x4& b, GegluState& s) {      // accumulator in the AGPR half
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0" FFR_PAD : "+a"(d), FFR_GS(s) : "v"(a), "v"(b));
}

__device__ __forceinline__ void mfma_acc_aab(f32x16& d, const bf16x8& a, const bf16x8& b) {                       // accumulator and B in the AGPR half
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" FFR_PAD : "+a"(d) : "v"(a), "a"(b));
}

// The last MFMAs' results before hipcc's own reads of the accumulators: 32 wait states, and EVERY tuple is an in/out operand of the
// statement -- a tuple whose last MFMA sits earlier in the stream is otherwise "defined" by that earlier statement, and hipcc is free
// to hoist the epilogue's v_accvgpr_read of it right behind that MFMA's ISSUE, 30 cycles before its result lands (seen: one stale
// register per row, off by exactly the last chunk's contribution)
template <int N>
__device__ __forceinline__ void ffr_settle(f32x16 (&acc)[N]) {
    static_assert(N == 10, "operand list written for 10 tuples");
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]), "+a"(acc[9]));
}

// erf-GELU of the gate times the value for one pair of hidden features, in 7 micro-steps of 1-4 plain fp32 VALU instructions
// (packed fp32 costs a wait state per dependent pair on gfx950 and is dearer than two plain ops beside MFMAs, MI355X_MICROARCH.md):
//   gelu(x) = x Phi(x) = max(x, 0) - |x| h(|x|),   h(a) = erfc(a / sqrt2) / 2 = 2^L(a),
// L a cubic fitted to log2(erfc(a / sqrt2) / 2) on [0, 9] (weighted towards the absolute error of a h(a)): |gelu error| <= 7.8e-5 for
// every x (tools/fit_gelu.py; the output is rounded to bf16, 2^-9 relative), all coefficients negative, so h -> 0 for large |x|.
// One transcendental per feature instead of the two (rcp + exp) of the Abramowitz-Stegun form in common.h, 7 VALU instead of 16.
#define FFR_GELU_C3 -0.026971418f
#define FFR_GELU_C2 -0.49073458f
#define FFR_GELU_C1 -1.1382852f
#define FFR_GELU_C0 -1.0007436f
template <int J>
__device__ __forceinline__ void geglu_step(GegluState& s, float v0, float v1, float x0, float x1, unsigned& out) {
    const float a0 = __builtin_fabsf(x0), a1 = __builtin_fabsf(x1);
    if constexpr (J == 0) {
        s.l0 = fmaf(a0, FFR_GELU_C3, FFR_GELU_C2);
        s.l1 = fmaf(a1, FFR_GELU_C3, FFR_GELU_C2);
    } else if constexpr (J == 1) {
        s.l0 = fmaf(a0, s.l0, FFR_GELU_C1);
        s.l1 = fmaf(a1, s.l1, FFR_GELU_C1);
    } else if constexpr (J == 2) {
        s.l0 = fmaf(a0, s.l0, FFR_GELU_C0);
        s.l1 = fmaf(a1, s.l1, FFR_GELU_C0);
        s.m0 = fmaxf(x0, 0.f);     // (one v_max each: ffn.hip is built with -fno-honor-nans, else hipcc canonicalises the operand first)
        s.m1 = fmaxf(x1, 0.f);
    } else if constexpr (J == 3) {
        s.l0 = __builtin_amdgcn_exp2f(s.l0);
        s.l1 = __builtin_amdgcn_exp2f(s.l1);
    } else if constexpr (J == 4) {
        s.l0 = fmaf(-a0, s.l0, s.m0);
        s.l1 = fmaf(-a1, s.l1, s.m1);
    } else if constexpr (J == 5) {
        s.l0 *= v0;
        s.l1 *= v1;
    } else {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 o = __builtin_convertvector(f32x2{s.l0, s.l1}, bf16x2);
        out = __builtin_bit_cast(unsigned, o);
    }
}

// ABL: developer ablation bits (tools/gpu_r4b.sh; only ABL = 0 is in the product library): 1 no GEGLU arithmetic, 2 no DMA behind the
// prologue's, 8 no projection MFMAs, 16 no FF-out MFMAs, 32 no barrier
// PRE / POST: a row-local C x C projection in front of / behind the feed-forward in the same launch (FFRowsParams::pre / post)
// POST = 2 (round 6): the trailing projection is a cross-attention's to_q behind the NEXT LayerNorm (attention.py:336 attn2(norm2(x))):
// y is stored (it is the residual stream), normalised in registers, projected, and q leaves in the attention kernels' head layout
template <int C, int ABL = 0, bool PRE = false, int POST = 0>
__global__ void __launch_bounds__(256, 1) ff_rows_kernel(FFRowsParams p) {
    using G = FFGeom<C>;
    constexpr int KS = G::KS, NKP = G::NKP, NCB = G::NCB, NCH = G::NCH, NPB = G::NPB, NBLK = G::NBLK, STAGE = G::STAGE;
    constexpr int PJ_BLK = G::PJ_BLK, PJ_ST = G::PJ_ST, PJ_KS = G::PJ_KS;
    constexpr int FB = PRE ? 1 : 0;     // LDS buffer of the feed-forward's chunk 0 (the leading projection's PJ_ST stages flip the parity)
#ifndef GL_FFN_RD
#define GL_FFN_RD 8
#endif
#ifndef GL_FFN_DMA_EVERY
#define GL_FFN_DMA_EVERY 3
#endif
    constexpr int RD = GL_FFN_RD;      // fragment ring (blocks): the read of block i + RD is issued behind MFMA i
    constexpr int NR = RD + 4;
    constexpr int DEV = GL_FFN_DMA_EVERY;   // the next stage's 16 DMA pieces per wave: one behind every DEV-th MFMA of the chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + (lane & 31);      // this lane's row (shared with lane ^ 32)
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- weights: stage 0 on its way while the rows are loaded and normalised
    u32x4 rs;
    {
        const unsigned long long a = (unsigned long long)(uintptr_t)p.stream;
        rs.x = (unsigned)a;
        rs.y = (unsigned)(a >> 32);           // base[47:32], stride 0
        rs.z = (unsigned)G::chain_bytes(PRE, POST != 0);     // num_records
        rs.w = 0x00020000u;                   // raw buffer, dword format (as __builtin_amdgcn_make_buffer_rsrc builds it for gfx950)
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // DMA cursors of this wave (it copies blocks wave, wave + 4, ... of every stage): LDS destination and stream offset of its next block
    const unsigned lds_w = lds0 + (unsigned)wave * 1024u;
    unsigned dma_lds = lds_w;
    unsigned dma_off = (unsigned)wave * 1024u;
#pragma unroll
    for (int i = 0; i < (PRE ? PJ_BLK : G::NBS) / 4; ++i) ffr_dma<true>(rs, lane16, dma_lds, dma_off);

    // ---- the wave's rows as B-operand fragments: lane holds x[row][16 ks + 8 h .. + 7]
    bf16x8 xf[KS];
    {
        const bf16* src = p.x + (size_t)row * p.ldx + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
    }
    // Round 6: no global load sits between an epilogue's stores. A load inside the per-feature-block loop makes hipcc wait vmcnt(0)
    // in front of its use, i.e. for every store issued before it (loads and stores share the counter): ten serialised round trips
    // per epilogue. The biases go through LDS, the leading projection's residual rows are fetched here, the other epilogues issue
    // all their residual loads in one block in front of their loop.
    uint2 prv[PRE ? NCB : 1][4];
    if constexpr (PRE) {
        const bf16* src = p.pre_res + (size_t)row * p.ld_pre_res + h * 4;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) prv[cb][j] = *reinterpret_cast<const uint2*>(src + cb * 32 + j * 8);
    }
    float* lbias = reinterpret_cast<float*>(smem + 2 * STAGE);       // [b2][pre_b][post_b]
    for (int i = t; i < C * (1 + (PRE ? 1 : 0) + (POST ? 1 : 0)); i += 256)
        lbias[i] = i < C ? p.b2[i] : (PRE && i < 2 * C) ? p.pre_b[i - C] : p.post_b[i - C * (PRE ? 2 : 1)];
    const float* lb2 = lbias;
    const float* lpre = lbias + C;
    const float* lpost = lbias + C * (PRE ? 2 : 1);
    const float gate = ((p.res || PRE) && p.gate) ? *p.gate : 1.f;
    const float pre_gate = (PRE && p.pre_gate) ? *p.pre_gate : 1.f;
    const float res_in_acc = (PRE && fabsf(gate) > 1e-20f) ? 1.f / gate : 0.f;   // chained form: the feed-forward's residual rides in its accumulator
    if (!PRE && p.normalize) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = bf2f(xf[ks][e]);
                s += v;
                ss = fmaf(v, v, ss);
            }
        s += __shfl_xor(s, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float mean = s * (1.f / C);
        const float var = fmaxf(ss * (1.f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float nm = -mean * rstd;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[ks][e] = f2bf(fmaf(bf2f(xf[ks][e]), rstd, nm));
    }
    bf16x8 xb;                                   // B operand of the bias step: k-slot 0 = 1
#pragma unroll
    for (int e = 0; e < 8; ++e) xb[e] = f2bf(0.f);
    xb[0] = f2bf(h == 0 ? 1.f : 0.f);

    f32x16 acc2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[cb][e] = 0.f;
    // two projection accumulator sets: chunk c multiplies into a1[c & 1] while the GEGLU of chunk c - 1 reads a1[(c - 1) & 1]; likewise
    // the GEGLU of chunk c writes the P^T operand P[c & 1] while the FF-out product of chunk c - 1 reads P[(c - 1) & 1]
    f32x16 a1[2][2];
    u32x4 P[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) P[i][0] = P[i][1] = u32x4{0u, 0u, 0u, 0u};
    GegluState gs;
    gs.l0 = gs.l1 = gs.m0 = gs.m1 = 0.f;
    const unsigned ra_st[2] = {lds0 + lane16, lds0 + (unsigned)STAGE + lane16};

    // A chained projection: acc[cb] += W-block(ks, cb) x xin[ks] over PJ_ST stages of PJ_BLK blocks; buffer parity of its first stage
    // BUF0; NEXT = blocks per wave of whatever follows its last stage in the stream (0: nothing), fetched under that last stage
    auto proj_phase = [&](f32x16 (&acc)[NCB], bf16x8 (&xin)[KS], auto buf0_c, auto next_c, auto skip0_c) {
        constexpr int BUF0 = decltype(buf0_c)::value, NEXT = decltype(next_c)::value;
        constexpr bool SKIP0 = decltype(skip0_c)::value;     // the first stage has been waited for already (in front of a store burst)
        static_for<PJ_ST>([&](auto sc) {
            constexpr int st = decltype(sc)::value, BUF = (BUF0 + st) & 1;
            constexpr int pieces = st + 1 < PJ_ST ? PJ_BLK / 4 : NEXT;
            constexpr int every = pieces > PJ_BLK / 4 ? 2 : (DEV < 4 ? 2 : 4);      // (16 blocks of a feed-forward stage do not fit one per four gaps)
            if constexpr (!(SKIP0 && st == 0)) asm volatile("s_waitcnt vmcnt(0)");
            if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
            const unsigned ra = ra_st[BUF];
            dma_lds = lds_w + (BUF ? 0u : (unsigned)STAGE);
            bf16x8 fr[NR];
            static_for<RD>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ffr_rd16<i * 1024>(fr[i % NR], ra);
            });
            static_for<PJ_BLK>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 4 == 0) {
                    constexpr int last = (i + RD - 1 < PJ_BLK - 1) ? i + RD - 1 : PJ_BLK - 1;
                    constexpr int need = (i + 3 < PJ_BLK - 1) ? i + 3 : PJ_BLK - 1;
                    ffr_wait_lgkm<last - need>();
                }
                constexpr int ks = st * PJ_KS + i / NCB, cb = i % NCB;
                mfma_acc_aab(acc[cb], fr[i % NR], xin[ks]);
                if constexpr (i + RD < PJ_BLK) ffr_rd16<(i + RD) * 1024>(fr[(i + RD) % NR], ra);
                if constexpr (i % every == 1 && i / every < pieces && !(ABL & 2)) ffr_dma(rs, lane16, dma_lds, dma_off);
            });
        });
        ffr_settle(acc);
    };
    // rows in accumulator layout (register 4 j + e of block cb = feature 32 cb + 8 j + 4 h + e) -> B operand fragments in the
    // permuted k-slot order (ff_kidx): slot 4 jj + e of k-step ks = register 4 (2 (ks & 1) + jj) + e of block ks >> 1
    auto to_fragments = [&](const float (&v)[NCB][16], bf16x8 (&dst)[KS], float scale, float shift) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[ks][e] = f2bf(fmaf(v[ks >> 1][4 * (2 * (ks & 1) + (e >> 2)) + (e & 3)], scale, shift));
    };

    // micro-step u (0..55) of the GEGLU of one chunk: call k = u / 7 handles block nb = k >> 2, feature pair (jj, pr); source = that
    // chunk's projection accumulators, destination dword 2 jj + pr of its P^T operand
    auto gstep = [&](auto uc, f32x16 (&src)[2], u32x4 (&dst)[2]) {
        constexpr int u = decltype(uc)::value, k = u / 7, j = u % 7, nb = k >> 2, jj = (k >> 1) & 1, pr = k & 1;
        constexpr int v0 = 4 * jj + 2 * pr, g0 = 8 + 4 * jj + 2 * pr;
        if constexpr (ABL & 1) {
            if constexpr (j == 6) {
                U2BF4 o;
                o.e[0] = f2bf(src[nb][v0] + src[nb][g0]);
                o.e[1] = f2bf(src[nb][v0 + 1] + src[nb][g0 + 1]);
                dst[nb][2 * jj + pr] = o.u.x;
            }
            return;
        }
        unsigned out = 0;
        geglu_step<j>(gs, src[nb][v0], src[nb][v0 + 1], src[nb][g0], src[nb][g0 + 1], out);
        if constexpr (j == 6) dst[nb][2 * jj + pr] = out;
    };

    // One chunk `it` (PAR = it & 1): 62 MFMAs in stream order. Slots 0..41 multiply the projection of chunk `it` into T = a1[PAR]
    // (k-step major; the 21st k-step is the bias), slots 42..61 the FF-out product of chunk it - 1 from P[PAR ^ 1]. Behind MFMA i:
    // the read of block i + 8, every fourth gap one DMA block of the next stage, and one GEGLU micro-step:
    //   gaps 0..36 : u = 19..55 of chunk it - 1 (source S = a1[PAR ^ 1])                     -> P[PAR ^ 1], complete long before slot 42
    //   gaps 43..61: u = 0..18 of chunk it (source T, whose last MFMAs are slots 40 / 41)   -> P[PAR]
    // FIRST: chunk 0 (no previous chunk: its FF-out blocks are zeros and P[1] = 0).
    auto chunk = [&](auto first_c, auto par_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        f32x16 (&T)[2] = a1[PAR];
        f32x16 (&S)[2] = a1[PAR ^ 1];
        asm volatile("s_waitcnt vmcnt(0)");            // this wave's blocks of the stage have landed
        if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();   // ... and everybody's; every wave is done reading the other stage
#ifdef GL_FFN_SLOW_SYNC
        __builtin_amdgcn_s_sleep(20);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
        __builtin_amdgcn_s_barrier();
#endif
        constexpr int BUF = PAR ^ FB;
        const unsigned ra = ra_st[BUF];
        dma_lds = lds_w + (BUF ? 0u : (unsigned)STAGE);   // the next stage goes where the previous one was
        bf16x8 fr[NR];
        static_for<RD>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ffr_rd16<i * 1024>(fr[i % NR], ra);
        });
        static_for<NBLK>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i % 4 == 0) {
                // blocks i .. i + 3 have arrived when at most the reads of blocks i + 4 .. i + RD - 1 are outstanding
                constexpr int last = (i + RD - 1 < NBLK - 1) ? i + RD - 1 : NBLK - 1;
                constexpr int need = (i + 3 < NBLK - 1) ? i + 3 : NBLK - 1;
                ffr_wait_lgkm<last - need>();
            }
            if constexpr (i < NPB) {
                constexpr int ks = i >> 1, nb = i & 1;
                if constexpr (!(ABL & 8)) {
                    if constexpr (ks == 0) mfma_first_ba(T[nb], fr[i % NR], xf[0], gs);
                    else if constexpr (ks < KS) mfma_acc_ba(T[nb], fr[i % NR], xf[ks], gs);
                    else mfma_acc_bv(T[nb], fr[i % NR], xb, gs);
                }
            } else {
                constexpr int ob = i - NPB, nb = ob / NCB, cb = ob % NCB;
                if constexpr (!(ABL & 16)) mfma_acc_aa(acc2[cb], fr[i % NR], P[PAR ^ 1][nb], gs);
            }
            if constexpr (i + RD < NBLK) ffr_rd16<(i + RD) * 1024>(fr[(i + RD) % NR], ra);
            if constexpr (i >= 1 && (i - 1) % DEV == 0 && (i - 1) / DEV < G::NBS / 4 && !(ABL & 2)) ffr_dma(rs, lane16, dma_lds, dma_off);
            if constexpr (i < 37) {
                if constexpr (!FIRST) gstep(std::integral_constant<int, 19 + i>{}, S, P[PAR ^ 1]);
            } else if constexpr (i > NPB) {
                gstep(std::integral_constant<int, i - NPB - 1>{}, T, P[PAR]);
            }
        });
    };
    static_assert(NPB == 42 && NBLK == 62, "the GEGLU micro-step placement above is written for 42 projection + 20 FF-out MFMAs per chunk");

    // Every load hipcc knows about (the x rows, the gate) has to be BACK before the first asm DMA of the main loop: its counted
    // vmcnt waits do not see those. And the x fragments have to be IN the AGPR half here, not copied there in front of their first
    // MFMA: a v_accvgpr_write needs wait states before an MFMA may read the register, and hipcc does not pad in front of an asm
    // statement (seen as lane-local NaNs -- whatever the previous kernel had left in those AGPRs -- that came and went between runs).
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0): the rows are back, this thread's bias words are in LDS (the first barrier publishes them)
    if constexpr (PRE) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(prv[cb][j].x), "+v"(prv[cb][j].y));
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(xf[ks]));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+a"(acc2[cb]));
    asm volatile("s_nop 7" : "+v"(xb));
    if constexpr (PRE) {
        // leading projection: t = pre_res + pre_gate (x Wpre^T + pre_b), written to mid_out, normalised into the feed-forward's operand
        proj_phase(acc2, xf, std::integral_constant<int, 0>{}, std::integral_constant<int, G::NBS / 4>{}, std::false_type{});
        // The feed-forward's residual is t itself. It rides in the accumulator -- Y^T starts at t / gate, so gate (Y^T + b2) is
        // t + gate (ff + b2) -- and t is neither stored nor read back; only when the gate is too small to divide by does t take the
        // way through mid_out (fp32 accumulation: the detour through t / gate costs ~2^-24 |t| per chunk, nothing next to bf16).
        float tv[NCB][16];
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(lpre + cb * 32 + j * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                U2BF4 r, o;
                r.u = prv[cb][j];
                const float bj[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.e[e] = f2bf(bf2f(r.e[e]) + pre_gate * (acc2[cb][4 * j + e] + bj[e]));
                    const float t_ = bf2f(o.e[e]);
                    tv[cb][4 * j + e] = t_;
                    s += t_;
                    ss = fmaf(t_, t_, ss);
                    acc2[cb][4 * j + e] = t_ * res_in_acc;   // (0 when the residual goes through mid_out)
                }
                if (res_in_acc == 0.f) *reinterpret_cast<uint2*>(p.mid_out + (size_t)row * p.ld_mid + cb * 32 + j * 8 + h * 4) = o.u;
            }
        }
        s += __shfl_xor(s, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float mean = s * (1.f / C);
        const float var = fmaxf(ss * (1.f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        to_fragments(tv, xf, rstd, -mean * rstd);
        __builtin_amdgcn_s_waitcnt(0x0F70);        // hipcc's loads and stores above are done before the next asm DMA (its waits do not count those)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(xf[ks]));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+a"(acc2[cb]));
        asm volatile("s_nop 7" : "+v"(xb));
    }
    chunk(std::true_type{}, std::integral_constant<int, 0>{});
#pragma unroll 1
    for (int it = 1; it + 1 < NCH; it += 2) {
        chunk(std::false_type{}, std::integral_constant<int, 1>{});
        chunk(std::false_type{}, std::integral_constant<int, 0>{});
    }
    chunk(std::false_type{}, std::integral_constant<int, 1>{});
    {
        // last stage (index NCH, even): the rest of the GEGLU of the last chunk, then its FF-out product; nothing left to fetch
        constexpr int PAR = NCH & 1, BUF = PAR ^ FB;
        f32x16 (&S)[2] = a1[PAR ^ 1];
        asm volatile("s_waitcnt vmcnt(0)");
        if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
        const unsigned ra = ra_st[BUF];
        dma_lds = lds_w + (BUF ? 0u : (unsigned)STAGE);
        bf16x8 fr[NR];
        static_for<RD>([&](auto ic) {
            constexpr int i = NPB + decltype(ic)::value;
            ffr_rd16<i * 1024>(fr[i % NR], ra);
        });
        static_for<37>([&](auto uc) { gstep(std::integral_constant<int, 19 + decltype(uc)::value>{}, S, P[PAR ^ 1]); });
        static_for<NBLK - NPB>([&](auto oc) {
            constexpr int i = NPB + decltype(oc)::value;
            constexpr int ob = i - NPB, nb = ob / NCB, cb = ob % NCB;
            if constexpr (ob % 4 == 0) {
                constexpr int last = (i + RD - 1 < NBLK - 1) ? i + RD - 1 : NBLK - 1;
                constexpr int need = (i + 3 < NBLK - 1) ? i + 3 : NBLK - 1;
                ffr_wait_lgkm<last - need>();
            }
            if constexpr (!(ABL & 16)) mfma_acc_aa(acc2[cb], fr[i % NR], P[PAR ^ 1][nb], gs);
            if constexpr (i + RD < NBLK) ffr_rd16<(i + RD) * 1024>(fr[(i + RD) % NR], ra);
            if constexpr (POST && ob % 2 == 1 && !(ABL & 2)) ffr_dma(rs, lane16, dma_lds, dma_off);   // the trailing projection's first stage: 10 blocks per wave
        });
        ffr_settle(acc2);
    }
    static_assert(!POST || (NBLK - NPB) / 2 == PJ_BLK / 4, "the last feed-forward stage fetches the trailing projection's first stage");

    // ---- epilogue: lane holds features 32 cb + 8 j + 4 h + e of its row. Copies under uniform branches
    // (a residual load behind a per-use guard is waited for in place)
    // y = res + gate (acc2 + b2); FINAL: stored to out (+ statistics); else kept (bf16-rounded) for the trailing projection
    // final_c: true / 1 = stored to out (+ statistics); false / 0 = kept (bf16-rounded) in yv for the trailing projection;
    // 2 = both, the row sums handed back in (sum_o, sq_o)
    float sum_o = 0.f, sq_o = 0.f;
    auto ff_result = [&](auto res_c, auto final_c, float (&yv)[NCB][16]) {
        constexpr bool RES = decltype(res_c)::value;
        constexpr int FMODE = (int)decltype(final_c)::value;
        constexpr bool FINAL = FMODE != 0, KEEP = FMODE != 1;
        const bf16* resp = PRE ? p.mid_out : p.res;
        const int ldr = PRE ? p.ld_mid : p.ldres;
        float s = 0.f, ss = 0.f;
        bf16* dst = p.out + (size_t)row * p.ldo + h * 4;
        uint2 rvv[RES ? NCB : 1][4];
        if constexpr (RES) {       // all residual loads in one block, one wait, then a loop that only stores
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j) rvv[cb][j] = *reinterpret_cast<const uint2*>(resp + (size_t)row * ldr + cb * 32 + j * 8 + h * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rvv[cb][j].x), "+v"(rvv[cb][j].y));
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(lb2 + cb * 32 + j * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc2[cb][4 * j] + bv[j].x, acc2[cb][4 * j + 1] + bv[j].y, acc2[cb][4 * j + 2] + bv[j].z, acc2[cb][4 * j + 3] + bv[j].w};
                if constexpr (RES) {
                    U2BF4 r;
                    r.u = rvv[cb][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(r.e[e]) + gate * v[e];
                } else if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gate;      // the residual was in the accumulator (t / gate)
                }
                U2BF4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.e[e] = f2bf(v[e]);
                    const float r = bf2f(o.e[e]);
                    if constexpr (FINAL) { s += r; ss = fmaf(r, r, ss); }
                    if constexpr (KEEP) { yv[cb][4 * j + e] = r; acc2[cb][4 * j + e] = 0.f; }
                }
                if constexpr (FINAL) *reinterpret_cast<uint2*>(dst + cb * 32 + j * 8) = o.u;
            }
        }
        if constexpr (FINAL) {
            s += __shfl_xor(s, 32, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (p.stats_out && h == 0) p.stats_out[(size_t)row * p.stats_ld] = make_float2(s, ss);
            sum_o = s;
            sq_o = ss;
        }
    };
    if constexpr (POST == 2) {
        // y = the feed-forward's result is stored AND kept; q = LN(y) Wq'^T + bq' (gamma / beta folded into Wq', bq') leaves in the
        // head layout [B*H][Tpad_q][DP]. The projection's first stage came in under the last feed-forward stage: it is waited for
        // here, in front of the store burst, and that stage then runs on its barrier alone
        float yv[NCB][16];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        using M2 = std::integral_constant<int, 2>;
        if (PRE ? res_in_acc == 0.f : p.res != nullptr) ff_result(std::true_type{}, M2{}, yv);
        else ff_result(std::false_type{}, M2{}, yv);
        const float mean = sum_o * (1.f / C);
        const float var = fmaxf(sq_o * (1.f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        to_fragments(yv, xf, rstd, -mean * rstd);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(xf[ks]));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+a"(acc2[cb]));
        asm volatile("s_nop 7" : "+v"(xb));
        proj_phase(acc2, xf, std::integral_constant<int, ((NCH & 1) ^ FB) ^ 1>{}, std::integral_constant<int, 0>{}, std::true_type{});
        constexpr int D = 40;
        static_assert(C % D == 0, "head geometry of the q epilogue");
        const int row0 = blockIdx.x * 128 + wave * 32;
        const int b = row0 / p.qT;
        const int tok = row0 - b * p.qT + (lane & 31);
        const size_t head_stride = (size_t)p.qTpad * p.qDP;
        bf16* base = p.q + (size_t)b * (C / D) * head_stride + (size_t)tok * p.qDP + 4 * h;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(lpost + cb * 32 + j * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f0 = cb * 32 + j * 8, hd = f0 / D, dd = f0 % D;
                U2BF4 o;
                o.e[0] = f2bf(acc2[cb][4 * j] + bv[j].x);
                o.e[1] = f2bf(acc2[cb][4 * j + 1] + bv[j].y);
                o.e[2] = f2bf(acc2[cb][4 * j + 2] + bv[j].z);
                o.e[3] = f2bf(acc2[cb][4 * j + 3] + bv[j].w);
                *reinterpret_cast<uint2*>(base + (size_t)hd * head_stride + dd) = o.u;
            }
        }
    } else if constexpr (!POST) {
        float unused[NCB][16];
        if (PRE ? res_in_acc == 0.f : p.res != nullptr) ff_result(std::true_type{}, std::true_type{}, unused);
        else ff_result(std::false_type{}, std::true_type{}, unused);
    } else {
        // trailing projection: out = post_res + (y Wpost^T + post_b)
        float yv[NCB][16];
        if (PRE ? res_in_acc == 0.f : p.res != nullptr) ff_result(std::true_type{}, std::false_type{}, yv);
        else ff_result(std::false_type{}, std::false_type{}, yv);
        to_fragments(yv, xf, 1.f, 0.f);
        // x_in of the trailing projection: fetched here, under the projection's MFMA stream (its stage waits are vmcnt(0) anyway)
        uint2 qrv[NCB][4];
        {
            const bf16* src = p.post_res + (size_t)row * p.ld_post_res + h * 4;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j) qrv[cb][j] = *reinterpret_cast<const uint2*>(src + cb * 32 + j * 8);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);        // (hipcc's loads above before the next asm DMA)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qrv[cb][j].x), "+v"(qrv[cb][j].y));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(xf[ks]));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+a"(acc2[cb]));
        asm volatile("s_nop 7" : "+v"(xb));
        proj_phase(acc2, xf, std::integral_constant<int, ((NCH & 1) ^ FB) ^ 1>{}, std::integral_constant<int, 0>{}, std::false_type{});
        float s = 0.f, ss = 0.f;
        bf16* dst = p.out + (size_t)row * p.ldo + h * 4;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(lpost + cb * 32 + j * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                U2BF4 r, o;
                r.u = qrv[cb][j];
                const float bj[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.e[e] = f2bf(bf2f(r.e[e]) + acc2[cb][4 * j + e] + bj[e]);
                    const float q_ = bf2f(o.e[e]);
                    s += q_;
                    ss = fmaf(q_, q_, ss);
                }
                *reinterpret_cast<uint2*>(dst + cb * 32 + j * 8) = o.u;
            }
        }
        if (p.stats_out) {
            s += __shfl_xor(s, 32, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (h == 0) p.stats_out[(size_t)row * p.stats_ld] = make_float2(s, ss);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Row-local q,k,v^T projection (round 6; reference ldm/modules/attention.py:167-186 SelfAttention.to_q / to_k / to_v behind
// BasicTransformerBlock.norm1 / GatedSelfAttentionDense.norm1, :366-368 SpatialTransformer.proj_in, :183-186 to_out + residual).
//
// As a tiled GEMM the fused q,k,v^T projection at the 64 x 64 level (M = 32768, N = 960, K = 320) stages both operands through
// the LDS-DMA once per 128 x 128 tile and runs at 0.40 PFLOP/s; the K = C projection in front of it (proj_in, or attn1.to_out with
// its residual) is one more launch and one more round trip of the residual stream. Here a wave keeps its 32 rows in registers
// as in ff_rows_kernel and runs the projections back to back:
//   PRE : t = pre_res + (x Wpre^T + pre_b)  -> mid_out (the residual stream), normalised in registers (LayerNorm without affine:
//         gamma / beta live in the folded q,k,v weights) -> operand of
//   q, k: features as MFMA rows (weights = A operand), so a lane holds 4 consecutive head dims of ONE token: the 8-byte store of the
//         row-major q layout and of the key-tile k layout (gemm.h ktile_off);
//   v^T : the SAME fragments with the operand roles exchanged (rows in the A slot, weights in the B slot -- both layouts are
//         lane = row / column, 8 k-slots per half), so a lane holds 16 TOKENS of one feature, contiguous under the 32-token
//         permutation of attn3_kernel (gemm.hip perm_tok4): two 16-byte stores per feature block.
// Weights: one stream of projection segments (5 stages x 40 blocks of 1 KiB each, k-step major), the main loop is proj_phase's
// fixed instruction order; the epilogues between the segments are hipcc's. A segment's first stage is fetched under the last
// stage of the segment before it and waited for BEFORE the epilogue's stores join the vmcnt counter, so the stores drain under
// the next segment's first MFMAs instead of in front of them.
#ifndef GL_QKV_ABL
#define GL_QKV_ABL 0      // developer ablation (tools/build_ffn_variant.sh -DGL_QKV_ABL=n): 1 no head stores, 2 no mid store, 4 no head MFMA streams
#endif
template <int C, int D, bool PRE, int NP>
__global__ void __launch_bounds__(256, 1) qkv_rows_kernel(QkvRowsParams p) {
    using G = FFGeom<C>;
    static_assert(C % D == 0 && D % 8 == 0, "head geometry");
    constexpr int KS = G::KS, NCB = G::NCB, PJ_BLK = G::PJ_BLK, PJ_ST = G::PJ_ST, PJ_KS = G::PJ_KS;
    constexpr int PSTAGE = PJ_BLK * 1024;
    constexpr int RD = 8, NR = RD + 4;
    constexpr int NSEG = (PRE ? 1 : 0) + NP;
    constexpr int NBIAS = NSEG * C;            // [pre_b][bias of q (, k, v)] as floats behind the three stage slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int row0 = blockIdx.x * 128 + wave * 32;     // the wave's first row (32 consecutive tokens of one sample: T % 32 == 0)
    const int row = row0 + l31;
    const unsigned lane16 = (unsigned)lane * 16u;

    u32x4 rs;
    {
        const unsigned long long a = (unsigned long long)(uintptr_t)p.stream;
        rs.x = (unsigned)a;
        rs.y = (unsigned)(a >> 32);
        rs.z = (unsigned)((size_t)NSEG * G::PJ_BYTES);
        rs.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned lds_w = lds0 + (unsigned)wave * 1024u;
    unsigned dma_lds = lds_w;
    unsigned dma_off = (unsigned)wave * 1024u;
#pragma unroll
    for (int i = 0; i < 2 * (PJ_BLK / 4); ++i) ffr_dma<true>(rs, lane16, dma_lds, dma_off);   // stages 0 and 1 (slots 0 and 1 are adjacent)

    // Every global LOAD of the kernel happens here, in front of the MFMA streams: the rows, the leading projection's residual rows
    // and the biases (into LDS). The epilogues between the streams then issue stores only -- a load among them would make hipcc
    // wait vmcnt(0) per feature block, i.e. for every store issued before it (measured: 65 us per launch instead of 3 x 5 + epilogues).
    bf16x8 xf[KS];
    {
        const bf16* src = p.x + (size_t)row * p.ldx + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
    }
    uint2 rv[PRE ? NCB : 1][4];
    const bool has_res = PRE && p.pre_res != nullptr;
    if constexpr (PRE) {
        if (has_res) {
            const bf16* src = p.pre_res + (size_t)row * p.ld_pre_res + h * 4;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[cb][j] = *reinterpret_cast<const uint2*>(src + cb * 32 + j * 8);
        } else {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[cb][j] = make_uint2(0u, 0u);
        }
    }
    float* lbias = reinterpret_cast<float*>(smem + 3 * PSTAGE);
    for (int i = t; i < NBIAS; i += 256) {
        float v;
        if (PRE && i < C) v = p.pre_b[i];
        else v = p.bias[i - (PRE ? C : 0)];
        lbias[i] = v;
    }
    if (!PRE && p.normalize) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = bf2f(xf[ks][e]);
                s += v;
                ss = fmaf(v, v, ss);
            }
        s += __shfl_xor(s, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float mean = s * (1.f / C);
        const float var = fmaxf(ss * (1.f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float nm = -mean * rstd;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[ks][e] = f2bf(fmaf(bf2f(xf[ks][e]), rstd, nm));
    }

    f32x16 acc[NCB];
    // Three LDS slots, the DMA two stages ahead: global stage g (segment-major, PJ_ST per segment) lives in slot g % 3 and is fetched
    // while stage g - 2 multiplies. At a segment boundary both of the next segment's first stages have landed BEFORE the epilogue
    // (the wait at the end of phase()), so those two stages run on barriers alone and the epilogue's stores -- which share the
    // vmcnt counter with the DMA and complete out of order with it -- get two stages of MFMAs to drain under.
    // (Measured and not kept: the next segment's THIRD stage fetched at the boundary as well, epilogue arithmetic first, then
    // vmcnt(0) and the 40 stores as one burst, three wait-free stages: 65 against 54-58 us -- the burst behind a wait is worse than
    // stores interleaved with their conversions.)
    const unsigned ra_st[3] = {lds0 + lane16, lds0 + (unsigned)PSTAGE + lane16, lds0 + 2u * (unsigned)PSTAGE + lane16};
    constexpr int NSTG = NSEG * PJ_ST;       // stages of the whole stream

    // One projection segment SEG: acc[cb] = sum over k-steps of W-block(ks, cb) x xin[ks] (SWAP: xin[ks] x W-block), first k-step with
    // C = 0. SKIP: its first two stages are known to have landed (the previous segment's closing wait)
    auto phase = [&](bf16x8 (&xin)[KS], auto seg_c, auto swap_c, auto skip_c) {
        constexpr int SEG = decltype(seg_c)::value;
        constexpr bool SWAP = decltype(swap_c)::value, SKIP = decltype(skip_c)::value;
        static_for<PJ_ST>([&](auto sc) {
            constexpr int st = decltype(sc)::value, g = SEG * PJ_ST + st, BUF = g % 3;
            constexpr int pieces = g + 2 < NSTG ? PJ_BLK / 4 : 0;
            if constexpr (!(SKIP && st < 2)) asm volatile("s_waitcnt vmcnt(0)");
            __builtin_amdgcn_s_barrier();
            const unsigned ra = ra_st[BUF];
            dma_lds = lds_w + (unsigned)((g + 2) % 3) * (unsigned)PSTAGE;
            bf16x8 fr[NR];
            static_for<RD>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ffr_rd16<i * 1024>(fr[i % NR], ra);
            });
            static_for<PJ_BLK>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 4 == 0) {
                    constexpr int last = (i + RD - 1 < PJ_BLK - 1) ? i + RD - 1 : PJ_BLK - 1;
                    constexpr int need = (i + 3 < PJ_BLK - 1) ? i + 3 : PJ_BLK - 1;
                    ffr_wait_lgkm<last - need>();
                }
                constexpr int ks = st * PJ_KS + i / NCB, cb = i % NCB;
                if constexpr (ks == 0) {
                    if constexpr (SWAP) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc[cb]) : "a"(xin[0]), "v"(fr[i % NR]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc[cb]) : "v"(fr[i % NR]), "a"(xin[0]));
                } else {
                    if constexpr (SWAP) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[cb]) : "a"(xin[ks]), "v"(fr[i % NR]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[cb]) : "v"(fr[i % NR]), "a"(xin[ks]));
                }
                if constexpr (i + RD < PJ_BLK) ffr_rd16<(i + RD) * 1024>(fr[(i + RD) % NR], ra);
                if constexpr (i % 2 == 1 && i / 2 < pieces) ffr_dma(rs, lane16, dma_lds, dma_off);
            });
        });
        ffr_settle(acc);
        if constexpr (SEG + 1 < NSEG) asm volatile("s_waitcnt vmcnt(0)");
    };
    auto pin_x = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(xf[ks]));
        asm volatile("s_nop 7\n\ts_nop 7");
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    using T_ = std::true_type;
    using F_ = std::false_type;

    // the rows, the residual rows and this thread's bias words are back (and in LDS) before the first asm DMA wait of the main loop;
    // the first stage's barrier publishes the biases
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0)
    if constexpr (PRE) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rv[cb][j].x), "+v"(rv[cb][j].y));
    }
    pin_x();
    if constexpr (PRE) {
        phase(xf, S0{}, F_{}, F_{});
        // t = pre_res + (x Wpre^T + pre_b), rounded to bf16: the residual stream. Lane holds features 32 cb + 8 j + 4 h + e of its row
        float tv[NCB][16];
        float s = 0.f, ss = 0.f;
        bf16* mrow = p.mid_out + (size_t)row * p.ld_mid + h * 4;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(lbias + cb * 32 + j * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                U2BF4 r, o;
                r.u = rv[cb][j];
                const float bj[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.e[e] = f2bf(acc[cb][4 * j + e] + bj[e] + bf2f(r.e[e]));
                    const float t_ = bf2f(o.e[e]);
                    tv[cb][4 * j + e] = t_;
                    s += t_;
                    ss = fmaf(t_, t_, ss);
                }
                if constexpr (GL_QKV_ABL & 2) asm volatile("" ::"v"(o.u.x), "v"(o.u.y));
                else *reinterpret_cast<uint2*>(mrow + cb * 32 + j * 8) = o.u;
            }
        }
        s += __shfl_xor(s, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (p.stats_out && h == 0) p.stats_out[row] = make_float2(s, ss);
        const float mean = s * (1.f / C);
        const float var = fmaxf(ss * (1.f / C) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float nm = -mean * rstd;
        // accumulator layout -> B / A operand fragments in the permuted k-slot order (ff_kidx, kperm)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[ks][e] = f2bf(fmaf(tv[ks >> 1][4 * (2 * (ks & 1) + (e >> 2)) + (e & 3)], rstd, nm));
        pin_x();
    }

    // ---- heads. b, token of the wave's rows: uniform sample, consecutive tokens
    const int b = row0 / p.T;
    const int tw = row0 - b * p.T;                 // multiple of 32
    const int tok = tw + l31;
    const float* hbias = lbias + (PRE ? C : 0);
    auto heads_qk = [&](auto which_c) {
        constexpr int WHICH = decltype(which_c)::value;
        const int tp = WHICH ? p.Tpad_k : p.Tpad_q;
        const size_t head_stride = (size_t)tp * p.DP;
        bf16* base = (WHICH ? p.k : p.q) + (size_t)b * p.H * head_stride +
                     (WHICH ? (size_t)(tok & ~63) * p.DP + (size_t)(tok & 63) * 8 + 4 * h : (size_t)tok * p.DP + 4 * h);
        const float* bias = hbias + WHICH * C + 4 * h;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + cb * 32 + j * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f0 = cb * 32 + j * 8, hd = f0 / D, dd = f0 % D;      // (compile-time after unrolling; + 4 h stays inside the head: D % 8 == 0)
                const float v[4] = {acc[cb][4 * j] + bv[j].x, acc[cb][4 * j + 1] + bv[j].y, acc[cb][4 * j + 2] + bv[j].z, acc[cb][4 * j + 3] + bv[j].w};
                U2BF4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.e[e] = f2bf(v[e]);
                bf16* dst = base + (size_t)hd * head_stride + (WHICH ? (size_t)(dd >> 3) * 512 : (size_t)dd);
                if constexpr (GL_QKV_ABL & 1) asm volatile("" ::"v"(o.u.x), "v"(o.u.y), "v"(dst));
                else *reinterpret_cast<uint2*>(dst) = o.u;
            }
        }
    };
    auto heads_vt = [&]() {
        // lane: feature 32 cb + l31, registers 4 j + e = token tw + 8 j + 4 h + e -> position tw + 16 h + 4 j + e of the permuted row
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int f = cb * 32 + l31, hd = f / D, dd = f - hd * D;
            const float bn = hbias[2 * C + f];
            bf16* dst = p.vt + ((size_t)(b * p.H + hd) * p.DPV + dd) * p.Tpad_k + tw + 16 * h;
            U4BF8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0.e[e] = f2bf(acc[cb][e] + bn);
                o1.e[e] = f2bf(acc[cb][8 + e] + bn);
            }
            if constexpr (GL_QKV_ABL & 1) asm volatile("" ::"v"(o0.u.x), "v"(o0.u.w), "v"(o1.u.x), "v"(o1.u.w), "v"(dst));
            else {
                *reinterpret_cast<uint4*>(dst) = o0.u;
                *reinterpret_cast<uint4*>(dst + 8) = o1.u;
            }
        }
    };
    if constexpr (NP == 1) {
        if constexpr (PRE) phase(xf, S1{}, F_{}, T_{}); else phase(xf, S0{}, F_{}, F_{});
        heads_qk(S0{});
    } else if constexpr ((GL_QKV_ABL & 4) != 0) {
        heads_qk(S0{});
        heads_qk(S1{});
        heads_vt();
    } else {
        if constexpr (PRE) phase(xf, S1{}, F_{}, T_{}); else phase(xf, S0{}, F_{}, F_{});
        heads_qk(S0{});
        if constexpr (PRE) phase(xf, S2{}, F_{}, T_{}); else phase(xf, S1{}, F_{}, T_{});
        heads_qk(S1{});
        if constexpr (PRE) phase(xf, S3{}, T_{}, T_{}); else phase(xf, S2{}, T_{}, T_{});
        heads_vt();
    }
}

template <int C>
__global__ void __launch_bounds__(256) qkv_rows_pack_kernel(const float* __restrict__ pre_w, const bf16* __restrict__ w, int np,
                                                            unsigned char* __restrict__ stream) {
    using G = FFGeom<C>;
    const size_t pre_bytes = pre_w ? G::PJ_BYTES : 0;
    const size_t n_pieces = (pre_bytes + (size_t)np * G::PJ_BYTES) / 16;
    for (size_t pc = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pc < n_pieces; pc += (size_t)gridDim.x * blockDim.x) {
        const size_t byte = pc * 16;
        const int L = (int)(byte >> 4) & 63, r = L & 31, h = L >> 5;
        const bool pre = byte < pre_bytes;
        const size_t sb = pre ? byte : byte - pre_bytes;
        const int seg = (int)(sb / G::PJ_BYTES);
        const int blk = (int)((sb % G::PJ_BYTES) >> 10);
        const int st = blk / G::PJ_BLK, i = blk % G::PJ_BLK, ks = st * G::PJ_KS + i / G::NCB, cb = i % G::NCB;
        U4BF8 o;
        if (pre) {
            const float* src = pre_w + (size_t)(cb * 32 + r) * C;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = f2bf(src[ff_kidx(ks, h, e, false)]);
        } else {
            // behind a leading projection the operand comes out of accumulator registers: permuted k-slots (ff_kidx)
            const bf16* src = w + (size_t)(seg * C + cb * 32 + r) * C;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = src[ff_kidx(ks, h, e, pre_w != nullptr)];
        }
        *reinterpret_cast<uint4*>(stream + byte) = o.u;
    }
}

}  // namespace

bool ff_rows_supported(int M, int C) { return C == 320 && M > 0 && M % 128 == 0; }

size_t ff_stream_bytes(int C) {
    if (C == 320) return FFGeom<320>::STREAM_BYTES;
    return 0;
}

size_t ff_chain_stream_bytes(int C, bool pre, bool post) {
    if (C == 320) return FFGeom<320>::chain_bytes(pre, post);
    return 0;
}

int ff_chain_pack_launch(const float* w1, const float* b1, const float* w2, const float* pre_w, const float* post_w, void* stream, int C, hipStream_t s) {
    if (C != 320) return set_error(GL_ERR_UNSUPPORTED, "ff_pack: no row-local feed-forward kernel for C = %d", C);
    if (post_w && !pre_w) return set_error(GL_ERR_UNSUPPORTED, "ff_pack: a trailing projection is built only together with a leading one");
    hipLaunchKernelGGL(ff_pack_kernel<320>, dim3(1024), dim3(256), 0, s, w1, b1, w2, pre_w, post_w, reinterpret_cast<unsigned char*>(stream));
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int ff_pack_launch(const float* w1, const float* b1, const float* w2, void* stream, int C, hipStream_t s) {
    return ff_chain_pack_launch(w1, b1, w2, nullptr, nullptr, stream, C, s);
}

#ifdef GL_FFN_ABLATE
static int g_ff_abl = 0;
void ff_rows_set_ablation(int abl) { g_ff_abl = abl; }
template <int ABL>
static int ff_abl_launch(const FFRowsParams& p, hipStream_t s) {
    using G = FFGeom<320>;
    auto kfn = ff_rows_kernel<320, ABL>;
    GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(kfn, dim3(p.M / 128), dim3(256), G::LDS_BYTES, s, p);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
#endif

int ff_rows_launch(const FFRowsParams& p, int C, hipStream_t s) {
#ifdef GL_FFN_ABLATE
    switch (g_ff_abl) {
        case 1: return ff_abl_launch<1>(p, s);
        case 2: return ff_abl_launch<2>(p, s);
        case 4: return ff_abl_launch<4>(p, s);
        case 8: return ff_abl_launch<8>(p, s);
        case 16: return ff_abl_launch<16>(p, s);
        case 24: return ff_abl_launch<24>(p, s);
        case 32: return ff_abl_launch<32>(p, s);
        case 3: return ff_abl_launch<3>(p, s);
        case 7: return ff_abl_launch<7>(p, s);
        case 6: return ff_abl_launch<6>(p, s);
        case 25: return ff_abl_launch<25>(p, s);
        case 31: return ff_abl_launch<31>(p, s);
        default: break;
    }
#endif
    if (!ff_rows_supported(p.M, C)) return set_error(GL_ERR_UNSUPPORTED, "ff_rows: M = %d, C = %d has no row-local kernel", p.M, C);
    if (p.ldx % 8 || p.ldo % 4 || (p.res && p.ldres % 4)) return set_error(GL_ERR_ARG, "ff_rows: row strides must keep 16-byte loads / 8-byte stores aligned");
    using G = FFGeom<320>;
    if (p.post && !p.pre) return set_error(GL_ERR_UNSUPPORTED, "ff_rows: a trailing projection is built only together with a leading one");
    if (p.post == 2 && (!p.post_b || !p.q || p.qT <= 0 || p.M % p.qT || p.qT % 32 || p.qDP < 40 || p.qDP % 4))
        return set_error(GL_ERR_ARG, "ff_rows: the trailing to_q projection needs its folded bias, the q buffer and its geometry");
    if (p.pre && (!p.normalize || !p.pre_b || !p.pre_res || !p.mid_out || p.ld_pre_res % 4 || p.ld_mid % 4))
        return set_error(GL_ERR_ARG, "ff_rows: the leading projection needs normalize = 1, its bias, residual and the buffer for its result");
    if (p.post == 1 && (!p.post_b || !p.post_res || p.ld_post_res % 4)) return set_error(GL_ERR_ARG, "ff_rows: the trailing projection needs its bias and residual");
    // once per process and kernel; never inside a stream capture (the first call of each form is an eager one)
#define FF_LAUNCH(PRE_, POST_)                                                                                               \
    do {                                                                                                                     \
        auto kfn = ff_rows_kernel<320, 0, PRE_, POST_>;                                                                      \
        static bool attr_done = false;                                                                                       \
        if (!attr_done) {                                                                                                    \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));         \
            attr_done = true;                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL(kfn, dim3(p.M / 128), dim3(256), G::LDS_BYTES, s, p);                                             \
    } while (0)
    if (p.pre && p.post == 2) FF_LAUNCH(true, 2);
    else if (p.pre && p.post) FF_LAUNCH(true, 1);
    else if (p.pre) FF_LAUNCH(true, 0);
    else FF_LAUNCH(false, 0);
#undef FF_LAUNCH
    GL_LAUNCH_CHECK();
    return GL_OK;
}


bool qkv_rows_supported(int M, int C, int d, int T) { return C == 320 && d == 40 && M > 0 && M % 128 == 0 && T > 0 && T % 128 == 0 && M % T == 0; }

size_t qkv_rows_stream_bytes(int C, bool pre, int np) {
    if (C != 320 || (np != 1 && np != 3)) return 0;
    return ((pre ? 1 : 0) + (size_t)np) * FFGeom<320>::PJ_BYTES;
}

int qkv_rows_pack_launch(const float* pre_w, const bf16* w, int np, void* stream, int C, hipStream_t s) {
    if (C != 320 || (np != 1 && np != 3) || !w || !stream) return set_error(GL_ERR_UNSUPPORTED, "qkv_rows_pack: C = %d, np = %d has no row-local projection kernel", C, np);
    hipLaunchKernelGGL(qkv_rows_pack_kernel<320>, dim3(512), dim3(256), 0, s, pre_w, w, np, reinterpret_cast<unsigned char*>(stream));
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int qkv_rows_launch(const QkvRowsParams& p, int C, hipStream_t s) {
    if (!qkv_rows_supported(p.M, C, p.d, p.T)) return set_error(GL_ERR_UNSUPPORTED, "qkv_rows: M = %d, C = %d, d = %d, T = %d has no row-local projection kernel", p.M, C, p.d, p.T);
    if (p.np != 1 && p.np != 3) return set_error(GL_ERR_ARG, "qkv_rows: np must be 1 (q) or 3 (q, k, v^T)");
    if (p.ldx % 8 || !p.x || !p.stream || !p.q || !p.bias || (p.np == 3 && (!p.k || !p.vt))) return set_error(GL_ERR_ARG, "qkv_rows: null pointer / unaligned rows");
    if (p.np == 3 && (p.vt_perm32 != 1 || p.Tpad_k % 32)) return set_error(GL_ERR_UNSUPPORTED, "qkv_rows: v^T is written in the 32-token form only (attn3_kernel)");
    if (p.H * p.d != C || p.DP < p.d || p.DP % 8 || (p.np == 3 && p.DPV < p.d)) return set_error(GL_ERR_ARG, "qkv_rows: head geometry");
    if (p.pre && (!p.pre_b || !p.mid_out || p.ld_mid % 4 || (p.pre_res && p.ld_pre_res % 4))) return set_error(GL_ERR_ARG, "qkv_rows: the leading projection needs its bias and the buffer for its result");
    using G = FFGeom<320>;
    constexpr int LDS = 3 * G::PJ_BLK * 1024 + 4 * 320 * (int)sizeof(float);
#define QKV_LAUNCH(PRE_, NP_)                                                                                                \
    do {                                                                                                                     \
        auto kfn = qkv_rows_kernel<320, 40, PRE_, NP_>;                                                                      \
        static bool attr_done = false;                                                                                       \
        if (!attr_done) {                                                                                                    \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));                  \
            attr_done = true;                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL(kfn, dim3(p.M / 128), dim3(256), LDS, s, p);                                                      \
    } while (0)
    if (p.pre && p.np == 3) QKV_LAUNCH(true, 3);
    else if (p.pre) QKV_LAUNCH(true, 1);
    else if (p.np == 3) QKV_LAUNCH(false, 3);
    else QKV_LAUNCH(false, 1);
#undef QKV_LAUNCH
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace gl
