// Row-local feed-forward for gfx950 (CDNA4, wave64): one launch for
//     y = res + gate * ( GEGLU( LN(x) W1^T + b1 ) W2^T + b2 )            (+ row statistics of y)
// i.e. the reference's FeedForward with the LayerNorm in front of it and the residual behind it
// (ldm/modules/attention.py:37-64 GEGLU / FeedForward, :333-338 BasicTransformerBlock._forward, :236-244 GatedSelfAttentionDense).
//
// Why: as two GEMMs the pair is bound by operand delivery, not by the matrix cores -- the 256 x 128 tiles of gemm_wide_kernel stage
// one byte through the LDS-DMA per 85 FLOP, the hidden [M][4C] activation goes out to HBM and comes back (2 x 84 MB at the 64 x 64
// level), and the erf-GEGLU epilogue of a K = 320 projection is as long as its K loop. Here a WAVE owns 32 complete rows from the
// first instruction to the last:
//   * its x rows live in registers as MFMA B-operand fragments (80 VGPRs at C = 320), normalised in place;
//   * per chunk of 32 hidden features:  S^T[64 x 32 rows] = W1c X^T (80 MFMAs), bias preloaded into the accumulators, erf-GEGLU on
//     the accumulators in registers, and the packed result IS the B operand of  Y^T[C x 32 rows] += W2c P^T  (40 MFMAs) --
//     v_mfma_f32_16x16x32_bf16 with the weights as the A operand leaves a lane with 4 consecutive features of ONE row, two such
//     accumulators are the 8 k-slots a lane supplies to the next product, and the pack kernel permutes W2's columns to match
//     (the S^T -> P^T trick of attention.hip);
//   * the FF-out accumulator Y^T (160 registers) stays in the AGPR half until the epilogue adds bias, gate and residual.
// So nothing but weights goes through LDS: one byte per 128 FLOP (4 waves x 32 rows share every weight fragment), no activation
// staging, no LDS hop for the hidden tile, no cross-wave data at all. The price is one wave per SIMD at up to 512 registers;
// latency is hidden inside the wave by a 2-group-deep fragment prefetch and by the GEGLU of chunk c-1 riding under the
// projection MFMAs of chunk c.
//
// The weights are a STREAM of 1 KiB fragment blocks in exactly the order the waves consume them (ff_pack_kernel): a block is one
// MFMA A operand in lane order, so an LDS-DMA instruction copies it verbatim and a fragment read is ds_read_b128 at lane * 16 --
// conflict free without any swizzle. Stage c = [W1 of chunk c : 4 KS blocks][W2 of chunk c-1 : C/16 blocks]; two stages in LDS,
// one barrier per chunk: vmcnt(0) -> barrier -> DMA of stage c+1 (spread over the chunk's MFMA groups) -> compute stage c.
#include "ffn.h"

#include <type_traits>
#include <utility>

namespace gl {

namespace {

template <int C>
struct FFGeom {
    static_assert(C % 64 == 0, "C must be a multiple of 64");
    static constexpr int KS = C / 32;            // k-steps of the GEGLU projection (K = C)
    static constexpr int NCF = C / 16;           // output-feature fragments of FF-out
    static_assert(NCF % 4 == 0, "FF-out fragments are read in groups of 4");
    static constexpr int NCH = 4 * C / 32;       // hidden chunks of 32 features
    static constexpr int NB = 4 * KS + NCF;      // 1 KiB blocks per stage
    static_assert(NB % 4 == 0, "every wave copies NB / 4 blocks of a stage");
    static constexpr int STAGE = NB * 1024;
    static_assert(STAGE < 65536, "fragment reads address a stage through the 16-bit offset field");
    static constexpr int NSTAGE = NCH + 1;       // stage c = [W1(c)][W2(c-1)], c = 0 .. NCH
    // b1 of every chunk as [chunk][nf][16] fp32, resident in LDS (+ one zero chunk: the last chunk prefetches "the next" bias)
    static constexpr int BIAS_BYTES = ((NCH + 1) * 256 + 1023) / 1024 * 1024;
    static constexpr size_t STREAM_BYTES = (size_t)NSTAGE * STAGE + BIAS_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE + BIAS_BYTES;
    static constexpr int NG = KS + NCF / 4;      // fragment groups of 4 per stage: KS projection k-steps, then NCF / 4 FF-out groups
};

// original GEGLU.proj row of fragment row r (0..15) of fragment nf (0: value, 1: gate, 2: value + 16, 3: gate + 16) of chunk c
__device__ __forceinline__ int w1_row(int C, int c, int nf, int r) { return (nf & 1) * 4 * C + c * 32 + (nf >> 1) * 16 + r; }

template <int C>
__global__ void __launch_bounds__(256) ff_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                      unsigned char* __restrict__ stream) {
    using G = FFGeom<C>;
    const size_t n_pieces = G::STREAM_BYTES / 16;
    for (size_t pc = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pc < n_pieces; pc += (size_t)gridDim.x * blockDim.x) {
        U4BF8 o;
        o.u = make_uint4(0, 0, 0, 0);
        const size_t byte = pc * 16;
        if (byte >= (size_t)G::NSTAGE * G::STAGE) {   // bias table
            const int f0 = (int)((byte - (size_t)G::NSTAGE * G::STAGE) / 4);   // first of 4 floats
            const int c = f0 >> 6, nf = (f0 >> 4) & 3, r = f0 & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < G::NCH) {
            v.x = b1[w1_row(C, c, nf, r)]; v.y = b1[w1_row(C, c, nf, r + 1)]; v.z = b1[w1_row(C, c, nf, r + 2)]; v.w = b1[w1_row(C, c, nf, r + 3)];
            }
            *reinterpret_cast<float4*>(stream + byte) = v;
            continue;
        }
        const int st = (int)(byte / G::STAGE);
        const int in = (int)(byte % G::STAGE);
        const int blk = in >> 10, L = (in >> 4) & 63, l15 = L & 15, q = L >> 4;
        if (blk < 4 * G::KS) {
            const int ks = blk >> 2, nf = blk & 3;
            if (st < G::NCH) {
                const float* src = w1 + (size_t)w1_row(C, st, nf, l15) * C + ks * 32 + q * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = f2bf(src[e]);
            }
        } else {
            const int cf = blk - 4 * G::KS, c = st - 1;
            if (c >= 0) {
                // k-slot 8 q + e of the P^T operand holds hidden feature 4 q + e (e < 4: from the first value / gate accumulator pair)
                // or 16 + 4 q + (e - 4) (from the second pair)
                const float* src = w2 + (size_t)(cf * 16 + l15) * 4 * C + c * 32;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = f2bf(src[e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4)]);
            }
        }
        *reinterpret_cast<uint4*>(stream + byte) = o.u;
    }
}

template <int OFF>
__device__ __forceinline__ void ffr_rd16(bf16x8& d, unsigned addr) {
    // (no "memory" clobber: an asm that may touch memory makes hipcc's waitcnt pass drain the pending LDS-DMA, gemm.hip)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void ffr_pin(bf16x8 (&d)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(d[i]));
}
template <int N>
__device__ __forceinline__ void ffr_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define FFR_BLDS16(rsrc, ldst, voff, soff)                                                                 \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(ldst), 16, \
                                             (int)(voff), (int)(soff), 0, 0)

// ABL: developer ablation bits (tools/gpu_r4b.sh; only ABL = 0 is in the product library): 1 no GEGLU arithmetic, 2 no DMA behind the
// first two stages, 4 no fragment reads behind each chunk's first two groups, 8 no projection MFMAs, 16 no FF-out MFMAs, 32 no barrier
template <int C, int ABL = 0>
__global__ void __launch_bounds__(256, 1) ff_rows_kernel(FFRowsParams p) {
    using G = FFGeom<C>;
    constexpr int KS = G::KS, NCF = G::NCF, NCH = G::NCH, STAGE = G::STAGE, NG = G::NG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15;
    const int q = lane >> 4;
    const int m0 = blockIdx.x * 128 + wave * 32;      // this wave's 32 rows
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- the wave's rows as B-operand fragments: lane (l15, q) holds x[m0 + 16 rb + l15][32 ks + 8 q .. + 7]
    bf16x8 xf[2][KS];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const bf16* src = p.x + (size_t)(m0 + rb * 16 + l15) * p.ldx + q * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[rb][ks] = *reinterpret_cast<const bf16x8*>(src + ks * 32);
    }
    const float gate = (p.res && p.gate) ? *p.gate : 1.f;

    // ---- weights: stage 0 and the bias table on their way while the rows are normalised
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stream), 0, (int)G::STREAM_BYTES, 0x00020000);
    auto dma_piece = [&](int st, int piece) {     // this wave's piece-th block of stage st (a wave copies blocks wave, wave + 4, ...)
        const int blk = wave + 4 * piece;
        if ((ABL & 2) && st > 1) return;
        FFR_BLDS16(rs, smem + (st & 1) * STAGE + blk * 1024, lane16, st * STAGE + blk * 1024);
    };
#pragma unroll
    for (int i = 0; i < G::NB / 4; ++i) dma_piece(0, i);
#pragma unroll
    for (int i = 0; i < (G::BIAS_BYTES / 1024 + 3) / 4; ++i) {
        const int blk = wave + 4 * i;
        if (blk < G::BIAS_BYTES / 1024) FFR_BLDS16(rs, smem + 2 * STAGE + blk * 1024, lane16, G::NSTAGE * STAGE + blk * 1024);
    }

    if (p.normalize) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = bf2f(xf[rb][ks][e]);
                    s += v;
                    ss = fmaf(v, v, ss);
                }
            s += __shfl_xor(s, 16, 64); ss += __shfl_xor(ss, 16, 64);
            s += __shfl_xor(s, 32, 64); ss += __shfl_xor(ss, 32, 64);
            const float mean = s * (1.f / C);
            const float var = fmaxf(ss * (1.f / C) - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.eps);
            const float nm = -mean * rstd;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[rb][ks][e] = f2bf(fmaf(bf2f(xf[rb][ks][e]), rstd, nm));
        }
    }

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned rbias = lds0 + 2 * STAGE + (unsigned)q * 16u;   // this lane's float4 of fragment 0 of chunk 0; + 64 per fragment, + 256 per chunk

    f32x4 acc2[NCF][2];
#pragma unroll
    for (int cf = 0; cf < NCF; ++cf) acc2[cf][0] = acc2[cf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two projection accumulator sets: chunk c multiplies into a1[c & 1] while the GEGLU of chunk c - 1 reads a1[(c - 1) & 1]
    f32x4 a1[2][4][2];
    float hv[2][8];        // GEGLU outputs of the chunk being activated, per row block, in k-slot order of the P^T operand
    bf16x8 P[2];

    // GEGLU call k (0..7) of a chunk: row block k >> 2, half hf = (k >> 1) & 1 (value = fragment 2 hf, gate = fragment 2 hf + 1),
    // element pair pr = k & 1
    auto geglu_call = [&](auto kc, f32x4 (&src)[4][2]) {
        constexpr int k = decltype(kc)::value, rb = k >> 2, hf = (k >> 1) & 1, pr = k & 1;
        if constexpr (ABL & 1) {
            hv[rb][4 * hf + 2 * pr] = src[2 * hf][rb][2 * pr] + src[2 * hf + 1][rb][2 * pr];
            hv[rb][4 * hf + 2 * pr + 1] = src[2 * hf][rb][2 * pr + 1] + src[2 * hf + 1][rb][2 * pr + 1];
            return;
        }
        const f32x2 r = geglu2(f32x2{src[2 * hf][rb][2 * pr], src[2 * hf][rb][2 * pr + 1]},
                               f32x2{src[2 * hf + 1][rb][2 * pr], src[2 * hf + 1][rb][2 * pr + 1]});
        hv[rb][4 * hf + 2 * pr] = r.x;
        hv[rb][4 * hf + 2 * pr + 1] = r.y;
    };
    auto pack_p = [&](int rb) {
#pragma unroll
        for (int e = 0; e < 8; ++e) P[rb][e] = f2bf(hv[rb][e]);
    };

    // One chunk `it` (PAR = it & 1). Its 15 fragment groups (10 projection k-steps, 5 FF-out groups of 4 fragments) each carry 8
    // MFMAs; the VALU work is spread over them so that no group is VALU-only:
    //   g = 0,2,4,6,8 : GEGLU calls 3..7 of chunk it - 1 (source a1[PAR ^ 1])     g = 1 / 9 : pack P[0] / P[1]
    //   g = 10,12,14  : GEGLU calls 0..2 of chunk it (source a1[PAR], complete after g = 9), carried to the next chunk in hv
    //   g = 9         : bias fragments of chunk it + 1 requested        g = 11, 13 : a1[PAR ^ 1] <- bias of chunk it + 1
    // MODE 0: the first chunk (no previous chunk: no FF-out product, calls 3..7 and packs skipped).
    auto chunk = [&](auto mode_c, auto par_c, int it) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool OUT = MODE != 0;
        f32x4 (&T)[4][2] = a1[PAR];
        f32x4 (&S)[4][2] = a1[PAR ^ 1];
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's pieces of the stage have landed (hipcc then knows nothing is pending)
        if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();   // ... and everybody's; every wave is done reading the other stage
        asm volatile("" ::: "memory");
        const unsigned ra = lds0 + (unsigned)(it & 1) * STAGE + lane16;

        bf16x8 fr[3][4];                           // fragment ring: group g lives in fr[g % 3], two groups in flight ahead of the multiply
        bf16x8 bfr[4];                             // the next chunk's projection bias (4 x float4 per lane)
        auto issue = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ffr_rd16<(g * 4 + i) * 1024>(fr[g % 3][i], ra);
            });
        };
        auto issue_bias = [&]() {
            static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ffr_rd16<i * 64>(bfr[i], rbias);
            });
            rbias += 256;
        };
        if constexpr (MODE == 0) {                 // the first chunk initialises its own accumulators
            issue_bias();
            ffr_wait_lgkm<0>();
            ffr_pin(bfr);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) T[nf][0] = T[nf][1] = __builtin_bit_cast(f32x4, bfr[nf]);
        }
        constexpr int G1 = OUT ? NG : KS;          // fragment groups this chunk reads
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);

        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            dma_piece(it + 1, g);                  // next stage, one block per wave and group
            if constexpr (g == KS - 1) issue_bias();
            if constexpr (g + 2 < G1 && !(ABL & 4)) issue(std::integral_constant<int, g + 2>{});
            constexpr bool heavy = (g % 2 == 0) && (g >= KS || OUT);   // a GEGLU call rides in this group
            if constexpr (g < G1) {
                // reads issued behind group g's: groups g + 1, g + 2 (where they exist) and, around g = KS - 1, the bias fragments
                constexpr int younger = 4 * ((G1 - 1 - g) < 2 ? (G1 - 1 - g) : 2) + ((g == KS - 1 || g == KS) ? 4 : 0);
                if constexpr (ABL & 4) ffr_wait_lgkm<0>();
                else ffr_wait_lgkm<younger>();
                ffr_pin(fr[g % 3]);
                if constexpr (heavy) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA,
                        __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);   // then a piece of the GEGLU (VALU / transcendental)
                    }
                }
                if constexpr (g < KS) {
                    // projection k-step g: S^T[4 fragments][2 row blocks] += W1 fragment x X fragment
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
                            if constexpr (!(ABL & 8)) T[nf][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][nf], xf[rb][g], T[nf][rb], 0, 0, 0);
                } else {
                    // FF-out group: Y^T[4 fragments][2 row blocks] += W2 fragment x P^T
                    constexpr int j = g - KS;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
                            if constexpr (!(ABL & 16)) acc2[4 * j + i][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][i], P[rb], acc2[4 * j + i][rb], 0, 0, 0);
                }
            } else if constexpr (g == KS) {
                ffr_wait_lgkm<0>();                // (first chunk: only the bias fragments are outstanding)
            }
            if constexpr (g < KS) {
                if constexpr (OUT && g % 2 == 0) geglu_call(std::integral_constant<int, 3 + g / 2>{}, S);
                if constexpr (OUT && g == 1) pack_p(0);
                if constexpr (OUT && g == KS - 1) pack_p(1);
            } else {
                if constexpr ((g - KS) % 2 == 0 && (g - KS) / 2 < 3) geglu_call(std::integral_constant<int, (g - KS) / 2>{}, T);
                if constexpr (g == KS + 1 || g == KS + 3) {
                    if constexpr (g == KS + 1) ffr_pin(bfr);
                    constexpr int n0 = g == KS + 1 ? 0 : 2;
#pragma unroll
                    for (int nf = n0; nf < n0 + 2; ++nf) S[nf][0] = S[nf][1] = __builtin_bit_cast(f32x4, bfr[nf]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    static_assert(KS == 10 && NG == 15, "the VALU placement above is written for 10 projection k-steps + 5 FF-out groups");
    static_assert(NCH % 2 == 0, "chunk parity of the last chunk");
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): scalar loads done (else hipcc drains lgkmcnt inside the loop, behind the fragment reads)
    chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
    for (int it = 1; it + 1 < NCH; it += 2) {
        chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, it);
        chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, it + 1);
    }
    chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, NCH - 1);
    {
        // last stage: the rest of the GEGLU of the last chunk (calls 0..2 ran behind its projection), then its FF-out product
        f32x4 (&S)[4][2] = a1[(NCH - 1) & 1];
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned ra = lds0 + (unsigned)(NCH & 1) * STAGE + lane16;
        static_for<5>([&](auto kc) { geglu_call(std::integral_constant<int, 3 + decltype(kc)::value>{}, S); });
        pack_p(0);
        pack_p(1);
        static_for<NCF / 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            bf16x8 fr[4];
            static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ffr_rd16<((KS + j) * 4 + i) * 1024>(fr[i], ra);
            });
            ffr_wait_lgkm<0>();
            ffr_pin(fr);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc2[4 * j + i][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i], P[rb], acc2[4 * j + i][rb], 0, 0, 0);
        });
    }

    // ---- epilogue: lane (l15, q) holds features 16 cf + 4 q .. + 3 of row m0 + 16 rb + l15. Two copies under one uniform branch
    // (a residual load behind a per-use guard is waited for in place)
    float4 b2v[NCF];
#pragma unroll
    for (int cf = 0; cf < NCF; ++cf) b2v[cf] = *reinterpret_cast<const float4*>(p.b2 + cf * 16 + q * 4);
    auto epilogue = [&](auto res_c) {
        constexpr bool RES = decltype(res_c)::value;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = m0 + rb * 16 + l15;
            uint2 rv[NCF];
            if constexpr (RES) {
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf) rv[cf] = *reinterpret_cast<const uint2*>(p.res + (size_t)row * p.ldres + cf * 16 + q * 4);
            }
            float s = 0.f, ss = 0.f;
            bf16* dst = p.out + (size_t)row * p.ldo + q * 4;
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) {
                float v[4] = {acc2[cf][rb][0] + b2v[cf].x, acc2[cf][rb][1] + b2v[cf].y, acc2[cf][rb][2] + b2v[cf].z, acc2[cf][rb][3] + b2v[cf].w};
                if constexpr (RES) {
                    U2BF4 r;
                    r.u = rv[cf];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(r.e[e]) + gate * v[e];
                }
                U2BF4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.e[e] = f2bf(v[e]);
                    const float r = bf2f(o.e[e]);
                    s += r;
                    ss = fmaf(r, r, ss);
                }
                *reinterpret_cast<uint2*>(dst + cf * 16) = o.u;
            }
            if (p.stats_out) {
                s += __shfl_xor(s, 16, 64); ss += __shfl_xor(ss, 16, 64);
                s += __shfl_xor(s, 32, 64); ss += __shfl_xor(ss, 32, 64);
                if (q == 0) p.stats_out[(size_t)row * p.stats_ld] = make_float2(s, ss);
            }
        }
    };
    if (p.res) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

}  // namespace

bool ff_rows_supported(int M, int C) { return C == 320 && M > 0 && M % 128 == 0; }

size_t ff_stream_bytes(int C) {
    if (C == 320) return FFGeom<320>::STREAM_BYTES;
    return 0;
}

int ff_pack_launch(const float* w1, const float* b1, const float* w2, void* stream, int C, hipStream_t s) {
    if (C != 320) return set_error(GL_ERR_UNSUPPORTED, "ff_pack: no row-local feed-forward kernel for C = %d", C);
    hipLaunchKernelGGL(ff_pack_kernel<320>, dim3(1024), dim3(256), 0, s, w1, b1, w2, reinterpret_cast<unsigned char*>(stream));
    GL_LAUNCH_CHECK();
    return GL_OK;
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
    auto kfn = ff_rows_kernel<320>;
    static bool attr_done = false;   // once per process; never inside a stream capture (the first call is an eager one)
    if (!attr_done) {
        GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        attr_done = true;
    }
    hipLaunchKernelGGL(kfn, dim3(p.M / 128), dim3(256), G::LDS_BYTES, s, p);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace gl
