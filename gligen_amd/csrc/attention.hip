// Fused (flash-style) multi-head attention forward for gfx950, bf16 in / fp32 softmax+accumulate.
//
// Replaces the materialised-softmax attention of the reference:
//   SelfAttention.forward   ldm/modules/attention.py:167-186   (attn1 and the GLIGEN fuser)
//   CrossAttention.forward  ldm/modules/attention.py:127-149   (attn2, 77 CLIP tokens)
// sim = (q k^T) * d^-0.5 ; softmax over keys ; out = attn v.   No mask, no dropout.
//
// Layouts (written by the projection GEMM epilogues, gemm.hip EPI_QK_HEADS / EPI_VT_HEADS):
//   q  [B*H][Tq_pad][DP]   head dim zero-padded to DP (48 / 80 / 160)
//   k  [B*H][Tk_pad/64][DP/8][64][8]   key-tile layout (gemm.h ktile_off): a 64-key tile is one contiguous block, 16-byte
//                          chunks chunk-major; for d = 40 column 40 holds 1.0 (attn_k_init_launch), see BIAS below
//   vt [B*H][DPV][Tk_pad]  V transposed, tokens permuted inside groups of 16 as
//                          [0-3, 8-11, 4-7, 12-15]  (DPV = 64 / 96 / 160)
//   o  [B][rows][ldo]      token-major, head h at columns [h*d, (h+1)*d)
//
// Per wave: 32 query rows, "swapped" products so the softmax axis is lane-local
//   S^T = K Q^T  (v_mfma_f32_32x32x16_bf16: A = K rows, B = Q rows): lane (q = lane&31) holds
//                16 keys per 32-key sub-tile; the other 16 sit in lane^32
//   O^T = V^T P^T: the S^T accumulator registers, converted to bf16 8 at a time, ARE the B
//                fragment (the key permutation above makes the matching V^T fragment one
//                ds_read_b128); online-softmax rescale factors are per lane.
// A 256-thread block (4 waves = 128 query rows) streams 64-key tiles of K and V^T through a
// 2-stage LDS ring (register-staged: loads of tile t+1 are in flight during the MFMAs of tile t).
#include "attention.h"
#include "gemm.h"

#include <cstdlib>
#include <type_traits>

namespace gl {

// max over the two lanes l and l^32 without the LDS round trip of a bpermute (gfx950 v_permlane32_swap)
__device__ __forceinline__ float max_xor32(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ONES (DPV > d): the first padding row of V^T (row d) holds 1.0 for every key, so the P V product
// itself accumulates the softmax denominator in O^T row d (same bf16 P as the numerator) and the
// per-lane VALU row sum disappears; see attn_vt_ones_launch.
//
// BIAS (DP > d, i.e. d = 40): the first padding column of the head dim carries the softmax stabiliser through the MFMA.
// Q is scaled by scale*log2(e) when its fragments are loaded, Q[:, d] := -m (the running row max, kept bf16-exact) and
// K[:, d] := 1 as the tile is written to LDS, so K Q^T comes out of the matrix core as s*c - m and the per-element
// v_fma in front of every v_exp disappears (32 of ~105 VALU instructions per 64-key tile; the kernel is bound by VALU
// + MFMA issue, which barely overlap here). m only has to be close to the row max, not equal to it: it is moved (and O^T
// rescaled) when a tile's scores exceed it by more than 2^6, which after the first tiles is rare.
template <int DP, int DPV, bool ONES>
__global__ void __launch_bounds__(256, DP == 48 ? 3 : 1) attn_kernel(AttnParams P) {  // d = 40: the bound trims the kernel to 124 VGPRs = four waves per SIMD
    constexpr bool BIAS = DP == 48 && ONES;   // d = 40 (column 40 is free)
    constexpr int KS = DP / 16;
    constexpr int DT = DPV / 32;
    constexpr int KROW = DP * 2 + 16;  // LDS row strides (bytes): +16 keeps b128 reads conflict-free
    constexpr int VROW = 64 * 2 + 16;
    constexpr int KBYTES = 64 * KROW;
    constexpr int VBYTES = DPV * VROW;
    constexpr int STAGE = KBYTES + VBYTES;
    constexpr int KCH = 64 * DP / 8;  // 16-byte chunks per K tile
    constexpr int NKC = (KCH + 255) / 256;
    constexpr int NVC = DPV * 8 / 256;
    static_assert(DPV * 8 % 256 == 0, "V tile must split evenly");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int lrow = lane & 31;
    const int half = lane >> 5;
    // XCD-aware block order: hardware places block i on XCD i % 8; renumber so that each XCD owns a contiguous range of
    // (batch, head, query block) -- all query blocks of a head then share one L2 and its K / V^T cross the fabric once
    int lin = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    }
    const int qblk = lin % P.nqb;
    const int bhi = lin / P.nqb;
    const int h = bhi % P.H;
    const int b = bhi / P.H;
    const size_t bh = (size_t)bhi;

    const bf16* __restrict__ Qg = P.q + bh * P.Tq_pad * DP;
    const bf16* __restrict__ Kg = P.k + bh * P.Tk_pad * DP;
    const bf16* __restrict__ Vg = P.vt + bh * DPV * P.Tk_pad;

    const int myq = qblk * 128 + wave * 32 + lrow;

    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)myq * DP + 16 * s + 8 * half);
    if constexpr (BIAS) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = f2bf((float)qf[s][e] * P.scale_log2e);
        if (half == 1) qf[2][0] = f2bf(0.f);   // column 40 = -m, m = 0 until the first tile has been seen
    }

    f32x16 ot[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    float m_run = BIAS ? 0.f : -1e30f;  // running max (BIAS: the stabiliser baked into Q column 40), in scaled (log2) units
    float l_run = 0.f;     // this lane's partial denominator (its 16+16 keys per tile)
    const float c = P.scale_log2e;

    const int nt = (P.Nk + 63) >> 6;

    typedef uint32_t kvec_t __attribute__((ext_vector_type(4 * NKC)));
    typedef uint32_t vvec_t __attribute__((ext_vector_type(4 * NVC)));
    kvec_t kreg; vvec_t vreg;
    auto load_tile = [&](int it, kvec_t& kreg, vvec_t& vreg) {
        const int kv0 = it << 6;
        const bf16* kp = Kg + (size_t)kv0 * DP;
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            int id = t + i * 256;
            if (id > KCH - 1) id = KCH - 1;  // tail threads re-read the last chunk (never stored)
            uint4 v = *reinterpret_cast<const uint4*>(kp + id * 8);
            kreg[4*i]=v.x; kreg[4*i+1]=v.y; kreg[4*i+2]=v.z; kreg[4*i+3]=v.w;
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            int id = t + i * 256;
            uint4 v = *reinterpret_cast<const uint4*>(Vg + (size_t)(id >> 3) * P.Tk_pad + kv0 + (id & 7) * 8);
            vreg[4*i]=v.x; vreg[4*i+1]=v.y; vreg[4*i+2]=v.z; vreg[4*i+3]=v.w;
        }
    };
    auto store_tile = [&](int buf, const kvec_t& kreg, const vvec_t& vreg) {
        unsigned char* ks = smem + buf * STAGE;
        unsigned char* vs = ks + KBYTES;
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            int id = t + i * 256;
            if (id < KCH) {
                int cch = id >> 6;        // key-tile layout: [DP / 8 chunks][64 keys]
                int row = id & 63;
                uint32_t k0 = kreg[4*i];
                if constexpr (BIAS) k0 = cch == 5 ? ((k0 & 0xffff0000u) | 0x3f80u) : k0;   // K[:, 40] = 1.0
                *reinterpret_cast<uint4*>(ks + row * KROW + cch * 16) = make_uint4(k0,kreg[4*i+1],kreg[4*i+2],kreg[4*i+3]);
            }
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            int id = t + i * 256;
            *reinterpret_cast<uint4*>(vs + (id >> 3) * VROW + (id & 7) * 16) = make_uint4(vreg[4*i],vreg[4*i+1],vreg[4*i+2],vreg[4*i+3]);
        }
    };

    load_tile(0, kreg, vreg);
    store_tile(0, kreg, vreg);
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nt;
        if (more) load_tile(it + 1, kreg, vreg);

        const unsigned char* ks = smem + buf * STAGE;
        const unsigned char* vs = ks + KBYTES;

        // ---- S^T = K Q^T for two 32-key sub-tiles
        f32x16 st[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (32 * u + lrow) * KROW + (2 * s + half) * 16);
                st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[u], 0, 0, 0);
            }
        }

        // ---- online softmax (per lane = per query row; keys split between lane and lane^32)
        const int kv0 = it << 6;
        if (kv0 + 64 > P.Nk) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int kv = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (kv >= P.Nk) st[u][r] = -1e30f;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[u][r]);
        mx = max_xor32(mx);
        bf16x8 pb0, pb1, pb2, pb3;
        if constexpr (BIAS) {
            // st already is s*c - m_run
            const bool move = it == 0 || mx > 6.f;
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
                const bf16 mb = f2bf(m_run + mx);
                const float m_upd = move ? (float)mb : m_run;
                const float delta = m_upd - m_run;   // exact: both are bf16 values
                if (it > 0) {                        // O^T is still zero in the first tile (and 2^-delta may overflow there)
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int i = 0; i < DT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[u][r] -= delta;
                m_run = m_upd;
                const bf16 nb = f2bf(-m_upd);
                if (half == 1) qf[2][0] = nb;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[u][r] = __builtin_amdgcn_exp2f(st[u][r]);
        } else {
        const float m_new = fmaxf(m_run, mx * c);
        // the running max stops growing after the first few tiles: rescale O only when some row's did
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
        }
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(fmaf(st[u][r], c, -m_new));
                if constexpr (!ONES) psum += p;
                st[u][r] = p;
            }
        if constexpr (!ONES) l_run += psum;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pb0[e] = f2bf(st[0][e]);
            pb1[e] = f2bf(st[0][8 + e]);
            pb2[e] = f2bf(st[1][e]);
            pb3[e] = f2bf(st[1][8 + e]);
        }

        // ---- O^T += V^T P^T
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const unsigned char* vrow = vs + (32 * i + lrow) * VROW + half * 16;
            bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(vrow);
            bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(vrow + 32);
            bf16x8 vf2 = *reinterpret_cast<const bf16x8*>(vrow + 64);
            bf16x8 vf3 = *reinterpret_cast<const bf16x8*>(vrow + 96);
            ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pb0, ot[i], 0, 0, 0);
            ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pb1, ot[i], 0, 0, 0);
            ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf2, pb2, ot[i], 0, 0, 0);
            ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf3, pb3, ot[i], 0, 0, 0);
        }

        if (more) store_tile(buf ^ 1, kreg, vreg);
        __syncthreads();
    }

    float l_tot;
    if constexpr (ONES) {
        // O^T row d: tile i = d / 32, offset off -> register ((off>>3)<<2 | off&3) of the half (off>>2)&1
        const int off = P.d & 31;
        const int rr = ((off >> 3) << 2) | (off & 3);
        float cand = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (i == (P.d >> 5) && r == rr) cand = ot[i][r];
        const float other = __shfl_xor(cand, 32, 64);
        l_tot = (((off >> 2) & 1) == half) ? cand : other;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.f / l_tot;
    if (myq < P.Nq) {
        bf16* orow = P.o + ((size_t)b * P.o_rows_per_b + myq) * P.ldo + h * P.d;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int dd0 = 32 * i + 8 * q4 + 4 * half;
                if (dd0 < P.d) {
                    U2BF4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.e[e] = f2bf(ot[i][4 * q4 + e] * inv);
                    *reinterpret_cast<uint2*>(orow + dd0) = o.u;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// v2 (round 3), d = 40: the same products and layouts, software-pipelined across key tiles.
//
// What bounds this kernel is INSTRUCTION ISSUE, not a pipe (profiles/r3: replacing the 32 v_exp per tile by v_mul changes nothing,
// removing the MFMAs removes 57 % of the time, the MFMA pipe is 46 % busy; a wave issues about one instruction per 4 clocks and a
// v_mfma_f32_32x32x16 covers 8 such slots, of which about 5 can hold other instructions of the same wave): per 64-key tile a wave
// has 14 MFMAs = 448 clocks that hide ~70 other instructions, everything beyond that is exposed at ~4 clocks apiece. So the loop
// is built to (a) interleave -- every iteration j holds three independent pieces of work in one straight-line region,
//     S(j) = K(j) Q^T            (MFMA)   into the "next" score registers
//     P(j-1) = exp2(S(j-1) - m)  (VALU)   from the "current" score registers
//     O^T += V^T(j-1) P^T(j-1)   (MFMA)   in two halves, each behind the exps that feed it
// so that the VALU work sits between the MFMAs instead of behind them -- and (b) to carry nothing else: K and V^T tiles come by
// LDS-DMA (buffer_load ... lds, no staging registers, no ds_write pass; 3-4 instructions per wave and tile, addressed from a few
// loop-invariant scalars), the ring slot is a compile-time constant of each of the two unrolled loop bodies, both score register
// sets alternate roles from one iteration to the next (nothing is copied).
// The rare stabiliser move (a tile's scores exceed the running one by > 2^6) sits at the end of the iteration, after PV(j-1):
// O^T -- then consistently at the old stabiliser -- and S(j) are rescaled there.
// Stage j of the two-slot ring holds {K(j), V^T(j-1)} and is fetched during iteration j-1. Fragment reads are inline asm (hipcc
// would drain the DMA in front of a visible LDS load, see gemm.hip) with counted lgkmcnt waits.
//   K tile   [DP/8 chunks][64 keys][16 B]: ds_read_b128 of chunk c for keys lrow is conflict free as it stands
//   V^T tile [DPV rows][128 B], 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (source-side swizzle of the DMA)
typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
#define GL_BLDS16(rsrc, ldst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(ldst), 16, (int)(voff), (int)(soff), 0, 0)

template <int OFF>
__device__ __forceinline__ void lds_rd16(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void pin_regs(bf16x8 (&d)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(d[i]));
}
// kf[2 s + u] <- 16-byte chunk 2 s + half of keys 32 u + lrow: K tile offset s * 2048 + u * 512 from the lane's base
template <int N, int I = 0, int BASE = 0>
__device__ __forceinline__ void lds_rd_k(bf16x8 (&d)[N], unsigned a) {
    if constexpr (I < N) {
        lds_rd16<((BASE + I) >> 1) * 2048 + ((BASE + I) & 1) * 512>(d[I], a);
        lds_rd_k<N, I + 1, BASE>(d, a);
    }
}
// vf[2 i + jj] <- V^T rows 32 i + lrow through the lane's two (swizzled) chunk addresses a0 (jj = 0), a1 (jj = 1)
template <int N, int I = 0>
__device__ __forceinline__ void lds_rd_v(bf16x8 (&d)[N], unsigned a0, unsigned a1) {
    if constexpr (I < N) {
        lds_rd16<(I >> 1) * 4096>(d[I], (I & 1) ? a1 : a0);
        lds_rd_v<N, I + 1>(d, a0, a1);
    }
}
template <int DP, int DPV, bool BIAS, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) attn2_kernel(AttnParams P) {
    constexpr int KS = DP / 16;            // k-steps of S^T = K Q^T
    constexpr int DT = DPV / 32;           // 32-row tiles of O^T
    constexpr int KP = DP / 8;             // K tile: 1-KiB DMA pieces (one 16-byte chunk of all 64 keys each)
    constexpr int VP = DPV / 8;            // V^T tile: 1-KiB DMA pieces (8 rows of 128 B each)
    constexpr int KBYTES = KP * 1024;
    constexpr int VBYTES = VP * 1024;
    constexpr int STAGE = KBYTES + VBYTES;
    constexpr int KR = (KP + NW - 1) / NW, VR = VP / NW;   // piece rounds: wave w takes pieces w, w + NW, ...
    static_assert(DPV % 32 == 0 && DP % 16 == 0 && DPV > DP - 8 && VP % NW == 0, "tile geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // (the only LDS of this kernel: offset 0, 128-byte aligned)

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lrow = lane & 31;
    const int half = lane >> 5;
    int lin = blockIdx.x;
    {   // XCD-aware block order, as above
        const int total = gridDim.x;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    }
    const int qblk = lin % P.nqb;
    const int bhi = lin / P.nqb;
    const int h = bhi % P.H;
    const int b = bhi / P.H;
    const size_t bh = (size_t)bhi;

    const bf16* __restrict__ Qg = P.q + bh * P.Tq_pad * DP;
    const bf16* __restrict__ Kg = P.k + bh * P.Tk_pad * DP;
    const bf16* __restrict__ Vg = P.vt + bh * DPV * P.Tk_pad;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, 0x80000000u, 0x00020000);

    const int nt = (P.Nk + 63) >> 6;
    const int myq = qblk * (NW * 32) + wave * 32 + lrow;
    const int myq_ld = myq < P.Tq_pad ? myq : P.Tq_pad - 1;     // (NW = 8: a 256-row block may reach past the 128-row padding)

    // ---- DMA of stage jj = {K(jj), V^T(jj-1)} into ring slot `slot`. Wave w moves K pieces w, w + NW, .. and V^T pieces w, w + NW, ..;
    // everything but jj is loop invariant: lane offsets in two / VR VGPRs, the wave's piece in one scalar.
    const lds_ptr lbase = (lds_ptr)smem;
    const int w1k = wave * 1024;
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned vvoff[VR];
#pragma unroll
    for (int i = 0; i < VR; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        vvoff[i] = (unsigned)(r * P.Tk_pad) * 2u + (unsigned)c * 16u;
    }
    auto issue_k = [&](int jj, int slot) {
        const lds_ptr st = lbase + slot * STAGE + w1k;
        const int so = jj * KBYTES + w1k;
#pragma unroll
        for (int i = 0; i < KR; ++i) {
            if ((i + 1) * NW <= KP) GL_BLDS16(rk, st + i * NW * 1024, lane16, so + i * NW * 1024);
            else if (wave < KP - i * NW) GL_BLDS16(rk, st + i * NW * 1024, lane16, so + i * NW * 1024);   // short last round (wave-uniform)
        }
    };
    auto issue_v = [&](int jj, int slot) {
        const lds_ptr st = lbase + slot * STAGE + KBYTES + w1k;
        const int so = (jj - 1) * 128;
#pragma unroll
        for (int i = 0; i < VR; ++i) GL_BLDS16(rv, st + i * NW * 1024, vvoff[i], so);
    };
    issue_k(0, 0);

    // ---- Q fragments, pre-multiplied by scale * log2(e) (scores come out of the matrix core in exp2 units)
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)myq_ld * DP + 16 * s + 8 * half);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = f2bf((float)qf[s][e] * P.scale_log2e);
    if constexpr (BIAS) {
        if (half == 1) qf[2][0] = f2bf(0.f);   // column 40 = -m, m = 0 until the first tile has been seen
    }

    f32x16 ot[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    float m_run = BIAS ? 0.f : -1e30f;   // BIAS: the stabiliser baked into Q column 40; else subtracted in front of the exps

    const unsigned k_lane = (unsigned)(lrow * 16 + half * 1024);
    const unsigned v_lane = KBYTES + (unsigned)(lrow * 128) + (unsigned)((half ^ ((lrow >> 1) & 7)) << 4);

    // One iteration. SLOT: ring slot of stage j (0 / 1; -1: j & 1 at run time, the once-only iterations). ISSUE: which parts of stage
    // j+1 exist (0 none, 1 K and V^T, 2 V^T only, 3 V^T and, if j+1 < nt, K). SM: softmax + PV of tile j-1 (scores in sc). QK: scores
    // of tile j into sn. TAILCK: tile j may be the partial last tile (run-time check). FIRSTMOVE: tile j is tile 0.
    auto iter = [&](int j, f32x16 (&sc)[2], f32x16 (&sn)[2], auto slot_c, auto issue_c, auto sm_c, auto qk_c, auto tail_c, auto first_c) {
        constexpr int SLOT = decltype(slot_c)::value, ISSUE = decltype(issue_c)::value;
        constexpr bool SM = decltype(sm_c)::value, QK = decltype(qk_c)::value, TAILCK = decltype(tail_c)::value, FIRSTMOVE = decltype(first_c)::value;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): stage j has landed (this wave's pieces; the barrier covers the others')
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int slot = SLOT >= 0 ? SLOT : (j & 1);
        if constexpr (ISSUE != 0) {           // stage j+1 into the slot every wave finished reading in iteration j-1
            if constexpr (ISSUE == 1) issue_k(j + 1, slot ^ 1);
            if constexpr (ISSUE == 3) { if (j + 1 < nt) issue_k(j + 1, slot ^ 1); }
            issue_v(j + 1, slot ^ 1);
        }
        const unsigned sb = (unsigned)(slot * STAGE);
        const unsigned ak = k_lane + sb;
        unsigned av[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) av[jj] = (v_lane + sb) ^ (unsigned)(jj << 5);

        bf16x8 kf[2 * KS], vfa[2 * DT], vfb[2 * DT];
        if constexpr (QK) lds_rd_k(kf, ak);
        if constexpr (SM) lds_rd_v(vfa, av[0], av[1]);     // key sub-tile 0 (keys 0..31 of tile j-1)
        // ---- region 1: S(j) on the matrix core  ||  exp2 of the first 32 keys of tile j-1
        if constexpr (QK) {
            if constexpr (SM) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DT));
            else asm volatile("s_waitcnt lgkmcnt(0)");
            pin_regs(kf);
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], zero, 0, 0, 0);
            sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[0], zero, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KS; ++s) {
                sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s], qf[s], sn[0], 0, 0, 0);
                sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s + 1], qf[s], sn[1], 0, 0, 0);
            }
        }
        bf16x8 pb0, pb1, pb2, pb3;
        if constexpr (SM) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pb0[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[0][e] : sc[0][e] - m_run));
                pb1[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[0][8 + e] : sc[0][8 + e] - m_run));
            }
            __builtin_amdgcn_sched_barrier(0);
            // the second sub-tile's V^T fragments (they may take the K fragments' registers: S(j)'s MFMAs are issued)
            lds_rd_v(vfb, av[2], av[3]);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DT));   // vfa has arrived
            pin_regs(vfa);
            // ---- region 2: first half of O^T += V^T P^T  ||  exp2 of the other 32 keys
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[2 * i], pb0, ot[i], 0, 0, 0);
                ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[2 * i + 1], pb1, ot[i], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pb2[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[1][e] : sc[1][e] - m_run));
                pb3[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[1][8 + e] : sc[1][8 + e] - m_run));
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)");
            pin_regs(vfb);
            // ---- region 3: second half of PV  ||  row maximum of S(j)
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[2 * i], pb2, ot[i], 0, 0, 0);
                ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[2 * i + 1], pb3, ot[i], 0, 0, 0);
            }
        }
        if constexpr (QK) {
            if constexpr (TAILCK) {
                const int kv0 = j << 6;
                if (kv0 + 64 > P.Nk) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kv = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (kv >= P.Nk) sn[u][r] = -1e30f;
                        }
                }
            }
            float mx = sn[0][0];     // (hipcc fuses the chain into v_max3_f32; an asm v_max3 costs an s_nop behind each)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sn[u][r]);
            mx = max_xor32(mx);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stabiliser move (rare after the first tiles). Everything accumulated so far -- O^T, including tile j-1 -- is at the
            // old stabiliser; S(j) is the only other thing that depends on it.
            if constexpr (BIAS) {
                // sn is s * c - m_run (the stabiliser rode through the MFMA in Q column 40)
                const bool move = FIRSTMOVE || mx > 6.f;
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const bf16 mb = f2bf(m_run + mx);
                    const float m_upd = move ? (float)mb : m_run;
                    const float delta = m_upd - m_run;   // exact: both are bf16 values
                    if constexpr (!FIRSTMOVE) {           // O^T is still zero in the first tile (and 2^-delta may overflow there)
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                        for (int i = 0; i < DT; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sn[u][r] -= delta;
                    m_run = m_upd;
                    const bf16 nb = f2bf(-m_upd);
                    if (half == 1) qf[2][0] = nb;
                }
            } else {
                // sn is s * c; the exps subtract m_run
                const bool move = FIRSTMOVE || mx - m_run > 6.f;
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const float m_upd = move ? mx : m_run;
                    if constexpr (!FIRSTMOVE) {
                        const float alpha = __builtin_amdgcn_exp2f(m_run - m_upd);
#pragma unroll
                        for (int i = 0; i < DT; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
                    }
                    m_run = m_upd;
                }
            }
        }
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using SR = std::integral_constant<int, -1>;     // slot = j & 1 at run time
    using I0 = std::integral_constant<int, 0>;      // nothing to fetch
    using I1 = std::integral_constant<int, 1>;      // stage j+1 = {K, V^T}
    using I2 = std::integral_constant<int, 2>;      // stage j+1 = {V^T} (j + 1 = nt)
    using I3 = std::integral_constant<int, 3>;      // stage j+1 = {V^T} and K if it exists
    f32x16 sa[2], sb2[2];
    iter(0, sb2, sa, S0{}, I3{}, F_{}, T_{}, T_{}, T_{});                      // S(0) -> sa
    if (nt == 1) {
        iter(1, sa, sb2, S1{}, I0{}, T_{}, F_{}, F_{}, F_{});                  // drain
    } else {
        int j = 1;
        for (; j + 1 <= nt - 2; j += 2) {                                      // j odd: stage j sits in slot 1
            iter(j, sa, sb2, S1{}, I1{}, T_{}, T_{}, F_{}, F_{});
            iter(j + 1, sb2, sa, S0{}, I1{}, T_{}, T_{}, F_{}, F_{});
        }
        if (j <= nt - 2) {
            iter(j, sa, sb2, S1{}, I1{}, T_{}, T_{}, F_{}, F_{});
            ++j;
            iter(j, sb2, sa, S0{}, I2{}, T_{}, T_{}, T_{}, F_{});              // j = nt-1 (even): S(nt-1) -> sa
            iter(j + 1, sa, sb2, S1{}, I0{}, T_{}, F_{}, F_{}, F_{});          // drain
        } else {
            iter(j, sa, sb2, S1{}, I2{}, T_{}, T_{}, T_{}, F_{});              // j = nt-1 (odd): S(nt-1) -> sb2
            iter(j + 1, sb2, sa, S0{}, I0{}, T_{}, F_{}, F_{}, F_{});          // drain
        }
    }

    // softmax denominator: O^T row d (the ones row of V^T); tile i = d / 32, offset off -> register ((off>>3)<<2 | off&3) of the half (off>>2)&1
    const int off = P.d & 31;
    const int rr = ((off >> 3) << 2) | (off & 3);
    float cand = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (i == (P.d >> 5) && r == rr) cand = ot[i][r];
    const float other = __shfl_xor(cand, 32, 64);
    const float l_tot = (((off >> 2) & 1) == half) ? cand : other;
    const float inv = 1.f / l_tot;
    if (myq < P.Nq) {
        bf16* orow = P.o + ((size_t)b * P.o_rows_per_b + myq) * P.ldo + h * P.d;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int dd0 = 32 * i + 8 * q4 + 4 * half;
                if (dd0 < P.d) {
                    U2BF4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.e[e] = f2bf(ot[i][4 * q4 + e] * inv);
                    *reinterpret_cast<uint2*>(orow + dd0) = o.u;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// v3 (round 5), d = 40: attn2_kernel with O^T += V^T P^T on v_mfma_f32_16x16x32_bf16.
//
// The 32x32x16 shape makes O^T a [64 x 32 queries] product for 40 real rows (+ the ones row): 37 % of the P V work multiplies padding,
// a V^T tile is 8 KB of DMA and LDS reads where 5.1 KB carry data, and O^T takes 32 accumulator registers. With 16-row MFMAs O^T is
// [48 x 16] per query half: 12 MFMAs of 16 clocks per 64-key tile instead of 8 of 32 (-25 % of the P V issue time, -14 % of the
// tile's), 6 KB per V^T tile (K + V^T stage 12 KB instead of 14), 6 fragment reads instead of 8, 24 accumulator registers.
// S^T = K Q^T stays on 32x32x16 (its k dimension is the head dim: 48 padded columns in three k-steps; a 32-wide k-step would pad
// 40 -> 64). The price is the operand hand-over: a 32x32 accumulator holds query q in lane q (keys 8 j + e) and lane q + 32 (keys
// 8 j + 4 + e), the 16x16x32 B operand wants query q & 15 of ONE 16-query half in all four 16-lane rows, each row with its own 8
// keys. v_permlane16_swap_b32 does exactly that exchange: with X = the packed exps of accumulator registers 0..7 and Y = those of
// registers 8..15, swapping the odd rows of X with the even rows of Y leaves X = the operand of queries 0..15 and Y = that of
// queries 16..31, k-slot groups (lane >> 4 = 0..3) = keys {0-3, 8-11}, {16-19, 24-27}, {4-7, 12-15}, {20-23, 28-31} of the 32-key
// sub-tile. The q,k,v^T projection stores V^T's tokens in that order (Epilogue::vt_perm32), so an A fragment (16 rows x 32 keys)
// is one ds_read_b128 per lane: row 16 i + (lane & 15), 16-byte chunk 4 u + (lane >> 4) of the tile's 128-byte row. 8 swaps per
// lane and tile buy 4 fewer MFMA issue slots' worth of matrix time -- what the kernel is short of is issue time under the MFMAs,
// so the ledger is: -64 MFMA clocks, +4 MFMA issues, +8 VALU, -2 ds_read, -1 DMA per wave and tile.
// Everything else is attn2_kernel's: pipeline (S(j) || exp2 of S(j-1) || P V (j-1)), LDS-DMA ring, lazy stabiliser in Q column 40,
// denominator from the ones row (row 40 of V^T = accumulator register 0 of row block 2 in lanes 32..47).
__device__ __forceinline__ void swap16(bf16x8& x, bf16x8& y) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t a = __builtin_bit_cast(u32x4_t, x), b = __builtin_bit_cast(u32x4_t, y);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const auto r = __builtin_amdgcn_permlane16_swap(a[w], b[w], false, false);   // odd 16-lane rows of a <-> even rows of b
        a[w] = r[0];
        b[w] = r[1];
    }
    x = __builtin_bit_cast(bf16x8, a);
    y = __builtin_bit_cast(bf16x8, b);
}
template <int N, int I = 0>
__device__ __forceinline__ void lds_rd_v16(bf16x8 (&d)[N], unsigned a) {
    if constexpr (I < N) {
        lds_rd16<I * 2048>(d[I], a);      // row block i: 16 rows of 128 bytes on
        lds_rd_v16<N, I + 1>(d, a);
    }
}
// DP / DPV: padded head dims of the q,k / v^T buffers (48 / 48 at d = 40, 80 / 96 at d = 80); BIAS: the stabiliser rides in the free Q column
// (d = 40), else it is subtracted in front of the exps. The ones row of V^T is row d (P.d) in both.
template <int DP, int DPV, bool BIAS, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) attn3_kernel(AttnParams P) {
    static_assert(DP % 16 == 0 && DPV % 16 == 0 && DPV > DP - 8, "tile geometry");
    constexpr int KS = DP / 16;            // k-steps of S^T = K Q^T
    constexpr int DB = DPV / 16;           // 16-row blocks of O^T
    constexpr int KP = DP / 8, VP = DPV / 8;   // 1-KiB DMA pieces per K / V^T tile
    constexpr int KBYTES = KP * 1024, VBYTES = VP * 1024, STAGE = KBYTES + VBYTES;
    constexpr int KR = (KP + NW - 1) / NW, VR = (VP + NW - 1) / NW;
    // LATE_V (d = 80): a sub-tile's V^T fragments are read when the registers of what ran before them are free (the K fragments behind
    // S(j)'s MFMAs, sub-tile 0's behind the first half of P V) instead of a region ahead: 48 fewer live registers, which is what lets
    // the pipelined form run at two waves per SIMD there (the early form spills 84); the read's latency then hides behind the
    // operand swaps and the SIMD's other wave only
    constexpr bool LATE_V = DPV > 64;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lrow = lane & 31;
    const int half = lane >> 5;
    const int l15 = lane & 15;
    const int g4 = lane >> 4;
    int lin = blockIdx.x;
    {   // XCD-aware block order, as above
        const int total = gridDim.x;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    }
    const int qblk = lin % P.nqb;
    const int bhi = lin / P.nqb;
    const int h = bhi % P.H;
    const int b = bhi / P.H;
    const size_t bh = (size_t)bhi;

    const bf16* __restrict__ Qg = P.q + bh * P.Tq_pad * DP;
    const bf16* __restrict__ Kg = P.k + bh * P.Tk_pad * DP;
    const bf16* __restrict__ Vg = P.vt + bh * DPV * P.Tk_pad;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, 0x80000000u, 0x00020000);

    const int nt = (P.Nk + 63) >> 6;
    const int q0 = qblk * (NW * 32) + wave * 32;
    const int myq = q0 + lrow;
    const int myq_ld = myq < P.Tq_pad ? myq : P.Tq_pad - 1;

    const lds_ptr lbase = (lds_ptr)smem;
    const int w1k = wave * 1024;
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned vvoff[VR];
#pragma unroll
    for (int i = 0; i < VR; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);          // (pieces >= VP of the short last round are never issued)
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        vvoff[i] = (unsigned)(r * P.Tk_pad) * 2u + (unsigned)c * 16u;
    }
    auto issue_k = [&](int jj, int slot) {
        const lds_ptr st = lbase + slot * STAGE + w1k;
        const int so = jj * KBYTES + w1k;
#pragma unroll
        for (int i = 0; i < KR; ++i) {
            if ((i + 1) * NW <= KP) GL_BLDS16(rk, st + i * NW * 1024, lane16, so + i * NW * 1024);
            else if (wave < KP - i * NW) GL_BLDS16(rk, st + i * NW * 1024, lane16, so + i * NW * 1024);
        }
    };
    auto issue_v = [&](int jj, int slot) {
        const lds_ptr st = lbase + slot * STAGE + KBYTES + w1k;
        const int so = (jj - 1) * 128;
#pragma unroll
        for (int i = 0; i < VR; ++i) {
            if ((i + 1) * NW <= VP) GL_BLDS16(rv, st + i * NW * 1024, vvoff[i], so);
            else if (wave < VP - i * NW) GL_BLDS16(rv, st + i * NW * 1024, vvoff[i], so);
        }
    };
    issue_k(0, 0);

    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)myq_ld * DP + 16 * s + 8 * half);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = f2bf((float)qf[s][e] * P.scale_log2e);
    if constexpr (BIAS) {
        if (half == 1) qf[2][0] = f2bf(0.f);   // column 40 = -m, m = 0 until the first tile has been seen
    }

    f32x4_t ot[2][DB];                     // [query half][row block]: lane holds query 16 qh + (lane & 15), rows 16 i + 4 (lane >> 4) + r
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[qh][i][r] = 0.f;
    float m_run = BIAS ? 0.f : -1e30f;   // BIAS: the stabiliser baked into Q column 40; else subtracted in front of the exps
    // Round 6, d = 40 (BIAS): the stabiliser follows the DENOMINATOR, not the row maximum. The per-tile maximum of the scores (15 v_max3 +
    // 6 v_max + a lane swap per tile: a sixth of the loop's VALU issue time, which is what bounds the kernel) only ever decided whether
    // exp2(s - m) might overflow; the running denominator l = row d of O^T says the same two tiles later: once it exceeds 2^LAZY_LOG2 the
    // stabiliser moves up by floor(log2 l) (O^T, the pending scores and the Q column rescaled as before). A single tile may overshoot by
    // ~2^60 before the products with V near fp32's range; should that ever happen the denominator comes out beyond 2^100 (or non-finite), and the workgroup
    // runs its tiles again with the exact per-tile maximum (safe: the loop of rounds 3-5). The first tile's maximum is still taken.
    constexpr bool LAZY = BIAS;
    constexpr int LAZY_LOG2 = 40;
    constexpr int D_ROW = BIAS ? 40 : DP;   // the ones row of V^T (P.d): 40 at DP = 48, 80 at DP = 80
    constexpr int OB = D_ROW >> 4, OG = (D_ROW & 15) >> 2, ORR = D_ROW & 3;
    bool overshoot = false;                 // a lazy move found a denominator beyond 2^100: some P v product may have left fp32's range

    const unsigned k_lane = (unsigned)(lrow * 16 + half * 1024);
    const unsigned v_lane = KBYTES + (unsigned)(l15 * 128) + (unsigned)((g4 ^ ((l15 >> 1) & 7)) << 4);

    // safe_c: the exact per-tile maximum (the only form without LAZY; with it, the rerun of a workgroup whose lazy pass overflowed)
    auto iter = [&](int j, f32x16 (&sc)[2], f32x16 (&sn)[2], auto slot_c, auto issue_c, auto sm_c, auto qk_c, auto tail_c, auto first_c, auto safe_c) {
        constexpr bool SAFE = !LAZY || decltype(safe_c)::value;
        constexpr int SLOT = decltype(slot_c)::value, ISSUE = decltype(issue_c)::value;
        constexpr bool SM = decltype(sm_c)::value, QK = decltype(qk_c)::value, TAILCK = decltype(tail_c)::value, FIRSTMOVE = decltype(first_c)::value;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): stage j has landed (this wave's pieces; the barrier covers the others')
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int slot = SLOT >= 0 ? SLOT : (j & 1);
        if constexpr (ISSUE != 0) {
            if constexpr (ISSUE == 1) issue_k(j + 1, slot ^ 1);
            if constexpr (ISSUE == 3) { if (j + 1 < nt) issue_k(j + 1, slot ^ 1); }
            issue_v(j + 1, slot ^ 1);
        }
        if constexpr (LAZY && SM && !SAFE) {
            {
                // the denominators accumulated through tile j - 2 (their MFMAs retired an iteration ago: no wait here): row d of O^T, lanes 16 OG ..
                const bool big = g4 == OG && (ot[0][OB][ORR] > __builtin_ldexpf(1.f, LAZY_LOG2) || ot[1][OB][ORR] > __builtin_ldexpf(1.f, LAZY_LOG2));
                if (__builtin_amdgcn_ballot_w64(big) != 0) {
                    const float la = __shfl(ot[0][OB][ORR], 16 * OG + l15, 64), lb = __shfl(ot[1][OB][ORR], 16 * OG + l15, 64);
                    const float lq = lrow < 16 ? la : lb;                        // this lane's query = lane & 31
                    overshoot |= !(lq < 1.0e30f);      // (sticky: the move below brings l back into range, not an O^T entry that has overflowed)
                    const float e = lq > __builtin_ldexpf(1.f, LAZY_LOG2) ? (float)((int)((__float_as_uint(lq) >> 23) & 0xffu) - 127) : 0.f;
                    const bf16 mb = f2bf(m_run + e);
                    const float m_upd = (float)mb;
                    const float delta = m_upd - m_run;   // exact: both are bf16 values
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    const auto ar = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
                    const float a0 = __uint_as_float(ar[0]), a1 = __uint_as_float(ar[1]);
#pragma unroll
                    for (int i = 0; i < DB; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ot[0][i][r] *= a0; ot[1][i][r] *= a1; }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[u][r] -= delta;          // the scores whose exps this iteration takes carry the old stabiliser
                    m_run = m_upd;
                    const bf16 nb = f2bf(-m_upd);
                    if (half == 1) qf[2][0] = nb;
                }
            }
        }
        const unsigned sb = (unsigned)(slot * STAGE);
        const unsigned ak = k_lane + sb;
        const unsigned av0 = v_lane + sb, av1 = av0 ^ 64u;     // key sub-tile 1: chunk index + 4

        bf16x8 kf[2 * KS], vfa[DB], vfb[LATE_V ? 1 : DB];
        if constexpr (QK) lds_rd_k(kf, ak);
        if constexpr (SM && !LATE_V) lds_rd_v16(vfa, av0);
        // ---- region 1: S(j) on the matrix core  ||  exp2 of the first 32 keys of tile j-1
        if constexpr (QK) {
            if constexpr (SM && !LATE_V) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(DB));
            else asm volatile("s_waitcnt lgkmcnt(0)");
            pin_regs(kf);
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], zero, 0, 0, 0);
            sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[0], zero, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KS; ++s) {
                sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s], qf[s], sn[0], 0, 0, 0);
                sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s + 1], qf[s], sn[1], 0, 0, 0);
            }
        }
        if constexpr (SM) {
            bf16x8 px, py;     // after swap16: px = P^T operand of queries 0..15, py = of queries 16..31
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                px[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[0][e] : sc[0][e] - m_run));
                py[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[0][8 + e] : sc[0][8 + e] - m_run));
            }
            if constexpr (!LATE_V) {
                swap16(px, py);
                __builtin_amdgcn_sched_barrier(0);
                lds_rd_v16(vfb, av1);                                   // the second sub-tile's V^T fragments
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(DB));      // vfa has arrived
            } else {
                __builtin_amdgcn_sched_barrier(0);
                lds_rd_v16(vfa, av0);                                   // (into the registers S(j)'s MFMAs have released)
                swap16(px, py);
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
            pin_regs(vfa);
            // ---- region 2: first half of O^T += V^T P^T  ||  exp2 of the other 32 keys
#pragma unroll
            for (int i = 0; i < DB; ++i) {
                ot[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfa[i], px, ot[0][i], 0, 0, 0);
                ot[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfa[i], py, ot[1][i], 0, 0, 0);
            }
            bf16x8 pz, pw;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pz[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[1][e] : sc[1][e] - m_run));
                pw[e] = f2bf(__builtin_amdgcn_exp2f(BIAS ? sc[1][8 + e] : sc[1][8 + e] - m_run));
            }
            if constexpr (!LATE_V) {
                swap16(pz, pw);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)");
                pin_regs(vfb);
            } else {
                __builtin_amdgcn_sched_barrier(0);
                lds_rd_v16(vfa, av1);                                   // (the first half's MFMAs are issued: their operands are free)
                swap16(pz, pw);
                asm volatile("s_waitcnt lgkmcnt(0)");
                pin_regs(vfa);
            }
            // ---- region 3: second half of PV  ||  row maximum of S(j)
            bf16x8 (&vf2)[DB] = *reinterpret_cast<bf16x8 (*)[DB]>(LATE_V ? &vfa[0] : &vfb[0]);
#pragma unroll
            for (int i = 0; i < DB; ++i) {
                ot[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf2[i], pz, ot[0][i], 0, 0, 0);
                ot[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf2[i], pw, ot[1][i], 0, 0, 0);
            }
        }
        if constexpr (QK) {
            if constexpr (TAILCK) {
                const int kv0 = j << 6;
                if (kv0 + 64 > P.Nk) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kv = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (kv >= P.Nk) sn[u][r] = -1e30f;
                        }
                }
            }
          if constexpr (FIRSTMOVE || SAFE) {
            float mx = sn[0][0];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sn[u][r]);
            mx = max_xor32(mx);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stabiliser move (rare after the first tiles). O^T holds query 16 qh + (lane & 15) in every 16-lane row: the factor of
            // this lane's query (lane & 31) goes to the rows that hold it -- one swap gives both halves' (the two lanes of a query agree:
            // mx went through max_xor32)
            auto rescale_o = [&](float alpha) {
                const auto ar = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
                const float a0 = __uint_as_float(ar[0]), a1 = __uint_as_float(ar[1]);
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ot[0][i][r] *= a0; ot[1][i][r] *= a1; }
            };
            if constexpr (BIAS) {
                // sn is s * c - m_run (the stabiliser rode through the MFMA in Q column 40)
                const bool move = FIRSTMOVE || mx > 6.f;
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const bf16 mb = f2bf(m_run + mx);
                    const float m_upd = move ? (float)mb : m_run;
                    const float delta = m_upd - m_run;   // exact: both are bf16 values
                    if constexpr (!FIRSTMOVE) rescale_o(__builtin_amdgcn_exp2f(-delta));
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sn[u][r] -= delta;
                    m_run = m_upd;
                    const bf16 nb = f2bf(-m_upd);
                    if (half == 1) qf[2][0] = nb;
                }
            } else {
                // sn is s * c; the exps subtract m_run
                const bool move = FIRSTMOVE || mx - m_run > 6.f;
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const float m_upd = move ? mx : m_run;
                    if constexpr (!FIRSTMOVE) rescale_o(__builtin_amdgcn_exp2f(m_run - m_upd));
                    m_run = m_upd;
                }
            }
          }
        }
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    f32x16 sa[2], sb2[2];
#define GL_ATTN3_TILES(SAFE_T) \
    iter(0, sb2, sa, S0{}, I3{}, F_{}, T_{}, T_{}, T_{}, SAFE_T{}); \
    if (nt == 1) { \
        iter(1, sa, sb2, S1{}, I0{}, T_{}, F_{}, F_{}, F_{}, SAFE_T{}); \
    } else { \
        int j = 1; \
        for (; j + 1 <= nt - 2; j += 2) { \
            iter(j, sa, sb2, S1{}, I1{}, T_{}, T_{}, F_{}, F_{}, SAFE_T{}); \
            iter(j + 1, sb2, sa, S0{}, I1{}, T_{}, T_{}, F_{}, F_{}, SAFE_T{}); \
        } \
        if (j <= nt - 2) { \
            iter(j, sa, sb2, S1{}, I1{}, T_{}, T_{}, F_{}, F_{}, SAFE_T{}); \
            ++j; \
            iter(j, sb2, sa, S0{}, I2{}, T_{}, T_{}, T_{}, F_{}, SAFE_T{}); \
            iter(j + 1, sa, sb2, S1{}, I0{}, T_{}, F_{}, F_{}, F_{}, SAFE_T{}); \
        } else { \
            iter(j, sa, sb2, S1{}, I2{}, T_{}, T_{}, T_{}, F_{}, SAFE_T{}); \
            iter(j + 1, sb2, sa, S0{}, I0{}, T_{}, F_{}, F_{}, F_{}, SAFE_T{}); \
        } \
    }
    GL_ATTN3_TILES(F_)
    if constexpr (LAZY) {
        // a denominator that is not a positive finite number: a tile overshot the lazy stabiliser by more than fp32 holds. The whole
        // workgroup (its waves share the barriers of the tile loop) runs its tiles again with the exact per-tile maximum
        const float c0 = ot[0][OB][ORR], c1 = ot[1][OB][ORR];
        // (below 2^100: O^T = sum P v stays finite with it for any |v| < 2^27; a lazy pass that went well leaves l < 2^(LAZY_LOG2 + two tiles' growth))
        const bool bad = overshoot || (g4 == OG && (!(c0 > 0.f && c0 < 1.0e30f) || !(c1 > 0.f && c1 < 1.0e30f)));
        // workgroup-wide OR through the first words of the (now idle) stage buffers -- not __syncthreads_or: its static LDS scratch would
        // move the dynamic buffer off address 0, which the fragment reads assume
        volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(smem);
        __builtin_amdgcn_s_barrier();                       // every wave is done reading the last stage
        const unsigned wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1u : 0u;     // (all lanes vote: outside the lane-0 branch)
        if (lane == 0) flags[wave] = wave_bad;
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): the flag is written
        __builtin_amdgcn_s_barrier();
        unsigned any_bad = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) any_bad |= flags[w];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                       // (the flags are read before the rerun's DMA overwrites them)
        if (any_bad) {
#pragma unroll
            for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[qh][i][r] = 0.f;
            m_run = 0.f;
            if (half == 1) qf[2][0] = f2bf(0.f);
            issue_k(0, 0);
            GL_ATTN3_TILES(T_)
        }
    }
#undef GL_ATTN3_TILES

    // softmax denominator = O^T row d (the ones row of V^T): row block d / 16, local row d % 16 = register (d % 4) of the 16-lane row
    // (d % 16) / 4   (d = 40: block 2, lanes 32..47, register 0; d = 80: block 5, lanes 0..15, register 0)
    const int ob = P.d >> 4, og = (P.d & 15) >> 2, orr = P.d & 3;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        float cand = 0.f;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i == ob && r == orr) cand = ot[qh][i][r];
        const float l_tot = __shfl(cand, 16 * og + l15, 64);
        const float inv = 1.f / l_tot;
        const int q = q0 + 16 * qh + l15;
        if (q < P.Nq) {
            bf16* orow = P.o + ((size_t)b * P.o_rows_per_b + q) * P.ldo + h * P.d;
#pragma unroll
            for (int i = 0; i < DB; ++i) {
                const int dd0 = 16 * i + 4 * g4;
                if (dd0 < P.d) {
                    U2BF4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.e[e] = f2bf(ot[qh][i][e] * inv);
                    *reinterpret_cast<uint2*>(orow + dd0) = o.u;
                }
            }
        }
    }
}

template <int DP, int DPV, bool BIAS, int NW>
static int launch_attn3(const AttnParams& P, int B, hipStream_t stream) {
    const size_t lds = 2 * ((DP / 8) * 1024 + (DPV / 8) * 1024);
    AttnParams Q = P;
    Q.nqb = cdiv(P.Nq, NW * 32);
    dim3 grid(Q.nqb * P.H * B);
    hipLaunchKernelGGL((attn3_kernel<DP, DPV, BIAS, NW>), grid, dim3(NW * 64), lds, stream, Q);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

template <int DP, int DPV, bool BIAS, int NW>
static int launch_attn2(const AttnParams& P, int B, hipStream_t stream) {
    const size_t lds = 2 * ((DP / 8) * 1024 + (DPV / 8) * 1024);
    auto kfn = attn2_kernel<DP, DPV, BIAS, NW>;
    static bool attr_done = false;  // once per instantiation; never inside a stream capture
    if (!attr_done && lds > 48 * 1024) {
        GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    AttnParams Q = P;
    Q.nqb = cdiv(P.Nq, NW * 32);
    dim3 grid(Q.nqb * P.H * B);
    hipLaunchKernelGGL(kfn, grid, dim3(NW * 64), lds, stream, Q);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

template <int DP, int DPV, bool ONES>
static int launch_attn(const AttnParams& P, int B, hipStream_t stream) {
    constexpr int KROW = DP * 2 + 16, VROW = 144;
    size_t lds = 2 * (64 * KROW + DPV * VROW);
    auto kfn = attn_kernel<DP, DPV, ONES>;
    static bool attr_done = false;  // once per instantiation; never inside a stream capture
    if (!attr_done && lds > 48 * 1024) {
        GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    AttnParams Q = P;
    Q.nqb = cdiv(P.Nq, 128);
    dim3 grid(Q.nqb * P.H * B);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, stream, Q);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int attn_dims(int d, int* DP, int* DPV) {
    switch (d) {
        case 40: *DP = 48; *DPV = 64; return GL_OK;
        case 80: *DP = 80; *DPV = 96; return GL_OK;
        case 160: *DP = 160; *DPV = 160; return GL_OK;
        default: return set_error(GL_ERR_UNSUPPORTED, "attention: head dim %d not supported (40, 80, 160)", d);
    }
}

// Row d of every [DPV][Tk_pad] V^T slab := 1.0 (no-op when DPV == d). Called once per buffer, right
// after allocation: the projection epilogue only ever writes rows < d.
__global__ void vt_ones_kernel(bf16* vt, int DPV, int Tk_pad, int d) {
    bf16* row = vt + ((size_t)blockIdx.x * DPV + d) * Tk_pad;
    for (int i = threadIdx.x; i < Tk_pad; i += blockDim.x) row[i] = (bf16)1.0f;
}
int attn_vt_ones_launch(bf16* vt, int BH, int d, int Tk_pad, hipStream_t stream, int DPV) {
    int dp, dpv;
    GL_TRY(attn_dims(d, &dp, &dpv));
    if (DPV) dpv = DPV;
    if (dpv == d) return GL_OK;
    hipLaunchKernelGGL(vt_ones_kernel, dim3(BH), dim3(256), 0, stream, vt, dpv, Tk_pad, d);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// GL_ATTN_V2 (developer A/B): 0 = the unpipelined kernel for every head dim, 1 = attn2_kernel with 4 waves (two workgroups per CU)
// for d = 40, 2 = attn2_kernel with 8 waves (one workgroup per CU: half the K / V^T DMA per query row), 3 (default) = attn3_kernel:
// attn2's pipeline with P V on 16x16x32 MFMAs (V^T buffers in the attn_vt_layout == 1 form) at d = 40 AND d = 80 (48 O^T registers
// instead of 96: the pipelined form fits two waves per SIMD there too), 4 = attn3_kernel at d = 40 only
static int attn_v2_mode() {
    static const int m = dev_env("GL_ATTN_V2") ? atoi(dev_env("GL_ATTN_V2")) : 3;
    return m;
}

// (short key sequences -- the 77 text tokens of the cross-attention: two tiles -- stay on the unpipelined kernel: nothing to
// pipeline, and its four workgroups per CU start and drain faster; 19.5 against 22.8 us at 4096 x 77, profiles/r3)
static bool attn_use_v2(int d, int Nk) { return Nk > 128 && ((d == 40 && attn_v2_mode() != 0) || (d == 80 && attn_v2_mode() == 3)); }

int attn_vt_layout(int d, int Nk, int* DPV) {
    int dp, dpv = 0;
    (void)attn_dims(d, &dp, &dpv);
    const int layout = (attn_use_v2(d, Nk) && attn_v2_mode() >= 3) ? 1 : 0;
    if (DPV) *DPV = (layout && d == 40) ? 48 : dpv;
    return layout;
}

const char* attn_kernel_name(int d, int Nk, int vt_layout) {
    int v2 = attn_use_v2(d, Nk) ? attn_v2_mode() : 0;
    if (vt_layout == 0 && v2 >= 3) v2 = 1;     // a caller whose V^T buffer is in the 16-token form gets the kernel that reads it
    if (d == 40) return v2 == 0 ? "attn_kernel<48, 64, true>" : v2 == 1 ? "attn2_kernel<48, 64, true, 4>" : v2 == 2 ? "attn2_kernel<48, 64, true, 8>" : "attn3_kernel<48, 48, true, 4>";
    if (d == 80) return (v2 >= 3 && vt_layout != 0) ? "attn3_kernel<80, 96, false, 4>" : "attn_kernel<80, 96, true>";
    return "attn_kernel<160, 160, false>";
}

// One-time init of a K buffer (key-tile layout): for d = 40 column 40 of every key := 1.0, the multiplier of the stabiliser that
// rides in Q column 40 (BIAS above). The projection epilogues only ever write columns < d. No-op for the other head dims.
__global__ void k_ones_kernel(bf16* k, int DP, int Tk_pad, int col) {
    bf16* slab = k + (size_t)blockIdx.x * Tk_pad * DP;
    for (int t = threadIdx.x; t < Tk_pad; t += blockDim.x) slab[ktile_off(0, Tk_pad, t, col, DP)] = (bf16)1.0f;
}
int attn_k_init_launch(bf16* k, int BH, int d, int Tk_pad, hipStream_t stream) {
    if (d != 40) return GL_OK;
    hipLaunchKernelGGL(k_ones_kernel, dim3(BH), dim3(256), 0, stream, k, 48, Tk_pad, 40);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int attn_launch(const AttnParams& P, int B, hipStream_t stream) {
    if (P.Nq <= 0 || P.Nk <= 0) return set_error(GL_ERR_ARG, "attention: empty Nq=%d Nk=%d", P.Nq, P.Nk);
    if (P.Tq_pad % 128 != 0 || P.Tk_pad % 64 != 0 || P.Tq_pad < P.Nq || P.Tk_pad < P.Nk)
        return set_error(GL_ERR_ARG, "attention: bad padding Tq_pad=%d Tk_pad=%d (Nq=%d Nk=%d)", P.Tq_pad, P.Tk_pad, P.Nq, P.Nk);
    int v2 = attn_use_v2(P.d, P.Nk) ? attn_v2_mode() : 0;
    // The V^T buffer's form decides between the two pipelined d = 40 kernels: a caller that laid it out per attn_vt_layout() == 1 gets
    // attn3_kernel, one that kept the 16-token form (the hoisted grounding-token K / V of gatedCA, older callers) gets attn2_kernel
    if (P.vt_layout == 1) {
        if (P.d == 40) return launch_attn3<48, 48, true, 4>(P, B, stream);
        if (P.d == 80) return launch_attn3<80, 96, false, 4>(P, B, stream);
        return set_error(GL_ERR_ARG, "attention: vt_layout 1 exists for head dims 40 and 80 (d = %d)", P.d);
    }
    if (v2 >= 3) v2 = 1;
    switch (P.d) {
        case 40:
            if (v2 == 1) return launch_attn2<48, 64, true, 4>(P, B, stream);
            if (v2 >= 2) return launch_attn2<48, 64, true, 8>(P, B, stream);
            return launch_attn<48, 64, true>(P, B, stream);
        case 80:   // (attn2_kernel<80, 96, false, *> needs ~320 registers: 64 spilled at the 256 of two waves per SIMD; not instantiated)
            return launch_attn<80, 96, true>(P, B, stream);
        case 160: return launch_attn<160, 160, false>(P, B, stream);
        default: return set_error(GL_ERR_UNSUPPORTED, "attention: head dim %d not supported (40, 80, 160)", P.d);
    }
}

}  // namespace gl
