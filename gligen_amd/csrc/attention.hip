// Fused (flash-style) multi-head attention forward for gfx950, bf16 in / fp32 softmax+accumulate.
//
// Replaces the materialised-softmax attention of the reference:
//   SelfAttention.forward   ldm/modules/attention.py:167-186   (attn1 and the GLIGEN fuser)
//   CrossAttention.forward  ldm/modules/attention.py:127-149   (attn2, 77 CLIP tokens)
// sim = (q k^T) * d^-0.5 ; softmax over keys ; out = attn v.   No mask, no dropout.
//
// Layouts (written by the projection GEMM epilogues, gemm.hip EPI_QK_HEADS / EPI_VT_HEADS):
//   q  [B*H][Tq_pad][DP]   head dim zero-padded to DP (48 / 80 / 160)
//   k  [B*H][Tk_pad][DP]
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
                int row = id / (DP / 8);
                int cch = id - row * (DP / 8);
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
int attn_vt_ones_launch(bf16* vt, int BH, int d, int Tk_pad, hipStream_t stream) {
    int dp, dpv;
    GL_TRY(attn_dims(d, &dp, &dpv));
    if (dpv == d) return GL_OK;
    hipLaunchKernelGGL(vt_ones_kernel, dim3(BH), dim3(256), 0, stream, vt, dpv, Tk_pad, d);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

const char* attn_kernel_name(int d) {
    return d == 40 ? "attn_kernel<48, 64, true>" : d == 80 ? "attn_kernel<80, 96, true>" : "attn_kernel<160, 160, false>";
}

int attn_launch(const AttnParams& P, int B, hipStream_t stream) {
    if (P.Nq <= 0 || P.Nk <= 0) return set_error(GL_ERR_ARG, "attention: empty Nq=%d Nk=%d", P.Nq, P.Nk);
    if (P.Tq_pad % 128 != 0 || P.Tk_pad % 64 != 0 || P.Tq_pad < P.Nq || P.Tk_pad < P.Nk)
        return set_error(GL_ERR_ARG, "attention: bad padding Tq_pad=%d Tk_pad=%d (Nq=%d Nk=%d)", P.Tq_pad, P.Tk_pad, P.Nq, P.Nk);
    switch (P.d) {
        case 40: return launch_attn<48, 64, true>(P, B, stream);
        case 80: return launch_attn<80, 96, true>(P, B, stream);
        case 160: return launch_attn<160, 160, false>(P, B, stream);
        default: return set_error(GL_ERR_UNSUPPORTED, "attention: head dim %d not supported (40, 80, 160)", P.d);
    }
}

}  // namespace gl
