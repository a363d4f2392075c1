// bf16 MFMA GEMM + implicit-GEMM 3x3 convolution for gfx950 (CDNA4, wave64).
//
//   C[M][N] = A[M][K] * W[N][K]^T   (+ fused epilogue)
//
// Replaces every nn.Linear / 1x1 conv / 3x3 conv the reference's UNet and VAE dispatch to
// cuBLAS / cuDNN (reference ldm/modules/diffusionmodules/openaimodel.py:116-232 ResBlock convs,
// ldm/modules/attention.py:37-64,102-186 projections and GEGLU, ldm/modules/diffusionmodules/
// model.py:82-141 VAE ResnetBlock).
//
// Kernels in this file (all: v_mfma_f32_*_bf16, fp32 accumulate, both operands K-contiguous in LDS with 128-byte rows whose
// 16-byte chunks are XOR-swizzled by ((row >> 1) & 7): ds_read_b128 fragment reads and LDS-DMA writes are conflict free):
//  * gemm_u_kernel ("v5", default): persistent work-item loop, buffer-descriptor LDS-DMA loader, inline-asm fragment
//    reads, load-first / LDS-staged epilogue, on-device autotuned tile / split-K / residency. See its header comment.
//  * gemm_p_kernel ("v3"): the same persistent loop with a flat-address global_load_lds loader and compiler-visible LDS
//    reads; used when an operand exceeds the 2 GiB a 32-bit buffer offset can address, and as an A/B reference.
//  * gemm_glds_kernel ("v2"): one 32x32-MFMA tile per workgroup; serves N < 128 problems (first / last convs' GEMMs,
//    PositionNet) and is the independent implementation `kbench check` compares the others against.
// Operand roles are exchanged w.r.t. the textbook mapping (MFMA "A" = weight rows, "B" = activation rows), so a lane owns
// ONE output row and groups of 4 CONSECUTIVE output columns: 8-byte bf16 stores, bias as float4, no transpose.
// The activation loader optionally performs the im2col gather of a 3x3 convolution over an NHWC tensor (stride 1|2,
// nearest-2x upsample of the source, channel concat of two sources, asymmetric padding), so convs never materialise
// im2col or concat / upsample copies in HBM. Split-K goes through an fp32 slab workspace + a deterministic reduce kernel.
#include "gemm.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>

namespace gl {

void epilogue_defaults(Epilogue& E) {
    E = Epilogue{};
    E.mode = EPI_ROWMAJOR;
    E.act = ACT_NONE;
    E.rows_per_b = 1;
    E.rpb_magic = 1; E.rpb_shift = 0;
}

void aoperand_rows(AOperand& A, const bf16* p, int K, int ld) {
    A = AOperand{};
    A.p0 = p;
    A.C0 = K;
    A.ld0 = ld;
    A.mode = A_ROWS;
}

// tokens are permuted inside groups of 16 so that the attention kernel's P^T fragment (taken
// straight from MFMA accumulators) lines up with one ds_read_b128 of V^T: [0-3,8-11,4-7,12-15].
// p32 (Epilogue::vt_perm32, attn3_kernel): groups of 32, quads in the order [0,2,4,6,1,3,5,7] -- the four 8-key k-slot groups of a
// v_mfma_f32_16x16x32 B operand assembled from a 32x32 S^T accumulator by v_permlane16_swap (attention.hip).
__device__ __forceinline__ int perm_tok4(int t0, int p32 = 0) {
    if (p32) {
        const int g = (t0 >> 2) & 7;
        return (t0 & ~31) | ((((g & 1) << 2) | (g >> 1)) << 2);
    }
    int g = (t0 >> 2) & 3;
    int gp = ((g & 1) << 1) | (g >> 1);
    return (t0 & ~15) | (gp << 2);
}

__device__ __forceinline__ void store_bf16x4(bf16* dst, const float v[4]) {
    U2BF4 o;
    o.e[0] = f2bf(v[0]);
    o.e[1] = f2bf(v[1]);
    o.e[2] = f2bf(v[2]);
    o.e[3] = f2bf(v[3]);
    *reinterpret_cast<uint2*>(dst) = o.u;
}

// m / E.rows_per_b through the multiplier gemm_launch prepared (round-up method: shift = ceil(log2 d), magic = floor(2^32 (2^shift - d) / d) + 1).
// A per-lane "/" by a kernel argument makes hipcc build the reciprocal in VGPRs in front of the K loop and keep it there.
__device__ __forceinline__ int div_rpb(const Epilogue& E, int m) {
    return (int)((__umulhi((unsigned)m, E.rpb_magic) + (unsigned)m) >> E.rpb_shift);
}

__device__ __forceinline__ void epi_finish4(const Epilogue& E, int m, int n0, float v[4]) {
    if (E.bias) {
        float4 b = *reinterpret_cast<const float4*>(E.bias + n0);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (E.bias2) {
        int bb = div_rpb(E, m);
        float4 b = *reinterpret_cast<const float4*>(E.bias2 + (size_t)bb * E.bias2_ld + n0);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (E.act == ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
    } else if (E.act == ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
    }
    switch (E.mode) {
        case EPI_ROWMAJOR: {
            if (E.remap_in) m = (m / E.remap_in) * E.remap_out + (m % E.remap_in) + E.remap_off;
            if (E.res) {
                U2BF4 r;
                r.u = *reinterpret_cast<const uint2*>(E.res + (size_t)m * E.ldres + n0);
                float g = E.gate ? *E.gate : 1.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = bf2f(r.e[i]) + g * v[i];
            }
            if (E.out_f32) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(E.out) + (size_t)m * E.ldo + n0) = o;
            } else {
                store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + n0, v);
            }
            break;
        }
        case EPI_QKV_HEADS:
        case EPI_QK_HEADS: {
            int which = n0 >= E.C;
            int c = n0 - which * E.C;
            int h = c / E.d;
            int dd = c - h * E.d;
            int b = m / E.T;
            int t = m - b * E.T + E.tok_off;
            int tp = which ? E.Tpad_k : E.Tpad_q;
            bf16* base = which ? E.k : E.q;
            const size_t off = (which || E.q_tiled) ? ktile_off((size_t)(b * E.H + h), tp, t, dd, E.DP) : ((size_t)(b * E.H + h) * tp + t) * E.DP + dd;
            store_bf16x4(base + off, v);
            break;
        }
        case EPI_VT_HEADS: {  // m = feature, n0 = first of 4 consecutive tokens
            int h = m / E.d;
            int dd = m - h * E.d;
            int b = n0 / E.T;
            int t0 = n0 - b * E.T + E.tok_off;
            bf16* base = reinterpret_cast<bf16*>(E.out);
            store_bf16x4(base + ((size_t)(b * E.H + h) * E.DPV + dd) * E.Tpad_k + perm_tok4(t0, E.vt_perm32), v);
            break;
        }
        case EPI_NCHW_F32: {
            int b = div_rpb(E, m);
            int pix = m - b * E.rows_per_b;
            float* o = reinterpret_cast<float*>(E.out);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + i < E.n_real) o[((size_t)b * E.n_real + n0 + i) * E.rows_per_b + pix] = v[i];
            break;
        }
    }
}

// Store half of epi_finish4 for epilogues that load bias / residual for ALL their fragments up front
// (v already holds bias and residual; for EPI_ROWMAJOR m is the remapped output row). A wave's vmcnt is one
// in-order counter for loads and stores, so a load issued after a store cannot be waited for without
// waiting for that store: an epilogue that alternates {load bias/residual, store} per fragment pays one
// memory round trip per fragment (measured: 18 of the 27 us of a 32768 x 320 x 320 GEMM).
// WITH_GELU = false: the conv instantiations (never followed by a GELU) keep the erf code out of their register allocation
template <bool WITH_GELU = true>
__device__ __forceinline__ void epi_store4(const Epilogue& E, int m, int n0, float v[4]) {
    if (E.act == ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
    } else if (WITH_GELU && E.act == ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
    }
    switch (E.mode) {
        case EPI_ROWMAJOR: {
            if (E.out_f32) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(E.out) + (size_t)m * E.ldo + n0) = o;
            } else {
                store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + n0, v);
            }
            break;
        }
        case EPI_QKV_HEADS:
        case EPI_QK_HEADS: {
            int which = n0 >= E.C;
            int c = n0 - which * E.C;
            int h = c / E.d;
            int dd = c - h * E.d;
            int b = m / E.T;
            int t = m - b * E.T + E.tok_off;
            int tp = which ? E.Tpad_k : E.Tpad_q;
            bf16* base = which ? E.k : E.q;
            const size_t off = (which || E.q_tiled) ? ktile_off((size_t)(b * E.H + h), tp, t, dd, E.DP) : ((size_t)(b * E.H + h) * tp + t) * E.DP + dd;
            store_bf16x4(base + off, v);
            break;
        }
        case EPI_VT_HEADS: {  // m = feature, n0 = first of 4 consecutive tokens
            int h = m / E.d;
            int dd = m - h * E.d;
            int b = n0 / E.T;
            int t0 = n0 - b * E.T + E.tok_off;
            bf16* base = reinterpret_cast<bf16*>(E.out);
            store_bf16x4(base + ((size_t)(b * E.H + h) * E.DPV + dd) * E.Tpad_k + perm_tok4(t0, E.vt_perm32), v);
            break;
        }
        case EPI_NCHW_F32: {
            int b = div_rpb(E, m);
            int pix = m - b * E.rows_per_b;
            float* o = reinterpret_cast<float*>(E.out);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + i < E.n_real) o[((size_t)b * E.n_real + n0 + i) * E.rows_per_b + pix] = v[i];
            break;
        }
    }
}

// GEGLU: weight/bias rows are pre-packed so that inside every 32-row tile the accumulator
// groups alternate value,gate,value,gate for the same 4 output features (see pack_geglu_rows).
__device__ __forceinline__ void epi_geglu4(const Epilogue& E, int m, int nv0, float val[4], float gate[4]) {
    if (E.bias) {
        float4 bv = *reinterpret_cast<const float4*>(E.bias + nv0);
        float4 bg = *reinterpret_cast<const float4*>(E.bias + nv0 + 8);
        val[0] += bv.x; val[1] += bv.y; val[2] += bv.z; val[3] += bv.w;
        gate[0] += bg.x; gate[1] += bg.y; gate[2] += bg.z; gate[3] += bg.w;
    }
    int j0 = (nv0 >> 5) * 16 + ((nv0 & 31) >> 4) * 8 + (nv0 & 7);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = val[i] * gelu_erf_f(gate[i]);
    store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + j0, o);
}

// Accumulator -> memory for one wave: lane (frow, fhalf) holds output row m0 + 32 i and, per 32x32
// tile, four groups of 4 consecutive columns starting at n0 + 32 j + 8 q.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], const Epilogue& E, int M, int N, int m0, int n0w,
                                              float* __restrict__ ws) {
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0w + j * 32;
            if (split) {
                float* dst = ws + ((size_t)blockIdx.z * M + m) * N;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int n0 = nb + 8 * q;
                    if (n0 < N)
                        *reinterpret_cast<float4*>(dst + n0) =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
            } else if (E.act == ACT_GEGLU) {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    int nv0 = nb + 16 * qq;
                    if (nv0 < N) {
                        float val[4], gate[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            val[e] = acc[i][j][8 * qq + e];
                            gate[e] = acc[i][j][8 * qq + 4 + e];
                        }
                        epi_geglu4(E, m, nv0, val, gate);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int n0 = nb + 8 * q;
                    if (n0 < N) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                        epi_finish4(E, m, n0, v);
                    }
                }
            }
        }
    }
}

// 128 zero bytes: the source of every out-of-range 16-byte chunk (conv padding taps, M / N tails)
// for the LDS-DMA loader below, which cannot write an immediate.
__device__ uint4 g_zero_chunk[8];

#define GL_GLDS16(gsrc, ldst)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),       \
                                     (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

// v2 main loop: both operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction = 8 tile rows x 128 B), no VGPR staging and no ds_write pass. The LDS image a
// wave-instruction writes is lane-linear, so the bank swizzle is applied to the per-lane SOURCE
// chunk (lane at position p of row r fetches logical chunk p ^ ((r>>1)&7)) and undone by the same
// XOR in the fragment reads. Two LDS stages: tile t+1 is in flight while tile t is multiplied.
template <int WM, int WN, int TM, int TN, int AMODE>
__global__ void __launch_bounds__(WM * WN * 64)
gemm_glds_kernel(AOperand A, const bf16* __restrict__ W, int M, int N, int K, Epilogue E,
                 float* __restrict__ ws, int kt_per_split, int tiles_n, int n_tiles) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int RPP = NT / 8;  // tile rows covered by one pass of 16-byte loads
    constexpr int XP = BM / RPP;
    constexpr int WP = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RPP % 16 == 0, "tile/loader mismatch");
    constexpr int STAGE = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int r0 = t >> 3;
    const int ch8 = ((t & 7) ^ ((r0 >> 1) & 7)) * 8;  // element offset of this lane's source chunk

    // XCD-aware tile order: hardware places block b on XCD b % 8; give every XCD a contiguous
    // range of tiles so the blocks that share an activation row-panel share an L2.
    int bid = blockIdx.x;
    {
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    const int m_base = tile_m * BM;
    const int n_base = tile_n * BN;

    const int nk = K >> 6;
    const int kt0 = blockIdx.z * kt_per_split;
    const int kt1 = min(nk, kt0 + kt_per_split);

    const bf16* zsrc = reinterpret_cast<const bf16*>(g_zero_chunk);

    int64_t xo0[XP], xo1[XP];
    int xb[XP], xy[XP], xx[XP];
    bool xv[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        int m = m_base + r0 + i * RPP;
        xv[i] = m < M;
        if constexpr (AMODE == A_ROWS) {
            xo0[i] = (int64_t)m * A.ld0;
            xo1[i] = (int64_t)m * A.ld1;
            xb[i] = xy[i] = xx[i] = 0;
        } else {
            int ox = m % A.Wo;
            int tmp = m / A.Wo;
            int oy = tmp % A.Ho;
            xb[i] = (tmp / A.Ho) * A.Hin;
            xy[i] = oy * A.stride - A.pad_lo;
            xx[i] = ox * A.stride - A.pad_lo;
            xo0[i] = xo1[i] = 0;
        }
    }
    const bf16* wp[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        int n = n_base + r0 + i * RPP;
        wp[i] = n < N ? W + (int64_t)n * K + ch8 : nullptr;
    }
    const int Cin = A.C0 + A.C1;
    const int Hup = A.Hin << A.ups;
    const int Wup = A.Win << A.ups;

    auto issue = [&](int kt, int buf) {
        unsigned char* xs = smem + buf * STAGE + wave * 1024;
        unsigned char* wsm = xs + BM * 128;
        const int k0 = kt << 6;
        if constexpr (AMODE == A_ROWS) {
            const bool first = k0 < A.C0;
            const bf16* base = (first ? A.p0 : A.p1) + (first ? k0 : k0 - A.C0) + ch8;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const bf16* src = xv[i] ? base + (first ? xo0[i] : xo1[i]) : zsrc;
                GL_GLDS16(src, xs + i * RPP * 128);
            }
        } else {
            const int tap = k0 / Cin;
            const int c = k0 - tap * Cin;
            const int ky = tap / 3;
            const int kx = tap - ky * 3;
            const bool first = c < A.C0;
            const bf16* base = (first ? A.p0 : A.p1) + (first ? c : c - A.C0) + ch8;
            const int ld = first ? A.ld0 : A.ld1;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int iy = xy[i] + ky;
                const int ix = xx[i] + kx;
                const bool ok = xv[i] && iy >= 0 && iy < Hup && ix >= 0 && ix < Wup;
                const int64_t pix = ((int64_t)(xb[i] + (iy >> A.ups))) * A.Win + (ix >> A.ups);
                const bf16* src = ok ? base + pix * ld : zsrc;
                GL_GLDS16(src, xs + i * RPP * 128);
            }
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const bf16* src = wp[i] ? wp[i] + k0 : zsrc;
            GL_GLDS16(src, wsm + i * RPP * 128);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fhalf = lane >> 5;

    // fragments of k-step s+1 are fetched from LDS while the MFMAs of k-step s issue
    auto compute = [&](int buf) {
        const unsigned char* xs = smem + buf * STAGE;
        const unsigned char* wsm = xs + BM * 128;
        auto ldfrag = [&](int s, bf16x8 (&xf)[TM], bf16x8 (&wf)[TN]) {
            const int c = 2 * s + fhalf;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int row = (wm * TM + i) * 32 + frow;
                xf[i] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int row = (wn * TN + j) * 32 + frow;
                wf[j] = *reinterpret_cast<const bf16x8*>(wsm + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
        };
        auto mma = [&](const bf16x8 (&xf)[TM], const bf16x8 (&wf)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        };
        bf16x8 xa[TM], wa[TN], xb2[TM], wb2[TN];
        ldfrag(0, xa, wa);
        ldfrag(1, xb2, wb2);
        mma(xa, wa);
        ldfrag(2, xa, wa);
        mma(xb2, wb2);
        ldfrag(3, xb2, wb2);
        mma(xa, wa);
        mma(xb2, wb2);
    };

    if (kt0 < kt1) {
        issue(kt0, 0);
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            // tile kt has landed (own DMA drained, then the barrier covers the other waves') and every
            // wave is done reading the other stage, which the next issue overwrites
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < kt1) issue(kt + 1, buf ^ 1);
            compute(buf);
        }
    }

    gemm_epilogue<TM, TN>(acc, E, M, N, m_base + wm * TM * 32 + frow, n_base + wn * TN * 32 + 4 * fhalf, ws);
}

// GEGLU epilogue for the 16x16-tile kernel: packed weight rows come in groups of 32 = 16 value
// features followed by the 16 matching gate features (pack_geglu layout 1), so accumulator tile
// 2g holds values and tile 2g+1 gates for the same (row, 4 features) in the same lane.
__device__ __forceinline__ void epi_geglu4_t16(const Epilogue& E, int m, int nv0, float val[4], float gate[4]) {
    if (E.bias) {
        float4 bv = *reinterpret_cast<const float4*>(E.bias + nv0);
        float4 bg = *reinterpret_cast<const float4*>(E.bias + nv0 + 16);
        val[0] += bv.x; val[1] += bv.y; val[2] += bv.z; val[3] += bv.w;
        gate[0] += bg.x; gate[1] += bg.y; gate[2] += bg.z; gate[3] += bg.w;
    }
    const int j0 = (nv0 >> 5) * 16 + (nv0 & 15);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = val[i] * gelu_erf_f(gate[i]);
    store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + j0, o);
}

// ---------------------------------------------------------------------------------------------
// v3: persistent tile loop. grid = min(#work items, 2 blocks x 256 CUs); a block walks work items
// (output tile x K split) item = blockIdx.x, + gridDim.x, ... and keeps ONE software pipeline
// running across item boundaries: the LDS-DMA of the next item's first K tile is issued before
// the last MFMAs of the current item, so its latency (and the other resident block's epilogue)
// hides behind matrix work instead of serialising with it. Tiles are 2x2 waves of TM x TN
// v_mfma_f32_16x16x32_bf16 fragments: BN = 160 divides every UNet channel count (320 / 640 /
// 960 / 1280 / 1920 / 2560 ...) exactly, BN = 128 serves GEGLU pairs and the VAE.
struct WorkDesc {
    int tiles_n;
    int splits;
    int kt_per_split;
    int n_items;
    // XCD partition of the item space. box < 0: contiguous ranges of the linear (tm, tn, z) order. Otherwise the 8 XCDs
    // form a 2^lgm x 2^lgn x 2^lgz grid over (M tiles, N tiles, K splits), box = lgm | lgn << 4, and each owns an
    // rm x tiles_n x rz box (tiles_n = N tiles PER BOX then), so that an activation panel is fetched by 2^lgn L2s and a
    // weight panel by 2^lgm (the K axis duplicates nothing). rz = splits when box < 0.
    int box, rm, rz;
};

template <int TM, int TN, int AMODE>
__global__ void __launch_bounds__(256, 2)
gemm_p_kernel(AOperand A, const bf16* __restrict__ W, int M, int N, int K, Epilogue E, float* __restrict__ ws, WorkDesc wd) {
    constexpr int BM = TM * 32;
    constexpr int BN = TN * 32;
    constexpr int XP = BM / 32;
    constexpr int WP = BN / 32;
    constexpr int STAGE = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const int r0 = t >> 3;
    const int ch8 = ((t & 7) ^ ((r0 >> 1) & 7)) * 8;  // element offset of this lane's source chunk
    const int nk = K >> 6;
    const bf16* zsrc = reinterpret_cast<const bf16*>(g_zero_chunk);
    const int Cin = A.C0 + A.C1;
    const int Hup = A.Hin << A.ups;
    const int Wup = A.Win << A.ups;

    // work item -> (tile_m, tile_n, split); items are renumbered so every XCD (block b sits on XCD b % 8,
    // and gridDim.x % 8 == 0 whenever a block runs more than one item) walks a contiguous range
    auto decode = [&](int w, int& tm, int& tn, int& z) {
        const int q = wd.n_items >> 3, r = wd.n_items & 7, xcd = w & 7, idx = w >> 3;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int tile = id / wd.splits;
        z = id - tile * wd.splits;
        tm = tile / wd.tiles_n;
        tn = tile - tm * wd.tiles_n;
    };

    // ---- load cursor
    int l_item = blockIdx.x, l_kt = 0, l_kt_end = 0;
    int xm[XP], xy[XP], xx[XP];  // A_ROWS: xm = row or -1.  A_CONV3: xm = b * Hin, (xy, xx) = top-left tap
    int wrow[WP];
    auto setup_load = [&](int item) {
        int tm, tn, z;
        decode(item, tm, tn, z);
        l_kt = z * wd.kt_per_split;
        l_kt_end = min(nk, l_kt + wd.kt_per_split);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int m = tm * BM + r0 + i * 32;
            const bool ok = m < M;
            if constexpr (AMODE == A_ROWS) {
                xm[i] = ok ? m : -1;
                xy[i] = xx[i] = 0;
            } else {
                const int ox = m % A.Wo;
                const int tmp = m / A.Wo;
                const int oy = tmp % A.Ho;
                xm[i] = (tmp / A.Ho) * A.Hin;
                xy[i] = ok ? oy * A.stride - A.pad_lo : -(1 << 20);
                xx[i] = ox * A.stride - A.pad_lo;
            }
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int n = tn * BN + r0 + i * 32;
            wrow[i] = n < N ? n : -1;
        }
    };

    auto issue = [&](int kt, int buf) {
        unsigned char* xs = smem + buf * STAGE + wave * 1024;
        unsigned char* wsm = xs + BM * 128;
        const int k0 = kt << 6;
        if constexpr (AMODE == A_ROWS) {
            const bool first = k0 < A.C0;
            const bf16* base = (first ? A.p0 : A.p1) + (first ? k0 : k0 - A.C0) + ch8;
            const int ld = first ? A.ld0 : A.ld1;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const bf16* src = xm[i] >= 0 ? base + (int64_t)xm[i] * ld : zsrc;
                GL_GLDS16(src, xs + i * 4096);
            }
        } else {
            const int tap = k0 / Cin;
            const int c = k0 - tap * Cin;
            const int ky = tap / 3;
            const int kx = tap - ky * 3;
            const bool first = c < A.C0;
            const bf16* base = (first ? A.p0 : A.p1) + (first ? c : c - A.C0) + ch8;
            const int ld = first ? A.ld0 : A.ld1;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int iy = xy[i] + ky;
                const int ix = xx[i] + kx;
                const bool ok = iy >= 0 && iy < Hup && ix >= 0 && ix < Wup;
                const int64_t pix = ((int64_t)(xm[i] + (iy >> A.ups))) * A.Win + (ix >> A.ups);
                const bf16* src = ok ? base + pix * ld : zsrc;
                GL_GLDS16(src, xs + i * 4096);
            }
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const bf16* src = wrow[i] >= 0 ? W + (int64_t)wrow[i] * K + k0 + ch8 : zsrc;
            GL_GLDS16(src, wsm + i * 4096);
        }
    };

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // fragment addressing: lane reads row (16 t + l15), 16-byte chunk 4 s + (lane >> 4); rows start at
    // multiples of 16, so the swizzle term ((row >> 1) & 7) is the lane constant l15 >> 1
    const int l15 = lane & 15;
    const int foff0 = l15 * 128 + ((((lane >> 4)) ^ (l15 >> 1)) << 4);
    const int foff1 = l15 * 128 + ((((lane >> 4) + 4) ^ (l15 >> 1)) << 4);
    const int xrow0 = wm * TM * 16 * 128;
    const int wrow0 = BM * 128 + wn * TN * 16 * 128;

    auto compute = [&](int buf) {
        const unsigned char* st = smem + buf * STAGE;
        bf16x8 xa[TM], wa[TN], xb[TM], wb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(st + xrow0 + i * 2048 + foff0);
#pragma unroll
        for (int j = 0; j < TN; ++j) wa[j] = *reinterpret_cast<const bf16x8*>(st + wrow0 + j * 2048 + foff0);
#pragma unroll
        for (int i = 0; i < TM; ++i) xb[i] = *reinterpret_cast<const bf16x8*>(st + xrow0 + i * 2048 + foff1);
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(st + wrow0 + j * 2048 + foff1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xa[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xb[i], acc[i][j], 0, 0, 0);
    };

    auto epilogue = [&](int tm, int tn, int z) {
        const int mrow = tm * BM + wm * TM * 16 + l15;
        const int ncol = tn * BN + wn * TN * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mrow + i * 16;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n0 = ncol + j * 16;
                if (n0 >= N) continue;
                if (wd.splits > 1) {
                    *reinterpret_cast<float4*>(ws + ((size_t)z * M + m) * N + n0) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else if (E.act == ACT_GEGLU) {
                    if constexpr (TN % 2 == 0) {
                        if ((j & 1) == 0) {
                            float val[4], gate[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                val[e] = acc[i][j][e];
                                gate[e] = acc[i][j + 1 < TN ? j + 1 : j][e];
                            }
                            epi_geglu4_t16(E, m, n0, val, gate);
                        }
                    }
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                    epi_finish4(E, m, n0, v);
                }
            }
        }
    };

    // ---- pipeline over the flattened (item, K tile) sequence
    if (l_item >= wd.n_items) return;
    setup_load(l_item);
    int c_item = l_item, c_tm, c_tn, c_z;
    decode(c_item, c_tm, c_tn, c_z);
    int c_left = l_kt_end - l_kt;
    issue(l_kt, 0);
    int buf = 0;
    for (;;) {
        // tile in stage `buf` has landed (own DMA drained, barrier covers the other waves') and every wave
        // has finished reading stage buf^1, which the next issue overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        bool more = true;
        if (++l_kt >= l_kt_end) {
            l_item += gridDim.x;
            more = l_item < wd.n_items;
            if (more) setup_load(l_item);
        }
        if (more) issue(l_kt, buf ^ 1);
        compute(buf);
        buf ^= 1;
        if (--c_left == 0) {
            epilogue(c_tm, c_tn, c_z);
            c_item += gridDim.x;
            if (c_item >= wd.n_items) break;
            zero_acc();
            decode(c_item, c_tm, c_tn, c_z);
            c_left = min(nk, (c_z + 1) * wd.kt_per_split) - c_z * wd.kt_per_split;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// v5 (unified persistent kernel, variant 4): 4 waves (2 M x 2 N), (TM*32) x (TN*32) tile, two LDS stages, two to four workgroups
// per CU. (Rounds 1-2 carried an 8-wave / three-stage geometry and a four-stage ring at one workgroup per CU in the same code;
// both lost every A/B and every autotune, DESIGN.md section 4, and were taken out in round 3.)
// What changed against v3/v4 is the loader. PMC showed v3 spending ~300 VALU/SALU instructions per
// K tile per wave on 64-bit DMA source addresses against 20-40 MFMAs (SQ_ACTIVE_INST_ANY ~ the
// whole budget of a wave, MFMA pipe 28 % busy), so both operands are fetched with
// buffer_load_dwordx4 ... lds through wave-uniform buffer descriptors: the per-lane byte offset
// (row, swizzled 16-byte chunk) is computed once per work item (conv: once per filter tap), the
// K-tile offset rides in the SGPR soffset, and a DMA costs one s_mov m0 + one buffer_load. Rows
// outside M / N and conv padding taps use an offset beyond num_records, for which the hardware
// returns zeros.
// LDS fragment reads in inline asm. hipcc cannot tell that a ds_read of one ring slot does not alias the LDS-DMA
// just issued into another, so with compiler-visible LDS loads it puts s_waitcnt vmcnt(0) in front of the first
// ds_read after every DMA issue: the "prefetch" is drained before the multiply it should run under (seen in the
// ISA, and in the ablation as DMA time + MFMA time adding up exactly). Asm reads are invisible to that pass; their
// completion is waited for by hand (counted lgkmcnt, LDS returns in order) and the consumers are pinned below the
// wait through "+v" operands (an MFMA is register-only, a "memory" clobber does not order it).
template <int OFF>
__device__ __forceinline__ void lds_rd16(bf16x8& d, unsigned addr) {
    // no "memory" clobber: an asm that "may touch memory" makes the waitcnt pass drain the pending DMA exactly like a
    // visible LDS load would; volatile keeps it ordered against the barrier, the DMA issues and the other asm statements
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N, int STRIDE, int I = 0>
__device__ __forceinline__ void lds_rd16_n(bf16x8 (&d)[N], unsigned addr) {
    if constexpr (I < N) {
        lds_rd16<I * STRIDE>(d[I], addr);
        lds_rd16_n<N, STRIDE, I + 1>(d, addr);
    }
}
// "this value has arrived": an empty asm that reads (and re-defines) a loaded value makes hipcc place the wait for it HERE, once,
// instead of in every guarded block that uses it later
__device__ __forceinline__ void epi_ready(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <int N>
__device__ __forceinline__ void pin_regs(bf16x8 (&d)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(d[i]));
}

#define GL_BLDS16(rsrc, ldst, voff, soff)                                                                   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(ldst), 16, \
                                             (int)(voff), (int)(soff), 0, 0)

// QKV (EPI_QKV_HEADS, A_ROWS only): its own instantiations, so that the other kernels' register allocation is untouched
template <int WMW, int TM, int TN, int AMODE, int NST, bool QKV = false>
__global__ void __launch_bounds__(WMW * 128, 2)
gemm_u_kernel(AOperand A, const bf16* __restrict__ W, int M, int N, int K, Epilogue E, float* __restrict__ ws, WorkDesc wd) {
    constexpr int NT = WMW * 128;               // threads
    constexpr int RPP = NT / 8;                 // tile rows per DMA pass (one 128-byte row per 8 lanes)
    constexpr int BM = WMW * TM * 16;
    constexpr int BN = TN * 32;
    constexpr int XP = BM / RPP;                // activation passes
    constexpr int WP = BN / RPP;                // weight passes
    constexpr int STAGE = (BM + BN) * 128;      // two stages: the two (to four) 4-wave workgroups of a CU hide each other's DMA latency
    constexpr unsigned SENT = 0x80000000u;      // >= num_records of every descriptor: reads as zeros
    static_assert(WMW == 2 && NST == 2, "one geometry: 4 waves, two LDS stages (deeper rings and the 8-wave form lost every A/B, DESIGN.md section 4)");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/loader mismatch");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const int r0 = t >> 3;                                          // row inside a DMA pass
    const unsigned chb = (((t & 7) ^ ((r0 >> 1) & 7)) * 16);        // byte offset of this lane's (swizzled) source chunk
    const int nk = K >> 6;
    const int Cin = A.C0 + A.C1;
    const int Hup = A.Hin << A.ups;
    const int Wup = A.Win << A.ups;

    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)A.p0, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(A.p1 ? A.p1 : A.p0), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x80000000u, 0x00020000);

    auto decode = [&](int w, int& tm, int& tn, int& z) {
        const int xcd = w & 7, idx = w >> 3;
        int id = idx;
        if (wd.box < 0) {
            const int q = wd.n_items >> 3, r = wd.n_items & 7;
            id += xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        }
        const int tile = id / wd.rz;
        z = id - tile * wd.rz;
        tm = tile / wd.tiles_n;
        tn = tile - tm * wd.tiles_n;
        if (wd.box >= 0) {
            const int lgm = wd.box & 15, lgn = wd.box >> 4;
            z += (xcd >> (lgm + lgn)) * wd.rz;
            tn += ((xcd >> lgm) & ((1 << lgn) - 1)) * wd.tiles_n;
            tm += (xcd & ((1 << lgm) - 1)) * wd.rm;
        }
    };

    // ---- load cursor: (item, K tile) plus, for the conv gather, the filter tap and channel offset of that tile
    int l_item = blockIdx.x, l_kt = 0, l_kt_end = 0;
    int l_tap = 0, l_cc = 0;     // A_CONV3: k = l_tap * Cin + l_cc.   A_ROWS: l_cc = k
    int xm[XP], xy[XP], xx[XP];  // A_ROWS: xm = row or -1.  A_CONV3: xm = b * Hin, (xy, xx) = top-left tap (xy very negative: no row)
    unsigned va[XP], vw[WP];     // per-lane byte offsets into the current activation source / the weight matrix

    // byte offsets of this lane's activation chunks for the current (tap, source)
    auto a_offsets = [&]() {
        const bool first = l_cc < A.C0;
        const int ld = first ? A.ld0 : A.ld1;
        if constexpr (AMODE == A_ROWS) {
#pragma unroll
            for (int i = 0; i < XP; ++i) va[i] = xm[i] >= 0 ? (unsigned)(xm[i] * ld) * 2u + chb : SENT;
        } else {
            const int ky = l_tap / 3;
            const int kx = l_tap - ky * 3;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int iy = xy[i] + ky;
                const int ix = xx[i] + kx;
                const bool ok = iy >= 0 && iy < Hup && ix >= 0 && ix < Wup;
                const int pix = (xm[i] + (iy >> A.ups)) * A.Win + (ix >> A.ups);
                va[i] = ok ? (unsigned)(pix * ld) * 2u + chb : SENT;
            }
        }
    };

    auto setup_load = [&](int item) {
        int tm, tn, z;
        decode(item, tm, tn, z);
        l_kt = z * wd.kt_per_split;
        l_kt_end = min(nk, l_kt + wd.kt_per_split);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int m = tm * BM + r0 + i * RPP;
            const bool ok = m < M;
            if constexpr (AMODE == A_ROWS) {
                xm[i] = ok ? m : -1;
                xy[i] = xx[i] = 0;
            } else {
                const int ox = m % A.Wo;
                const int tmp = m / A.Wo;
                const int oy = tmp % A.Ho;
                xm[i] = (tmp / A.Ho) * A.Hin;
                xy[i] = ok ? oy * A.stride - A.pad_lo : -(1 << 20);
                xx[i] = ox * A.stride - A.pad_lo;
            }
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int n = tn * BN + r0 + i * RPP;
            vw[i] = (n < N) ? (unsigned)(n * K) * 2u + chb : SENT;
        }
        const int k0 = l_kt << 6;
        if constexpr (AMODE == A_ROWS) {
            l_tap = 0;
            l_cc = k0;
        } else {
            l_tap = k0 / Cin;
            l_cc = k0 - l_tap * Cin;
        }
        a_offsets();
    };

    // DMA of K tile l_kt into ring slot `slot`, then advance the cursor by one tile
    bool more = true;
    auto issue_next = [&](int slot_) {
        // the cursor is wave-uniform by construction; say so, or one value merged through a divergent
        // branch makes hipcc wrap every buffer_load in a waterfall loop (descriptor / soffset / m0 "divergent")
        const int slot = __builtin_amdgcn_readfirstlane(slot_);
        l_cc = __builtin_amdgcn_readfirstlane(l_cc);
        l_kt = __builtin_amdgcn_readfirstlane(l_kt);
        unsigned char* xs = smem + slot * STAGE + wave * 1024;
        unsigned char* wsm = xs + BM * 128;
        const bool first = l_cc < A.C0;
        const int soff = (first ? l_cc : l_cc - A.C0) * 2;
        if (first) {
#pragma unroll
            for (int i = 0; i < XP; ++i) GL_BLDS16(ra0, xs + i * RPP * 128, va[i], soff);
        } else {
#pragma unroll
            for (int i = 0; i < XP; ++i) GL_BLDS16(ra1, xs + i * RPP * 128, va[i], soff);
        }
        const int woff = l_kt << 7;
#pragma unroll
        for (int i = 0; i < WP; ++i) GL_BLDS16(rw, wsm + i * RPP * 128, vw[i], woff);
        // advance
        l_cc += 64;
        if (++l_kt >= l_kt_end) {
            l_item += gridDim.x;
            more = l_item < wd.n_items;
            if (more) setup_load(l_item);
        } else if constexpr (AMODE == A_CONV3) {
            if (l_cc == Cin) {
                l_cc = 0;
                ++l_tap;
                a_offsets();
            } else if (l_cc == A.C0) {
                a_offsets();
            }
        } else {
            if (A.C1 && l_cc == A.C0) a_offsets();
        }
    };

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // fragment addressing: lane reads row (16 t + l15), 16-byte chunk 4 s + (lane >> 4); rows start at
    // multiples of 16, so the swizzle term ((row >> 1) & 7) is the lane constant l15 >> 1
    const int l15 = lane & 15;
    const int foff0 = l15 * 128 + ((((lane >> 4)) ^ (l15 >> 1)) << 4);
    const int foff1 = l15 * 128 + ((((lane >> 4) + 4) ^ (l15 >> 1)) << 4);
    const int xrow0 = wm * TM * 16 * 128;
    const int wrow0 = BM * 128 + wn * TN * 16 * 128;

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // swapv (wave-uniform, EPI_QKV_HEADS items in the V third): MFMA operand roles exchanged -- lane (l15, q) then holds
    // feature l15 and tokens 4q .. 4q+3 of every 16x16 tile (the v^T store pattern) instead of token l15, features 4q .. 4q+3
    auto compute = [&](int slot, bool swapv) {
        const unsigned st = lds0 + slot * STAGE;
        const unsigned ax0 = st + xrow0 + foff0, ax1 = st + xrow0 + foff1;
        const unsigned aw0 = st + wrow0 + foff0, aw1 = st + wrow0 + foff1;
        bf16x8 xa[TM], wa[TN], xb[TM], wb[TN];
        lds_rd16_n<TM, 2048>(xa, ax0);
        lds_rd16_n<TN, 2048>(wa, aw0);
        lds_rd16_n<TM, 2048>(xb, ax1);
        lds_rd16_n<TN, 2048>(wb, aw1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN));  // the k-step 0 fragments have returned
        pin_regs(xa);
        pin_regs(wa);
        __builtin_amdgcn_sched_barrier(0);
        if (!QKV || !swapv) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xa[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i], wa[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        pin_regs(xb);
        pin_regs(wb);
        __builtin_amdgcn_sched_barrier(0);
        if (!QKV || !swapv) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xb[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[i], wb[j], acc[i][j], 0, 0, 0);
        }
    };

    // Folded LayerNorm (gemm.h, Epilogue::ln_stats): (mean, rstd) of the item's BM rows from the producer's partial sums, into the
    // ring slot whose K tile was multiplied last. TPR threads per row add the partials in a fixed order.
    // ln_fetch (when the compute cursor enters an item: the loads fly under its K loop) + ln_prepare (in front of its epilogue)
    // (the loaded pairs stay in registers untouched until ln_prepare: an add right behind the load would make hipcc wait for it there)
    constexpr int LNV = 5;
    float2 ln_v[LNV];
    float ln_s = 0.f, ln_q = 0.f;
    auto ln_fetch = [&](int tm) {
        constexpr int TPR = NT / BM;
        const int row = t / TPR, part = t % TPR;
        const int m = tm * BM + row;
        ln_s = ln_q = 0.f;
#pragma unroll
        for (int i = 0; i < LNV; ++i) {
            const int nb = part + i * TPR;
            ln_v[i] = make_float2(0.f, 0.f);
            if (m < M && nb < E.ln_nb) ln_v[i] = E.ln_stats[(size_t)m * E.ln_ld + nb];
        }
        if (m < M)
            for (int nb = part + LNV * TPR; nb < E.ln_nb; nb += TPR) {   // (more than LNV * TPR column blocks: the 16x16 / 8x8 levels)
                const float2 v = E.ln_stats[(size_t)m * E.ln_ld + nb];
                ln_s += v.x; ln_q += v.y;
            }
    };
    auto ln_prepare = [&](int tm, int slot_done) -> const float2* {
        float2* lst = reinterpret_cast<float2*>(smem + slot_done * STAGE);
        constexpr int TPR = NT / BM;
        __builtin_amdgcn_s_barrier();          // every wave is done with the fragments of that slot
        const int row = t / TPR, part = t % TPR;
        float s = ln_s, q = ln_q;
#pragma unroll
        for (int i = 0; i < LNV; ++i) { s += ln_v[i].x; q += ln_v[i].y; }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (part == 0) {
            const float mean = s * E.ln_inv_c;
            const float var = fmaxf(q * E.ln_inv_c - mean * mean, 0.f);
            lst[row] = make_float2(mean, rsqrtf(var + E.ln_eps));
        }
        __syncthreads();
        return lst;
    };

    // q / k scatter of a QKV item outside the V third (the EPI_QK_HEADS store, no residual forms; a bias only with the folded LayerNorm)
    // Both head-layout epilogues fetch every coefficient (bias, csum, row statistics) BEFORE their first store and make hipcc wait
    // for them right there (epi_ready): vmcnt counts stores too, so a coefficient load between stores -- or a wait the compiler
    // re-inserts in every guarded block because the guard's other path has not waited -- makes each store wait for the
    // acknowledgement of the one before it (28 of the 32 stores of an item did, round 3).
    auto epilogue_qk = [&](int tm, int tn, const float2* lst) {
        const int mrow = tm * BM + wm * TM * 16 + l15;
        const int ncol = tn * BN + wn * TN * 16 + (lane >> 4) * 4;
        float2 st[TM];
        int rb[TM], rt[TM];   // sample and token of this lane's row in each row block
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            st[i] = lst ? lst[wm * TM * 16 + i * 16 + l15] : make_float2(0.f, 1.f);
            const int m = mrow + i * 16;
            rb[i] = m / E.T;
            rt[i] = m - rb[i] * E.T + E.tok_off;
        }
        float4 cs[TN], bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n0 = ncol + j * 16;
            cs[j] = bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 < N) {
                if (lst) cs[j] = *reinterpret_cast<const float4*>(E.ln_csum + n0);
                if (E.bias) bj[j] = *reinterpret_cast<const float4*>(E.bias + n0);   // (W beta of a folded LayerNorm, with or without statistics)
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) { epi_ready(cs[j]); epi_ready(bj[j]); }
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(st[i].x), "+v"(st[i].y));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n0 = ncol + j * 16;
            if (n0 >= N) continue;
            const int which = n0 >= E.C;
            const int cc = n0 - which * E.C;
            const int h = cc / E.d;
            const int dd = cc - h * E.d;
            bf16* base = which ? E.k : E.q;
            const int tp = which ? E.Tpad_k : E.Tpad_q;
            const bool tiled = which || E.q_tiled;   // keys: key-tile layout (gemm.h ktile_off)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (mrow + i * 16 >= M) continue;
                const int b = rb[i], t = rt[i];
                float v[4] = {st[i].y * (acc[i][j][0] - st[i].x * cs[j].x) + bj[j].x, st[i].y * (acc[i][j][1] - st[i].x * cs[j].y) + bj[j].y,
                              st[i].y * (acc[i][j][2] - st[i].x * cs[j].z) + bj[j].z, st[i].y * (acc[i][j][3] - st[i].x * cs[j].w) + bj[j].w};
                const size_t off = tiled ? ktile_off((size_t)(b * E.H + h), tp, t, dd, E.DP) : ((size_t)(b * E.H + h) * tp + t) * E.DP + dd;
                store_bf16x4(base + off, v);
            }
            __builtin_amdgcn_sched_barrier(0);   // addresses just in time: hoisting all TM x TN of them costs registers (spills at 4 x 5)
        }
    };

    // v^T store of an operand-swapped item (EPI_QKV_HEADS): feature n = column, 4 consecutive tokens per lane
    auto epilogue_vt = [&](int tm, int tn, const float2* lst) {
        const int nfeat = tn * BN + wn * TN * 16 + l15;
        const int mtok = tm * BM + wm * TM * 16 + (lane >> 4) * 4;
        float csv[TN], bnv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nfeat + j * 16;
            csv[j] = (lst && n < N) ? E.ln_csum[n] : 0.f;
            bnv[j] = (E.bias && n < N) ? E.bias[n] : 0.f;
        }
        float4 s01[TM], s23[TM];   // (mean, rstd) of the lane's four tokens of each row block: rows (wm TM + i) 16 + 4 q .. + 3 of the item
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            s01[i] = make_float4(0.f, 1.f, 0.f, 1.f);
            s23[i] = s01[i];
            if (lst) {
                s01[i] = *reinterpret_cast<const float4*>(lst + wm * TM * 16 + i * 16 + (lane >> 4) * 4);
                s23[i] = *reinterpret_cast<const float4*>(lst + wm * TM * 16 + i * 16 + (lane >> 4) * 4 + 2);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(csv[j]), "+v"(bnv[j]));
#pragma unroll
        for (int i = 0; i < TM; ++i) { epi_ready(s01[i]); epi_ready(s23[i]); }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nfeat + j * 16;
            if (n >= N) continue;
            const int cfeat = n - 2 * E.C;
            const int h = cfeat / E.d;
            const int dd = cfeat - h * E.d;
            const float cs = csv[j], bn = bnv[j];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m0 = mtok + i * 16;
                if (m0 >= M) continue;
                const int b = m0 / E.T;
                const int t0 = m0 - b * E.T + E.tok_off;
                float v[4] = {s01[i].y * (acc[i][j][0] - s01[i].x * cs) + bn, s01[i].w * (acc[i][j][1] - s01[i].z * cs) + bn,
                              s23[i].y * (acc[i][j][2] - s23[i].x * cs) + bn, s23[i].w * (acc[i][j][3] - s23[i].z * cs) + bn};
                store_bf16x4(E.vt + ((size_t)(b * E.H + h) * E.DPV + dd) * E.Tpad_k + perm_tok4(t0, E.vt_perm32), v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Epilogue: every load (bias, time-embedding bias, residual, gate) of the wave's TM x TN fragments is issued
    // before the first store (see epi_store4).
    auto epilogue = [&](int tm, int tn, int z) {
        const int mrow = tm * BM + wm * TM * 16 + l15;
        const int ncol = tn * BN + wn * TN * 16 + (lane >> 4) * 4;
        if (wd.splits > 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    if (n0 >= N) continue;
                    *reinterpret_cast<float4*>(ws + ((size_t)z * M + m) * N + n0) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
            return;
        }
        if (E.act == ACT_GEGLU) {
            if constexpr (TN % 2 == 0) {
                float4 bv[TN / 2], bg[TN / 2];
#pragma unroll
                for (int jj = 0; jj < TN / 2; ++jj) {
                    const int n0 = ncol + jj * 32;
                    bv[jj] = bg[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (E.bias && n0 < N) {
                        bv[jj] = *reinterpret_cast<const float4*>(E.bias + n0);
                        bg[jj] = *reinterpret_cast<const float4*>(E.bias + n0 + 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = mrow + i * 16;
                    if (m >= M) continue;
#pragma unroll
                    for (int jj = 0; jj < TN / 2; ++jj) {
                        const int n0 = ncol + jj * 32;
                        if (n0 >= N) continue;
                        const float bvv[4] = {bv[jj].x, bv[jj].y, bv[jj].z, bv[jj].w};
                        const float bgg[4] = {bg[jj].x, bg[jj].y, bg[jj].z, bg[jj].w};
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const f32x2 r = geglu2(f32x2{acc[i][2 * jj][e] + bvv[e], acc[i][2 * jj][e + 1] + bvv[e + 1]},
                                                   f32x2{acc[i][2 * jj + 1][e] + bgg[e], acc[i][2 * jj + 1][e + 1] + bgg[e + 1]});
                            o[e] = r.x;
                            o[e + 1] = r.y;
                        }
                        const int j0 = (n0 >> 5) * 16 + (n0 & 15);
                        store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + j0, o);
                    }
                }
            }
            return;
        }
        float4 bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n0 = ncol + j * 16;
            bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (E.bias && n0 < N) bj[j] = *reinterpret_cast<const float4*>(E.bias + n0);
        }
        int mo[TM];  // output row (EPI_ROWMAJOR row remap applied)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mrow + i * 16;
            mo[i] = (AMODE == A_ROWS && E.mode == EPI_ROWMAJOR && E.remap_in) ? (m / E.remap_in) * E.remap_out + (m % E.remap_in) + E.remap_off : m;
        }
        if (AMODE == A_CONV3 && E.bias2) {  // + broadcast per-sample bias (ResBlock time embedding); never combined with a residual; convs only (gemm_p_launch)
            float4 b2[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                const int bb = div_rpb(E, m);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    b2[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < M && n0 < N) b2[i][j] = *reinterpret_cast<const float4*>(E.bias2 + (size_t)bb * E.bias2_ld + n0);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    if (n0 >= N) continue;
                    float v[4] = {acc[i][j][0] + bj[j].x + b2[i][j].x, acc[i][j][1] + bj[j].y + b2[i][j].y,
                                  acc[i][j][2] + bj[j].z + b2[i][j].z, acc[i][j][3] + bj[j].w + b2[i][j].w};
                    epi_store4<AMODE == A_ROWS>(E, mo[i], n0, v);
                }
            }
        } else if (E.mode == EPI_ROWMAJOR && E.res) {
            // (the fallback of epilogue_staged: fp32 / NCHW outputs, GELU, ragged N. The 128 x 160 tile fetches the residual one row
            // block at a time -- all 20 pieces at once do not fit its registers)
            constexpr bool ROWWISE = TM == 4 && TN == 5;
            uint2 rs[TM][TN];
            const float g = (AMODE == A_ROWS && E.gate) ? *E.gate : 1.f;
            auto fetch_row = [&](int i) {
                const int m = mrow + i * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    rs[ROWWISE ? 0 : i][j] = make_uint2(0, 0);
                    if (m < M && n0 < N) rs[ROWWISE ? 0 : i][j] = *reinterpret_cast<const uint2*>(E.res + (size_t)mo[i] * E.ldres + n0);
                }
            };
            if constexpr (!ROWWISE) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fetch_row(i);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                if constexpr (ROWWISE) fetch_row(i);
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    if (n0 >= N) continue;
                    U2BF4 r;
                    r.u = rs[ROWWISE ? 0 : i][j];
                    float v[4] = {acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w};
                    if (E.act == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(r.e[e]) + g * v[e];
                    if (E.out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(E.out) + (size_t)mo[i] * E.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)mo[i] * E.ldo + n0, v);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    if (n0 >= N) continue;
                    float v[4] = {acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w};
                    epi_store4<AMODE == A_ROWS>(E, mo[i], n0, v);
                }
            }
        }
    };

    // Row-major bf16 outputs leave through LDS: a fragment-layout store writes 16 rows x 32 B per instruction (measured
    // 2.7 TB/s for the tile set of one 32768 x 320 output, tools/storebench.hip), whole-row pieces of 16 B per lane reach 3.9,
    // and the residual is read the same way. Each wave stages one 16-row fragment row of its sub-tile at a time, in fp32
    // (single rounding, as before), in a private piece of the ring slot whose K tile was multiplied last.
    constexpr int WCOLS = TN * 16;                 // columns of a wave's sub-tile
    constexpr int LPR = WCOLS / 8;                 // lanes per staged row (8 columns each)
    constexpr int RPI = 64 / LPR;                  // rows per read instruction
    constexpr int NRI = (16 + RPI - 1) / RPI;      // read instructions per 16-row fragment row
    constexpr int RS = WCOLS * 4 + 16;             // staged row stride (bytes): +16 keeps the b128 writes conflict-free
    constexpr int EPW = 16 * RS;                   // bytes per wave
    static_assert(WMW * 2 * (EPW + 16 * LPR * 8) <= STAGE, "epilogue staging (+ row-statistics scratch) does not fit one ring slot");
    auto epilogue_staged = [&](int tm, int tn, int slot_done) -> bool {
        if (wd.splits > 1 || E.mode != EPI_ROWMAJOR || E.out_f32 || (N & 7) || (E.act == ACT_GEGLU && (TN & 1)) || E.act == ACT_GELU) return false;
        if (E.res && E.act == ACT_SILU) return false;   // (no caller combines them; the fragment-layout epilogue handles it)
        unsigned char* ep = smem + slot_done * STAGE + wave * EPW;
        if constexpr ((TN & 1) == 0) {
            // value / gate fragments alternate (16 columns each), so a wave's TN*16 accumulator columns become TN*8 output
            // columns: val * gelu(gate) is formed in the fragment layout and staged, then leaves as 16 B per lane
            if (E.act == ACT_GEGLU) {
                constexpr int OW = TN * 8;                 // output columns of the wave
                constexpr int LPG = OW / 8;                // lanes per staged row
                constexpr int RSG = OW * 4 + 16;
                if (E.res || E.bias2 || E.remap_in || ((N >> 1) & 7)) return false;
                const int mrow = tm * BM + wm * TM * 16;
                const int ncb = tn * BN + wn * WCOLS;      // first accumulator column (a multiple of 32)
                const int rr = lane / LPG, cc = lane - rr * LPG;
                const int ocol = (ncb >> 1) + cc * 8;
                __builtin_amdgcn_s_barrier();
                float4 bv[TN / 2], bg[TN / 2];
#pragma unroll
                for (int jj = 0; jj < TN / 2; ++jj) {
                    const int n0 = ncb + jj * 32 + (lane >> 4) * 4;
                    bv[jj] = bg[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (E.bias && n0 < N) {
                        bv[jj] = *reinterpret_cast<const float4*>(E.bias + n0);
                        bg[jj] = *reinterpret_cast<const float4*>(E.bias + n0 + 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int jj = 0; jj < TN / 2; ++jj) {
                        const f32x2 r0 = geglu2(f32x2{acc[i][2 * jj][0] + bv[jj].x, acc[i][2 * jj][1] + bv[jj].y},
                                                f32x2{acc[i][2 * jj + 1][0] + bg[jj].x, acc[i][2 * jj + 1][1] + bg[jj].y});
                        const f32x2 r1 = geglu2(f32x2{acc[i][2 * jj][2] + bv[jj].z, acc[i][2 * jj][3] + bv[jj].w},
                                                f32x2{acc[i][2 * jj + 1][2] + bg[jj].z, acc[i][2 * jj + 1][3] + bg[jj].w});
                        *reinterpret_cast<float4*>(ep + l15 * RSG + (jj * 16 + (lane >> 4) * 4) * 4) = make_float4(r0.x, r0.y, r1.x, r1.y);
                    }
                    const int m = mrow + i * 16 + rr;
                    if (rr < 16 && m < M && ocol < (N >> 1)) {
                        const float4 a = *reinterpret_cast<const float4*>(ep + rr * RSG + cc * 32);
                        const float4 b = *reinterpret_cast<const float4*>(ep + rr * RSG + cc * 32 + 16);
                        U4BF8 o;
                        o.e[0] = f2bf(a.x); o.e[1] = f2bf(a.y); o.e[2] = f2bf(a.z); o.e[3] = f2bf(a.w);
                        o.e[4] = f2bf(b.x); o.e[5] = f2bf(b.y); o.e[6] = f2bf(b.z); o.e[7] = f2bf(b.w);
                        *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + ocol) = o.u;
                    }
                }
                return true;
            }
        }
        const int mrow = tm * BM + wm * TM * 16;               // first row of the wave's sub-tile
        const int ncb = tn * BN + wn * WCOLS;                  // first column
        const int rr = lane / LPR, cc = lane - rr * LPR;       // read side: row inside a read instruction, 8-column group
        float2* scr = reinterpret_cast<float2*>(smem + slot_done * STAGE + WMW * 2 * EPW) + wave * 16 * LPR;   // row-statistics scratch (Epilogue::stats_out)
        const int ncol = ncb + cc * 8;
        const bool col_ok = rr < RPI && ncol < N;
        // every wave is done with the fragments of the last K tile before any wave overwrites the slot
        __builtin_amdgcn_s_barrier();
        // ---- loads first (one in-order vmcnt for loads and stores): bias, time-embedding bias, residual
        float4 bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n0 = ncb + j * 16 + (lane >> 4) * 4;
            bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (E.bias && n0 < N) bj[j] = *reinterpret_cast<const float4*>(E.bias + n0);
        }
        const float g = (AMODE == A_ROWS && E.gate) ? *E.gate : 1.f   /* (gate and row remap exist for row GEMMs only: gemm_launch rejects them on a conv) */;
        auto out_row = [&](int i, int k) {  // output row of read instruction k of fragment row i, or -1
            const int r = k * RPI + rr;
            const int m = mrow + i * 16 + r;
            if (!(col_ok && r < 16 && m < M)) return -1;
            return (AMODE == A_ROWS && E.remap_in) ? (m / E.remap_in) * E.remap_out + (m % E.remap_in) + E.remap_off : m;
        };
        // the residual is prefetched for two fragment rows at a time (register budget); the second pair's loads queue behind
        // the first pair's stores, one extra store round trip per work item instead of one per fragment
        // Two copies of the row loop, chosen by a uniform branch: with the residual its loads have no "or zeros" alternative, so
        // hipcc cannot fold the bf16 unpacking into the block that loads (which made it wait for every load right behind its issue:
        // 32 of the epilogue's loads in the round-2 form, one memory round trip each)
        constexpr int PF = (TM >= 2 && !(TM == 4 && TN == 5)) ? 2 : 1;   // (128 x 160: one row block at a time, the tile sits at the 256-register limit)
        auto rows = [&](auto res_c) {
        constexpr bool HAS_RES = decltype(res_c)::value;
        uint4 rs[PF][NRI];
        auto prefetch_res = [&](int i0) {
            if constexpr (!HAS_RES) return;
#pragma unroll
            for (int ii = 0; ii < PF; ++ii)
#pragma unroll
                for (int k = 0; k < NRI; ++k) {
                    // UNCONDITIONAL per lane (a lane without an output row reads the tensor's first 16 bytes and never uses them):
                    // behind a per-lane guard hipcc unpacks the bf16 pairs inside the guarded block, i.e. waits for every load
                    // right behind its issue -- the residual reads of an item then cost one memory round trip EACH
                    const int mo = out_row(i0 + ii, k);
                    rs[ii][k] = *reinterpret_cast<const uint4*>(E.res + (mo >= 0 ? (size_t)mo * E.ldres + ncol : (size_t)0));
                }
        };
        auto res_ready = [&]() {   // one wait for the pair in front of its first store block (see epi_ready)
            if constexpr (!HAS_RES) return;
#pragma unroll
            for (int ii = 0; ii < PF; ++ii)
#pragma unroll
                for (int k = 0; k < NRI; ++k) asm volatile("" : "+v"(rs[ii][k].x), "+v"(rs[ii][k].y), "+v"(rs[ii][k].z), "+v"(rs[ii][k].w));
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i % PF == 0) prefetch_res(i);
            // fragment layout -> LDS: lane (l15, q) holds row l15, columns 16 j + 4 q .. + 3
            const int mf = mrow + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 v = make_float4(acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w);
                if (AMODE == A_CONV3 && E.bias2) {
                    const int n0 = ncb + j * 16 + (lane >> 4) * 4;
                    if (mf < M && n0 < N) {
                        const float4 b2 = *reinterpret_cast<const float4*>(E.bias2 + (size_t)div_rpb(E, mf) * E.bias2_ld + n0);
                        v.x += b2.x; v.y += b2.y; v.z += b2.z; v.w += b2.w;
                    }
                }
                *reinterpret_cast<float4*>(ep + l15 * RS + (j * 16 + (lane >> 4) * 4) * 4) = v;
            }
            if (i % PF == 0) res_ready();
            // LDS -> whole-row pieces: 8 consecutive columns per lane
#pragma unroll
            for (int k = 0; k < NRI; ++k) {
                const int r = k * RPI + rr;
                const int mo = out_row(i, k);
                if (mo < 0) continue;
                const float4 a = *reinterpret_cast<const float4*>(ep + r * RS + cc * 32);
                const float4 b = *reinterpret_cast<const float4*>(ep + r * RS + cc * 32 + 16);
                float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                if constexpr (!HAS_RES) {
                    if (E.act == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                    }
                }
                U4BF8 o;
                if constexpr (HAS_RES) {
                    U4BF8 rv;
                    rv.u = rs[i % PF][k];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.e[e] = f2bf(bf2f(rv.e[e]) + g * v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.e[e] = f2bf(v[e]);
                }
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(E.out) + (size_t)mo * E.ldo + ncol) = o.u;
                if constexpr (AMODE == A_ROWS) {
                    if (E.stats_out) {   // this lane's 8 FINAL outputs of row r -> (sum, sum of squares), through the wave's scratch
                        float s8 = 0.f, q8 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = bf2f(o.e[e]); s8 += f; q8 += f * f; }
                        scr[r * LPR + cc] = make_float2(s8, q8);
                    }
                }
            }
            if constexpr (AMODE == A_ROWS) {
                if (E.stats_out) {
                    // lanes 0..15 add the LPR pieces of "their" row in a fixed order: the row's partial over this wave's column block
                    __builtin_amdgcn_wave_barrier();
                    const int m = mrow + i * 16 + lane;
                    if (lane < 16 && m < M && ncb < N) {     // (a wave whose column block lies beyond N has nothing to report)
                        float s = 0.f, q = 0.f;
#pragma unroll
                        for (int c = 0; c < LPR; ++c) { const float2 v2 = scr[lane * LPR + c]; s += v2.x; q += v2.y; }
                        const int mo2 = E.remap_in ? (m / E.remap_in) * E.remap_out + (m % E.remap_in) + E.remap_off : m;
                        E.stats_out[(size_t)mo2 * E.stats_ld + (tn * 2 + wn)] = make_float2(s, q);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        };
        if (E.res) rows(std::true_type{});
        else rows(std::false_type{});
        return true;
    };

    if (l_item >= wd.n_items) return;
    setup_load(l_item);
    int c_item = l_item, c_tm, c_tn, c_z;
    decode(c_item, c_tm, c_tn, c_z);
    bool c_swap = QKV && c_tn * BN >= 2 * E.C;   // BN divides 2C (checked on the host): an item is all V or no V
    if constexpr (QKV) {
        if (E.ln_stats) ln_fetch(c_tm);
    }
    int c_left = l_kt_end - l_kt;
    int slot_c = 0;      // ring slot of the tile being multiplied; the next tile's DMA goes into the other one
    if (more) issue_next(slot_c);
    for (;;) {
        c_left = __builtin_amdgcn_readfirstlane(c_left);
        slot_c = __builtin_amdgcn_readfirstlane(slot_c);
        more = __builtin_amdgcn_readfirstlane((int)more) != 0;
        // vmcnt(0), unconditional and through the builtin so that hipcc KNOWS nothing is pending: behind an asm wait its scoreboard
        // carries the epilogue's bias / residual loads (whose destination VGPRs the fragments reuse) around the loop and it re-waits
        // vmcnt(0) in front of the first fragment register write, i.e. right after the next DMA issue
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // tile c is in LDS for every wave after this barrier, and every wave has finished reading tile c-1
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) issue_next(slot_c ^ 1);
        compute(slot_c, c_swap);
        const int slot_done = slot_c;
        slot_c ^= 1;
        if (--c_left == 0) {
            if constexpr (QKV) {
                // folded LayerNorm (the head-layout epilogues, unsplit by construction): row statistics first
                const float2* lst = E.ln_stats ? ln_prepare(c_tm, slot_done) : nullptr;
                if (c_swap) epilogue_vt(c_tm, c_tn, lst);
                else epilogue_qk(c_tm, c_tn, lst);
            } else {
                if (!epilogue_staged(c_tm, c_tn, slot_done)) epilogue(c_tm, c_tn, c_z);
            }
            c_item += gridDim.x;
            if (c_item >= wd.n_items) break;
            zero_acc();
            decode(c_item, c_tm, c_tn, c_z);
            c_swap = QKV && c_tn * BN >= 2 * E.C;
            if constexpr (QKV) {
                if (E.ln_stats) ln_fetch(c_tm);
            }
            c_left = min(nk, (c_z + 1) * wd.kt_per_split) - c_z * wd.kt_per_split;
        }
    }
}

// (mean, rstd) of row m from the producer's partial sums (Epilogue::ln_stats), fixed summation order
__device__ __forceinline__ float2 ln_row_stats(const Epilogue& E, int m) {
    float s = 0.f, q = 0.f;
    for (int nb = 0; nb < E.ln_nb; ++nb) {
        const float2 v = E.ln_stats[(size_t)m * E.ln_ld + nb];
        s += v.x; q += v.y;
    }
    const float mean = s * E.ln_inv_c;
    const float var = fmaxf(q * E.ln_inv_c - mean * mean, 0.f);
    return make_float2(mean, rsqrtf(var + E.ln_eps));
}

// Deterministic split-K reduction + epilogue: one thread per (row, group of 4 columns).
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, Epilogue E) {
    const int groups = N >> 2;
    const int64_t total = (int64_t)M * groups;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int m = (int)(idx / groups);
        int n0 = (int)(idx - (int64_t)m * groups) << 2;
        if (E.act == ACT_GEGLU && E.geglu16) {
            if (n0 & 16) continue;  // gate groups are consumed by their value group
            float val[4] = {0, 0, 0, 0}, gate[4] = {0, 0, 0, 0};
            for (int z = 0; z < splits; ++z) {
                const float* p = ws + ((size_t)z * M + m) * N + n0;
                float4 a = *reinterpret_cast<const float4*>(p);
                float4 b = *reinterpret_cast<const float4*>(p + 16);
                val[0] += a.x; val[1] += a.y; val[2] += a.z; val[3] += a.w;
                gate[0] += b.x; gate[1] += b.y; gate[2] += b.z; gate[3] += b.w;
            }
            epi_geglu4_t16(E, m, n0, val, gate);
        } else if (E.act == ACT_GEGLU) {
            if (n0 & 8) continue;  // gate groups are consumed by their value group
            float val[4] = {0, 0, 0, 0}, gate[4] = {0, 0, 0, 0};
            for (int z = 0; z < splits; ++z) {
                const float* p = ws + ((size_t)z * M + m) * N + n0;
                float4 a = *reinterpret_cast<const float4*>(p);
                float4 b = *reinterpret_cast<const float4*>(p + 8);
                val[0] += a.x; val[1] += a.y; val[2] += a.z; val[3] += a.w;
                gate[0] += b.x; gate[1] += b.y; gate[2] += b.z; gate[3] += b.w;
            }
            epi_geglu4(E, m, n0, val, gate);
        } else {
            // Row-major bf16 outputs (every split-K conv and projection of the UNet): bias, per-sample bias, residual and gate are
            // requested BEFORE the slabs, unconditionally (an absent operand reads the slab area and is never used -- behind
            // `if (E.res)` hipcc would unpack inside the branch and wait there); they arrive while the slabs are summed
            const bool fast = E.mode == EPI_ROWMAJOR && !E.out_f32 && !E.remap_in && E.act != ACT_GELU && !E.ln_stats;
            // (no `if (fast)` around the loads either: a value that is "loaded or zero" is the pattern that gets waited for in place)
            const float4 pb = *reinterpret_cast<const float4*>((fast && E.bias) ? E.bias + n0 : ws);
            const float4 pb2 = *reinterpret_cast<const float4*>((fast && E.bias2) ? E.bias2 + (size_t)div_rpb(E, m) * E.bias2_ld + n0 : ws);
            const uint2 pr = *reinterpret_cast<const uint2*>((fast && E.res) ? reinterpret_cast<const void*>(E.res + (size_t)m * E.ldres + n0)
                                                                              : reinterpret_cast<const void*>(ws));
            const float pg = *((fast && E.res && E.gate) ? E.gate : ws);
            // slabs four at a time, all four loads in flight before the first add (a slab beyond `splits` re-reads the last one and
            // counts as zero): one load per loop trip is one L2 round trip per slab and thread, 8-16 of them in a row at the 8 x 8
            // level. Summation order stays z = 0 .. splits - 1.
            float v[4] = {0, 0, 0, 0};
            const float* p0 = ws + (size_t)m * N + n0;
            const size_t zs = (size_t)M * N;
            for (int z = 0; z < splits; z += 4) {
                float4 a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(p0 + (size_t)min(z + u, splits - 1) * zs);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = z + u < splits;
                    v[0] += ok ? a[u].x : 0.f; v[1] += ok ? a[u].y : 0.f; v[2] += ok ? a[u].z : 0.f; v[3] += ok ? a[u].w : 0.f;
                }
            }
            if (E.ln_stats) {   // folded LayerNorm (gemm.h): rstd (acc - mean csum); epi_finish4 adds the folded bias
                const float2 st = ln_row_stats(E, m);
                const float4 cs = *reinterpret_cast<const float4*>(E.ln_csum + n0);
                v[0] = st.y * (v[0] - st.x * cs.x); v[1] = st.y * (v[1] - st.x * cs.y);
                v[2] = st.y * (v[2] - st.x * cs.z); v[3] = st.y * (v[3] - st.x * cs.w);
            }
            if (fast) {
                if (E.bias) { v[0] += pb.x; v[1] += pb.y; v[2] += pb.z; v[3] += pb.w; }
                if (E.bias2) { v[0] += pb2.x; v[1] += pb2.y; v[2] += pb2.z; v[3] += pb2.w; }
                if (E.act == ACT_SILU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
                }
                if (E.res) {
                    U2BF4 r;
                    r.u = pr;
                    const float g = E.gate ? pg : 1.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = bf2f(r.e[i]) + g * v[i];
                }
                store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + n0, v);
            } else {
                epi_finish4(E, m, n0, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv3x3 "halo" kernel (stride 1, pad 1, power-of-two images up to 64 wide): 8 waves, ONE workgroup per CU, a 256 x (TN*32)
// output tile = 256 consecutive pixels (whole image rows / whole small images) x BN output channels.
// The implicit-GEMM kernel above stages the activation operand once per FILTER TAP (128 rows x 128 B per K tile next to 160
// weight rows: 36 KB of LDS-DMA per 128x160x64 multiply, and the DMA path -- about 53 GB/s per CU -- bounds it). Here a
// 64-channel chunk of the tile's input pixels INCLUDING the one-pixel halo is staged once ((R+2) x (W+2) rows of 128 B,
// 324-400 rows for the shapes below) and the nine taps are walked inside LDS: the fragment read of tap (ky, kx) is the same
// read shifted by ky (W+2) + kx halo rows. Per 256x160x64 multiply the DMA then brings one 20 KB weight tile plus 1/9 of a
// halo: 2.5x fewer bytes per FLOP.
//   LDS: two halo buffers (448 rows each = 7 DMA passes of 64 rows) + two weight slots.
//   K order inside a work item: chunk major, taps inside (weights stay [O][tap][I]: the tap only moves the SGPR offset).
//   Per (chunk, tap):  vmcnt(0) -> barrier -> DMA of the next weight tile + ONE pass of the next chunk's halo (taps 0..6)
//                      -> fragment reads + 40 MFMAs per wave.
//   Split-K splits by chunks and goes through the same fp32 slabs + splitk_reduce_kernel as the kernel above.
struct HaloDesc {
    int tiles_n, splits, chunks_per_split, n_items;
    int lgW, lgH;   // image width / height (powers of two)
};

// GN = true (round 6): GroupNorm-apply + SiLU as the conv's prologue (reference openaimodel.py:212-232: GroupNorm32 -> SiLU -> conv).
// The RAW tensor is staged; every lane then normalises, in LDS, exactly the 16-byte piece its own DMA wrote (pass p of the next
// chunk is issued with tap p, has landed behind tap p + 1's vmcnt(0) and is rewritten during tap p + 1: no extra barrier, the
// chunk is first read two or more barriers later), from the chunk's 64 x (a, c) coefficients that one more 512-byte DMA brings
// into LDS beside the halo. Padding pixels never pass through the transform and stay zero, as the reference pads the ACTIVATED
// tensor. An item's first chunk (all seven passes issued at once by begin_item) is rewritten in front of its first barrier.
template <int TN, int NW, bool GN = false>
__global__ void __launch_bounds__(NW * 64, 1)
conv_halo_kernel(AOperand A, const bf16* __restrict__ W, int M, int N, int K, Epilogue E, float* __restrict__ ws, HaloDesc hd) {
    constexpr int NT = NW * 64;
    constexpr int RPP = NT / 8;                 // rows per DMA pass (one 128-byte row per 8 lanes)
    constexpr int TM = 32 / NW;                 // 16-row fragments per wave: NW/2 waves along M, 2 along N
    constexpr int BM = 256;
    constexpr int BN = TN * 32;
    constexpr int HROWS = 448;
    constexpr int HPASS = HROWS / RPP;          // halo DMA passes
    constexpr int HPT = (HPASS + 6) / 7;        // halo passes issued with each of the taps 0..6
    constexpr int HBUF = HROWS * 128;           // bytes per halo buffer
    constexpr int WSLOT = BN * 128;
    constexpr int WP = (BN + RPP - 1) / RPP;    // weight DMA passes; the last one is partial when BN % RPP != 0
    constexpr int WREM = BN % RPP;
    constexpr unsigned SENT = 0x80000000u;
    constexpr int COEF0 = 2 * HBUF + 2 * WSLOT;  // GN: two 1 KiB coefficient slots behind the weight slots (a DMA writes 64 x 16 B)
    static_assert(HPASS * RPP == HROWS && HPT * 7 >= HPASS && WREM % 8 == 0, "loader geometry");
    static_assert(!GN || (HPT == 1 && HPASS == 7), "the prologue rewrites pass p of the next chunk during tap p + 1");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const int l15 = lane & 15;
    const int q = lane >> 4;
    const int r0 = t >> 3;                                          // row inside a DMA pass
    const unsigned chb = (((t & 7) ^ ((r0 >> 1) & 7)) * 16);        // byte offset of this lane's (swizzled) source chunk
    // The halo buffer is read at nine different row shifts (the filter taps), so its swizzle has to stay conflict free under any
    // shift: chunk ^= row & 6 does (a lane group of ds_read_b128 mixes rows {0-3, 12-15} of one 16-byte column with rows {4-11} of the
    // next; the column's low bit then tells the two sets apart and row mod 8 separates the rows inside each, whatever the first row
    // is). With the (row >> 1) & 7 swizzle of the aligned tiles only shifts that are multiples of 4 rows are conflict free: PMC
    // showed 23-25 % of this kernel's LDS cycles as bank conflicts (profiles/r3_final/pmc_mfma_report.txt), on the port that bounds it.
    const unsigned chb_h = (((t & 7) ^ (r0 & 6)) * 16);                  // (halo rows of a DMA pass: h = pass * 64 + r0)
    const int Cin = A.C0 + A.C1;

    const int Wd = 1 << hd.lgW, Hd = 1 << hd.lgH;
    const int R = BM >> hd.lgW;                 // image rows per tile
    const int lgHB = min(hd.lgH, 8 - hd.lgW);   // rows per halo block: a tile is R / HB whole images, or HB = R rows of one
    const int HB = 1 << lgHB;
    const int W2 = Wd + 2;
    const int blkrows = (HB + 2) * W2;
    const int NH = (R >> lgHB) * blkrows;       // halo rows in use (<= 400)

    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)A.p0, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(A.p1 ? A.p1 : A.p0), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x80000000u, 0x00020000);

    // ---- halo row h = pass * RPP + r0 of a tile  <->  source pixel (first pixel of the tile) + hrel, if it is a real pixel: not a
    // left / right padding column, not beyond NH, not above / below the image (whole-image blocks: always padding there; a block
    // of R rows inside an image: padding only when the tile starts at the top / ends at the bottom of its image).
    // Recomputed per DMA pass (one pass per tile in the steady state) with multiply-shift divisions -- ceil(2^22 / d) is exact for
    // h < 2^22 / d, i.e. for every h < 448 with d <= 400 (a 16-bit reciprocal is NOT: 395 * ceil(65536 / 396) >> 16 = 1) -- rather
    // than kept in seven registers that the 256-register budget does not have. tests/test_host_cpu.py restates this index math.
    const unsigned inv_w2 = ((1u << 22) + W2 - 1) / W2, inv_blk = ((1u << 22) + blkrows - 1) / blkrows;
    // fragment rows: pixel p of the tile sits at halo row hr (tap (0,0)); tap (ky, kx) adds ky * W2 + kx. A wave's TM * 16 pixels lie
    // inside one halo block (a block has >= 64 pixels), so fragment i is fragment 0 plus a wave-uniform number of halo rows:
    // 16 pixels further along the image row, wrapping into the next halo row(s) every Wd pixels
    int hr00;
    {
        const int p = wm * TM * 16 + l15;
        const int blk = p >> (lgHB + hd.lgW);
        const int yl = (p >> hd.lgW) & (HB - 1);
        const int x = p & (Wd - 1);
        hr00 = blk * blkrows + yl * W2 + x;
    }
    auto hr_delta = [&](int i) -> int {   // halo rows from fragment 0 to fragment i (scalar): the wave's first pixel is a multiple
        const int px = i * 16;            // of 64, so x + (px mod Wd) never carries into the next image row
        return (px >> hd.lgW) * W2 + (px & (Wd - 1));
    };
    const int foff0 = l15 * 128 + (((q) ^ (l15 >> 1)) << 4);
    const int foff1 = l15 * 128 + (((q + 4) ^ (l15 >> 1)) << 4);
    const int wrow0 = wn * TN * 16 * 128;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- work item state
    int m0 = 0, c_tn = 0, c_z = 0;
    bool top_ok = false, bot_ok = false;
    unsigned vw0 = SENT;              // byte offset of this lane's chunk of weight row (tile's first row + r0); pass i adds i * RPP rows
    auto setup = [&](int item) {
        const int tile = item / hd.splits;
        c_z = item - tile * hd.splits;
        const int tm = tile / hd.tiles_n;
        c_tn = tile - tm * hd.tiles_n;
        m0 = tm * BM;
        const int y0 = (m0 >> hd.lgW) & (Hd - 1);
        top_ok = y0 != 0;
        bot_ok = y0 + HB != Hd;
        vw0 = (unsigned)((c_tn * BN + r0) * K) * 2u + chb;     // N % BN == 0 (host): every weight row of the tile exists
    };
    auto issue_halo = [&](int p, int cc_, int hb_) {     // pass p of the 64-channel chunk at channel cc into halo buffer hb
        const int cc = __builtin_amdgcn_readfirstlane(cc_);
        const int hb = __builtin_amdgcn_readfirstlane(hb_);
        const bool first = cc < A.C0;
        const unsigned ld2 = (unsigned)(first ? A.ld0 : A.ld1) * 2u;
        const int soff = (first ? cc : cc - A.C0) * 2;
        int h = p * RPP + r0;
        // GN: the row map really is recomputed per pass -- without this hipcc hoists the seven item-invariant (pixel, mask) pairs out of
        // the chunk loop, and with the prologue's registers on top of the accumulators they no longer fit (spills inside the K loop)
        if constexpr (GN) asm volatile("" : "+v"(h));
        const int blk = (int)(((unsigned)h * inv_blk) >> 22);
        const int rem = h - blk * blkrows;
        const int hy = (int)(((unsigned)rem * inv_w2) >> 22);
        const int hx = rem - hy * W2;
        const bool ok = h < NH && hx >= 1 && hx <= Wd && (hy != 0 || top_ok) && (hy != HB + 1 || bot_ok);
        const int hrel = ((blk << lgHB) + hy - 1) * Wd + hx - 1;
        const unsigned va = ok ? __umul24((unsigned)(m0 + hrel), ld2) + chb_h : SENT;
        unsigned char* dst = smem + hb * HBUF + (p * RPP + wave * 8) * 128;
        if (first) GL_BLDS16(ra0, dst, va, soff);
        else GL_BLDS16(ra1, dst, va, soff);
    };
    auto issue_w = [&](int tap, int cc_, int ws_) {      // weight tile of (tap, chunk) into weight slot ws
        const int cc = __builtin_amdgcn_readfirstlane(cc_);
        const int wsl = __builtin_amdgcn_readfirstlane(ws_);
        const int soff = (tap * Cin + cc) * 2;
        unsigned char* dst = smem + 2 * HBUF + wsl * WSLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            if (WREM && i == WP - 1 && wave >= WREM / 8) continue;   // wave-uniform
            GL_BLDS16(rw, dst + i * RPP * 128, vw0, soff + i * RPP * K * 2);
        }
    };

    // GN: the 64 x (a, c) coefficients of channel chunk cc of the tile's sample (a 256-pixel tile lies inside one image: H W >= 256,
    // checked on the host) into coefficient slot hb: 512 contiguous bytes, lanes 0..31 of wave 0 (the other lanes write zeros)
    auto issue_coef = [&](int cc_, int hb_) {
        if constexpr (GN) {
            if (wave == 0) {
                const int cc = __builtin_amdgcn_readfirstlane(cc_);
                const int hb = __builtin_amdgcn_readfirstlane(hb_);
                const int smp = m0 >> (hd.lgW + hd.lgH);
                // (flat-address DMA: a fourth buffer descriptor would cost four SGPRs this kernel does not have; lanes 32..63 fetch the
                // same 512 bytes again into the slot's unused upper half)
                // scalar base + 32-bit lane offset, the lane id taken from v_mbcnt on the spot: any lane constant that lives across the
                // tap loop for this one instruction is a register the loop does not have (it was spilled, and its reload drained the DMA)
                unsigned lo;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo));
                lo = (lo & 31u) * 16u;
                GL_GLDS16(reinterpret_cast<const unsigned char*>(A.gn) + (size_t)(unsigned)((smp * Cin + cc) * 8) + lo, smem + COEF0 + hb * 1024);
            }
        }
    };
    // GN: rewrite the 16-byte piece this lane's DMA of pass p wrote into halo buffer hb (8 channels of halo row p RPP + r0) as
    // bf16(silu(x a + c)) -- the arithmetic of gn_apply_kernel (norm.hip), so the conv multiplies the very values the two-pass form
    // would have read back from HBM. Lanes whose halo row is padding hold zeros and skip. LDS accesses in asm: a compiler-visible
    // LDS access behind an LDS-DMA issue would drain vmcnt(0) (see lds_rd16).
    // pass = PC::value + prt: the compile-time part rides in the DS instructions' offset field (a per-pass address in a VGPR is
    // loop invariant, hipcc hoists all seven out of the chunk loop and spills them), prt is the run-time pass of the first-chunk loop.
    // Three steps so that the caller can put matrix work between them: xf_read (five LDS reads), xf_math (after the caller's lgkmcnt
    // wait; branch-free: lanes whose halo row is padding compute on their zeros and select zero -- MFMAs can then be interleaved), xf_write.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct XF { f32x4 a0, a1, c0, c1; };
    struct XP1 { u32x4 v; unsigned addr; bool ok; };
    // rsh: 0 = the piece this lane's own DMA wrote (a row shift reaches another wave's piece: valid behind a barrier only)
    auto xf_piece = [&](auto PC, int prt, int hb, int rsh, XP1& x) {
        constexpr int PI = decltype(PC)::value;
        int h = (PI + prt) * RPP + r0 + rsh;
        asm volatile("" : "+v"(h));       // (recomputed per call, see issue_halo)
        const int blk = (int)(((unsigned)h * inv_blk) >> 22);
        const int rem = h - blk * blkrows;
        const int hy = (int)(((unsigned)rem * inv_w2) >> 22);
        const int hx = rem - hy * W2;
        x.ok = h < NH && hx >= 1 && hx <= Wd && (hy != 0 || top_ok) && (hy != HB + 1 || bot_ok);
        x.addr = lds0 + hb * HBUF + (prt * RPP + rsh) * 128 + t * 16;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x.v) : "v"(x.addr), "n"(PI * RPP * 128));
    };
    auto xf_coef = [&](int hb, XF& x) {
        const unsigned caddr = lds0 + COEF0 + hb * 1024 + (((t & 7) ^ (r0 & 6)) << 6);
        asm volatile("ds_read_b128 %0, %1" : "=v"(x.a0) : "v"(caddr));
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(x.c0) : "v"(caddr));
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(x.a1) : "v"(caddr));
        asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(x.c1) : "v"(caddr));
    };
    auto xf_wait = [&](XF& x, XP1& p) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.v), "+v"(x.a0), "+v"(x.a1), "+v"(x.c0), "+v"(x.c1)); };
    auto xf_math = [&](const XF& x, const XP1& p) -> u32x4 {
        const float a[8] = {x.a0[0], x.a0[1], x.a0[2], x.a0[3], x.a1[0], x.a1[1], x.a1[2], x.a1[3]};
        const float c[8] = {x.c0[0], x.c0[1], x.c0[2], x.c0[3], x.c1[0], x.c1[1], x.c1[2], x.c1[3]};
        union { u32x4 u; bf16 e[8]; } o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned w = p.v[i];
            const float x0 = __uint_as_float(w << 16), x1 = __uint_as_float(w & 0xffff0000u);
            o.e[2 * i] = f2bf(silu_f(fmaf(x0, a[2 * i], c[2 * i])));
            o.e[2 * i + 1] = f2bf(silu_f(fmaf(x1, a[2 * i + 1], c[2 * i + 1])));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o.u[i] = p.ok ? o.u[i] : 0u;
        return o.u;
    };
    auto xf_write = [&](auto PC, const XP1& p, u32x4 o) {
        constexpr int PI = decltype(PC)::value;
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(p.addr), "v"(o), "n"(PI * RPP * 128));
    };

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // One (chunk, tap) tile: all 2 x (TM + TN) fragment reads go out at once (every wave of the workgroup reads at the same time and
    // keeps the LDS port full), the MFMAs of K step 0 start when its fragments are in, and hipcc is left to schedule the rest: it
    // runs the MFMAs of K step 1 behind the NEXT tile's barrier and DMA issue. Hand-placed alternatives were all slower on
    // MI355X (profiles/r2_final/halo_schedules.txt): the two wave groups half a tile out of phase (each wave gets one
    // ds_read_b128 out per ~45 clocks, four waves cannot fill the port), activation fragments of the next tile read ahead with
    // the reads interleaved between the MFMA rows, and four 128x80 waves (one per SIMD: nothing to issue while one waits).
    auto compute = [&](int tapoff, int hb, int wsl) {
        const unsigned hbase = lds0 + hb * HBUF;
        const unsigned wbase = lds0 + 2 * HBUF + wsl * WSLOT + wrow0;
        unsigned ax0[TM], ax1[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned hr = (unsigned)(hr00 + hr_delta(i) + tapoff);
            ax0[i] = hbase + hr * 128u + (((unsigned)q ^ (hr & 6u)) << 4);
            ax1[i] = ax0[i] ^ 64u;    // k-step 1 = chunks 4..7: (q + 4) ^ s = (q ^ s) ^ 4
        }
        const unsigned aw0 = wbase + foff0, aw1 = wbase + foff1;
        bf16x8 xa[TM], wa[TN], xb[TM], wb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) lds_rd16<0>(xa[i], ax0[i]);
        lds_rd16_n<TN, 2048>(wa, aw0);
#pragma unroll
        for (int i = 0; i < TM; ++i) lds_rd16<0>(xb[i], ax1[i]);
        lds_rd16_n<TN, 2048>(wb, aw1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN));  // the k-step 0 fragments have returned
        pin_regs(xa);
        pin_regs(wa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xa[i], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)");
        pin_regs(xb);
        pin_regs(wb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xb[i], acc[i][j], 0, 0, 0);
    };

    // GN variant: the same tile with K step 1 deferred BY HAND (what hipcc does by itself above: its MFMAs sink behind the next barrier
    // down to the first inline asm). Written out because the prologue's arithmetic has to sit between those MFMAs: k1_mfma(lo, hi)
    // issues MFMAs lo..hi-1 of the pending K step 1 (fragments xbp / wbp stay in registers across the barrier).
    bf16x8 xbp[TM], wbp[TN];
    auto k1_mfma = [&](auto LO, auto HI) {
#pragma unroll
        for (int ij = decltype(LO)::value; ij < decltype(HI)::value; ++ij)
            acc[ij / TN][ij % TN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbp[ij % TN], xbp[ij / TN], acc[ij / TN][ij % TN], 0, 0, 0);
    };
    auto compute_k0 = [&](int tapoff, int hb, int wsl) {   // fragment reads of both K steps, K step 0's MFMAs; K step 1 stays pending
        const unsigned hbase = lds0 + hb * HBUF;
        const unsigned wbase = lds0 + 2 * HBUF + wsl * WSLOT + wrow0;
        unsigned ax0[TM], ax1[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned hr = (unsigned)(hr00 + hr_delta(i) + tapoff);
            ax0[i] = hbase + hr * 128u + (((unsigned)q ^ (hr & 6u)) << 4);
            ax1[i] = ax0[i] ^ 64u;
        }
        const unsigned aw0 = wbase + foff0, aw1 = wbase + foff1;
        bf16x8 xa[TM], wa[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) lds_rd16<0>(xa[i], ax0[i]);
        lds_rd16_n<TN, 2048>(wa, aw0);
#pragma unroll
        for (int i = 0; i < TM; ++i) lds_rd16<0>(xbp[i], ax1[i]);
        lds_rd16_n<TN, 2048>(wbp, aw1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN));
        pin_regs(xa);
        pin_regs(wa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xa[i], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)");
        pin_regs(xbp);
        pin_regs(wbp);
        __builtin_amdgcn_sched_barrier(0);
    };

    auto store_row = [&](int m, int n0, float (&v)[4]) {   // EPI_ROWMAJOR only (checked on the host)
        if (E.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        }
        if (E.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(E.out) + (size_t)m * E.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
        else store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + n0, v);
    };
    auto epilogue = [&]() {
        const int mrow = m0 + wm * TM * 16 + l15;
        const int ncol = c_tn * BN + wn * TN * 16 + q * 4;
        if (hd.splits > 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    *reinterpret_cast<float4*>(ws + ((size_t)c_z * M + m) * N + n0) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
            return;
        }
        float4 bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (E.bias) bj[j] = *reinterpret_cast<const float4*>(E.bias + ncol + j * 16);
        }
        if (E.bias2) {  // + broadcast per-sample bias (ResBlock time embedding); never combined with a residual
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                const float* b2p = E.bias2 + (size_t)div_rpb(E, m) * E.bias2_ld + ncol;
                float4 b2[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b2[j] = *reinterpret_cast<const float4*>(b2p + j * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v[4] = {acc[i][j][0] + bj[j].x + b2[j].x, acc[i][j][1] + bj[j].y + b2[j].y,
                                  acc[i][j][2] + bj[j].z + b2[j].z, acc[i][j][3] + bj[j].w + b2[j].w};
                    store_row(m, ncol + j * 16, v);
                }
            }
        } else if (E.res) {
            const float g = E.gate ? *E.gate : 1.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
                uint2 rs[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) rs[j] = *reinterpret_cast<const uint2*>(E.res + (size_t)m * E.ldres + ncol + j * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = ncol + j * 16;
                    U2BF4 r;
                    r.u = rs[j];
                    float v[4] = {acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w};
                    if (E.act == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(r.e[e]) + g * v[e];
                    if (E.out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(E.out) + (size_t)m * E.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + n0, v);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v[4] = {acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w};
                    store_row(m, ncol + j * 16, v);
                }
            }
        }
    };

    // ---- persistent loop over work items. Per (chunk, tap), one barrier:
    //   vmcnt(0) -> barrier -> DMA of the next weight tile (+ HPT passes of the next chunk's halo, taps 0..6) -> compute
    int item = blockIdx.x;
    if (item >= hd.n_items) return;
    int hb = 0, wsl = 0;              // halo buffer / weight slot of the tile about to be multiplied
    int cc = 0, cc_end = 0;
    auto begin_item = [&](int it) {   // prologue DMA of an item: the whole first halo + the first weight tile
        setup(it);
        cc = c_z * hd.chunks_per_split * 64;
        cc_end = min(Cin, cc + hd.chunks_per_split * 64);
#pragma unroll
        for (int p = 0; p < HPASS; ++p) issue_halo(p, cc, hb);
        issue_coef(cc, hb);
        issue_w(0, cc, wsl);
    };
    begin_item(item);
    zero_acc();
    for (;;) {
        hb = __builtin_amdgcn_readfirstlane(hb);
        wsl = __builtin_amdgcn_readfirstlane(wsl);
        cc = __builtin_amdgcn_readfirstlane(cc);
        cc_end = __builtin_amdgcn_readfirstlane(cc_end);
        const bool next_chunk = cc + 64 < cc_end;
        auto one_tap = [&](auto TC) {
            constexpr int tap = decltype(TC)::value;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this tile's weights (tap 0: and the chunk's halo) have landed
            bool first_chunk = false;
            if constexpr (GN) {
                first_chunk = tap == 0 && cc == c_z * hd.chunks_per_split * 64;
                if (tap == 0 && first_chunk) {   // an item's first chunk (still raw): all seven pieces, before anyone reads them
                    __builtin_amdgcn_s_barrier();      // (the coefficients are wave 0's DMA: landed for everybody only behind a barrier)
                    XF x;
                    xf_coef(hb, x);
#pragma unroll 1
                    for (int p = 0; p < HPASS; ++p) {
                        XP1 pc;
                        xf_piece(std::integral_constant<int, 0>{}, p, hb, 0, pc);
                        xf_wait(x, pc);
                        xf_write(std::integral_constant<int, 0>{}, pc, xf_math(x, pc));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)");
                }
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tap < 8) issue_w(tap + 1, cc, wsl ^ 1);
            else if (next_chunk) issue_w(0, cc + 64, wsl ^ 1);
            if (tap < 7 && next_chunk) {
#pragma unroll
                for (int pp = 0; pp < HPT; ++pp)
                    if (tap * HPT + pp < HPASS) issue_halo(tap * HPT + pp, cc + 64, hb ^ 1);
                if (tap == 0) issue_coef(cc + 64, hb ^ 1);
            }
            if constexpr (GN) {
                constexpr int NMF = TM * TN, HALF = NMF / 2;
                using I0 = std::integral_constant<int, 0>;
                using IH = std::integral_constant<int, HALF>;
                using IN = std::integral_constant<int, NMF>;
                constexpr int PX = (tap >= 1 && tap <= HPASS) ? tap - 1 : -1;   // the pass whose piece is rewritten during this tap
                if (PX >= 0 && next_chunk) {
                    // the piece of pass tap - 1 (landed behind this tap's vmcnt(0)): its reads fly under the first half of the pending
                    // MFMAs, its arithmetic is interleaved with the second half
                    constexpr int PXC = PX >= 0 ? PX : 0;
                    XF x;
                    XP1 p1;
                    xf_piece(std::integral_constant<int, PXC>{}, 0, hb ^ 1, 0, p1);
                    xf_coef(hb ^ 1, x);
                    k1_mfma(I0{}, IH{});
                    xf_wait(x, p1);
                    const u32x4 o = xf_math(x, p1);
                    k1_mfma(IH{}, IN{});
#pragma unroll
                    for (int g = 0; g < NMF - HALF; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // eight VALU
                    }
                    xf_write(std::integral_constant<int, PXC>{}, p1, o);
                } else if (!first_chunk) {
                    k1_mfma(I0{}, IN{});
                }
                compute_k0((tap / 3) * W2 + tap % 3, hb, wsl);
            } else {
                compute((tap / 3) * W2 + tap % 3, hb, wsl);
            }
            wsl ^= 1;
        };
        one_tap(std::integral_constant<int, 0>{}); one_tap(std::integral_constant<int, 1>{}); one_tap(std::integral_constant<int, 2>{});
        one_tap(std::integral_constant<int, 3>{}); one_tap(std::integral_constant<int, 4>{}); one_tap(std::integral_constant<int, 5>{});
        one_tap(std::integral_constant<int, 6>{}); one_tap(std::integral_constant<int, 7>{}); one_tap(std::integral_constant<int, 8>{});
        cc += 64;
        hb ^= 1;
        if (cc < cc_end) continue;
        if constexpr (GN) k1_mfma(std::integral_constant<int, 0>{}, std::integral_constant<int, TM * TN>{});   // the item's last K step 1
        // item done: start the next item's DMA (into the halo buffer / weight slot not read by the last tile), then store
        const int done_m0 = m0, done_tn = c_tn, done_z = c_z;
        item += gridDim.x;
        const bool more = item < hd.n_items;
        if (more) begin_item(item);
        {   // the epilogue addresses the finished item
            const int nm0 = m0, ntn = c_tn, nz = c_z;
            m0 = done_m0; c_tn = done_tn; c_z = done_z;
            epilogue();
            m0 = nm0; c_tn = ntn; c_z = nz;
        }
        if (!more) break;
        zero_acc();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// "Wide" GEMM (A_ROWS, round 2): the geometry of conv_halo_kernel for plain row-major activations. 8 waves, ONE workgroup per CU,
// 256 x (TN*32) tile, two LDS stages of (256 + BN) x 128 B, one barrier per K tile:
//     vmcnt(0) -> barrier -> DMA of the next K tile into the other stage -> all fragment reads -> MFMAs
// (hipcc runs the second K step's MFMAs behind the next barrier, as in the halo kernel). Against gemm_u_kernel's 128-row tiles
// the weight tile is fetched once per 256 rows: 52 KB of LDS-DMA per 256x160x64 multiply instead of 72 KB. For the long-K,
// wide-N problems (FF-out, the GEGLU projections of the 32x32 / 16x16 levels); whole tiles only (M % 256 == 0, N % BN == 0).
// LDS ring depth of the wide kernel. 3 (two K tiles in flight, counted vmcnt wait) was built and measured: 3-5 % SLOWER than 2 on
// every GEGLU projection (profiles/r3/wide_ring_kbench.txt) -- the K loop is not waiting for the fabric, it is sharing the LDS
// port between 128 KB of fragment reads and 48 KB of DMA writes per tile next to 0.43 us of MFMAs. tools/build_variant.sh NAME
// -DGL_WIDE_NST=3 rebuilds that arm.
#ifndef GL_WIDE_NST
#define GL_WIDE_NST 2
#endif
struct WideDesc {
    int tiles_n, splits, kt_per_split, n_items;
    int xcd;   // 1: every XCD walks a contiguous range of the (tile_m, tile_n) order (see launch_wide)
};

template <int TN>
__global__ void __launch_bounds__(512, 1)
gemm_wide_kernel(AOperand A, const bf16* __restrict__ W, int M, int N, int K, Epilogue E, float* __restrict__ ws, WideDesc wd) {
    constexpr int TM = 4;
    constexpr int BM = 256;
    constexpr int BN = TN * 32;
    constexpr int XP = BM / 64;                 // activation DMA passes (64 rows each)
    constexpr int WP = (BN + 63) / 64;          // weight DMA passes; the last one is partial when BN % 64 != 0
    constexpr int WREM = BN % 64;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int NST = GL_WIDE_NST;            // LDS ring depth: NST - 1 K tiles in flight behind the one being multiplied

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const int l15 = lane & 15;
    const int q = lane >> 4;
    const int r0 = t >> 3;                                          // row inside a DMA pass
    const unsigned chb = (((t & 7) ^ ((r0 >> 1) & 7)) * 16);        // byte offset of this lane's (swizzled) source chunk
    const int nk = K >> 6;

    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)A.p0, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(A.p1 ? A.p1 : A.p0), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x80000000u, 0x00020000);

    const int foff0 = l15 * 128 + (((q) ^ (l15 >> 1)) << 4);
    const int foff1 = l15 * 128 + (((q + 4) ^ (l15 >> 1)) << 4);
    const int xrow0 = wm * TM * 16 * 128;
    const int wrow0 = BM * 128 + wn * TN * 16 * 128;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- work item state
    int m0 = 0, c_tn = 0, c_z = 0;
    unsigned vx0 = 0, vx1 = 0, vw0 = 0;   // byte offsets of this lane's chunk of row (tile's first row + r0) in p0 / p1 / W; pass i adds 64 rows
    auto setup = [&](int item_) {
        int item = item_;
        if (wd.xcd) {
            // block b sits on XCD b % 8 and the grid is a multiple of 8, so item & 7 is this workgroup's XCD: give every XCD a
            // contiguous range of the (tile_m, tile_n) order -- the 32 workgroups of an XCD then walk the column tiles of the same
            // one or two 256-row stripes together and the stripe crosses the fabric once, not once per XCD
            const int q = wd.n_items >> 3, r = wd.n_items & 7, xcd = item & 7, idx = item >> 3;
            item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int tile = item / wd.splits;
        c_z = item - tile * wd.splits;
        const int tm = tile / wd.tiles_n;
        c_tn = tile - tm * wd.tiles_n;
        m0 = tm * BM;
        vx0 = (unsigned)((m0 + r0) * A.ld0) * 2u + chb;
        vx1 = (unsigned)((m0 + r0) * A.ld1) * 2u + chb;
        vw0 = (unsigned)((c_tn * BN + r0) * K) * 2u + chb;
    };
    auto issue = [&](int kt_, int st_) {     // K tile kt into stage st
        const int kt = __builtin_amdgcn_readfirstlane(kt_);
        const int st = __builtin_amdgcn_readfirstlane(st_);
        const int k0 = kt << 6;
        const bool first = k0 < A.C0;
        unsigned char* xs = smem + st * STAGE + wave * 1024;
        if (first) {
#pragma unroll
            for (int i = 0; i < XP; ++i) GL_BLDS16(ra0, xs + i * 64 * 128, vx0, k0 * 2 + i * 64 * A.ld0 * 2);
        } else {
#pragma unroll
            for (int i = 0; i < XP; ++i) GL_BLDS16(ra1, xs + i * 64 * 128, vx1, (k0 - A.C0) * 2 + i * 64 * A.ld1 * 2);
        }
        unsigned char* wsm = xs + BM * 128;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            if (WREM && i == WP - 1 && wave >= WREM / 8) continue;   // wave-uniform
            GL_BLDS16(rw, wsm + i * 64 * 128, vw0, k0 * 2 + i * 64 * K * 2);
        }
    };

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // K step 1 of a tile is multiplied one tile late: its MFMAs go out right behind the next barrier and DMA issue, and the next
    // tile's fragment reads fly under them (the order hipcc finds by itself in the unrolled halo kernel; this loop is not unrolled)
    bf16x8 xb[TM], wb[TN];
    auto step1 = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xb[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](int stg, bool pending) {
        const unsigned st = lds0 + stg * STAGE;
        const unsigned ax0 = st + xrow0 + foff0, ax1 = st + xrow0 + foff1;
        const unsigned aw0 = st + wrow0 + foff0, aw1 = st + wrow0 + foff1;
        bf16x8 xa[TM], wa[TN];
        if (pending) step1();
        lds_rd16_n<TM, 2048>(xa, ax0);
        lds_rd16_n<TN, 2048>(wa, aw0);
        lds_rd16_n<TM, 2048>(xb, ax1);
        lds_rd16_n<TN, 2048>(wb, aw1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN));  // the k-step 0 fragments have returned
        pin_regs(xa);
        pin_regs(wa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // (keeps K step 0's MFMAs in front of the wait below: they only need the first TM + TN reads)
        // every fragment read of this stage has RETURNED before the wave reaches the next barrier: behind that barrier the stage is
        // overwritten by DMA, and K step 1's fragments are consumed (the asm reads are invisible to hipcc's own lgkmcnt tracking)
        asm volatile("s_waitcnt lgkmcnt(0)");
        pin_regs(xb);
        pin_regs(wb);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto finish = [&]() {   // the item's last K step 1
        asm volatile("s_waitcnt lgkmcnt(0)");
        pin_regs(xb);
        pin_regs(wb);
        __builtin_amdgcn_sched_barrier(0);
        step1();
    };

    // lst: (mean, rstd) of the item's 256 rows in LDS when the LayerNorm in front of this GEMM is folded into it (gemm.h)
    auto epilogue = [&](const float2* lst) {
        const int mrow = m0 + wm * TM * 16 + l15;
        const int ncol = c_tn * BN + wn * TN * 16 + q * 4;
        if (wd.splits > 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow + i * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float4*>(ws + ((size_t)c_z * M + m) * N + ncol + j * 16) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            return;
        }
        if constexpr (TN % 2 == 0) {
            if (E.act == ACT_GEGLU) {   // geglu16 weight layout: fragment 2 jj = 16 value features, 2 jj + 1 = their gates
                float4 bv[TN / 2], bg[TN / 2];
#pragma unroll
                for (int jj = 0; jj < TN / 2; ++jj) {
                    bv[jj] = bg[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (E.bias) {
                        bv[jj] = *reinterpret_cast<const float4*>(E.bias + ncol + jj * 32);
                        bg[jj] = *reinterpret_cast<const float4*>(E.bias + ncol + jj * 32 + 16);
                    }
                }
                if (lst) {   // folded LayerNorm: acc <- rstd (acc - mean csum), in place, before bias and GEGLU
#pragma unroll
                    for (int jj = 0; jj < TN / 2; ++jj) {
                        const float4 cv = *reinterpret_cast<const float4*>(E.ln_csum + ncol + jj * 32);
                        const float4 cg = *reinterpret_cast<const float4*>(E.ln_csum + ncol + jj * 32 + 16);
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const float2 st = lst[wm * TM * 16 + i * 16 + l15];
                            acc[i][2 * jj][0] = st.y * (acc[i][2 * jj][0] - st.x * cv.x); acc[i][2 * jj][1] = st.y * (acc[i][2 * jj][1] - st.x * cv.y);
                            acc[i][2 * jj][2] = st.y * (acc[i][2 * jj][2] - st.x * cv.z); acc[i][2 * jj][3] = st.y * (acc[i][2 * jj][3] - st.x * cv.w);
                            acc[i][2 * jj + 1][0] = st.y * (acc[i][2 * jj + 1][0] - st.x * cg.x); acc[i][2 * jj + 1][1] = st.y * (acc[i][2 * jj + 1][1] - st.x * cg.y);
                            acc[i][2 * jj + 1][2] = st.y * (acc[i][2 * jj + 1][2] - st.x * cg.z); acc[i][2 * jj + 1][3] = st.y * (acc[i][2 * jj + 1][3] - st.x * cg.w);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = mrow + i * 16;
#pragma unroll
                    for (int jj = 0; jj < TN / 2; ++jj) {
                        const int n0 = ncol + jj * 32;
                        const f32x2 r0 = geglu2(f32x2{acc[i][2 * jj][0] + bv[jj].x, acc[i][2 * jj][1] + bv[jj].y},
                                                f32x2{acc[i][2 * jj + 1][0] + bg[jj].x, acc[i][2 * jj + 1][1] + bg[jj].y});
                        const f32x2 r1 = geglu2(f32x2{acc[i][2 * jj][2] + bv[jj].z, acc[i][2 * jj][3] + bv[jj].w},
                                                f32x2{acc[i][2 * jj + 1][2] + bg[jj].z, acc[i][2 * jj + 1][3] + bg[jj].w});
                        float o[4] = {r0.x, r0.y, r1.x, r1.y};
                        const int j0 = (n0 >> 5) * 16 + (n0 & 15);
                        store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + j0, o);
                    }
                }
                return;
            }
        }
        float4 bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (E.bias) bj[j] = *reinterpret_cast<const float4*>(E.bias + ncol + j * 16);
        }
        const float g = (E.res && E.gate) ? *E.gate : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mrow + i * 16;
            uint2 rs[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                rs[j] = make_uint2(0, 0);
                if (E.res) rs[j] = *reinterpret_cast<const uint2*>(E.res + (size_t)m * E.ldres + ncol + j * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n0 = ncol + j * 16;
                float v[4] = {acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w};
                if (E.act == ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                if (E.res) {
                    U2BF4 r;
                    r.u = rs[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(r.e[e]) + g * v[e];
                }
                if (E.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(E.out) + (size_t)m * E.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
                else store_bf16x4(reinterpret_cast<bf16*>(E.out) + (size_t)m * E.ldo + n0, v);
            }
        }
    };

    // ---- persistent loop over work items
    int item = blockIdx.x;
    if (item >= wd.n_items) return;
    int stg = 0;
    int kt = 0, kt_end = 0;
    // folded LayerNorm: this thread's share of the partial sums of row t >> 1 of the item whose K loop starts (the loads fly under
    // it); the finished item's pair is handed to its epilogue before the next item's fetch overwrites it
    // (the loaded pairs stay in registers untouched until the epilogue: an add right behind the load would make hipcc wait for it there)
    constexpr int LNV = 5;
    float2 ln_v[LNV];
    float ln_s = 0.f, ln_q = 0.f;
    auto ring_next = [](int s_) { return NST == 2 ? s_ ^ 1 : (s_ == NST - 1 ? 0 : s_ + 1); };
    // DMA instructions one K tile costs this wave (the wait below leaves exactly one tile in flight)
    const bool w_short = WREM && wave >= WREM / 8;
    auto begin_item = [&](int it) {
        setup(it);
        kt = c_z * wd.kt_per_split;
        kt_end = min(nk, kt + wd.kt_per_split);
        issue(kt, stg);
        if (NST == 3 && kt + 1 < kt_end) issue(kt + 1, ring_next(stg));
        if (E.ln_stats) {
            const float2* src = E.ln_stats + (size_t)(m0 + (t >> 1)) * E.ln_ld;
            ln_s = ln_q = 0.f;
#pragma unroll
            for (int i = 0; i < LNV; ++i) {
                const int nb = (t & 1) + 2 * i;
                ln_v[i] = make_float2(0.f, 0.f);
                if (nb < E.ln_nb) ln_v[i] = src[nb];
            }
            for (int nb = (t & 1) + 2 * LNV; nb < E.ln_nb; nb += 2) {   // (more than 10 column blocks: C = 1280)
                const float2 v = src[nb];
                ln_s += v.x; ln_q += v.y;
            }
        }
    };
    begin_item(item);
    zero_acc();
    // one K tile: wait for it, barrier, send the tile NST - 1 ahead, multiply. An item's first tile is its own copy of the body
    // (first = true): it drains the counter through the builtin -- the previous item's epilogue stores sit in the same counter as
    // the DMA (loads and stores do not retire in order with each other), and hipcc then KNOWS that nothing is pending, the
    // statistics prefetch of begin_item included; behind an asm wait its scoreboard would carry those loads around the K loop and
    // drain in front of every register copy it places on the back edge. The other tiles leave the tile behind them in flight.
    int stg_done = 0;
    auto tile = [&](auto first_c) {
        constexpr bool first = decltype(first_c)::value;
        stg = __builtin_amdgcn_readfirstlane(stg);
        kt = __builtin_amdgcn_readfirstlane(kt);
        kt_end = __builtin_amdgcn_readfirstlane(kt_end);
        if (NST == 3 && !first && kt + 1 < kt_end) {
            if (w_short) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XP + WP - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XP + WP) : "memory");
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (NST == 3) { if (kt + 2 < kt_end) issue(kt + 2, ring_next(ring_next(stg))); }
        else if (kt + 1 < kt_end) issue(kt + 1, stg ^ 1);
        compute(stg, !first);
        stg_done = stg;
        stg = ring_next(stg);
        ++kt;
    };
    for (;;) {
        tile(std::true_type{});
        while (kt < kt_end) tile(std::false_type{});
        finish();
        // item done: start the next item's first tile (into the stage the last tile was not read from), then store
        const int done_m0 = m0, done_tn = c_tn, done_z = c_z;
        float done_s = ln_s, done_q = ln_q;
#pragma unroll
        for (int i = 0; i < LNV; ++i) { done_s += ln_v[i].x; done_q += ln_v[i].y; }
        item += gridDim.x;
        const bool more = item < wd.n_items;
        if (more) begin_item(item);
        {
            const int nm0 = m0, ntn = c_tn, nz = c_z;
            m0 = done_m0; c_tn = done_tn; c_z = done_z;
            const float2* lst = nullptr;
            if (E.ln_stats && wd.splits == 1) {
                // folded LayerNorm: (mean, rstd) of the finished item's 256 rows from the producer's partial sums, into the stage the
                // last K tile was read from (free until the DMA behind the next barrier); two threads per row, fixed summation order
                float2* l = reinterpret_cast<float2*>(smem + stg_done * STAGE);
                __builtin_amdgcn_s_barrier();          // every wave is done with that stage's fragments
                const int row = t >> 1, part = t & 1;
                float s = done_s, qq = done_q;
                s += __shfl_xor(s, 1, 64); qq += __shfl_xor(qq, 1, 64);
                if (part == 0) {
                    const float mean = s * E.ln_inv_c;
                    const float var = fmaxf(qq * E.ln_inv_c - mean * mean, 0.f);
                    l[row] = make_float2(mean, rsqrtf(var + E.ln_eps));
                }
                __syncthreads();
                lst = l;
            }
            epilogue(lst);
            m0 = nm0; c_tn = ntn; c_z = nz;
        }
        if (!more) break;
        zero_acc();
    }
}



static int g_gemm_variant = -1;  // 1: LDS-DMA v2 (one tile per workgroup), 2: persistent v3, 4: v5 buffer-DMA persistent (default)
void gemm_set_variant(int v) { g_gemm_variant = v; }
static int gemm_variant() {
    if (g_gemm_variant < 0) {
        const char* e = dev_env("GL_GEMM_VARIANT");
        g_gemm_variant = e ? atoi(e) : 4;
    }
    return g_gemm_variant;
}
// GEGLU weight-row packing the current main-loop variant expects (pack_geglu_launch layout argument)
int gemm_geglu_layout() { return gemm_variant() >= 2 ? 1 : 0; }
bool gemm_supports_qkv() { return gemm_variant() == 4; }

static int g_force_tm = 0, g_force_tn = 0, g_force_splits = 0;  // developer override (kbench sweeps)
static int g_force_grid = 0;
void gemm_force_cfg(int tm, int tn, int splits) { g_force_tm = tm; g_force_tn = tn; g_force_splits = splits; }
void gemm_force_grid(int g) { g_force_grid = g; }
void gemm_set_autotune(int on);
static thread_local int g_last_cfg[3] = {0, 0, 0};      // per calling thread: one engine context per thread
static thread_local int g_last_stats_nb = 0;
int gemm_last_stats_nb() { return g_last_stats_nb; }
static thread_local char g_last_name[96] = "gemm";
const char* gemm_last_kernel_name() { return g_last_name; }
void gemm_last_cfg(int* tm, int* tn, int* splits) { *tm = g_last_cfg[0]; *tn = g_last_cfg[1]; *splits = g_last_cfg[2]; }

namespace {

struct Cfg {
    int bm, bn;
    float speed;
};
const Cfg kCfgs[4] = {{128, 128, 1.0f}, {128, 64, 0.8f}, {64, 64, 0.55f}, {128, 32, 0.45f}};

template <int WM, int WN, int TM, int TN>
int launch_cfg(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, float* ws,
               int splits, int kt_per_split, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    dim3 grid(tiles_m * tiles_n, 1, splits);
    dim3 block(WM * WN * 64);
    size_t lds = 2 * (BM + BN) * 128;
    const int n_tiles = tiles_m * tiles_n;
#define GL_LAUNCH_ONE(KFN, ...)                                                                                  \
    do {                                                                                                         \
        auto kfn = KFN;                                                                                          \
        static bool attr_done = false; /* once per instantiation; never inside a stream capture */              \
        if (!attr_done && lds > 48 * 1024) {                                                                     \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_done = true;                                                                                    \
        }                                                                                                        \
        hipLaunchKernelGGL(kfn, grid, block, lds, stream, A, W, M, N, K, E, ws, kt_per_split, tiles_n, ##__VA_ARGS__); \
    } while (0)
    if (A.mode == A_ROWS) GL_LAUNCH_ONE((gemm_glds_kernel<WM, WN, TM, TN, A_ROWS>), n_tiles);
    else GL_LAUNCH_ONE((gemm_glds_kernel<WM, WN, TM, TN, A_CONV3>), n_tiles);
#undef GL_LAUNCH_ONE
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace

namespace {

template <int TM, int TN>
int launch_p(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, float* ws, const WorkDesc& wd,
             hipStream_t stream) {
    const int cap = g_force_grid ? g_force_grid : 512;
    dim3 grid(wd.n_items < cap ? wd.n_items : cap);
    dim3 block(256);
    const size_t lds = 2 * (TM * 32 + TN * 32) * 128;
#define GL_LAUNCH_P(KFN)                                                                                         \
    do {                                                                                                         \
        auto kfn = KFN;                                                                                          \
        static bool attr_done = false; /* once per instantiation; never inside a stream capture */              \
        if (!attr_done && lds > 48 * 1024) {                                                                     \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_done = true;                                                                                    \
        }                                                                                                        \
        hipLaunchKernelGGL(kfn, grid, block, lds, stream, A, W, M, N, K, E, ws, wd);                             \
    } while (0)
    if (A.mode == A_ROWS) GL_LAUNCH_P((gemm_p_kernel<TM, TN, A_ROWS>));
    else GL_LAUNCH_P((gemm_p_kernel<TM, TN, A_CONV3>));
#undef GL_LAUNCH_P
    GL_LAUNCH_CHECK();
    return GL_OK;
}

template <int WMW, int TM, int TN, int NST>
int launch_u(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, float* ws, const WorkDesc& wd,
             hipStream_t stream) {
    constexpr int NT = WMW * 128, RPP = NT / 8, BM = WMW * TM * 16, BN = TN * 32;
    const int cap = g_force_grid ? g_force_grid : 512;   // workgroups resident per launch: two per CU unless the tuner says otherwise
    dim3 grid(wd.n_items < cap ? wd.n_items : cap);
    dim3 block(NT);
    const size_t lds = NST * (BM + BN) * 128;
#define GL_LAUNCH_U(KFN)                                                                                         \
    do {                                                                                                         \
        auto kfn = KFN;                                                                                          \
        static bool attr_done = false; /* once per instantiation; never inside a stream capture */              \
        if (!attr_done && lds > 48 * 1024) {                                                                     \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_done = true;                                                                                    \
        }                                                                                                        \
        hipLaunchKernelGGL(kfn, grid, block, lds, stream, A, W, M, N, K, E, ws, wd);                             \
    } while (0)
    if (E.mode == EPI_QKV_HEADS || (E.mode == EPI_QK_HEADS && E.ln_stats)) {
        // (the 128 x 160 tile is not built for QKV: with the second MFMA form it needs more than 256 registers)
        // q-only / q,k projections behind a folded LayerNorm run here too: these instantiations hold the statistics code
        if constexpr (TM == 4 && TN == 5) return set_error(GL_ERR_UNSUPPORTED, "gemm: no 128x160 tile for EPI_QKV_HEADS");
        else GL_LAUNCH_U((gemm_u_kernel<WMW, TM, TN, A_ROWS, NST, true>));
    } else if (A.mode == A_ROWS) GL_LAUNCH_U((gemm_u_kernel<WMW, TM, TN, A_ROWS, NST>));
    else GL_LAUNCH_U((gemm_u_kernel<WMW, TM, TN, A_CONV3, NST>));
#undef GL_LAUNCH_U
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// Tile shape + K split for the persistent kernel: minimise a cycle model of
//   (items per block) x (K tiles per item x cycles per K tile + fixed per-item cost) + split-K reduce pass
// over the tile shapes {128,64} x {160,128} and a few split counts.
// Autotuner: the first (eager, non-capturing) launch of every distinct problem times the feasible tile / split /
// residency candidates on the device with the caller's own buffers and caches the winner; captured launches and
// GL_GEMM_AUTOTUNE=0 use the analytic cost model below. A launch only writes E.out (and the split-K workspace),
// and the engine never aliases E.out with an input, so re-running a launch is idempotent.
// The halo kernel's problems: 3x3, stride 1, pad 1, power-of-two images up to 64 wide whose 256-pixel tiles are whole rows of one
// image or whole images, plain row-major epilogue (bias, per-sample bias, residual, SiLU, fp32 slabs for split-K).
static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}
bool halo_eligible(const AOperand& A, int M, int N, int K, const Epilogue& E) {
    if (A.mode != A_CONV3 || A.stride != 1 || A.ups || A.pad_lo != 1 || A.Ho != A.Hin || A.Wo != A.Win) return false;
    const int lgW = ilog2_exact(A.Win), lgH = ilog2_exact(A.Hin);
    if (lgW < 3 || lgW > 6 || lgH < 0 || M % 256) return false;
    const int R = 256 >> lgW, HB = std::min(A.Hin, R);
    if ((R / HB) * (HB + 2) * (A.Win + 2) > 448) return false;
    if (A.C0 % 64 || A.C1 % 64 || (N % 160 && N % 128)) return false;   // whole 160- or 128-wide tiles only
    if (E.mode != EPI_ROWMAJOR || E.remap_in || (E.act != ACT_NONE && E.act != ACT_SILU) || (E.bias2 && E.res)) return false;
    if (E.bias2 && (E.rows_per_b != A.Hin * A.Win)) return false;
    const size_t a_rows = (size_t)M;
    return a_rows * (size_t)std::max(A.ld0, A.ld1) * 2 < 0x7fff0000ull && (size_t)N * K * 2 < 0x7fff0000ull;
}

int launch_halo(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, float* ws, size_t ws_bytes, int want_splits,
                hipStream_t stream) {
    const int tn = N % 160 == 0 ? 5 : 4;
    const int bn = tn * 32;
    HaloDesc hd;
    hd.lgW = ilog2_exact(A.Win);
    hd.lgH = ilog2_exact(A.Hin);
    hd.tiles_n = cdiv(N, bn);
    const int tiles = (M / 256) * hd.tiles_n;
    const int nch = (A.C0 + A.C1) / 64;
    int sp = want_splits;
    if (sp <= 0) {   // one work item per CU where the chunks allow it
        sp = 1;
        while (tiles * sp < 200 && nch / (sp * 2) >= 2) sp *= 2;
    }
    sp = std::max(1, std::min(sp, nch));
    if (!ws) sp = 1;
    while (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_bytes) --sp;
    hd.chunks_per_split = cdiv(nch, sp);
    hd.splits = cdiv(nch, hd.chunks_per_split);
    hd.n_items = tiles * hd.splits;
    const int halo_waves = 8;
    g_last_cfg[0] = 8; g_last_cfg[1] = tn; g_last_cfg[2] = hd.splits;
    snprintf(g_last_name, sizeof g_last_name, "conv_halo_kernel<%d, %d%s>%s", tn, halo_waves, A.gn ? ", gn" : "", hd.splits > 1 ? " + splitk_reduce_kernel" : "");
    dim3 grid(std::min(hd.n_items, 256)), block(halo_waves * 64);
    const size_t lds = 2 * 7 * 64 * 128 + 2 * bn * 128 + (A.gn ? 2048 : 0);
#define GL_LAUNCH_HALO(KFN)                                                                                      \
    do {                                                                                                         \
        auto kfn = KFN;                                                                                          \
        static bool attr_done = false;                                                                           \
        if (!attr_done) {                                                                                        \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_done = true;                                                                                    \
        }                                                                                                        \
        hipLaunchKernelGGL(kfn, grid, block, lds, stream, A, W, M, N, K, E, ws, hd);                             \
    } while (0)
    if (A.gn) {
        if (tn == 5) GL_LAUNCH_HALO((conv_halo_kernel<5, 8, true>));
        else GL_LAUNCH_HALO((conv_halo_kernel<4, 8, true>));
    } else {
        if (tn == 5) GL_LAUNCH_HALO((conv_halo_kernel<5, 8>));
        else GL_LAUNCH_HALO((conv_halo_kernel<4, 8>));
    }
#undef GL_LAUNCH_HALO
    GL_LAUNCH_CHECK();
    if (hd.splits > 1) {
        int64_t total = (int64_t)M * (N / 4);
        int blocks = (int)fmin((double)cdiv64(total, 256), 4096.0);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, hd.splits, M, N, E);
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

// The wide kernel's problems: row-major activations, whole 256 x BN tiles, plain / residual / SiLU / GEGLU row-major epilogues.
bool wide_eligible(const AOperand& A, int M, int N, int K, const Epilogue& E) {
    if (A.mode != A_ROWS || M % 256 || (N % 160 && N % 128)) return false;
    if (A.C1 && A.C0 % 64) return false;
    if (E.mode != EPI_ROWMAJOR || E.remap_in || E.bias2) return false;
    if (E.act == ACT_GEGLU) { if (N % 128 || !E.geglu16 || E.res || E.out_f32) return false; }
    else if (E.act != ACT_NONE && E.act != ACT_SILU) return false;
    return (size_t)M * (size_t)std::max(A.ld0, A.ld1) * 2 < 0x7fff0000ull && (size_t)N * K * 2 < 0x7fff0000ull;
}

int launch_wide(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, float* ws, size_t ws_bytes, int want_splits,
                hipStream_t stream) {
    const int tn = (E.act != ACT_GEGLU && N % 160 == 0) ? 5 : 4;
    const int bn = tn * 32;
    WideDesc wd;
    wd.tiles_n = N / bn;
    const int tiles = (M / 256) * wd.tiles_n;
    const int nk = K / 64;
    int sp = want_splits;
    if (sp <= 0) {
        sp = 1;
        while (tiles * sp < 200 && nk / (sp * 2) >= 4) sp *= 2;
    }
    sp = std::max(1, std::min(sp, nk));
    if (!ws || E.act == ACT_GEGLU) sp = 1;
    while (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_bytes) --sp;
    wd.kt_per_split = cdiv(nk, sp);
    wd.splits = cdiv(nk, wd.kt_per_split);
    wd.n_items = tiles * wd.splits;
    // Item order against the 8 XCD L2s (profiles/r3/wide_ring_kbench.txt, per-problem fabric traffic in profiles/r3_final/).
    // Linear order with tiles_n a multiple of 8 is weight-stationary by accident: XCD x only ever sees the column tiles = x mod 8,
    // an eighth of the weight matrix stays in its L2 and the activations cross the fabric 8 times -- the cheaper side when the
    // weights are the larger operand (32x32 / 16x16 levels: 6.5 / 26 MB of weights against 10 / 5 MB of activations; contiguous
    // ranges measured 7-8 % slower there). At 64x64 (tiles_n = 20, 1.6 MB of weights, 21 MB of activations) linear order sends
    // every stripe to every XCD for nothing: contiguous ranges are 4-5 % faster.
    static const char* xcd_env = dev_env("GL_WIDE_XCD");
    wd.xcd = xcd_env ? atoi(xcd_env) : (wd.tiles_n % 8 != 0 && wd.n_items >= 512);
    if (wd.n_items < 256 || std::min(wd.n_items, 256) % 8) wd.xcd = 0;
    g_last_cfg[0] = 8; g_last_cfg[1] = tn; g_last_cfg[2] = wd.splits;
    snprintf(g_last_name, sizeof g_last_name, "gemm_wide_kernel<%d>%s", tn, wd.splits > 1 ? " + splitk_reduce_kernel" : "");
    dim3 grid(std::min(wd.n_items, 256)), block(512);
    const size_t lds = GL_WIDE_NST * (256 + bn) * 128;
#define GL_LAUNCH_WIDE(KFN)                                                                                      \
    do {                                                                                                         \
        auto kfn = KFN;                                                                                          \
        static bool attr_done = false;                                                                           \
        if (!attr_done) {                                                                                        \
            GL_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_done = true;                                                                                    \
        }                                                                                                        \
        hipLaunchKernelGGL(kfn, grid, block, lds, stream, A, W, M, N, K, E, ws, wd);                             \
    } while (0)
    if (tn == 5) GL_LAUNCH_WIDE(gemm_wide_kernel<5>);
    else GL_LAUNCH_WIDE(gemm_wide_kernel<4>);
#undef GL_LAUNCH_WIDE
    GL_LAUNCH_CHECK();
    if (wd.splits > 1) {
        int64_t total = (int64_t)M * (N / 4);
        int blocks = (int)fmin((double)cdiv64(total, 256), 4096.0);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, wd.splits, M, N, E);
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

struct TunedCfg { int c, sp, grid; };
static std::unordered_map<std::string, TunedCfg> g_tuned;   // process-wide, guarded by g_tune_mu (ctypes drops the GIL during calls)
static std::mutex g_tune_mu;
static int g_autotune = -1;
void gemm_set_autotune_impl(int on) { g_autotune = on; }

// Does this problem go to conv_halo_kernel? (one predicate for the launcher and for gemm_gn_prologue_supported)
// eligible 3x3 convs with M >= 256 * GL_CONV_HALO (default 8; 0 = never) go to the halo kernel: at M = 512 (the 8 x 8 level) its
// 16 tiles x deep split lose to the 64 x 160 tiles of gemm_u_kernel
static bool routes_to_halo(const AOperand& A, int M, int N, int K, const Epilogue& E) {
    static const int halo = dev_env("GL_CONV_HALO") ? atoi(dev_env("GL_CONV_HALO")) : 8;
    if (!halo || gemm_variant() != 4 || g_force_tm || N < 128 || (E.bias2 && E.res)) return false;
    return halo_eligible(A, M, N, K, E) && M >= halo * 256;   // (halo_eligible includes the 2 GiB operand limit of the buffer loader)
}
int gemm_p_launch(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E, float* ws, size_t ws_bytes,
                  hipStream_t stream) {
    // candidates 0-3: 4 waves on a 2-stage ring, two (or three) workgroups per CU.
    // Measured and dropped (twice: round 1 sweeps, round 2 on-device autotune over all 107 problems of the benchmark, 0 wins):
    // the same tiles on a 4-stage ring (three K tiles in flight) with ONE workgroup per CU for the <= 256-item problems of the
    // 16x16 / 8x8 UNet levels; and (round 1) an 8-wave 256-row tile on a 3-stage ring.
    // candidate 4 (round 3): 64 x 64 tiles, 32 KB of LDS, up to four workgroups per CU -- for the M = 2048 / 512 problems of the
    // 16x16 / 8x8 levels, whose 128 / 64-row tiles leave each CU one or two K-tile-deep latency chains (v5 kernel only)
    constexpr int NC = 5;
    static const int kTm[NC] = {4, 4, 2, 2, 2}, kTn[NC] = {5, 4, 5, 4, 2};
    static const int kSp[10] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    const int nk = K / 64;
    // v5 addresses both operands through 32-bit buffer offsets: every operand must be < 2 GiB
    const size_t a_rows = A.mode == A_CONV3 ? (size_t)(M / (A.Ho * A.Wo)) * A.Hin * A.Win : (size_t)M;
    const bool fits32 = a_rows * (size_t)std::max(A.ld0, A.ld1) * 2 < 0x7fff0000ull && (size_t)N * K * 2 < 0x7fff0000ull;
    // v5's epilogue has no bias2 + residual form, and the per-sample bias only in its conv instantiations
    const bool use_u = gemm_variant() == 4 && fits32 && !(E.bias2 && (E.res || A.mode == A_ROWS));
    // (GL_CONV_HALO_SPLITS=n forces the halo kernel's K split)
    static const int halo_splits = dev_env("GL_CONV_HALO_SPLITS") ? atoi(dev_env("GL_CONV_HALO_SPLITS")) : 0;
    if (use_u && routes_to_halo(A, M, N, K, E))
        return launch_halo(A, W, M, N, K, E, ws, ws_bytes, halo_splits, stream);
    // The wide kernel takes the GEGLU projections (GL_GEMM_WIDE=1, default): 0.78-0.82x the time of gemm_u_kernel's 128x128 tiles at the
    // 64x64 / 32x32 levels, even below. Everything else it is eligible for is slower there (narrow N: 256-row tiles leave CUs idle or
    // need a K split) or within 4 % (FF-out): GL_GEMM_WIDE=2 sends all of it for A/B runs, 0 none. (profiles/r2_final/wide_kbench.txt)
    static const int wide = dev_env("GL_GEMM_WIDE") ? atoi(dev_env("GL_GEMM_WIDE")) : 1;
    static const int wide_splits = dev_env("GL_GEMM_WIDE_SPLITS") ? atoi(dev_env("GL_GEMM_WIDE_SPLITS")) : 0;
    if (wide && use_u && !g_force_tm && (wide >= 2 || E.act == ACT_GEGLU) && wide_eligible(A, M, N, K, E))
        return launch_wide(A, W, M, N, K, E, ws, ws_bytes, wide_splits, stream);
    static const int xcd_boxes = dev_env("GL_GEMM_XCD_BOXES") ? atoi(dev_env("GL_GEMM_XCD_BOXES")) : 1;

    auto feasible = [&](int c, int& sp) {
        const int tm = kTm[c], tn = kTn[c];
        if (c == 4 && !use_u) return false;
        if (E.act == ACT_GEGLU && (tn & 1)) return false;
        if (E.mode == EPI_QKV_HEADS && (sp > 1 || (2 * E.C) % (tn * 32) || (tm == 4 && tn == 5))) return false;   // an item must not straddle the k | v boundary
        if (E.mode == EPI_QK_HEADS && E.ln_stats && (sp > 1 || (tm == 4 && tn == 5))) return false;                   // (same kernel family, no V third)
        if (sp > 1 && (!ws || nk / sp < 2 || (size_t)sp * M * N * sizeof(float) > ws_bytes)) return false;
        const int kps = cdiv(nk, sp);
        sp = cdiv(nk, kps);
        return true;
    };
    auto run_cfg = [&](int c, int sp, int grid_cap) -> int {
        WorkDesc wd;
        const int tm = kTm[c], tn = kTn[c];
        wd.tiles_n = cdiv(N, tn * 32);
        wd.kt_per_split = cdiv(nk, sp);
        wd.splits = cdiv(nk, wd.kt_per_split);
        wd.n_items = cdiv(M, tm * 32) * wd.tiles_n * wd.splits;
        wd.box = -1; wd.rm = 0; wd.rz = wd.splits;
        if (use_u && xcd_boxes) {
            // fabric-side bytes ~ A_bytes * (#N bands) + W_bytes * (#M bands); only exact partitions (all boxes equal)
            const int tiles_m = cdiv(M, tm * 32), tiles_n = wd.tiles_n;
            const double a_bytes = (double)a_rows * (A.C0 + A.C1) * 2, w_bytes = (double)N * K * 2;
            double best = 1e300;
            for (int lgm = 3; lgm >= 0; --lgm)
                for (int lgn = 3 - lgm; lgn >= 0; --lgn) {
                    const int lgz = 3 - lgm - lgn;
                    if (tiles_m % (1 << lgm) || tiles_n % (1 << lgn) || wd.splits % (1 << lgz)) continue;
                    const double cost = a_bytes * (1 << lgn) + w_bytes * (1 << lgm);
                    if (cost < best) {
                        best = cost;
                        wd.box = lgm | lgn << 4;
                        wd.rm = tiles_m >> lgm; wd.tiles_n = tiles_n >> lgn; wd.rz = wd.splits >> lgz;
                    }
                }
        }
        g_last_cfg[0] = tm; g_last_cfg[1] = tn; g_last_cfg[2] = wd.splits;
        // row statistics for a folded LayerNorm downstream: only the staged row-major epilogue of gemm_u_kernel produces them
        // (one partial per row and wave column block of tn * 16 columns)
        g_last_stats_nb = (E.stats_out && use_u && wd.splits == 1 && A.mode == A_ROWS && E.mode == EPI_ROWMAJOR && !E.out_f32 && E.act != ACT_GEGLU &&
                           E.act != ACT_GELU && !(E.res && E.act == ACT_SILU) /* (epilogue_staged does not take that combination) */ &&
                           N % (tn * 16) == 0 && N / (tn * 16) <= E.stats_ld)
                              ? N / (tn * 16) : 0;
        if (E.ln_stats && (!use_u || A.mode != A_ROWS || wd.splits > 1 || (tm == 4 && tn == 5) || (E.mode != EPI_QKV_HEADS && E.mode != EPI_QK_HEADS)))
            return set_error(GL_ERR_UNSUPPORTED, "gemm: the folded-LayerNorm epilogue exists for the head layouts of gemm_u_kernel and the GEGLU form of gemm_wide_kernel");
        Epilogue Ek = E;                                   // what the kernels see: no statistics pointer unless this launch produces them
        if (!g_last_stats_nb) Ek.stats_out = nullptr;     // (the staged epilogue writes whenever the pointer is set)
        const Epilogue& E = Ek;
        if (use_u && E.mode == EPI_QKV_HEADS) snprintf(g_last_name, sizeof g_last_name, "gemm_u_kernel<2, %d, %d, 0, 2, true>", tm, tn);
        else if (use_u) snprintf(g_last_name, sizeof g_last_name, "gemm_u_kernel<2, %d, %d, %d, 2, false>", tm, tn, A.mode);
        else snprintf(g_last_name, sizeof g_last_name, "gemm_p_kernel<%d, %d, %d>", tm, tn, A.mode);
        if (wd.splits > 1) strncat(g_last_name, " + splitk_reduce_kernel", sizeof g_last_name - strlen(g_last_name) - 1);
        const int saved_grid = g_force_grid;
        if (grid_cap) g_force_grid = grid_cap;
        int rc;
        if (use_u) {
            switch (c) {
                case 0: rc = launch_u<2, 4, 5, 2>(A, W, M, N, K, E, ws, wd, stream); break;
                case 1: rc = launch_u<2, 4, 4, 2>(A, W, M, N, K, E, ws, wd, stream); break;
                case 2: rc = launch_u<2, 2, 5, 2>(A, W, M, N, K, E, ws, wd, stream); break;
                case 3: rc = launch_u<2, 2, 4, 2>(A, W, M, N, K, E, ws, wd, stream); break;
                case 4: rc = launch_u<2, 2, 2, 2>(A, W, M, N, K, E, ws, wd, stream); break;
                default: g_force_grid = saved_grid; return set_error(GL_ERR_UNSUPPORTED, "gemm: unknown tile candidate");
            }
        } else {
            switch (c) {
                case 0: rc = launch_p<4, 5>(A, W, M, N, K, E, ws, wd, stream); break;
                case 1: rc = launch_p<4, 4>(A, W, M, N, K, E, ws, wd, stream); break;
                case 2: rc = launch_p<2, 5>(A, W, M, N, K, E, ws, wd, stream); break;
                default: rc = launch_p<2, 4>(A, W, M, N, K, E, ws, wd, stream); break;
            }
        }
        g_force_grid = saved_grid;
        GL_TRY(rc);
        if (wd.splits > 1) {
            int64_t total = (int64_t)M * (N / 4);
            int blocks = (int)fmin((double)cdiv64(total, 256), 4096.0);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, wd.splits, M, N, E);
            GL_LAUNCH_CHECK();
        }
        return GL_OK;
    };

    // ---- developer override (kbench sweeps): exactly this tile / split if it fits the problem
    if (g_force_tm) {
        for (int c = 0; c < NC; ++c) {
            if (kTm[c] != g_force_tm || kTn[c] != g_force_tn) continue;
            for (int si = 0; si < 10; ++si) {
                int sp = kSp[si];
                if (g_force_splits && sp != g_force_splits) continue;
                if (!feasible(c, sp)) continue;
                if (g_force_splits && sp != g_force_splits) continue;
                if (!g_force_splits && si > 0) break;  // no split given: unsplit
                return run_cfg(c, sp, 0);
            }
        }
    }

    // ---- autotuned choice
    if (g_autotune < 0) {
        const char* e = dev_env("GL_GEMM_AUTOTUNE");
        g_autotune = e ? atoi(e) : 1;
    }
    char key[160];
    snprintf(key, sizeof key, "%d,%d,%d|%d,%d,%d,%d,%d,%d,%d|%d,%d,%d,%d,%d,%d|%d%s", M, N, K, A.mode, A.C0, A.C1, A.stride, A.ups, A.Win,
             A.Hin, E.mode, E.act, E.res != nullptr, E.bias2 != nullptr, E.out_f32, E.gate != nullptr, (int)use_u,
             (E.mode == EPI_QK_HEADS && E.ln_stats) ? "|ln" : "");   // (q-only projection behind a folded LayerNorm: another kernel family)
    std::unique_lock<std::mutex> tune_lock(g_tune_mu);
    if (g_tuned.empty() && use_u && !dev_env("GL_GEMM_NO_TABLE")) {
        // shipped choices for the problems of the benchmark configurations (generated by tools/make_tuned_table.py from an
        // autotune log taken on MI355X with 10 timed launches per candidate): deterministic kernel selection run to run
        static const struct { const char* key; int c, sp, grid; } kTable[] = {
#include "gemm_tuned.inc"
        };
        for (const auto& e : kTable) g_tuned.emplace(e.key, TunedCfg{e.c, e.sp, e.grid});
    }
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) {
        const TunedCfg t = it->second;
        tune_lock.unlock();
        return run_cfg(t.c, t.sp, t.grid);
    }

    // ---- analytic model (also the capture-time / tuning-off fallback)
    double best_t = 1e30;
    int best_c = -1, best_sp = 1;
    for (int c = 0; c < 4; ++c) {   // (the analytic model was fitted without the 64 x 64 candidate: the autotuner alone may pick it)
        const int tm = kTm[c], tn = kTn[c];
        const int bm = tm * 32, bn = tn * 32;
        const int tiles = cdiv(M, bm) * cdiv(N, bn);
        for (int si = 0; si < 10; ++si) {
            int sp = kSp[si];
            if (!feasible(c, sp)) continue;
            const int kps = cdiv(nk, sp);
            const int items = tiles * sp;
            const int per_block = cdiv(items, 512);
            // cycles per K tile of one block with two blocks per CU, fitted to kbench sweeps on MI355X
            double t_kt = 0.15 * bm * bn + 500.0;
            if (items <= 256) t_kt *= 0.75;
            const double t_item = kps * t_kt + (E.act == ACT_GEGLU ? 9000.0 : 6000.0) * (bm * bn / 20480.0);
            double tt = per_block * t_item;
            if (sp > 1) tt += 6000.0 + (double)sp * M * N * 8.0 / 2000.0;  // fp32 slabs out and back + reduce launch
            if (tt < best_t) { best_t = tt; best_c = c; best_sp = sp; }
        }
    }
    if (best_c < 0) return set_error(GL_ERR_ARG, "gemm: no tile configuration for M=%d N=%d K=%d", M, N, K);

    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(stream, &cap);
    if (!g_autotune || cap != hipStreamCaptureStatusNone) {
        tune_lock.unlock();
        return run_cfg(best_c, best_sp, 0);
    }

    // timing events live on the device that is current for this call (the context's device), for this tuning pass only;
    // the lock is held across the pass: two threads tuning at once would time each other's launches
    GL_HIP(hipDeviceSynchronize());     // a quiet chip: work another execution context has in flight on its own stream must not share the timed launches
    hipEvent_t g_tune_ev[2];
    GL_HIP(hipEventCreate(&g_tune_ev[0]));
    GL_HIP(hipEventCreate(&g_tune_ev[1]));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); } } ev_guard{g_tune_ev};
    TunedCfg win{best_c, best_sp, 0};
    float win_ms = 1e30f;
    static const int tune_reps = dev_env("GL_GEMM_TUNE_REPS") ? std::max(1, atoi(dev_env("GL_GEMM_TUNE_REPS"))) : 3;
    // GL_GEMM_TUNE_CORUN=1 (developer): the candidates are timed as TWO streams running the same launches side by side -- the cost of a
    // launch while another batch shares the chip (bench.py / the CLI keep two batches in flight), not its latency on a quiet chip: a
    // tile that leaves CUs to the neighbour is credited for it. Adds resident-grid caps below two workgroups per CU to the candidates.
    static const bool corun = dev_env("GL_GEMM_TUNE_CORUN") && atoi(dev_env("GL_GEMM_TUNE_CORUN")) != 0;
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    struct CoGuard { hipStream_t& s; hipEvent_t &a, &b; ~CoGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); if (s) (void)hipStreamDestroy(s); } } co_guard{s2, ev_fork, ev_join};
    if (corun) {
        GL_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        GL_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        GL_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    for (int c = 0; c < NC; ++c) {
        const int tiles = cdiv(M, kTm[c] * 32) * cdiv(N, kTn[c] * 32);
        if (c == 4 && ((size_t)M * N > ((size_t)1 << 23) || N % 64)) continue;   // small problems only (M N <= 8 M outputs: 2048 x 3840, 8192 x 640 ..)
        int last_sp = -1;
        for (int si = 0; si < 10; ++si) {
            int sp = kSp[si];
            if (!feasible(c, sp) || sp == last_sp) continue;
            last_sp = sp;
            if (sp > 1 && tiles * sp > 4096) continue;      // splitting an already over-subscribed grid never paid
            for (int gi = 0; gi < (corun ? 5 : 3); ++gi) {
                const int grid = gi == 0 ? 0 : gi == 1 ? 768 : gi == 2 ? 1024 : gi == 3 ? 256 : 128;
                if (gi == 1 && (kTm[c] * 32 + kTn[c] * 32 > 192 || tiles * sp <= 512)) continue;  // 3 workgroups/CU need <= 48 KB LDS each
                if (gi == 2 && (kTm[c] * 32 + kTn[c] * 32 > 128 || tiles * sp <= 768)) continue;  // 4 workgroups/CU: the 64 x 64 tile (32 KB)
                if (gi == 3 && tiles * sp <= 256) continue;                                        // (co-run tuning) one workgroup per CU
                if (gi == 4 && tiles * sp <= 128) continue;                                        // (co-run tuning) half the CUs
                GL_TRY(run_cfg(c, sp, grid));  // warm-up (also sets the kernel's LDS attribute outside the timed region)
                GL_HIP(hipEventRecord(g_tune_ev[0], stream));
                if (corun) {
                    GL_HIP(hipEventRecord(ev_fork, stream));
                    GL_HIP(hipStreamWaitEvent(s2, ev_fork, 0));
                    for (int r = 0; r < tune_reps; ++r) {
                        GL_TRY(run_cfg(c, sp, grid));
                        hipStream_t keep = stream;
                        stream = s2;                       // run_cfg launches on `stream` (captured by reference)
                        const int rc2 = run_cfg(c, sp, grid);
                        stream = keep;
                        GL_TRY(rc2);
                    }
                    GL_HIP(hipEventRecord(ev_join, s2));
                    GL_HIP(hipStreamWaitEvent(stream, ev_join, 0));
                } else
                for (int r = 0; r < tune_reps; ++r) GL_TRY(run_cfg(c, sp, grid));
                GL_HIP(hipEventRecord(g_tune_ev[1], stream));
                GL_HIP(hipEventSynchronize(g_tune_ev[1]));
                float ms = 0.f;
                GL_HIP(hipEventElapsedTime(&ms, g_tune_ev[0], g_tune_ev[1]));
                if (ms < win_ms) { win_ms = ms; win = TunedCfg{c, sp, grid}; }
            }
        }
    }
    g_tuned[key] = win;
    tune_lock.unlock();
    static const bool tune_log = dev_env("GL_GEMM_TUNE_LOG") != nullptr;
    if (tune_log)
        fprintf(stderr, "[gemm autotune] %s -> %dx%d / %d splits @%d (%.1f us; model said %dx%d / %d) cfg %d %d %d\n", key, kTm[win.c] * 32,
                kTn[win.c] * 32, win.sp, win.grid ? win.grid : 512, win_ms * 1e3f / tune_reps, kTm[best_c] * 32, kTn[best_c] * 32, best_sp,
                win.c, win.sp, win.grid);
    return run_cfg(win.c, win.sp, win.grid);
}

}  // namespace

void gemm_set_autotune(int on) { gemm_set_autotune_impl(on); }

// AOperand::gn (GroupNorm-apply + SiLU inside the conv's loader) exists in conv_halo_kernel only, for tiles that lie inside one
// image (H W a multiple of 256: the tile's 256 pixels share one sample's coefficients)
bool gemm_gn_prologue_supported(const AOperand& A, int M, int N, int K, const Epilogue& E) {
    return K % 64 == 0 && routes_to_halo(A, M, N, K, E) && (A.Hin * A.Win) % 256 == 0;
}

// Can a GEMM with this epilogue consume raw rows + row statistics instead of LayerNorm'ed rows (Epilogue::ln_stats)?
// Mirrors the routing of gemm_launch: the head-layout epilogues of gemm_u_kernel, the GEGLU epilogue of gemm_wide_kernel.
bool gemm_ln_fold_supported(const AOperand& A, int M, int N, int K, const Epilogue& E) {
    if (gemm_variant() != 4 || A.mode != A_ROWS || A.C1 || N < 128 || K % 64) return false;
    if ((size_t)M * (size_t)A.ld0 * 2 >= 0x7fff0000ull || (size_t)N * K * 2 >= 0x7fff0000ull) return false;
    static const int wide = dev_env("GL_GEMM_WIDE") ? atoi(dev_env("GL_GEMM_WIDE")) : 1;
    if (E.act == ACT_GEGLU) return wide && E.mode == EPI_ROWMAJOR && wide_eligible(A, M, N, K, E);
    return E.mode == EPI_QKV_HEADS || E.mode == EPI_QK_HEADS;
}

int gemm_launch(const AOperand& A, const bf16* W, int M, int N, int K, const Epilogue& E_in, float* ws,
                size_t ws_bytes, hipStream_t stream) {
    g_last_stats_nb = 0;
    Epilogue E = E_in;
    if (E.rows_per_b < 1) return set_error(GL_ERR_ARG, "gemm: rows_per_b=%d", E.rows_per_b);
    {   // divide-free m / rows_per_b for the epilogues (div_rpb)
        const unsigned d = (unsigned)E.rows_per_b;
        int sh = 0;
        while ((1ull << sh) < d) ++sh;
        E.rpb_shift = sh;
        E.rpb_magic = (unsigned)((((1ull << sh) - d) << 32) / d + 1);
    }
    if (M <= 0 || N <= 0 || K <= 0) return set_error(GL_ERR_ARG, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    if (E.ln_stats && (!E.ln_csum || !E.bias || E.ln_nb < 1 || E.ln_ld < E.ln_nb || !gemm_ln_fold_supported(A, M, N, K, E)))
        return set_error(GL_ERR_UNSUPPORTED, "gemm: folded LayerNorm needs csum + folded bias + statistics, and an epilogue that applies them");
    if (K % 64 != 0) return set_error(GL_ERR_ARG, "gemm: K=%d must be a multiple of 64", K);
    if (N % 4 != 0) return set_error(GL_ERR_ARG, "gemm: N=%d must be a multiple of 4", N);
    if (A.gn && !gemm_gn_prologue_supported(A, M, N, K, E))
        return set_error(GL_ERR_UNSUPPORTED, "gemm: the GroupNorm prologue (AOperand::gn) exists in conv_halo_kernel only (3x3, stride 1, H W %% 256 == 0)");
    if (A.mode == A_CONV3) {
        if ((A.C0 + A.C1) % 64 != 0 || A.C0 % 64 != 0 || K != 9 * (A.C0 + A.C1))
            return set_error(GL_ERR_ARG, "conv3x3: channels (%d,%d) must be multiples of 64 and K=9*Cin (K=%d)", A.C0, A.C1, K);
    } else {
        if (K != A.C0 + A.C1 || (A.C1 && A.C0 % 64 != 0))
            return set_error(GL_ERR_ARG, "gemm: K=%d does not match operand channels (%d,%d)", K, A.C0, A.C1);
    }
    if (A.mode == A_CONV3 && (E.gate || E.remap_in))
        return set_error(GL_ERR_UNSUPPORTED, "conv3x3: the gated residual and the row remap are row-GEMM epilogues");
    if (E.act == ACT_GELU && (E.res || E.bias2 || A.mode != A_ROWS))
        return set_error(GL_ERR_UNSUPPORTED, "gemm: the GELU epilogue has no residual / broadcast-bias form");
    if (E.act == ACT_GEGLU && (N % 32 != 0 || E.mode != EPI_ROWMAJOR))
        return set_error(GL_ERR_ARG, "gemm: GEGLU epilogue needs packed N %% 32 == 0 (N=%d)", N);
    if (E.act == ACT_GEGLU && E.geglu16 != gemm_geglu_layout())
        return set_error(GL_ERR_STATE, "gemm: GEGLU weights were packed for a different main-loop variant");
    if (E.mode == EPI_QKV_HEADS) {
        if (A.mode != A_ROWS || A.C1) return set_error(GL_ERR_ARG, "gemm: EPI_QKV_HEADS takes a single row-major activation operand");
        const size_t a_bytes = (size_t)M * A.ld0 * 2;
        if (!gemm_supports_qkv() || a_bytes >= 0x7fff0000ull || (size_t)N * K * 2 >= 0x7fff0000ull || N != 3 * E.C || (2 * E.C) % 128 || !E.vt || !E.q || !E.k ||
            E.T % 64 || M % E.T)
            return set_error(GL_ERR_UNSUPPORTED, "gemm: EPI_QKV_HEADS needs the v5 main loop, N = 3C with 2C %% 128 == 0, tokens per sample %% 64 == 0");
    }
    if (gemm_variant() >= 2 && (N >= 128 || E.act == ACT_GEGLU)) return gemm_p_launch(A, W, M, N, K, E, ws, ws_bytes, stream);

    // pick the tile: padding efficiency x relative tile speed x chip fill
    int best = 0;
    float best_score = -1.f;
    for (int c = 0; c < 4; ++c) {
        if (c == 3 && N > 32) continue;
        if (c != 3 && N <= 32) continue;
        const Cfg& cf = kCfgs[c];
        double tm = cdiv(M, cf.bm), tn = cdiv(N, cf.bn);
        double pad = ((double)M * N) / (tm * cf.bm * tn * cf.bn);
        double fill = fmin(1.0, tm * tn / 256.0);
        float score = (float)(pad * cf.speed * (0.35 + 0.65 * fill));
        if (score > best_score) { best_score = score; best = c; }
    }
    const Cfg& cf = kCfgs[best];
    snprintf(g_last_name, sizeof g_last_name, "gemm_glds_kernel<%dx%d, %d>", cf.bm, cf.bn, A.mode);
    const int tiles = cdiv(M, cf.bm) * cdiv(N, cf.bn);
    const int nk = K / 64;
    int splits = 1;
    if (ws && tiles < 256 && nk >= 8) {
        splits = min(min(cdiv(512, tiles), nk / 4), 16);
        while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    }
    int kt_per_split = cdiv(nk, splits);
    splits = cdiv(nk, kt_per_split);
    g_last_cfg[0] = cf.bm / 32; g_last_cfg[1] = cf.bn / 32; g_last_cfg[2] = splits;

    int rc;
    switch (best) {
        case 0: rc = launch_cfg<2, 2, 2, 2>(A, W, M, N, K, E, ws, splits, kt_per_split, stream); break;
        case 1: rc = launch_cfg<2, 2, 2, 1>(A, W, M, N, K, E, ws, splits, kt_per_split, stream); break;
        case 2: rc = launch_cfg<2, 2, 1, 1>(A, W, M, N, K, E, ws, splits, kt_per_split, stream); break;
        default: rc = launch_cfg<4, 1, 1, 1>(A, W, M, N, K, E, ws, splits, kt_per_split, stream); break;
    }
    GL_TRY(rc);
    if (splits > 1) {
        int64_t total = (int64_t)M * (N / 4);
        int blocks = (int)fmin((double)cdiv64(total, 256), 4096.0);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, splits, M, N, E);
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

int gemm_launch_t(const bf16* Wrows, int Mw, const bf16* X, int Nx, int K, const Epilogue& E, hipStream_t stream) {
    AOperand A;
    aoperand_rows(A, Wrows, K, K);
    return gemm_launch(A, X, Mw, Nx, K, E, nullptr, 0, stream);
}

}  // namespace gl
