// HBM-bound normalisation kernels for gfx950: GroupNorm(32)+SiLU over NHWC bf16 (two-source
// channel concat supported), LayerNorm (with the GLIGEN [visual ; grounding] row concat), and a
// row softmax for the VAE's single-head attention.
//
// Reference semantics:
//   GroupNorm32 (fp32 statistics, eps 1e-5)      ldm/modules/diffusionmodules/util.py:223-226
//   Normalize   (GroupNorm 32 groups, eps 1e-6)  ldm/modules/attention.py:76-77, model.py:38-39
//   nn.LayerNorm (eps 1e-5)                      ldm/modules/attention.py:225-226,309-311
// All statistics are accumulated in fp32 (cross-block combination in fp64) with a fixed
// reduction order, so results are bit-reproducible run to run.
#include "norm.h"

namespace gl {

// Both kernels use the same thread shape: block = C8 * R threads, thread (cx, ry) owns the 8 channels [8cx, 8cx+8)
// (at most two groups) and walks pixels ry, ry+R, ... of its block's pixel range, UNR pixels per trip with all UNR
// 16-byte loads issued before the first use (these kernels are latency-bound otherwise: one load in flight per thread
// ran at 1.1 TB/s).
constexpr int GN_UNR = 8;
static int gn_rows(int C8) { return C8 >= 256 ? 1 : 256 / C8; }
int gn_nsplit_c(int HW, int C8) { return max(1, min(64, cdiv(HW, gn_rows(C8) * GN_UNR))); }
int gn_nsplit(int HW) { return max(1, min(64, HW / 8)); }  // upper bound over all channel counts (workspace sizing)
size_t gn_partial_bytes(int B, int HW) { return (size_t)B * 64 * 32 * 2 * sizeof(float); }

// ---- GroupNorm statistics: grid (nsplit, B)
__global__ void gn_stats_kernel(GNParams P, int nsplit, int C8, int R, int cpg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* tl = reinterpret_cast<float4*>(smem);
    const int t = threadIdx.x;
    const int cx = t % C8;
    const int ry = t / C8;
    const int b = blockIdx.y;
    const int c0 = cx * 8;
    const bool first = c0 < P.C0;
    const int ld = first ? P.C0 : P.C1;
    const bf16* base = (first ? P.x0 + c0 : P.x1 + (c0 - P.C0)) + (size_t)b * P.HW * ld;
    const int per = (P.HW + nsplit - 1) / nsplit;
    const int p0 = blockIdx.x * per;
    const int p1 = min(P.HW, p0 + per);
    const int g_lo = c0 / cpg;
    bool hi[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) hi[p] = (c0 + 2 * p) / cpg != g_lo;

    float s_lo = 0.f, q_lo = 0.f, s_hi = 0.f, q_hi = 0.f;
    for (int p = p0 + ry; p < p1; p += R * GN_UNR) {
        // UNCONDITIONAL loads (a trip's pixels beyond the block's range re-read its last pixel and count as zero): behind a
        // per-lane guard hipcc folds the bf16 unpacking into the guarded block and waits for every load right behind its issue --
        // one load in flight per thread, the latency-bound form the unrolling was meant to avoid (seen in the ISA, round 3)
        U4BF8 v[GN_UNR];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) v[u].u = *reinterpret_cast<const uint4*>(base + (size_t)min(p + u * R, p1 - 1) * ld);
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const bool ok = p + u * R < p1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = ok ? bf2f(v[u].e[e]) : 0.f;
                if (hi[e >> 1]) { s_hi += f; q_hi += f * f; }
                else            { s_lo += f; q_lo += f * f; }
            }
        }
    }
    tl[t] = make_float4(s_lo, q_lo, s_hi, q_hi);
    __syncthreads();
    if (t < 32) {
        const int g = t;
        const int cx_lo = (g * cpg) / 8;
        const int cx_hi = ((g + 1) * cpg - 1) / 8;
        float s = 0.f, q = 0.f;
        for (int x = max(0, cx_lo - 1); x <= cx_hi; ++x) {
            const int gl_x = (x * 8) / cpg;
            for (int y = 0; y < R; ++y) {
                float4 v = tl[y * C8 + x];
                if (gl_x == g) { s += v.x; q += v.y; }
                if (gl_x + 1 == g) { s += v.z; q += v.w; }
            }
        }
        float* dst = P.partial + (((size_t)b * nsplit + blockIdx.x) * 32 + g) * 2;
        dst[0] = s;
        dst[1] = q;
    }
}

// ---- GroupNorm apply (+SiLU): grid (nblk, B), same thread shape; y = x * a + c with a = rstd * gamma, c = beta - mean * a
__global__ void gn_apply_kernel(GNParams P, int nsplit, int C8, int R, int cpg, int per) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ double red_s[4][32][2];
    const int t = threadIdx.x;
    const int b = blockIdx.y;
    // per-group sums over the nsplit partials: 4 x 32 threads, each 16 independent loads, then a fixed-order combine
    if (t < 128) {
        const int g = t & 31, part = t >> 5;
        double s = 0.0, q = 0.0;
        const float2* src = reinterpret_cast<const float2*>(P.partial) + (size_t)b * nsplit * 32 + g;
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = src[(size_t)min(part + 4 * i, nsplit - 1) * 32];   // unconditional: all 16 in flight together
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = part + 4 * i < nsplit;
            s += ok ? (double)v[i].x : 0.0;
            q += ok ? (double)v[i].y : 0.0;
        }
        red_s[part][g][0] = s;
        red_s[part][g][1] = q;
    }
    __syncthreads();
    if (t < 32) {
        const double s = ((red_s[0][t][0] + red_s[1][t][0]) + red_s[2][t][0]) + red_s[3][t][0];
        const double q = ((red_s[0][t][1] + red_s[1][t][1]) + red_s[2][t][1]) + red_s[3][t][1];
        const double n = (double)P.HW * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[t] = (float)mean;
        rstd_s[t] = (float)(1.0 / sqrt(var + (double)P.eps));
    }
    __syncthreads();
    const int cx = t % C8;
    const int ry = t / C8;
    const int c0 = cx * 8;
    const int C = P.C0 + P.C1;
    const bool first = c0 < P.C0;
    const int ld = first ? P.C0 : P.C1;
    const bf16* base = (first ? P.x0 + c0 : P.x1 + (c0 - P.C0)) + (size_t)b * P.HW * ld;
    bf16* obase = P.y + (size_t)b * P.HW * C + c0;
    float a[8], c[8];
    {
        const float4 g0 = *reinterpret_cast<const float4*>(P.gamma + c0);
        const float4 g1 = *reinterpret_cast<const float4*>(P.gamma + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(P.beta + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(P.beta + c0 + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c0 + e) / cpg;
            a[e] = rstd_s[g] * gm[e];
            c[e] = bt[e] - mean_s[g] * a[e];
        }
    }
    const int p0 = blockIdx.x * per;
    const int p1 = min(P.HW, p0 + per);
    for (int p = p0 + ry; p < p1; p += R * GN_UNR) {
        U4BF8 v[GN_UNR];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u)   // unconditional, as in gn_stats_kernel (the store below keeps its guard)
            v[u].u = *reinterpret_cast<const uint4*>(base + (size_t)min(p + u * R, p1 - 1) * ld);
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int pp = p + u * R;
            if (pp < p1) {
                U4BF8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = fmaf(bf2f(v[u].e[e]), a[e], c[e]);
                    if (P.silu) y = silu_f(y);
                    o.e[e] = f2bf(y);
                }
                *reinterpret_cast<uint4*>(obase + (size_t)pp * C) = o.u;
            }
        }
    }
}

// ---- fused GroupNorm for the low-resolution levels (HW <= 256: the 16x16 and 8x8 UNet levels, 43 of the 61 GroupNorms of a
// forward): one workgroup per (sample, group) reads its cpg x HW slab twice (statistics, then normalise + SiLU; the slab is
// a few KB and L1/L2-resident), so the statistics never leave the CU and one launch + one dependent-kernel boundary go away.
// Same fixed-order fp32 / fp64 reduction discipline as the two-kernel path.
template <int NT>
__global__ void __launch_bounds__(NT) gn_small_kernel(GNParams P, int cpg) {
    constexpr int NW = NT / 64;
    __shared__ double red_s[NW][2];
    __shared__ float stat_s[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = blockIdx.x, b = blockIdx.y;
    const int C = P.C0 + P.C1;
    const int hp = cpg >> 2;                 // 4-channel pieces per pixel in this group (cpg % 4 == 0, C0 % 4 == 0)
    const int n_q = P.HW * hp;
    const int c_base = g * cpg;
    auto src_of = [&](int pix, int c) -> const bf16* {
        return c < P.C0 ? P.x0 + ((size_t)b * P.HW + pix) * P.C0 + c : P.x1 + ((size_t)b * P.HW + pix) * P.C1 + (c - P.C0);
    };
    float s = 0.f, q = 0.f;
    for (int idx = t; idx < n_q; idx += NT) {
        const int pix = idx / hp;
        const int c = c_base + 4 * (idx - pix * hp);
        const uint2 v = *reinterpret_cast<const uint2*>(src_of(pix, c));
        const float a0 = __uint_as_float(v.x << 16), a1 = __uint_as_float(v.x & 0xffff0000u);
        const float a2 = __uint_as_float(v.y << 16), a3 = __uint_as_float(v.y & 0xffff0000u);
        s += (a0 + a1) + (a2 + a3);
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { red_s[wave][0] = s; red_s[wave][1] = q; }
    __syncthreads();
    if (t == 0) {
        double ss = 0.0, qq = 0.0;
        for (int w = 0; w < NW; ++w) { ss += red_s[w][0]; qq += red_s[w][1]; }
        const double n = (double)P.HW * cpg;
        const double mean = ss / n;
        double var = qq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat_s[0] = (float)mean;
        stat_s[1] = (float)(1.0 / sqrt(var + (double)P.eps));
    }
    __syncthreads();
    const float mean = stat_s[0], rstd = stat_s[1];
    for (int idx = t; idx < n_q; idx += NT) {
        const int pix = idx / hp;
        const int c = c_base + 4 * (idx - pix * hp);
        const uint2 v = *reinterpret_cast<const uint2*>(src_of(pix, c));
        const float4 gm = *reinterpret_cast<const float4*>(P.gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(P.beta + c);
        float y0 = (__uint_as_float(v.x << 16) - mean) * rstd * gm.x + bt.x;
        float y1 = (__uint_as_float(v.x & 0xffff0000u) - mean) * rstd * gm.y + bt.y;
        float y2 = (__uint_as_float(v.y << 16) - mean) * rstd * gm.z + bt.z;
        float y3 = (__uint_as_float(v.y & 0xffff0000u) - mean) * rstd * gm.w + bt.w;
        if (P.silu) { y0 = silu_f(y0); y1 = silu_f(y1); y2 = silu_f(y2); y3 = silu_f(y3); }
        union { uint2 u; bf16 e[4]; } o;
        o.e[0] = f2bf(y0);
        o.e[1] = f2bf(y1);
        o.e[2] = f2bf(y2);
        o.e[3] = f2bf(y3);
        *reinterpret_cast<uint2*>(P.y + ((size_t)b * P.HW + pix) * C + c) = o.u;
    }
}

// ---- GroupNorm as the consumer's prologue (round 6): the statistics pass stays, the apply pass does not. The conv that follows
// (conv_halo_kernel, gemm.hip) normalises + SiLUs its input tile while staging it, from per-(sample, channel) coefficients
//     y = silu(x * a + c),  a = rstd * gamma,  c = beta - mean * a          (reference openaimodel.py:212-232: GroupNorm32 -> SiLU -> conv)
// written here as coef[b][C / 8][16]: a of 8 consecutive channels, then their c -- 64 bytes per 16-byte activation chunk, so a
// 64-channel chunk of the conv's K loop is one contiguous 512-byte block that the conv fetches by LDS-DMA beside its halo.
// Same fp64 fixed-order combination of the partials as gn_apply_kernel: (a, c) are bit-identical to the ones that kernel uses.
__global__ void gn_coef_kernel(GNParams P, int nsplit, int cpg) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ double red_s[4][32][2];
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    if (t < 128) {
        const int g = t & 31, part = t >> 5;
        double s = 0.0, q = 0.0;
        const float2* src = reinterpret_cast<const float2*>(P.partial) + (size_t)b * nsplit * 32 + g;
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = src[(size_t)min(part + 4 * i, nsplit - 1) * 32];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = part + 4 * i < nsplit;
            s += ok ? (double)v[i].x : 0.0;
            q += ok ? (double)v[i].y : 0.0;
        }
        red_s[part][g][0] = s;
        red_s[part][g][1] = q;
    }
    __syncthreads();
    if (t < 32) {
        const double s = ((red_s[0][t][0] + red_s[1][t][0]) + red_s[2][t][0]) + red_s[3][t][0];
        const double q = ((red_s[0][t][1] + red_s[1][t][1]) + red_s[2][t][1]) + red_s[3][t][1];
        const double n = (double)P.HW * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[t] = (float)mean;
        rstd_s[t] = (float)(1.0 / sqrt(var + (double)P.eps));
    }
    __syncthreads();
    const int C = P.C0 + P.C1;
    float* dst = P.coef + (size_t)b * C * 2;
    for (int c = t; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float a = rstd_s[g] * P.gamma[c];
        const float cc = P.beta[c] - mean_s[g] * a;
        dst[(c >> 3) * 16 + (c & 7)] = a;
        dst[(c >> 3) * 16 + 8 + (c & 7)] = cc;
    }
}

// HW <= 256 (the 16 x 16 level): gn_small_kernel's statistics half -- one workgroup per (sample, group), one read of its slab --
// writing the group's coefficients directly: one launch, no partials
template <int NT>
__global__ void __launch_bounds__(NT) gn_small_coef_kernel(GNParams P, int cpg) {
    constexpr int NW = NT / 64;
    __shared__ double red_s[NW][2];
    __shared__ float stat_s[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = blockIdx.x, b = blockIdx.y;
    const int C = P.C0 + P.C1;
    const int hp = cpg >> 2;
    const int n_q = P.HW * hp;
    const int c_base = g * cpg;
    auto src_of = [&](int pix, int c) -> const bf16* {
        return c < P.C0 ? P.x0 + ((size_t)b * P.HW + pix) * P.C0 + c : P.x1 + ((size_t)b * P.HW + pix) * P.C1 + (c - P.C0);
    };
    float s = 0.f, q = 0.f;
    for (int idx = t; idx < n_q; idx += NT) {
        const int pix = idx / hp;
        const int c = c_base + 4 * (idx - pix * hp);
        const uint2 v = *reinterpret_cast<const uint2*>(src_of(pix, c));
        const float a0 = __uint_as_float(v.x << 16), a1 = __uint_as_float(v.x & 0xffff0000u);
        const float a2 = __uint_as_float(v.y << 16), a3 = __uint_as_float(v.y & 0xffff0000u);
        s += (a0 + a1) + (a2 + a3);
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { red_s[wave][0] = s; red_s[wave][1] = q; }
    __syncthreads();
    if (t == 0) {
        double ss = 0.0, qq = 0.0;
        for (int w = 0; w < NW; ++w) { ss += red_s[w][0]; qq += red_s[w][1]; }
        const double n = (double)P.HW * cpg;
        const double mean = ss / n;
        double var = qq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat_s[0] = (float)mean;
        stat_s[1] = (float)(1.0 / sqrt(var + (double)P.eps));
    }
    __syncthreads();
    if (t < cpg) {
        const int c = c_base + t;
        const float a = stat_s[1] * P.gamma[c];
        const float cc = P.beta[c] - stat_s[0] * a;
        float* dst = P.coef + (size_t)b * C * 2;
        dst[(c >> 3) * 16 + (c & 7)] = a;
        dst[(c >> 3) * 16 + 8 + (c & 7)] = cc;
    }
}

static int gn_check(const GNParams& P, int* cpg_out) {
    const int C = P.C0 + P.C1;
    if (C % 64 != 0 || (P.C1 && P.C0 % 8 != 0)) return set_error(GL_ERR_ARG, "groupnorm: C=%d (C0=%d) unsupported", C, P.C0);
    const int cpg = C / 32;
    if (cpg < 4 || (cpg & 1)) return set_error(GL_ERR_UNSUPPORTED, "groupnorm: %d channels per group (need an even number >= 4)", cpg);
    if (C / 8 > 1024) return set_error(GL_ERR_ARG, "groupnorm: C=%d too large", C);
    *cpg_out = cpg;
    return GL_OK;
}

int groupnorm_coef_launch(const GNParams& P, hipStream_t stream) {
    int cpg;
    GL_TRY(gn_check(P, &cpg));
    if (!P.coef) return set_error(GL_ERR_ARG, "groupnorm_coef: no coefficient buffer");
    const int C = P.C0 + P.C1, C8 = C / 8;
    if (P.HW <= 256 && (P.C0 & 3) == 0 && (cpg & 3) == 0) {
        if (P.HW * cpg >= 8192) hipLaunchKernelGGL(gn_small_coef_kernel<1024>, dim3(32, P.B), dim3(1024), 0, stream, P, cpg);
        else hipLaunchKernelGGL(gn_small_coef_kernel<256>, dim3(32, P.B), dim3(256), 0, stream, P, cpg);
        GL_LAUNCH_CHECK();
        return GL_OK;
    }
    const int R = gn_rows(C8);
    const int nsplit = gn_nsplit_c(P.HW, C8);
    const int T = C8 * R;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, P.B), dim3(T), T * sizeof(float4), stream, P, nsplit, C8, R, cpg);
    GL_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_coef_kernel, dim3(P.B), dim3(256), 0, stream, P, nsplit, cpg);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
int groupnorm_coef_launches(int HW, int C0, int C1) {
    const int cpg = (C0 + C1) / 32;
    return (HW <= 256 && (C0 & 3) == 0 && (cpg & 3) == 0) ? 1 : 2;
}

int groupnorm_launch(const GNParams& P, hipStream_t stream) {
    const int C = P.C0 + P.C1;
    if (C % 64 != 0 || (P.C1 && P.C0 % 8 != 0)) return set_error(GL_ERR_ARG, "groupnorm: C=%d (C0=%d) unsupported", C, P.C0);
    const int cpg = C / 32;
    const int C8 = C / 8;
    if (cpg < 4 || (cpg & 1)) return set_error(GL_ERR_UNSUPPORTED, "groupnorm: %d channels per group (need an even number >= 4)", cpg);
    if (C8 > 1024) return set_error(GL_ERR_ARG, "groupnorm: C=%d too large", C);
    if (P.HW <= 256 && (P.C0 & 3) == 0 && (cpg & 3) == 0) {
        if (P.HW * cpg >= 8192) hipLaunchKernelGGL(gn_small_kernel<1024>, dim3(32, P.B), dim3(1024), 0, stream, P, cpg);
        else hipLaunchKernelGGL(gn_small_kernel<256>, dim3(32, P.B), dim3(256), 0, stream, P, cpg);
        GL_LAUNCH_CHECK();
        return GL_OK;
    }
    const int R = gn_rows(C8);
    const int nsplit = gn_nsplit_c(P.HW, C8);
    const int T = C8 * R;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, P.B), dim3(T), T * sizeof(float4), stream, P, nsplit, C8, R, cpg);
    GL_LAUNCH_CHECK();
    // apply: one trip (R * GN_UNR pixels) per block until the grid holds ~2048 blocks, then several trips per block
    const int trips = cdiv(P.HW, R * GN_UNR);
    const int nblk = max(1, min(trips, max(1, 2048 / P.B)));
    const int per = cdiv(cdiv(P.HW, nblk), R) * R;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(P.HW, per), P.B), dim3(T), 0, stream, P, nsplit, C8, R, cpg, per);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- LayerNorm: one wave per output row.
__global__ void __launch_bounds__(256) ln_kernel(LNParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= (int64_t)P.B * P.Tpad) return;
    const int b = (int)(r / P.Tpad);
    const int t = (int)(r - (int64_t)b * P.Tpad);
    const int C8 = P.C >> 3;
    const int ldx = P.ldx ? P.ldx : P.C, ldy = P.ldy ? P.ldy : P.C;
    bf16* dst = P.y + r * ldy;
    for (int ch = C8 + lane; ch < (ldy >> 3); ch += 64) *reinterpret_cast<uint4*>(dst + ch * 8) = make_uint4(0, 0, 0, 0);
    const bf16* src;
    if (t < P.N1) src = P.x + ((size_t)b * P.N1 + t) * ldx;
    else if (t < P.N1 + P.N2) src = P.x2 + ((size_t)b * P.N2 + (t - P.N1)) * P.C;
    else {
        for (int ch = lane; ch < C8; ch += 64) *reinterpret_cast<uint4*>(dst + ch * 8) = make_uint4(0, 0, 0, 0);
        return;
    }
    float v[3][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = lane + 64 * i;
        if (ch < C8) {
            U4BF8 u;
            u.u = *reinterpret_cast<const uint4*>(src + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(u.e[e]); s += v[i][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    const float mean = wave_sum(s) / P.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = lane + 64 * i;
        if (ch < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / P.C + P.eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = lane + 64 * i;
        if (ch < C8) {
            U4BF8 o;
            if (P.gamma) {
                const float4 g0 = *reinterpret_cast<const float4*>(P.gamma + ch * 8);
                const float4 g1 = *reinterpret_cast<const float4*>(P.gamma + ch * 8 + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(P.beta + ch * 8);
                const float4 b1 = *reinterpret_cast<const float4*>(P.beta + ch * 8 + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = f2bf((v[i][e] - mean) * rstd * gm[e] + bt[e]);
            } else {   // no affine: gamma and beta live in the consumer's folded weights (gemm.h Epilogue::ln_stats)
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = f2bf((v[i][e] - mean) * rstd);
            }
            *reinterpret_cast<uint4*>(dst + ch * 8) = o.u;
        }
    }
}

int layernorm_launch(const LNParams& P, hipStream_t stream) {
    if (P.C % 8 != 0 || P.C > 1536) return set_error(GL_ERR_ARG, "layernorm: C=%d unsupported", P.C);
    if (P.Tpad < P.N1 + P.N2) return set_error(GL_ERR_ARG, "layernorm: Tpad=%d < %d+%d", P.Tpad, P.N1, P.N2);
    if ((P.ldx && (P.ldx < P.C || P.ldx % 8)) || (P.ldy && (P.ldy < P.C || P.ldy % 8))) return set_error(GL_ERR_ARG, "layernorm: bad row strides");
    const int64_t rows = (int64_t)P.B * P.Tpad;
    hipLaunchKernelGGL(ln_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, stream, P);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- row softmax (fp32 -> bf16), one block per row
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ S, bf16* __restrict__ Pm, int cols, float scale) {
    __shared__ float red[4];
    const float* row = S + (size_t)blockIdx.x * cols;
    bf16* out = Pm + (size_t)blockIdx.x * cols;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float mx = -1e30f;
    for (int c = t; c < cols; c += 256) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = t; c < cols; c += 256) s += __expf((row[c] - mx) * scale);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = t; c < cols; c += 256) out[c] = f2bf(__expf((row[c] - mx) * scale) * inv);
}

int softmax_rows_launch(const float* S, bf16* Pm, int rows, int cols, float scale, hipStream_t stream) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, S, Pm, cols, scale);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace gl
