// storebench — throughput of the GEMM epilogue's global store / residual-load patterns. Developer tool.
//   pattern 0: MFMA-fragment stores, lane (r = l&15, q = l>>4) writes 8 B at row r, byte q*8 of a 32-B run; 20 stores/lane
//   pattern 1: same addresses, 20 loads then 20 stores (residual + out)
//   pattern 2: row-contiguous 16-B stores: a wave writes 4 full 256-B row pieces per instruction
//   pattern 3: row-contiguous 16-B loads + stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HC(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)

template <int PAT>
__global__ void __launch_bounds__(256) k(const char* __restrict__ src, char* __restrict__ dst, int ld_bytes, int tiles_n) {
    // block = 128 x 160 bf16 output tile (4 waves as 2 x 2, wave tile 64 x 80), like the GEMM
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const size_t row0 = (size_t)tm * 128 + (wave >> 1) * 64;
    const int colb0 = (tn * 160 + (wave & 1) * 80) * 2;
    if (PAT <= 1) {
        const int r = lane & 15, q = lane >> 4;
        uint2 v[20];
        if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    v[i * 5 + j] = *reinterpret_cast<const uint2*>(src + (row0 + i * 16 + r) * ld_bytes + colb0 + j * 32 + q * 8);
        } else {
#pragma unroll
            for (int x = 0; x < 20; ++x) v[x] = make_uint2(lane + x, wave);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                uint2 o = v[i * 5 + j];
                o.x += 1;
                *reinterpret_cast<uint2*>(dst + (row0 + i * 16 + r) * ld_bytes + colb0 + j * 32 + q * 8) = o;
            }
    } else {
        // wave tile 64 rows x 160 B: 10 lanes x 16 B per row -> 6 rows per instruction (60 lanes), 11 instructions
        const int rr = lane / 10, cc = lane % 10;
        uint4 v[11];
#pragma unroll
        for (int x = 0; x < 11; ++x) {
            const int row = x * 6 + rr;
            v[x] = make_uint4(lane, x, wave, 0);
            if (PAT == 3 && lane < 60 && row < 64) v[x] = *reinterpret_cast<const uint4*>(src + (row0 + row) * ld_bytes + colb0 + cc * 16);
        }
#pragma unroll
        for (int x = 0; x < 11; ++x) {
            const int row = x * 6 + rr;
            uint4 o = v[x];
            o.x += 1;
            if (lane < 60 && row < 64) *reinterpret_cast<uint4*>(dst + (row0 + row) * ld_bytes + colb0 + cc * 16) = o;
        }
    }
}

int main() {
    const int M = 32768, N = 320;
    const size_t bytes = (size_t)M * N * 2;
    char *src, *dst;
    HC(hipMalloc(&src, bytes));
    HC(hipMalloc(&dst, bytes));
    HC(hipMemset(src, 1, bytes));
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1));
    const char* names[4] = {"frag 8-B stores            ", "frag 8-B loads + stores    ", "row 16-B stores            ", "row 16-B loads + stores    "};
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            HC(hipEventRecord(e0, 0));
            for (int it = 0; it < 10; ++it) {
                dim3 grid((M / 128) * (N / 160)), block(256);
                if (pat == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, src, dst, N * 2, N / 160);
                if (pat == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, src, dst, N * 2, N / 160);
                if (pat == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, src, dst, N * 2, N / 160);
                if (pat == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, src, dst, N * 2, N / 160);
            }
            HC(hipEventRecord(e1, 0));
            HC(hipEventSynchronize(e1));
            float ms;
            HC(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double moved = (double)bytes * ((pat & 1) ? 2 : 1);
        printf("%s : %7.2f us per 32768x320 bf16 tile set, %6.2f TB/s\n", names[pat], best * 100.0, moved / (best * 1e-4) * 1e-12 * 1e-3 * 1e3 / 1e3 * 1e3);
    }
    return 0;
}
