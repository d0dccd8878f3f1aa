// How much does an output-tile store cost by access pattern?  One workgroup per CU (256 threads) writes
// 128 KiB tiles of a row-major bf16 matrix [M][N] (N = 2048) in four ways; prints us per tile per CU.
//   0: per instruction 16 rows x 32 B (8 B per lane)     -- the GEMM epilogue's store_tile_t on bf16
//   1: per instruction 16 rows x 64 B (16 B per lane)
//   2: per instruction 4 rows x 256 B (16 B per lane)
//   3: per instruction 2 rows x 512 B (16 B per lane, one full tile row per 32 lanes)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o tools/micro/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PAT>
__global__ void __launch_bounds__(256) k(unsigned short *C, int ldc, int tiles_n, int ntiles, int rounds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = 0; r < rounds; ++r) {
        const int t = (blockIdx.x + r * gridDim.x) % ntiles;
        const int tm = t / tiles_n, tn = t % tiles_n;
        unsigned short *base = C + (size_t)tm * 256 * ldc + tn * 256;
        if constexpr (PAT == 0) {          // wave w: rows w*64.., 64 rows x 256 cols = 4 x 16 tiles of 16x16
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 16; ++j) {
                    uint2 v = make_uint2(lane + i, j + r);
                    *reinterpret_cast<uint2 *>(base + (size_t)(w * 64 + i * 16 + (lane & 15)) * ldc + j * 16 + (lane >> 4) * 4) = v;
                }
        } else if constexpr (PAT == 1) {
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 8; ++j) {
                    uint4 v = make_uint4(lane + i, j + r, 0, 1);
                    *reinterpret_cast<uint4 *>(base + (size_t)(w * 64 + i * 16 + (lane & 15)) * ldc + j * 32 + (lane >> 4) * 8) = v;
                }
        } else if constexpr (PAT == 2) {   // 4 rows x 256 B per instruction
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 2; ++j) {
                    uint4 v = make_uint4(lane + i, j + r, 0, 1);
                    *reinterpret_cast<uint4 *>(base + (size_t)(w * 64 + i * 4 + (lane >> 4)) * ldc + j * 128 + (lane & 15) * 8) = v;
                }
        } else {                            // 2 rows x 512 B
            for (int i = 0; i < 32; ++i) {
                uint4 v = make_uint4(lane + i, r, 0, 1);
                *reinterpret_cast<uint4 *>(base + (size_t)(w * 64 + i * 2 + (lane >> 5)) * ldc + (lane & 31) * 8) = v;
            }
        }
    }
}
int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const int M = 32768, N = 2048, tiles_n = N / 256, ntiles = (M / 256) * tiles_n, rounds = 8;
    unsigned short *C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            switch (pat) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds); break;
                default: hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%d workgroups, pattern %d: %.2f us per 128 KiB tile per CU (%.2f TB/s aggregate)\n", nwg, pat, best * 1e3 / rounds,
               (double)nwg * rounds * 131072 / (best * 1e-3) / 1e12);
    }
    return 0;
}
