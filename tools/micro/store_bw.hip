// store_bw.hip -- HBM write bandwidth by store pattern (is the GEMM epilogue's 2.4 TB/s the
// memory system's limit or the pattern's?).  M x N bf16 output, 256x256 tiles, 8 waves per tile.
//   pattern 0: every wave instruction writes 1 KiB contiguous (16 B per lane)
//   pattern 1: the ring GEMM's epilogue: a 16x16 tile per instruction, 8 B per lane = 16 rows x 32 B
//   pattern 2: whole 128-byte rows: 8 lanes x 16 B per row, 8 rows per instruction
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/store_bw tools/micro/store_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT>
__global__ void __launch_bounds__(512) store_kernel(unsigned short *C, int M, int N, int tiles_m, int tiles_n) {
    const int t = blockIdx.x;
    if (t >= tiles_m * tiles_n) return;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w & 1, wn = w >> 1;                         // 2 x 4 waves, 128 x 64 per wave
    const int r0 = tm * 256 + wm * 128, c0 = tn * 256 + wn * 64;
    const uint4 v4 = make_uint4(lane, w, t, 7);
    const uint2 v2 = make_uint2(lane, t);
    if (PAT == 0) {            // 16 KiB per wave as 16 contiguous 1-KiB pieces (not a matrix layout: bandwidth only)
        char *base = reinterpret_cast<char *>(C) + ((size_t)t * 8 + w) * 16384;
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4 *>(base + i * 1024 + lane * 16) = v4;
    } else if (PAT == 1) {     // 8 x 4 tiles of 16 x 16, lane -> (row lane >> 2, 4 columns (lane & 3) * 4)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + i * 16 + (lane >> 2), c = c0 + j * 16 + (lane & 3) * 4;
                if (r < M) *reinterpret_cast<uint2 *>(C + (size_t)r * N + c) = v2;
            }
    } else {                   // 16 instructions of 8 rows x 128 B
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + i * 8 + (lane >> 3), c = c0 + (lane & 7) * 8;
            if (r < M) *reinterpret_cast<uint4 *>(C + (size_t)r * N + c) = v4;
        }
    }
}

int main() {
    const int M = 29696, N = 17920;
    const int tiles_m = M / 256, tiles_n = N / 256;
    unsigned short *C;
    CK(hipMalloc(&C, (size_t)M * N * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat = 0; pat < 3; ++pat) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) {
                if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, C, M, N, tiles_m, tiles_n);
                else if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, C, M, N, tiles_m, tiles_n);
                else hipLaunchKernelGGL(store_kernel<2>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, C, M, N, tiles_m, tiles_n);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("pattern %d: %.1f us per %.2f GB  = %.2f TB/s\n", pat, ms * 100, (double)M * N * 2 / 1e9, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e12);
        }
    }
    return 0;
}
