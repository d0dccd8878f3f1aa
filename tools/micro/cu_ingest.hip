// Microbenchmark: what ONE CU can ingest from L2 in the access pattern of a GEMM operand stream -- pieces of 8 rows x 128 B
// (rows `ld` bytes apart, K slabs of 128 B walked left to right), every workgroup of an XCD reading the SAME tile (as the
// workgroups of a GEMM share operands in their XCD's L2) -- by how the bytes are moved:
//   mode 0  global_load_dwordx4 -> registers (nothing else)
//   mode 1  global_load_lds_dwordx4 (LDS-DMA) into a ring of slabs, counted vmcnt, one s_barrier per slab
//   mode 2  global_load_dwordx4 -> registers -> ds_write_b128 into the ring, one barrier per slab
// with 4 or 8 waves per workgroup and 1 or 2 workgroups per CU.  Prints GB/s and B/clk (at 2.1 GHz) per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o cu_ingest tools/micro/cu_ingest.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16_off(const void *gptr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n" ::"s"(lds_off), "v"(gptr) : "memory");
}

// tile = ROWS rows x K bytes (row pitch ld); slab s = columns [128 s, 128 s + 128): ROWS / 8 pieces, piece p -> wave p % NW
template <int MODE, int ROWS, int NW, int NS>
__global__ void __launch_bounds__(64 * NW) ingest_kernel(const char *__restrict__ base, size_t tile_bytes, int ld, int nslab, int reps, int *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PPW = ROWS / 8 / NW;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *tile = base + (size_t)(blockIdx.x & 7) * tile_bytes;          // XCD b % 8: one tile per XCD
    const int prow = lane >> 3, scol = ((lane & 7) ^ prow) * 16;
    const char *src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) src[i] = tile + (size_t)((w * PPW + i) * 8 + prow) * ld + scol;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
    constexpr unsigned STAGE = ROWS * 128;
    i32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        if constexpr (MODE == 0) {
            for (int s = 0; s < nslab; s += NS) {
                i32x4 v[NS][PPW];
#pragma unroll
                for (int u = 0; u < NS; ++u)
#pragma unroll
                    for (int i = 0; i < PPW; ++i) v[u][i] = *reinterpret_cast<const i32x4 *>(src[i] + (size_t)min(s + u, nslab - 1) * 128);
#pragma unroll
                for (int u = 0; u < NS; ++u)
#pragma unroll
                    for (int i = 0; i < PPW; ++i) acc ^= v[u][i];
            }
        } else if constexpr (MODE == 1) {
            constexpr int D = NS - 1;
#pragma unroll
            for (int s = 0; s < D; ++s)
#pragma unroll
                for (int i = 0; i < PPW; ++i) dma16_off(src[i] + (size_t)s * 128, lds0 + s * STAGE + (w * PPW + i) * 1024);
            unsigned slot = 0;
            for (int s = 0; s < nslab; ++s) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PPW) : "memory");
                asm volatile("s_barrier" ::: "memory");
                const unsigned ns = slot == 0 ? NS - 1 : slot - 1;
                const int sn = min(s + D, nslab - 1);
#pragma unroll
                for (int i = 0; i < PPW; ++i) dma16_off(src[i] + (size_t)sn * 128, lds0 + ns * STAGE + (w * PPW + i) * 1024);
                slot = slot + 1 == NS ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        } else {
            constexpr int D = NS - 1;                       // slabs in registers in flight
            i32x4 v[D][PPW];
#pragma unroll
            for (int u = 0; u < D; ++u)
#pragma unroll
                for (int i = 0; i < PPW; ++i) v[u][i] = *reinterpret_cast<const i32x4 *>(src[i] + (size_t)min(u, nslab - 1) * 128);
            for (int s0 = 0; s0 < nslab; s0 += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    i32x4 *dst = reinterpret_cast<i32x4 *>(lds + ((s0 + u) % NS) * STAGE);
#pragma unroll
                    for (int i = 0; i < PPW; ++i) dst[(w * PPW + i) * 64 + lane] = v[u][i];
#pragma unroll
                    for (int i = 0; i < PPW; ++i) v[u][i] = *reinterpret_cast<const i32x4 *>(src[i] + (size_t)min(s0 + u + D, nslab - 1) * 128);
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
    }
    if (MODE != 0) acc = *reinterpret_cast<i32x4 *>(lds + lane * 16);
    if (acc[0] == 0x12345678) sink[blockIdx.x] = acc[1];
}

template <int MODE, int ROWS, int NW, int NS>
void run(const char *src, size_t tile_bytes, int ld, int nslab, int nblocks, int *sink, const char *name) {
    const int reps = 40;
    const size_t smem = MODE == 0 ? 16 : (size_t)NS * ROWS * 128;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ingest_kernel<MODE, ROWS, NW, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ingest_kernel<MODE, ROWS, NW, NS>), dim3(nblocks), dim3(64 * NW), smem, 0, src, tile_bytes, ld, nslab, 2, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((ingest_kernel<MODE, ROWS, NW, NS>), dim3(nblocks), dim3(64 * NW), smem, 0, src, tile_bytes, ld, nslab, reps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ROWS * 128 * nslab * reps;                   // per workgroup
    const double per_cu = bytes * (nblocks / 256.0) / (ms * 1e-3);
    printf("%-34s rows %3d waves %d ring %d  WGs %3d: %6.1f GB/s per CU = %5.1f B/clk @2.1GHz  (%5.2f TB/s chip)  %s\n", name, ROWS, NW, NS, nblocks, per_cu / 1e9,
           per_cu / 2.1e9, per_cu * 256 / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int ld = 3072, nslab = 24;                                          // K = 1536 bf16
    const size_t tile_bytes = (size_t)256 * ld;
    char *src; int *sink;
    hipMalloc(&src, tile_bytes * 8);
    hipMalloc(&sink, 4096 * 4);
    hipMemset(src, 1, tile_bytes * 8);
    for (int nb : {256, 512}) {
        run<0, 128, 4, 4>(src, tile_bytes, ld, nslab, nb, sink, "loads -> registers");
        run<0, 256, 8, 4>(src, tile_bytes, ld, nslab, nb, sink, "loads -> registers");
        run<1, 128, 4, 4>(src, tile_bytes, ld, nslab, nb, sink, "LDS-DMA ring");
        run<1, 256, 4, 3>(src, tile_bytes, ld, nslab, nb, sink, "LDS-DMA ring");
        run<1, 256, 8, 3>(src, tile_bytes, ld, nslab, nb, sink, "LDS-DMA ring");
        run<1, 128, 4, 8>(src, tile_bytes, ld, nslab, nb, sink, "LDS-DMA ring");
        run<2, 128, 4, 4>(src, tile_bytes, ld, nslab, nb, sink, "loads -> registers -> ds_write");
        run<2, 256, 8, 3>(src, tile_bytes, ld, nslab, nb, sink, "loads -> registers -> ds_write");
        run<2, 256, 4, 3>(src, tile_bytes, ld, nslab, nb, sink, "loads -> registers -> ds_write");
    }
    return 0;
}
