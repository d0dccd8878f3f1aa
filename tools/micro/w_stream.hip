// Microbenchmark: the W operand stream of a one-round GEMM of a few hundred tokens -- every workgroup pulls ITS OWN strip of
// 256 rows x K bytes (nothing shared: every line misses the XCD's L2) through an LDS-DMA ring with D slabs in flight, in the
// slab kernel's piece shape (8 rows x 128 B per instruction) -- by how the strip lies in memory:
//   layout 0  row-major, row pitch = K bytes (3072: the weights as the GEMM kernels read them): a slab = 256 lines 3 KB apart
//   layout 1  slab-major: slab s of a strip = 256 x 128 B = 32 KiB contiguous
// cold: every repetition reads a fresh region of a 4 GiB arena (HBM); warm: the same 200 MB again (Infinity Cache).
// Prints GB/s per CU.   build: hipcc --offload-arch=gfx950 -O3 -o w_stream tools/micro/w_stream.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void dma16_off(const void *gptr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n" ::"s"(lds_off), "v"(gptr) : "memory");
}

template <int LAYOUT, int NS>
__global__ void __launch_bounds__(256) stream_kernel(const char *__restrict__ base, size_t strip_bytes, int kbytes, int nslab, int *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PPW = 8, D = NS - 1;                     // 32 pieces of a 256-row slab over 4 waves
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *strip = base + (size_t)blockIdx.x * strip_bytes;
    const int prow = lane >> 3, scol = ((lane & 7) ^ prow) * 16;
    const char *src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (w * PPW + i) * 8 + prow;
        src[i] = LAYOUT == 0 ? strip + (size_t)row * kbytes + scol : strip + (size_t)row * 128 + scol;
    }
    const size_t sstep = LAYOUT == 0 ? 128 : (size_t)256 * 128;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma16_off(src[i] + (size_t)s * sstep, lds0 + s * 32768 + (w * PPW + i) * 1024);
    unsigned slot = 0;
    for (int s = 0; s < nslab; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PPW) : "memory");
        asm volatile("s_barrier" ::: "memory");
        const unsigned ns = slot == 0 ? NS - 1 : slot - 1;
        const int sn = min(s + D, nslab - 1);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma16_off(src[i] + (size_t)sn * sstep, lds0 + ns * 32768 + (w * PPW + i) * 1024);
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    const int v = *reinterpret_cast<int *>(lds + lane * 16);
    if (v == 0x12345678) sink[blockIdx.x] = v;
}

template <int LAYOUT, int NS>
void run(const char *arena, size_t arena_bytes, int nwg, bool cold, int *sink) {
    const int kbytes = 3072, nslab = kbytes / 128;
    const size_t strip = (size_t)256 * kbytes, per_launch = strip * nwg;
    const size_t smem = (size_t)NS * 32768;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_kernel<LAYOUT, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int reps = 12;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float tot = 0;
    for (int r = -2; r < reps; ++r) {
        const size_t off = cold ? ((size_t)(r + 2) * per_launch) % (arena_bytes - per_launch) : 0;
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<LAYOUT, NS>), dim3(nwg), dim3(256), smem, 0, arena + off, strip, kbytes, nslab, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 0) tot += ms;
    }
    const double us = tot / reps * 1e3, gbs = (double)strip / (us * 1e-6) / 1e9;
    printf("layout %s ring %d (%3d KiB in flight) WGs %3d %s: %6.2f us per launch  %6.1f GB/s per CU  %5.2f TB/s chip  %s\n", LAYOUT ? "slab-major" : "row-major ", NS, (NS - 1) * 32,
           nwg, cold ? "cold" : "warm", us, gbs, gbs * nwg / 1e3, hipGetErrorString(hipGetLastError()));
}

int main() {
    const size_t arena_bytes = (size_t)4 << 30;
    char *arena; int *sink;
    hipMalloc(&arena, arena_bytes);
    hipMalloc(&sink, 4096 * 4);
    hipMemset(arena, 1, arena_bytes);
    for (int nwg : {210, 70}) for (bool cold : {true, false}) {
        run<0, 3>(arena, arena_bytes, nwg, cold, sink);
        run<1, 3>(arena, arena_bytes, nwg, cold, sink);
        run<0, 5>(arena, arena_bytes, nwg, cold, sink);
        run<1, 5>(arena, arena_bytes, nwg, cold, sink);
    }
    return 0;
}
