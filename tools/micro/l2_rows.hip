// Microbenchmark: the small-batch score GEMM's load pattern without the arithmetic.
// 256 workgroups x 256 threads; per K chunk (256 B per row) a workgroup reads 32 "A" rows
// (shared by all workgroups of the same parity) and 32 "B" rows (its own), rows `stride`
// bytes apart, PF chunks in flight.  Prints the time per chunk and GB/s per CU for the plain
// 4 KiB row stride and for padded strides.
// build: hipcc --offload-arch=gfx950 -O3 -o l2_rows tools/micro/l2_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PF, bool TILED>
__global__ void rows_kernel(const char *__restrict__ A, const char *__restrict__ B, size_t stride, int nchunk,
                            float4 *sink) {
    const int tid = threadIdx.x;
    const int row = tid >> 4, seg = tid & 15;   // 16 rows x 256 B per wave-pass; 256 threads: 16 rows per pass
    const char *a0 = A + (size_t)((blockIdx.x & 1) * 32 + row) * stride + seg * 16;
    // TILED: the workgroup's B tile is one contiguous [chunk][32 rows][256 B] block
    const char *b0 = TILED ? B + (size_t)blockIdx.x * 32 * stride + tid * 16
                           : B + (size_t)(blockIdx.x * 32 + row) * stride + seg * 16;
    const size_t bstep = TILED ? 8192 : 256, bhalf = TILED ? 4096 : 16 * stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r[PF][4];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        r[p][0] = *reinterpret_cast<const float4 *>(a0 + p * 256);
        r[p][1] = *reinterpret_cast<const float4 *>(a0 + 16 * stride + p * 256);
        r[p][2] = *reinterpret_cast<const float4 *>(b0 + p * bstep);
        r[p][3] = *reinterpret_cast<const float4 *>(b0 + bhalf + p * bstep);
    }
    for (int c0 = 0; c0 + 2 * PF <= nchunk; c0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc.x += r[p][j].x; acc.y += r[p][j].y; acc.z += r[p][j].z; acc.w += r[p][j].w; }
            const int c = c0 + p + PF;
            r[p][0] = *reinterpret_cast<const float4 *>(a0 + c * 256);
            r[p][1] = *reinterpret_cast<const float4 *>(a0 + 16 * stride + c * 256);
            r[p][2] = *reinterpret_cast<const float4 *>(b0 + c * bstep);
            r[p][3] = *reinterpret_cast<const float4 *>(b0 + bhalf + c * bstep);
        }
    }
    if (acc.x == 12345.678f) sink[blockIdx.x * 256 + tid] = acc;
}

template <bool TILED>
void sweep() {
    const int nblocks = 256, nchunk = 16;
    printf(TILED ? "-- B tile contiguous per workgroup\n" : "-- B rows strided\n");   // 16 chunks of 256 B = one 4 KiB row, as d = 1024 floats
    for (size_t stride : {(size_t)4096, (size_t)4096 + 256}) {
        char *A, *B;
        float4 *sink;
        hipMalloc(&A, 64 * stride + 4096);
        hipMalloc(&B, (size_t)nblocks * 32 * stride + 4096);
        hipMalloc(&sink, 16 * 256 * nblocks);
        hipMemset(A, 0, 64 * stride);
        hipMemset(B, 0, (size_t)nblocks * 32 * stride);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 50;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rows_kernel<4, TILED>), dim3(nblocks), dim3(256), 0, 0, A, B, stride, nchunk, sink);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rows_kernel<4, TILED>), dim3(nblocks), dim3(256), 0, 0, A, B, stride, nchunk, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("row stride %5zu B: %6.2f us per launch (16 chunks), %5.1f GB/s per CU (incl. launch)\n", stride, us,
               16.0 * 16384 / (us * 1e-6) / 1e9);
        // long rows: 64 chunks, to separate the per-chunk cost from the launch
        hipFree(A); hipFree(B);
        const size_t stride2 = stride * 4;
        hipMalloc(&A, 64 * stride2 + 4096);
        hipMalloc(&B, (size_t)nblocks * 32 * stride2 + 4096);
        hipMemset(A, 0, 64 * stride2);
        hipMemset(B, 0, (size_t)nblocks * 32 * stride2);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rows_kernel<4, TILED>), dim3(nblocks), dim3(256), 0, 0, A, B, stride2, 64, sink);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rows_kernel<4, TILED>), dim3(nblocks), dim3(256), 0, 0, A, B, stride2, 64, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double us2 = ms * 1e3 / reps;
        printf("row stride %5zu B: %6.2f us per launch (64 chunks) -> %5.3f us per chunk, %5.1f GB/s per CU marginal\n", stride2, us2,
               (us2 - us) / 48, 16384 / ((us2 - us) / 48 * 1e-6) / 1e9);
        hipFree(A); hipFree(B); hipFree(sink);
    }
}

int main() {
    sweep<false>();
    sweep<true>();
    return 0;
}
