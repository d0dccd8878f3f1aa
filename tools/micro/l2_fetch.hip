// Microbenchmark: how fast can one CU pull a private, L2-resident region?
// Each workgroup (one per CU) streams its own `bytes_per_block` region `reps` times with
// W waves, each keeping D 16-byte-per-lane loads in flight.  Prints GB/s per CU and in total.
// build: hipcc --offload-arch=gfx950 -O3 -o l2_fetch tools/micro/l2_fetch.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int D>
__global__ void fetch_kernel(const float4 *__restrict__ src, size_t vec_per_block, int reps, float4 *sink) {
    const float4 *base = src + (size_t)blockIdx.x * vec_per_block;
    const int nthr = blockDim.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + (size_t)(D - 1) * nthr < vec_per_block; i += (size_t)D * nthr) {
            float4 v[D];
#pragma unroll
            for (int u = 0; u < D; ++u) v[u] = base[i + (size_t)u * nthr];
#pragma unroll
            for (int u = 0; u < D; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (acc.x == 12345.678f) sink[blockIdx.x * nthr + threadIdx.x] = acc;
}

template <int D>
float run(const float4 *src, size_t vpb, int nblocks, int threads, int reps, float4 *sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fetch_kernel<D>, dim3(nblocks), dim3(threads), 0, 0, src, vpb, 2, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(fetch_kernel<D>, dim3(nblocks), dim3(threads), 0, 0, src, vpb, reps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv) {
    const int nblocks = 256, reps = 64;
    for (size_t kb : {64, 128}) {
        const size_t vpb = kb * 1024 / 16;
        float4 *src, *sink;
        hipMalloc(&src, vpb * 16 * nblocks);
        hipMalloc(&sink, 16 * 1024 * nblocks);
        hipMemset(src, 0, vpb * 16 * nblocks);
        for (int waves : {2, 4, 8, 16}) {
            const int threads = waves * 64;
            float ms[4];
            ms[0] = run<1>(src, vpb, nblocks, threads, reps, sink);
            ms[1] = run<2>(src, vpb, nblocks, threads, reps, sink);
            ms[2] = run<4>(src, vpb, nblocks, threads, reps, sink);
            ms[3] = run<8>(src, vpb, nblocks, threads, reps, sink);
            const int Ds[4] = {1, 2, 4, 8};
            for (int j = 0; j < 4; ++j) {
                const double bytes = (double)vpb * 16 * reps;   // per block
                printf("region %3zu KiB/CU  waves %2d  in-flight %d x16B/lane: %7.1f GB/s per CU  %6.2f TB/s total\n", kb, waves,
                       Ds[j], bytes / (ms[j] * 1e-3) / 1e9, bytes * nblocks / (ms[j] * 1e-3) / 1e12);
            }
        }
        hipFree(src); hipFree(sink);
    }
    return 0;
}
