// Microbenchmark: what HBM gives RANDOM 1 KiB rows of a large store (the re-rank's access: 4 640 candidate rows of a
// 212 GB SQ8 store per query), by how a wave asks for them and by how many waves a CU holds.
//   piece 1024: one dwordx4 instruction = one whole row (64 lanes x 16 B)
//   piece  128: one instruction = a 128-byte piece of 8 rows; the 8 pieces of a row follow in later instructions
//               (rerank_sq8_kernel's pattern: lane r consumes candidate r's bytes piece by piece)
//   piece  256 / 512: 4 / 2 rows per instruction
// Plain loads into registers, U instructions in flight per wave, W waves per CU (launch bounds + LDS padding).
// build: hipcc --offload-arch=gfx950 -O3 -o random_rows tools/micro/random_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PIECE, int U>
__global__ void __launch_bounds__(64) rows_kernel(const char *__restrict__ base, const unsigned *__restrict__ idx, int tiles_per_wave,
                                                  int lds_pad, float4 *sink) {
    extern __shared__ char pad[];
    const int lane = threadIdx.x;
    constexpr int RPI = 1024 / PIECE;          // rows per instruction
    constexpr int LPR = 64 / RPI;              // lanes per row
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lds_pad < 0) pad[lane] = 1;            // keeps the allocation
    for (int t = 0; t < tiles_per_wave; ++t) {
        const unsigned *my = idx + ((size_t)blockIdx.x * tiles_per_wave + t) * 64;     // 64 rows per tile
        // instruction j of piece p covers rows [j * RPI, (j + 1) * RPI): 64 / RPI instructions per piece, 1024 / PIECE pieces
        constexpr int NI = 64 / RPI, NP = 1024 / PIECE, TOT = NI * NP;                 // always 64 instructions per tile
        float4 r[U];
#pragma unroll
        for (int i0 = 0; i0 < TOT; i0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u, p = i / NI, j = i % NI;
                const size_t row = my[j * RPI + lane / LPR];
                r[u] = *reinterpret_cast<const float4 *>(base + row * 1024 + (size_t)p * PIECE + (lane % LPR) * 16);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += r[u].x; acc.y += r[u].y; acc.z += r[u].z; acc.w += r[u].w; }
        }
    }
    if (acc.x == 12345.678f) sink[blockIdx.x * 64 + lane] = acc;
}

template <int PIECE, int U>
void run(const char *base, const unsigned *idx, int nwaves, int tiles_per_wave, int waves_per_cu, float4 *sink) {
    const int lds = 160 * 1024 / waves_per_cu - 512;
    hipFuncSetAttribute((const void *)rows_kernel<PIECE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rows_kernel<PIECE, U>), dim3(nwaves), dim3(64), lds, 0, base, idx, tiles_per_wave, lds, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((rows_kernel<PIECE, U>), dim3(nwaves), dim3(64), lds, 0, base, idx, tiles_per_wave, lds, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)nwaves * tiles_per_wave * 64 * 1024;
    printf("piece %4d B  %2d in flight  %2d waves/CU: %.3f ms  %.0f GB/s\n", PIECE, U, waves_per_cu, ms, bytes / ms / 1e6);
}

int main() {
    const size_t nb = (size_t)48 << 20;                       // 48 Mi rows x 1 KiB = 48 GiB
    char *base;
    if (hipMalloc(&base, nb * 1024) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(base, 1, nb * 1024);
    const int nwaves = 1024 * 73, tiles_per_wave = 1;        // the re-rank's grid: 1024 queries x 73 tiles of 64 candidates
    std::vector<unsigned> h((size_t)nwaves * tiles_per_wave * 64);
    unsigned long long s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (unsigned)(s % nb); }
    unsigned *idx;
    hipMalloc(&idx, h.size() * 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float4 *sink;
    hipMalloc(&sink, (size_t)nwaves * 64 * 16);
    for (int wpc : {6, 9, 16, 32}) {
        run<1024, 8>(base, idx, nwaves, tiles_per_wave, wpc, sink);
        run<1024, 16>(base, idx, nwaves, tiles_per_wave, wpc, sink);
        run<512, 16>(base, idx, nwaves, tiles_per_wave, wpc, sink);
        run<256, 16>(base, idx, nwaves, tiles_per_wave, wpc, sink);
        run<128, 8>(base, idx, nwaves, tiles_per_wave, wpc, sink);
        run<128, 16>(base, idx, nwaves, tiles_per_wave, wpc, sink);
    }
    return 0;
}
