// Probe kernel for tools/micro/small_wg_overlap.py: one-wave workgroups that do nothing but `iters` dependent MFMAs (MFMA-bound, no
// memory, no LDS), with NV accumulator tiles kept live so that the wave's VGPR footprint is a parameter (NV = 4: ~40 VGPRs,
// NV = 24: ~128).  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libspin.so tools/micro/spin.hip
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ void __launch_bounds__(64) spin_kernel(int iters, float *sink) {
    f32x4 acc[NV];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[v], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) s += acc[v][0] + acc[v][1] + acc[v][2] + acc[v][3];
    if (s == 12345.678f) sink[blockIdx.x] = s;
}

extern "C" int spin_launch(int nwg, int iters, int nv, float *sink, void *stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nv <= 4) hipLaunchKernelGGL((spin_kernel<4>), dim3(nwg), dim3(64), 0, st, iters, sink);
    else hipLaunchKernelGGL((spin_kernel<24>), dim3(nwg), dim3(64), 0, st, iters / 6, sink);
    return (int)hipGetLastError();
}
