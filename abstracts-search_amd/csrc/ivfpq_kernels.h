// ivfpq_kernels.h -- gfx950 (CDNA4, wave64) kernels of the IVF-PQ search path.
//
// Stages of IndexIVFPQ.search (reference call sites: Makefile:32 `index tune`,
// README.md:28 query-time app; arithmetic restated in oracle/ivfpq_oracle.c):
//
//   ip_gemm_kernel   S = Q . C^T, exact f32 on v_mfma_f32_16x16x4_f32
//                    (bitwise an ascending-k fmaf chain = the oracle's dot)
//   select_kernel    best-K of each row of S under (score desc, index asc), K <= 64
//   select_big_kernel  the same for 64 < K <= 4096 (threshold + bitonic sort)
//   to_f16_rows_kernel / [ring GEMM, f16] / select_refine_kernel
//                    large batches: the same coarse result from f16 MFMA scores +
//                    exact re-scoring of the centroids within a proven error margin
//   lut_kernel       LUT[q][m][j] = <q_m, codebook[m][j]>
//   scan_kernel      stream PQ codes of the probed lists, 64 LDS table
//                    look-ups per code, per-wave register top-k       (HBM/LDS bound);
//                    k > 64: ALL mode stores every (score, id), select_pairs_kernel
//                    keeps the k best
//   merge_kernel     k-way merge of per-slice / per-shard partial top-k and of the
//                    re-ranked candidates of IndexRefineFlat
//   pq_encode_kernel Index.add: residual + nearest codeword per sub-vector
//
// Inverted-list layout in HBM ("group-interleaved"): a list is padded to
// groups of 64 codes; group g is NCH = ceil(M/16) chunks of 1 KiB, chunk c
// holding bytes [16c, 16c+16) of the 64 codes, 16 B per code, so that lane j
// of a wave reads code j with NCH perfectly coalesced global_load_dwordx4
// (64 lanes x 16 B = 1 KiB per instruction).  ids[group*64 + lane] is int64.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace mi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr unsigned EMPTY_POS = 0xFFFFFFFFu;
constexpr int64_t EMPTY_ID = INT64_MAX;
#define MI_NEG_INF (-__builtin_huge_valf())

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ unsigned readlane_u(unsigned v, int l) {
    return (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
// LDS-DMA issued from inline asm: lane i's 16 (4) bytes at gptr land at
// lds_wave_base + 16 i (4 i).  hipcc treats the builtin form as an LDS write and puts
// s_waitcnt vmcnt(0) in front of every later LDS read; with the asm form the waits
// are ours (counted s_waitcnt vmcnt(N): loads and DMAs retire in issue order).
__device__ __forceinline__ void dma16_lds(const void *gptr, void *lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n" ::"s"(m0v), "v"(gptr) : "memory");
}
__device__ __forceinline__ void dma4_lds(const void *gptr, void *lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n" ::"s"(m0v), "v"(gptr) : "memory");
}
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v & 0xffffffffu), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// lane i <- lane i-1 (lane 0 keeps its own value): one v_mov_b32_dpp wave_shr:1
// instead of a ds_bpermute round trip through the LDS crossbar.
__device__ __forceinline__ int wave_shr1_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ float wave_shr1_f(float v) {
    return __int_as_float(wave_shr1_i(__float_as_int(v)));
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ int wave_reduce_add_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int lane_prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// order-preserving float <-> unsigned map (for LDS atomicMax on scores)
__device__ __forceinline__ unsigned f2o(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// ---------------------------------------------------------------------
// S[na][nb] = A[na][d] . B[nb][d]^T, exact f32.
// Workgroup tile BM x BN, K chunk BK, double-buffered through LDS (row-major,
// padded).  Lane (i = lane&15, g = lane>>4) feeds A[i][k0+g] to the MFMA, which
// then consumes k in ascending order: every output is an ascending-k fmaf
// chain from +0.  Small batches (one 16x16 tile per SIMD) are bound by that chain:
// d/4 dependent MFMAs at 40 cycles each, which the evaluation order forbids splitting;
// the 32x32 workgroup tile for 16 < na <= 128 gives every CU a tile with the fewest
// bytes to fetch, and the K loop is organised around never stalling the chain (below).
// Grid: 8 * tiles_m * ceil(tiles_n/8); block b runs on XCD b%8, and all
// blocks of one XCD walk the M tiles of the same B strip (L2 reuse of B).
// ---------------------------------------------------------------------
// ---------------------------------------------------------------------
// LUT[q][m][j] = <q_m, codebook[m][j]>  (ascending-t fmaf chain from +0).
// Work unit = (sub-quantiser m, tile of `qtile` queries); 256 threads: thread j
// keeps codeword j of sub-quantiser m in registers and walks the queries.
// Runs either as its own kernel or as extra workgroups appended to the coarse
// GEMM launch (the GEMM of a small batch is latency-bound with one wave per
// SIMD; the LUT workgroups fill the idle issue slots and a launch is saved).
// ---------------------------------------------------------------------
struct LutArgs {
    const float *q;         // [nq][d]
    const float *codebook;  // [M][256][dsub]
    float *lut;             // [nq][M][256]; null = no LUT work
    int nq, d, M, dsub, qtile, nblocks;
};

template <int DSUB>
__device__ __forceinline__ void lut_block(const LutArgs &a, int blk) {
    // 256 codewords per (m, query tile): one 256-thread workgroup, or four
    // 64-thread ones when appended to the wave-per-tile GEMM launch
    const int parts = 256 / (int)blockDim.x;
    const int j = (blk % parts) * (int)blockDim.x + threadIdx.x;
    blk /= parts;
    const int m = blk % a.M;
    const int q0 = (blk / a.M) * a.qtile;
    const int q1 = min(a.nq, q0 + a.qtile);
    float cb[DSUB];
    const float *cp = a.codebook + ((size_t)m * 256 + j) * DSUB;
#pragma unroll
    for (int t = 0; t < DSUB; ++t) cb[t] = cp[t];
    // the tile's query sub-vectors go through LDS in one cooperative load: fetched one query
    // at a time inside the loop, every query cost a dependent global-memory round trip
    // (1024 queries: 46 us -> 25 us; the bench step, where these workgroups ride in the GEMM
    // launch: 3.80 -> 3.88 M QPS)
    constexpr int QT = 16;
    __shared__ float qsm[QT * DSUB];
    const int nqt = q1 - q0;   // <= a.qtile <= QT (host)
    for (int i = threadIdx.x; i < nqt * DSUB; i += blockDim.x)
        qsm[i] = a.q[(size_t)(q0 + i / DSUB) * a.d + m * DSUB + i % DSUB];
    __syncthreads();
    for (int qi = 0; qi < nqt; ++qi) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < DSUB; ++t) acc = __builtin_fmaf(qsm[qi * DSUB + t], cb[t], acc);
        a.lut[((size_t)(q0 + qi) * a.M + m) * 256 + j] = acc;
    }
}

// Gather mode (re-ranking): M tile tm is ONE query (row tm of A) and its B rows are the
// rows idx[tm][0..kc) of B (negative = empty slot: any row, the caller ignores the score);
// S[tm][c] = <A[tm], B[idx[tm][c]]> with exactly the arithmetic of the plain mode.
struct GatherArgs {
    const int64_t *idx;   // [tiles_m][kc] or null = plain GEMM
    int kc;
};

// TB = f16_t: B is stored as IEEE half (IndexScalarQuantizer QT_fp16, the refine store of
// "Refine(SQfp16)"): the loads move half the bytes, every element is widened to f32 exactly
// on its way into LDS and the arithmetic is the same f32 chain -- score = chain(q_k * (float)x16_k).
template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, int PF, typename TB = float>
__global__ void __launch_bounds__(WAVES_M *WAVES_N * 64)
    ip_gemm_kernel(const float *__restrict__ A, int na, const TB *__restrict__ B, int nb,
                   int d, float *__restrict__ S, int64_t ldS, int tiles_m, int tiles_n, int gemm_blocks,
                   LutArgs la, GatherArgs ga) {
    if ((int)blockIdx.x >= gemm_blocks) {  // appended LUT workgroups (dsub <= 16 only)
        const int blk = blockIdx.x - gemm_blocks;
        if (la.dsub == 16) lut_block<16>(la, blk);
        else if (la.dsub == 8) lut_block<8>(la, blk);
        else if (la.dsub == 4) lut_block<4>(la, blk);
        return;
    }
    constexpr int BM = 16 * WM * WAVES_M;
    constexpr int BN = 16 * WN * WAVES_N;
    constexpr int KQ = BK / 4;  // float4 per tile row
    constexpr int NT = WAVES_M * WAVES_N * 64;
    // LDS tiles are row-major [rows][ST], ST = BK + 4 floats (ST/4 odd).  The staging
    // store is one ds_write_b128 per loaded float4 (consecutive lanes = consecutive k-quads
    // of one row: whole 128 B lines on the global side, consecutive banks on the LDS side),
    // the operand read is a ds_read_b32 with immediate
    // offsets (16 rows x 4 consecutive k -> banks 4*(odd*row) + lg: all 64).
    constexpr int ST = BK + 4;
    static_assert((ST / 4) % 2 == 1, "row stride must be an odd number of float4");
    constexpr int CA = (BM * KQ + NT - 1) / NT;
    constexpr int CB = (BN * KQ + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * ST];
    float *As = smem;                 // [2][BM][ST]
    float *Bs = smem + 2 * BM * ST;   // [2][BN][ST]

    const int bid = blockIdx.x;
    const int xcd = bid & 7, jb = bid >> 3;
    const int tm = jb % tiles_m;
    const int tn = (jb / tiles_m) * 8 + xcd;
    if (tn >= tiles_n) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wmi = w % WAVES_M, wni = w / WAVES_M;
    const int li = lane & 15, lg = lane >> 4;

    // PF register sets: the global loads of K chunk t+PF are issued while chunk t
    // is multiplied, so PF chunks of memory latency are covered (small batches
    // run one wave per SIMD: nothing else hides it).
    float4 ra[PF][CA], rb[PF][CB];
    f32x4 acc[WM][WN];
#pragma unroll
    for (int x = 0; x < WM; ++x)
#pragma unroll
        for (int y = 0; y < WN; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // row pointers of this thread's staging loads (rows are clamped, never guarded; a thread
    // without a tile row re-reads a clamped row)
    const float *pa_row[CA];
    const TB *pb_row[CB];
#pragma unroll
    for (int u = 0; u < CA; ++u) {
        const int idx = tid + u * NT;
        const int gr = ga.idx ? tm : min(m0 + idx / KQ, na - 1);
        pa_row[u] = A + (size_t)gr * d + (idx % KQ) * 4;
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) {
        const int idx = tid + u * NT;
        int64_t gr = min(n0 + idx / KQ, nb - 1);
        if (ga.idx)   // clamped both ways: an id that is not a position in B must not fault
            gr = max(min(ga.idx[(size_t)tm * ga.kc + min(n0 + idx / KQ, ga.kc - 1)], (int64_t)nb - 1), (int64_t)0);
        pb_row[u] = B + (size_t)gr * d + (idx % KQ) * 4;
    }
    // `full` (a literal at every call site) marks a K chunk that lies inside d: its
    // loads carry no guard at all.  A guarded load is an exec-masked branch, after which
    // hipcc falls back to s_waitcnt vmcnt(0) and the PF-deep prefetch is gone.
    auto gload = [&](int k0, float4(&pa)[CA], float4(&pb)[CB], bool full) {
#pragma unroll
        for (int u = 0; u < CA; ++u) {
            const int k = k0 + ((tid + u * NT) % KQ) * 4;
            if (full) pa[u] = *reinterpret_cast<const float4 *>(pa_row[u] + k0);
            else pa[u] = (k < d) ? *reinterpret_cast<const float4 *>(pa_row[u] + k0)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int k = k0 + ((tid + u * NT) % KQ) * 4;
            if constexpr (sizeof(TB) == 4) {
                if (full) pb[u] = *reinterpret_cast<const float4 *>(pb_row[u] + k0);
                else pb[u] = (k < d) ? *reinterpret_cast<const float4 *>(pb_row[u] + k0)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 hv = {(f16_t)0.f, (f16_t)0.f, (f16_t)0.f, (f16_t)0.f};
                if (full || k < d) hv = *reinterpret_cast<const h4 *>(pb_row[u] + k0);
                pb[u] = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
            }
        }
    };
    auto sstore = [&](int buf, const float4(&pa)[CA], const float4(&pb)[CB]) {
#pragma unroll
        for (int u = 0; u < CA; ++u) {
            int idx = tid + u * NT;
            if ((BM * KQ) % NT == 0 || idx < BM * KQ) {
                int row = idx / KQ;
                int kq = idx % KQ;
                *reinterpret_cast<float4 *>(As + (buf * BM + row) * ST + kq * 4) = pa[u];
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            int idx = tid + u * NT;
            if ((BN * KQ) % NT == 0 || idx < BN * KQ) {
                int row = idx / KQ;
                int kq = idx % KQ;
                *reinterpret_cast<float4 *>(Bs + (buf * BN + row) * ST + kq * 4) = pb[u];
            }
        }
    };
    auto compute = [&](int buf) {
        const float *ab = As + (buf * BM + wmi * WM * 16 + li) * ST + lg;
        const float *bb = Bs + (buf * BN + wni * WN * 16 + li) * ST + lg;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float a[WM], b[WN];
#pragma unroll
            for (int x = 0; x < WM; ++x) a[x] = ab[x * 16 * ST + kk * 4];
#pragma unroll
            for (int y = 0; y < WN; ++y) b[y] = bb[y * 16 * ST + kk * 4];
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
    };

    // LDS hand-off between the waves of the workgroup: own LDS writes retired, then the
    // barrier.  (__syncthreads() would also drain vmcnt, i.e. the prefetched chunks.)
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int nk = (d + BK - 1) / BK;
    if (PF * BK <= d) {
#pragma unroll
        for (int s = 0; s < PF; ++s) gload(s * BK, ra[s], rb[s], true);
    } else {
#pragma unroll
        for (int s = 0; s < PF; ++s)
            if (s < nk) gload(s * BK, ra[s], rb[s], false);
    }
    // One 16x16 tile per wave (small batches): every output is ONE dependent MFMA chain
    // (40 cycles per link, 16 links per K chunk) and a wave has nothing else to run, so
    // the whole chunk hand-off has to hide inside the chain.  Step t therefore starts
    // with the barrier -- every LDS operation it waits for was issued a full step ago --
    // then reads the operands of chunk t+1 (stored during step t-1) into the second
    // register set, stores chunk t+2 into the buffer chunk t was read from, requests
    // chunk t+PF, and only then issues chunk t's 16 MFMAs from registers: nothing inside
    // or behind the chain waits for LDS or for the other waves.
    constexpr bool PIPE = (WM * WN == 1) && (PF % 2 == 0) && PF >= 4;
    if constexpr (PIPE) {
        constexpr int NKK = BK / 4;
        float opa[2][NKK], opb[2][NKK];
        auto read_ops = [&](int buf, float(&pa)[NKK], float(&pb)[NKK]) {
            const float *ab = As + (buf * BM + wmi * 16 + li) * ST + lg;
            const float *bb = Bs + (buf * BN + wni * 16 + li) * ST + lg;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                pa[kk] = ab[kk * 4];
                pb[kk] = bb[kk * 4];
            }
        };
        auto mma_chunk = [&](const float(&pa)[NKK], const float(&pb)[NKK]) {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk], pb[kk], acc[0][0], 0, 0, 0);
        };
        sstore(0, ra[0], rb[0]);
        lds_barrier();
        read_ops(0, opa[0], opb[0]);
        if (1 < nk) sstore(1, ra[1], rb[1]);
        int t0 = 0;
        for (; (t0 + 2 * PF) * BK <= d; t0 += PF) {   // steady state: branch-free, waits stay counted
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;   // t0 is a multiple of PF (even): buffer / register-set parities are static
                lds_barrier();
                read_ops((u + 1) & 1, opa[(u + 1) & 1], opb[(u + 1) & 1]);          // chunk t+1
                sstore(u & 1, ra[(u + 2) % PF], rb[(u + 2) % PF]);                    // chunk t+2
                gload((t + PF) * BK, ra[u], rb[u], true);                             // chunk t+PF
                mma_chunk(opa[u & 1], opb[u & 1]);
                // issue order inside the step: the chain first, everything else in the 40-cycle
                // gaps between its links (an MFMA behind 20 fresh LDS operations would otherwise
                // wait for lgkmcnt(0): the counter only encodes up to 15)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                }
            }
        }
        for (; t0 < nk; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                if (t < nk) {
                    lds_barrier();
                    if (t + 1 < nk) read_ops((u + 1) & 1, opa[(u + 1) & 1], opb[(u + 1) & 1]);
                    if (t + 2 < nk) sstore(u & 1, ra[(u + 2) % PF], rb[(u + 2) % PF]);
                    if (t + PF < nk) gload((t + PF) * BK, ra[u], rb[u], false);
                            mma_chunk(opa[u & 1], opb[u & 1]);
                }
            }
        }
    } else {
        sstore(0, ra[0], rb[0]);
        lds_barrier();
        int t0 = 0;
        for (; (t0 + 2 * PF) * BK <= d; t0 += PF) {   // steady state: branch-free, waits stay counted
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                gload((t + PF) * BK, ra[u], rb[u], true);   // set u was stored to LDS last step
                compute(t & 1);
                sstore((t + 1) & 1, ra[(u + 1) % PF], rb[(u + 1) % PF]);
                lds_barrier();
            }
        }
        for (; t0 < nk; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                if (t < nk) {
                    if (t + PF < nk) gload((t + PF) * BK, ra[u], rb[u], false);
                    compute(t & 1);
                    if (t + 1 < nk) sstore((t + 1) & 1, ra[(u + 1) % PF], rb[(u + 1) % PF]);
                    lds_barrier();
                }
            }
        }
    }
    // D layout of 16x16x4: lane holds rows (lane>>4)*4 + r, column lane&15
#pragma unroll
    for (int x = 0; x < WM; ++x)
#pragma unroll
        for (int y = 0; y < WN; ++y) {
            int col = n0 + (wni * WN + y) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int trow = (wmi * WM + x) * 16 + lg * 4 + r;   // row inside the tile
                if (ga.idx) {
                    if (trow == 0 && col < ga.kc) S[(size_t)tm * ldS + col] = acc[x][y][r];
                } else {
                    const int row = m0 + trow;
                    if (row < na && col < nb) S[(size_t)row * ldS + col] = acc[x][y][r];
                }
            }
        }
}

// ---------------------------------------------------------------------
// wave-level sorted list, one entry per lane, best first (lane 0 = best).
// insert keeps the list sorted; entries pushed past lane 63 are dropped.
// ---------------------------------------------------------------------
__device__ __forceinline__ void wave_insert_i32(float &ls, int &li, int lane, int k, float cs, int ci) {
    bool before = (ls > cs) || (ls == cs && li < ci);
    int r = __popcll(__ballot(before));
    if (r >= k) return;  // wave-uniform
    float us = wave_shr1_f(ls);
    int ui = wave_shr1_i(li);
    if (lane > r) {
        ls = us;
        li = ui;
    } else if (lane == r) {
        ls = cs;
        li = ci;
    }
}

// ---------------------------------------------------------------------
// Best K entries of each row of S under (score desc, column asc); one
// 256-thread workgroup per row.  Threshold paths for K <= 64 and K <= 256 (the host
// sends 64 < K <= 4096 to select_big_kernel); anything else: select_by_insertion.
// Unfilled: index -1, score -FLT_MAX.  out_i32 / out_i64 / out_s may be null.
// With list tables (coarse quantiser of a search): also emits, per row, the
// probe tables the scan kernel walks -- first group and length of every
// probed list and the exclusive prefix sum of their group counts.
// ---------------------------------------------------------------------
struct ProbeTables {
    const int32_t *list_goff;  // [nlist+1] in, null = no tables
    const int32_t *list_len;   // [nlist]   in
    int32_t *p_goff;           // [rows][K]   out
    int32_t *p_len;            // [rows][K]   out
    int32_t *p_prefix;         // [rows][K+1] out
};

// Per-row probe tables for the scan kernel from the selected list numbers
// idx_row[0..K) (global or LDS): first group and length of every probed list,
// exclusive prefix of their group counts.  Whole 256-thread workgroup.
__device__ __forceinline__ void emit_probe_tables(const ProbeTables &pt, int64_t row, int K,
                                                  const int32_t *idx_row, int *wtot) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (K + 255) / 256;
    const int b = tid * per;
    int sum = 0;
    // K <= 256 (one entry per thread): the list's first group, group count and length are
    // fetched once, together, and kept -- the second loop below would pay another dependent
    // memory round trip for the same three words
    int g0_1 = 0, ng_1 = 0, len_1 = 0;
    if (per == 1) {
        if (tid < K) {
            const int l = idx_row[tid];
            if (l >= 0) {
                g0_1 = pt.list_goff[l];
                ng_1 = pt.list_goff[l + 1] - g0_1;
                len_1 = pt.list_len[l];
            }
        }
        sum = ng_1;
    } else {
        for (int i = 0; i < per; ++i)
            if (b + i < K) {
                int l = idx_row[b + i];
                if (l >= 0) sum += pt.list_goff[l + 1] - pt.list_goff[l];
            }
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int ww = 0; ww < w; ++ww) run += wtot[ww];
    if (per == 1) {
        if (tid < K) {
            pt.p_goff[(size_t)row * K + tid] = g0_1;
            pt.p_len[(size_t)row * K + tid] = len_1;
            pt.p_prefix[(size_t)row * (K + 1) + tid] = run;
            if (tid == K - 1) pt.p_prefix[(size_t)row * (K + 1) + K] = run + ng_1;
        }
        return;
    }
    for (int i = 0; i < per; ++i)
        if (b + i < K) {
            int l = idx_row[b + i];
            int g0 = 0, ng = 0, len = 0;
            if (l >= 0) {
                g0 = pt.list_goff[l];
                ng = pt.list_goff[l + 1] - g0;
                len = pt.list_len[l];
            }
            pt.p_goff[(size_t)row * K + b + i] = g0;
            pt.p_len[(size_t)row * K + b + i] = len;
            pt.p_prefix[(size_t)row * (K + 1) + b + i] = run;
            run += ng;
            if (b + i == K - 1) pt.p_prefix[(size_t)row * (K + 1) + K] = run;
        }
}

// search_preassigned: the coarse result comes from outside (e.g. merged from
// centroid slices computed on several GPUs); only the probe tables are needed.
__global__ void __launch_bounds__(256)
    probe_tables_kernel(ProbeTables pt, int K, const int32_t *__restrict__ idx) {
    __shared__ int wtot[4];
    emit_probe_tables(pt, blockIdx.x, K, idx + (size_t)blockIdx.x * K, wtot);
}

// Sliced selection (launch_select, few long rows): one workgroup walking a 65 536-column row is a chain of 16 tiles (37 us); 16
// workgroups take a 4 096-column slice each (one tile: the row stays in registers) and a second pass picks the K best of the
// 16 K results.  Pass 1: mod = slices per row (the slice's first column is added to the indices).  Pass 2: remap = the slices'
// indices [rows][mod K]; equal scores keep their order: inside a slice they come index-ascending, and slices ascend.
struct SelSlices {
    int mod;
    const int32_t *remap;
};


// Last-resort selection for any K and n: per-wave sorted lists with serial insertion, 64
// results per pass over the row (pass p keeps the best 64 among the entries strictly after
// the last entry of pass p-1).  Whole 256-thread workgroup; c_s / c_i: 256 LDS slots,
// o_s / o_i: 64.  Slow (one pass over the row per 64 results): only rows with more tied
// scores than the threshold paths have survivor slots end up here.
__device__ __noinline__ void select_by_insertion(const float *__restrict__ r, int n, int K, int64_t row,
                                                 int32_t *__restrict__ out_i32, int64_t *__restrict__ out_i64,
                                                 float *__restrict__ out_s, int idx_off, float *c_s, int *c_i,
                                                 float *o_s, int *o_i, const int32_t *__restrict__ remap = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = uniform_i(tid >> 6);
    float *m_s = c_s;
    int *m_i = c_i;
    bool has_bound = false;
    float bs = 0.f;
    int bi = 0;
    for (int p0 = 0; p0 < K; p0 += 64) {
        const int kp = min(64, K - p0);
        float ls = MI_NEG_INF, thr = MI_NEG_INF;
        int li = INT_MAX;
        for (int base = w * 64; base < n; base += 256) {
            int c = base + lane;
            bool valid = c < n;
            float s = valid ? r[c] : 0.f;
            bool pf = valid && (s >= thr);
            if (has_bound) pf = pf && (s < bs || (s == bs && c > bi));
            unsigned long long mask = __ballot(pf);
            while (mask) {
                int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                float cs = readlane_f(s, src);
                if (!(cs >= thr)) continue;
                wave_insert_i32(ls, li, lane, kp, cs, base + src);
                thr = readlane_f(ls, kp - 1);
            }
        }
        m_s[tid] = ls;
        m_i[tid] = li;
        if (tid < 64) {
            o_s[tid] = MI_NEG_INF;
            o_i[tid] = INT_MAX;
        }
        __syncthreads();
        if (lane < kp && li != INT_MAX) {
            int rank = 0;
            for (int ww = 0; ww < 4; ++ww)
#pragma unroll 8
                for (int j = 0; j < kp; ++j) {
                    float js = m_s[ww * 64 + j];
                    int ji = m_i[ww * 64 + j];
                    rank += (js > ls) || (js == ls && ji < li);
                }
            if (rank < kp) {
                o_s[rank] = ls;
                o_i[rank] = li;
            }
        }
        __syncthreads();
        if (tid < kp) {
            int oi = o_i[tid];
            float os = o_s[tid];
            size_t o = (size_t)row * K + p0 + tid;
            const int sel = oi == INT_MAX ? -1 : remap ? remap[oi] : oi + idx_off;
            if (out_i32) out_i32[o] = sel;
            if (out_i64) out_i64[o] = (int64_t)sel;
            if (out_s) out_s[o] = oi == INT_MAX ? -FLT_MAX : os;
        }
        has_bound = true;
        bs = o_s[kp - 1];
        bi = o_i[kp - 1];
        __syncthreads();
    }
}

constexpr int SEL_CAP = 1024;  // survivor slots of the fast path

__global__ void __launch_bounds__(256)
    select_kernel(const float *__restrict__ S, int64_t ldS, int n, int K, int32_t *__restrict__ out_i32,
                  int64_t *__restrict__ out_i64, float *__restrict__ out_s, ProbeTables pt, int idx_off, SelSlices sl) {
    __shared__ float c_s[SEL_CAP];   // survivors (fast path) / wave lists (fallback: first 256)
    __shared__ int c_i[SEL_CAP];
    __shared__ int c_rank[SEL_CAP];
    __shared__ unsigned gkey4[1024];
    unsigned *gkey = gkey4;
    __shared__ float o_s[256];
    __shared__ int o_i[256];
    __shared__ int wtot[4];
    __shared__ int c_cnt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t row = blockIdx.x;
    const float *r = S + row * ldS;
    // sliced selection of a few long rows (SelSlices): pass 1 -- this "row" is slice row % mod of a longer one, its columns start
    // at (row % mod) * n; pass 2 -- the row holds the slices' results, column c stands for the index remap[c]
    if (sl.mod > 1) idx_off += (int)(row % sl.mod) * n;
    const int32_t *remap = sl.remap ? sl.remap + row * ldS : nullptr;

    // ---- fast path (K <= 256): the K-th largest of the 256 per-thread maxima
    // is a lower bound of the K-th largest element, so everything below it is
    // dropped with one compare; the few survivors are ranked by counting.
    bool done = false, tables_done = false;
    if (K <= 256) {
        // thread t owns 4 interleaved groups: elements c with c%256 == t, (c/256)%4 == g.
        // K <= 64 uses the 256 per-thread maxima, larger K the 1024 group maxima
        // (the bound is only tight when there are several times more groups than K).
        // The row is walked in tiles of 256 x VPT with VPT unconditional (clamped) loads
        // per thread in flight at once; a row of one tile (n <= 4096: every coarse
        // quantiser up to IVF4096) stays in registers for the survivor pass.
        constexpr int VPT = 16, TILE = 256 * VPT;
        const bool one_tile = n <= TILE;
        unsigned key[VPT];
        auto load_tile = [&](int base) {
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int c = base + j * 256 + tid;
                const float v = r[min(c, n - 1)];
                key[j] = (c < n && v == v) ? f2o(v) : 0u;
            }
        };
        unsigned gm[4] = {0u, 0u, 0u, 0u};
        for (int base = 0; base < n; base += TILE) {
            load_tile(base);
#pragma unroll
            for (int j = 0; j < VPT; ++j) gm[j & 3] = max(gm[j & 3], key[j]);
        }
        o_s[tid] = MI_NEG_INF;
        o_i[tid] = INT_MAX;
        if (tid == 0) c_cnt = 0;
        unsigned T0 = 0;
        if (K <= 64) {
            gkey[tid] = max(max(gm[0], gm[1]), max(gm[2], gm[3]));
            __syncthreads();
            // (rotated per wave, so that a fold below decorrelates the waves' columns)
            const unsigned k0 = gkey[lane], k1 = gkey[64 + ((lane + 16) & 63)], k2 = gkey[128 + ((lane + 32) & 63)],
                           k3 = gkey[192 + ((lane + 48) & 63)];
            if (K <= 16) {
                // any K distinct elements bound the K-th largest from below: the 64 column
                // maxima do, at one ballot per descent step instead of four
                const unsigned f = max(max(k0, k1), max(k2, k3));
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned t = T0 | (1u << bit);
                    const int c = __popcll(__ballot(f >= t));
                    if (c >= K) T0 = t;
                    if (c == K) break;
                }
            } else {
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned t = T0 | (1u << bit);
                    const int c = __popcll(__ballot(k0 >= t)) + __popcll(__ballot(k1 >= t)) +
                                  __popcll(__ballot(k2 >= t)) + __popcll(__ballot(k3 >= t));
                    if (c >= K) T0 = t;
                    if (c == K) break;  // t already separates exactly K group maxima
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) gkey4[g * 256 + tid] = gm[g];
            __syncthreads();
            unsigned kk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) kk[i] = gkey4[i * 64 + lane];
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned t = T0 | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) c += __popcll(__ballot(kk[i] >= t));
                if (c >= K) T0 = t;
                if (c == K) break;
            }
        }
        // survivors: key >= T0 (NaN has key 0 and never survives); one LDS atomic per
        // wave and tile reserves the slots of all VPT ballots
        for (int base = 0; base < n; base += TILE) {
            if (!one_tile) load_tile(base);
            unsigned long long m[VPT];
            int tot = 0;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                m[j] = __ballot(key[j] != 0u && key[j] >= T0);
                tot += __popcll(m[j]);
            }
            if (tot) {
                int o = 0;
                if (lane == 0) o = atomicAdd(&c_cnt, tot);
                o = uniform_i(o);
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    if (m[j]) {
                        const int pos = o + lane_prefix_count(m[j]);
                        if (((m[j] >> lane) & 1ull) && pos < SEL_CAP) {
                            c_s[pos] = o2f(key[j]);
                            c_i[pos] = base + j * 256 + tid;
                        }
                        o += __popcll(m[j]);
                    }
                }
            }
        }
        __syncthreads();
        const int Sn = c_cnt;
        if (Sn <= SEL_CAP) {
            for (int e = tid; e < Sn; e += 256) c_rank[e] = 0;
            __syncthreads();
            if (Sn > 0) {
                const int P = max(1, 256 / Sn);  // thread groups sharing the j range (Sn < 256)
                for (int e0 = 0; e0 < Sn; e0 += 256) {
                    const int part = (Sn < 256) ? tid / Sn : 0;
                    const int e = (Sn < 256) ? tid - part * Sn : e0 + tid;
                    if (part < P && e < Sn) {
                        const float es = c_s[e];
                        const int ei = c_i[e];
                        const int j0 = (Sn * part) / P, j1 = (Sn * (part + 1)) / P;
                        int rk = 0;
#pragma unroll 8
                        for (int j = j0; j < j1; ++j) {
                            const float js = c_s[j];
                            const int ji = c_i[j];
                            rk += (js > es) || (js == es && ji < ei);
                        }
                        if (rk) atomicAdd(&c_rank[e], rk);
                    }
                }
            }
            __syncthreads();
            for (int e = tid; e < Sn; e += 256) {
                const int rk = c_rank[e];
                if (rk < K) {
                    o_s[rk] = c_s[e];
                    o_i[rk] = c_i[e];
                }
            }
            __syncthreads();
            if (tid < K) {
                const int oi = o_i[tid];
                const float os = o_s[tid];
                const size_t o = (size_t)row * K + tid;
                const int sel = oi == INT_MAX ? -1 : remap ? remap[oi] : oi + idx_off;
                if (out_i32) out_i32[o] = sel;
                if (out_i64) out_i64[o] = (int64_t)sel;
                if (out_s) out_s[o] = oi == INT_MAX ? -FLT_MAX : os;
                c_rank[tid] = sel;   // the selection stays in LDS for the probe tables
            }
            __syncthreads();
            if (pt.list_goff) {
                emit_probe_tables(pt, row, K, c_rank, wtot);
                tables_done = true;
            }
            done = true;
        }
    }

    // ---- K > 256 reaches this kernel only through select_big_kernel's overflow; a
    // pathological row with > SEL_CAP tied survivors falls through to here as well.
    if (!done) select_by_insertion(r, n, K, row, out_i32, out_i64, out_s, idx_off, c_s, c_i, o_s, o_i, remap);
    if (pt.list_goff && !tables_done) emit_probe_tables(pt, row, K, out_i32 + (size_t)row * K, wtot);
}

// ---------------------------------------------------------------------
// Best K of each row for 256 < K <= SELB_CAP (large nprobe: the recall >= 0.95 operating
// points probe a quarter of the lists; flat search with a large k).  Same contract as
// select_kernel, one 256-thread workgroup per row, three steps:
//   1. one pass over the row keeps 16 maxima per thread (thread t, slot j: the elements
//      c = t + 256 (j + 16 i)): 4096 group maxima -- the elements themselves when
//      n <= 4096, which then stay in registers;
//   2. the K-th largest of the 4096 maxima by a bitwise descent on order-preserving keys
//      (per step: 16 ballots per wave, one LDS word per wave, one barrier).  Any K distinct
//      elements bound the K-th largest element from below, so everything under it is
//      dropped by one compare (exact for n <= 4096; ~K(1 + 1/7) survivors at K = 1024 of
//      65536);
//   3. survivors become 64-bit keys (score key << 32 | ~column: descending order = score
//      desc, column asc) and are sorted by a bitonic network in LDS; the first K leave.
// More than SELB_CAP survivors (masses of tied scores): select_by_insertion.
// ---------------------------------------------------------------------
constexpr int SELB_CAP = 4096;
constexpr int SELP_CAP = 8192;   // select_pairs_kernel<32>: the largest k of an IVF-PQ search (candidate lists of the refine stage)

__global__ void __launch_bounds__(256)
    select_big_kernel(const float *__restrict__ S, int64_t ldS, int n, int K, int32_t *__restrict__ out_i32,
                      int64_t *__restrict__ out_i64, float *__restrict__ out_s, ProbeTables pt, int idx_off) {
    __shared__ unsigned long long skey[SELB_CAP];   // survivors; afterwards the selection (int32)
    __shared__ int wcnt[2][4];
    __shared__ int wtot[4];
    __shared__ int c_cnt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = uniform_i(tid >> 6);
    const int64_t row = blockIdx.x;
    const float *r = S + row * ldS;
    constexpr int VPT = 16, TILE = 256 * VPT;
    const bool one_tile = n <= TILE;
    unsigned key[VPT];
    auto load_tile = [&](int base) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = base + j * 256 + tid;
            const float v = r[min(c, n - 1)];
            key[j] = (c < n && v == v) ? f2o(v) : 0u;
        }
    };
    unsigned gm[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) gm[j] = 0u;
    for (int base = 0; base < n; base += TILE) {
        load_tile(base);
#pragma unroll
        for (int j = 0; j < VPT; ++j) gm[j] = max(gm[j], key[j]);
    }
    if (tid == 0) c_cnt = 0;
    // K-th largest group maximum (0 when fewer than K groups hold anything: keep all)
    unsigned T0 = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned t = T0 | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) c += __popcll(__ballot(gm[j] >= t));
        if (lane == 0) wcnt[bit & 1][w] = c;
        __syncthreads();
        c = wcnt[bit & 1][0] + wcnt[bit & 1][1] + wcnt[bit & 1][2] + wcnt[bit & 1][3];
        if (c >= K) T0 = t;
        if (c == K) break;
    }
    for (int base = 0; base < n; base += TILE) {
        if (!one_tile) load_tile(base);
        unsigned long long m[VPT];
        int tot = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            m[j] = __ballot(key[j] != 0u && key[j] >= T0);
            tot += __popcll(m[j]);
        }
        if (tot) {
            int o = 0;
            if (lane == 0) o = atomicAdd(&c_cnt, tot);
            o = uniform_i(o);
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                if (m[j]) {
                    const int pos = o + lane_prefix_count(m[j]);
                    if (((m[j] >> lane) & 1ull) && pos < SELB_CAP)
                        skey[pos] = ((unsigned long long)key[j] << 32) | (unsigned)~(unsigned)(base + j * 256 + tid);
                    o += __popcll(m[j]);
                }
            }
        }
    }
    __syncthreads();
    const int Sn = c_cnt;
    if (Sn > SELB_CAP) {   // wave-uniform, workgroup-uniform
        __syncthreads();
        float *f = reinterpret_cast<float *>(skey);
        select_by_insertion(r, n, K, row, out_i32, out_i64, out_s, idx_off, f, reinterpret_cast<int *>(f + 256),
                            f + 512, reinterpret_cast<int *>(f + 768));
        if (pt.list_goff) emit_probe_tables(pt, row, K, out_i32 + (size_t)row * K, wtot);
        return;
    }
    int P = 64;
    while (P < Sn) P <<= 1;
    for (int e = Sn + tid; e < P; e += 256) skey[e] = 0ull;   // below every survivor (their score key is never 0)
    __syncthreads();
    // bitonic network, descending
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = skey[lo], b = skey[hi];
                const bool desc = (lo & k2) == 0;
                if ((a < b) == desc) {
                    skey[lo] = b;
                    skey[hi] = a;
                }
            }
            __syncthreads();
        }
    // results (K <= 4096: at most 16 per thread, through registers so that the selection can
    // take the place of the keys)
    constexpr int RPT = SELB_CAP / 256;
    unsigned long long res[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int e = tid + i * 256;
        res[i] = (e < K && e < Sn) ? skey[e] : 0ull;
    }
    __syncthreads();
    int32_t *sel = reinterpret_cast<int32_t *>(skey);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int e = tid + i * 256;
        if (e < K) {
            const bool filled = res[i] != 0ull;
            const int col = (int)~(unsigned)res[i];
            const int si = filled ? col + idx_off : -1;
            const size_t o = (size_t)row * K + e;
            if (out_i32) out_i32[o] = si;
            if (out_i64) out_i64[o] = (int64_t)si;
            if (out_s) out_s[o] = filled ? o2f((unsigned)(res[i] >> 32)) : -FLT_MAX;
            sel[e] = si;
        }
    }
    __syncthreads();
    if (pt.list_goff) emit_probe_tables(pt, row, K, sel, wtot);
}

// ---------------------------------------------------------------------
// Best K (score desc, id asc) of each row of (score, id) pairs -- the rows
// scan_kernel<.., ALL> stores when k > 64: row q holds p_prefix[q][nprobe] * 64 pairs,
// NaN scores are padding.  One pass over the codes instead of one per 64 results.
// select_big_kernel's scheme (4096 group maxima -> K-th largest by block-wide descent ->
// survivors -> bitonic sort) on (score key, id) keys; the ids of the survivors only are
// fetched.  More survivors than slots (masses of tied scores) take the exact route: the
// K-th largest score key T by a descent over the whole row, then -- if the entries
// tied at T still do not fit -- the id of the last tied entry to keep by a descent over
// the id bits; exactly K entries survive.
// ---------------------------------------------------------------------
// VPT_ = 16: K <= 4096 (CAP = 256 VPT_ survivor slots, as many group maxima); VPT_ = 32: K <= 8192 -- the refine stage's
// candidate lists at the recall >= 0.95 operating point of the whole 207 M index (k * k_factor_rf of several thousand).
// SET_ (the first stage of IndexRefine: the candidate list only feeds the re-rank, whose result does not depend on the
// order of its candidates): the same K entries, written in no particular order and without their scores -- the exact
// K-th key by a descent over the survivors in LDS instead of the 91-stage bitonic sort of 8192 slots (1.24 ms of the
// 4.6 ms whole-index refine step).  Ties at the cut that do not all fit keep the sorted route (ids decide).
#ifdef MI_SELP_TS   // profiling build only (tools/micro/selp_stamps.py): s_memtime at the phase boundaries of every workgroup
__device__ unsigned long long selp_ts[8 * 4096];
#define SELP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) selp_ts[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SELP_STAMP(i) do {} while (0)
#endif
// NT_ threads per workgroup (CAP = NT_ VPT_ slots): <32, ., 256> and <16, ., 512> hold the same 8192 slots -- twice the waves
// per row halve every per-thread chain (ballots per descent step, loads per pass, output rounds)
template <int VPT_, bool SET_ = false, int NT_ = 256>
__global__ void __launch_bounds__(NT_, SET_ ? (NT_ == 256 ? 2 : 4) : 1)
    select_pairs_kernel(const float *__restrict__ S, const int64_t *__restrict__ IDS, int64_t ld,
                        const int32_t *__restrict__ p_prefix, int nprobe, int K, float *__restrict__ D,
                        int64_t *__restrict__ I, int64_t ldo, const int32_t *__restrict__ p_goff = nullptr,
                        const int64_t *__restrict__ list_ids = nullptr) {
    constexpr int CAP = NT_ * VPT_;
    constexpr int NW = NT_ / 64;
    // the survivors: key + id (sorted mode: the sort moves both), or key + 32-bit column in the set mode, whose ids are
    // fetched when they are written out -- 64 instead of 96 KiB of LDS at 8192 slots: two workgroups per CU, and this
    // kernel spends two thirds of its wave cycles waiting (memory round trips, 70 block-wide barriers)
    using SlotT = std::conditional_t<SET_, int32_t, int64_t>;
    __shared__ unsigned sk[CAP];
    __shared__ SlotT sid[CAP];
    __shared__ int wcnt[2][NW];
    __shared__ unsigned wmm[2][2][NW];
    __shared__ int c_cnt, c_eq;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = uniform_i(tid >> 6);
    const int64_t row = blockIdx.x;
    SELP_STAMP(0);
    const int n = p_prefix[row * (nprobe + 1) + nprobe] * 64;
    const float *r = S + row * ld;
    // the id of column c: the row of ids the scan stored beside the scores, or -- IDS null: the all-scores scan of an
    // inner-product search stores scores only -- read from the lists themselves through the probe tables: column c is lane
    // c % 64 of the query's group c / 64, i.e. group (c / 64 - prefix[p]) of probe p's list, whose groups start at p_goff[p]
    // (the scan's own addressing).  Only the survivors' ids are ever read: 5 120 of 25.6 k at the recall-0.95 point, where
    // storing every id cost the scan 16 of its 84 bytes per code.
    // (the two probe tables of the row in LDS when they fit: the search is then registers and LDS, the id itself one load)
    constexpr int TAB = 256;
    __shared__ int s_pre[TAB + 1], s_goff[TAB];
    const int32_t *pre_g = p_prefix + row * (nprobe + 1);
    const bool tab_lds = !IDS && nprobe <= TAB;
    if (tab_lds) {
        for (int p = tid; p <= nprobe; p += NT_) s_pre[p] = pre_g[p];
        for (int p = tid; p < nprobe; p += NT_) s_goff[p] = p_goff[row * nprobe + p];
        __syncthreads();
    }
    auto id_at = [&](int c) -> int64_t {
        if (IDS) return IDS[row * ld + c];
        const int t = c >> 6;
        int lo = 0, hi = nprobe;                              // last p with prefix[p] <= t
        if (tab_lds) {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_pre[mid] <= t) lo = mid; else hi = mid;
            }
            return list_ids[(size_t)(s_goff[lo] + (t - s_pre[lo])) * 64 + (c & 63)];
        }
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre_g[mid] <= t) lo = mid; else hi = mid;
        }
        return list_ids[(size_t)(p_goff[row * nprobe + lo] + (t - pre_g[lo])) * 64 + (c & 63)];
    };
    constexpr int VPT = VPT_, TILE = NT_ * VPT;
    // Which column a thread's j-th value of a tile is, is free (the survivors carry their column): four CONSECUTIVE columns per
    // lane, one 16-byte load -- 25.6 k scores per query as 32 four-byte loads per thread and round ran at 0.75 TB/s chip-wide
    // (profiles/r04_select_pairs_stamps.txt: the first pass was 43 % of the kernel).
    auto col_of = [&](int base, int j) { return base + ((j >> 2) * NT_ + tid) * 4 + (j & 3); };
    const bool wide = ((reinterpret_cast<uintptr_t>(r) | (uintptr_t)(ld * 4)) & 15) == 0;   // (workgroup-uniform) rows 16-byte aligned
    auto load_tile = [&](int base, unsigned (&key)[VPT]) {
        if (wide) {   // (workgroup-uniform) the tile's loads first, then the keys: per load, hipcc waits for each before the next
            float4 raw[VPT / 4];
#pragma unroll
            for (int j4 = 0; j4 < VPT / 4; ++j4) {
                const int c0 = col_of(base, 4 * j4);
                raw[j4] = *reinterpret_cast<const float4 *>(r + (c0 < n ? c0 : 0));   // n is a multiple of 64: all four or none
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j4 = 0; j4 < VPT / 4; ++j4) {
                const bool in = col_of(base, 4 * j4) < n;
                const float v[4] = {raw[j4].x, raw[j4].y, raw[j4].z, raw[j4].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned kx = f2o(v[q]);
                    key[4 * j4 + q] = (in && v[q] == v[q]) ? kx : 0u;
                }
            }
            return;
        }
#pragma unroll
        for (int j4 = 0; j4 < VPT / 4; ++j4) {
            const int c0 = col_of(base, 4 * j4);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = r[min(c0 + q, max(n - 1, 0))];
#pragma unroll
            for (int q = 0; q < 4; ++q) key[4 * j4 + q] = (c0 + q < n && v[q] == v[q]) ? f2o(v[q]) : 0u;
        }
    };
    // up to RT tiles of keys stay in registers between the two passes (32 k scores at VPT 32: the refine stage's rows): the
    // compaction pass then reads nothing
    constexpr int RT = 4;
    const bool resident = n <= RT * TILE;
    unsigned kk[RT][VPT];
    unsigned gm[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) gm[j] = 0u;
    if (resident && wide) {
        // every load of the thread first (columns past n: the row's first line, key 0), then the keys: written as load +
        // convert per tile, each 16-byte load was followed by `s_waitcnt vmcnt(0)` and the NaN test's branches -- one memory
        // latency per load, 32 in a row (the first pass was 36 % of the kernel: profiles/r04_select_pairs_stamps.txt)
        float4 raw[RT][VPT / 4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j4 = 0; j4 < VPT / 4; ++j4) {
                const int c0 = col_of(t * TILE, 4 * j4);
                raw[t][j4] = *reinterpret_cast<const float4 *>(r + (c0 < n ? c0 : 0));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j4 = 0; j4 < VPT / 4; ++j4) {
                const bool in = col_of(t * TILE, 4 * j4) < n;       // n is a multiple of 64: all four or none
                const float v[4] = {raw[t][j4].x, raw[t][j4].y, raw[t][j4].z, raw[t][j4].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned kx = f2o(v[q]);
                    kk[t][4 * j4 + q] = (in && v[q] == v[q]) ? kx : 0u;
                }
            }
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j = 0; j < VPT; ++j) gm[j] = max(gm[j], kk[t][j]);
    } else if (resident) {
#pragma unroll
        for (int t = 0; t < RT; ++t) load_tile(t * TILE, kk[t]);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j = 0; j < VPT; ++j) gm[j] = max(gm[j], kk[t][j]);
    } else {
        for (int base = 0; base < n; base += TILE) {
            load_tile(base, kk[0]);
#pragma unroll
            for (int j = 0; j < VPT; ++j) gm[j] = max(gm[j], kk[0][j]);
        }
    }
    if (tid == 0) {
        c_cnt = 0;
        c_eq = 0;
    }
    int ph = 0;
    SELP_STAMP(1);
    auto block_sum = [&](int wave_total) -> int {   // one barrier; alternating slots
        if (lane == 0) wcnt[ph][w] = wave_total;
        __syncthreads();
        int c = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) c += wcnt[ph][i];
        ph ^= 1;
        return c;
    };
    // The K-th largest of the block's keys v (0 = no key; 0 when fewer than K keys): a bitwise descent on the count of keys >= t,
    // started below the common prefix of the largest and the smallest key -- the scores of one query share their sign, exponent
    // and often a few mantissa bits: a third of the 32 block-wide steps (with the count of keys, the prefix bits are decided:
    // count >= K sets every one bit of the prefix, a zero bit of it can never be set).  An exact count ends it early.
    auto kth_largest = [&](const unsigned (&v)[VPT_]) -> unsigned {
        unsigned mx = 0u, mn = ~0u;
        int c = 0;
#pragma unroll
        for (int j = 0; j < VPT_; ++j) {
            mx = max(mx, v[j]);
            mn = min(mn, v[j] != 0u ? v[j] : ~0u);
            c += __popcll(__ballot(v[j] != 0u));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
            mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
        }
        if (lane == 0) {
            wcnt[ph][w] = c;
            wmm[ph][0][w] = mx;
            wmm[ph][1][w] = mn;
        }
        __syncthreads();
        c = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            c += wcnt[ph][i];
            mx = max(mx, wmm[ph][0][i]);
            mn = min(mn, wmm[ph][1][i]);
        }
        ph ^= 1;
        if (c < K) return 0u;
        const unsigned diff = mx ^ mn;
        int bit = diff ? 31 - __clz((int)diff) : -1;
        unsigned T = diff ? (mx & ~((2u << bit) - 1u)) : mx;
        if (c == K) return T;
        // (two bits a step -- three thresholds, one barrier -- measured slower: 19.4 -> 26.8 us.  A step is bound by the CU's
        // scalar unit, two scalar instructions per ballot whatever the number of waves that share the keys.)
        for (; bit >= 0; --bit) {
            const unsigned t = T | (1u << bit);
            int cc = 0;
#pragma unroll
            for (int j = 0; j < VPT_; ++j) cc += __popcll(__ballot(v[j] >= t));
            cc = block_sum(cc);
            if (cc >= K) T = t;
            if (cc == K) break;
        }
        return T;
    };
    const unsigned T0 = kth_largest(gm);
    // survivors of `keep(key, column)` into sk / sid
    // (pure: `keep` has no side effects -- the ballots are then taken twice, once for the wave's total and once for the
    // positions, instead of VPT 64-bit masks held across the slot reservation: 64 SGPRs at VPT 32, which the compiler spilt)
    auto compact = [&](auto keep, auto pure) {
        auto tile = [&](int base, const unsigned (&key)[VPT]) {
            if constexpr (decltype(pure)::value) {
                int tot = 0;
#pragma unroll
                for (int j = 0; j < VPT; ++j) tot += __popcll(__ballot(key[j] != 0u && keep(key[j], col_of(base, j))));
                if (tot) {
                    int o = 0;
                    if (lane == 0) o = atomicAdd(&c_cnt, tot);
                    o = uniform_i(o);
#pragma unroll
                    for (int j = 0; j < VPT; ++j) {
                        const bool kp = key[j] != 0u && keep(key[j], col_of(base, j));
                        const unsigned long long m = __ballot(kp);
                        if (m) {
                            const int pos = o + lane_prefix_count(m);
                            if (kp && pos < CAP) {
                                sk[pos] = key[j];
                                sid[pos] = (SlotT)col_of(base, j);
                            }
                            o += __popcll(m);
                        }
                    }
                }
                return;
            }
            unsigned long long m[VPT];
            int tot = 0;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                m[j] = __ballot(key[j] != 0u && keep(key[j], col_of(base, j)));
                tot += __popcll(m[j]);
            }
            if (tot) {
                int o = 0;
                if (lane == 0) o = atomicAdd(&c_cnt, tot);
                o = uniform_i(o);
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    if (m[j]) {
                        const int pos = o + lane_prefix_count(m[j]);
                        if (((m[j] >> lane) & 1ull) && pos < CAP) {
                            sk[pos] = key[j];
                            sid[pos] = (SlotT)col_of(base, j);   // the column; its id is fetched below (set mode: on output)
                        }
                        o += __popcll(m[j]);
                    }
                }
            }
        };
        if (resident) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
                if (t * TILE < n) tile(t * TILE, kk[t]);
        } else {
            for (int base = 0; base < n; base += TILE) {
                load_tile(base, kk[0]);
                tile(base, kk[0]);
            }
        }
        __syncthreads();
        // ids of the survivors: independent loads, all of a thread's in flight at once (fetched inside the loop above
        // every load sat in front of its own LDS store: a DRAM latency per survivor column)
        if constexpr (!SET_) {
            const int got = min(c_cnt, CAP);
            for (int e = tid; e < got; e += NT_) sid[e] = id_at((int)sid[e]);
            __syncthreads();
        }
    };
    SELP_STAMP(2);
    compact([&](unsigned kx, int) { return kx >= T0; }, std::true_type{});
    SELP_STAMP(3);
    int Sn = c_cnt;
    if (Sn > CAP) {   // workgroup-uniform
        // exact route: counts over the whole row, streamed (rare: masses of tied scores)
        auto count_row = [&](auto pred) -> int {
            int c = 0;
            for (int base = 0; base < n; base += NT_) {
                const int col = base + tid;
                bool p = false;
                if (col < n) {
                    const float v = r[col];
                    const unsigned kx = v == v ? f2o(v) : 0u;
                    p = kx != 0u && pred(kx, col);
                }
                c += __popcll(__ballot(p));
            }
            return block_sum(c);
        };
        unsigned T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned t = T | (1u << bit);
            const int c = count_row([&](unsigned kx, int) { return kx >= t; });
            if (c >= K) T = t;
            if (c == K) break;
        }
        const int cgt = count_row([&](unsigned kx, int) { return kx > T; });
        const int ceq = count_row([&](unsigned kx, int) { return kx == T; });
        if (tid == 0) c_cnt = 0;
        __syncthreads();
        if (cgt + ceq <= CAP) {
            compact([&](unsigned kx, int) { return kx >= T; }, std::true_type{});
        } else {
            // the (K - cgt)-th smallest id among the entries tied at T: the largest U with
            // fewer than that many tied ids below it
            const int need = K - cgt;
            int64_t U = 0;
            for (int bit = 62; bit >= 0; --bit) {
                const int64_t t = U | ((int64_t)1 << bit);
                const int c = count_row([&](unsigned kx, int col) { return kx == T && id_at(col) < t; });
                if (c < need) U = t;
            }
            const int below = count_row([&](unsigned kx, int col) { return kx == T && id_at(col) < U; });
            const int take_eq = need - below;   // entries with id == U to keep (duplicated ids)
            compact([&](unsigned kx, int col) {
                if (kx > T) return true;
                if (kx != T) return false;
                const int64_t id = id_at(col);
                if (id < U) return true;
                if (id > U) return false;
                return atomicAdd(&c_eq, 1) < take_eq;
            }, std::false_type{});
        }
        Sn = c_cnt;   // <= CAP now
    }
    if constexpr (SET_) {
        // unordered output: everything when the survivors are no more than K, else the entries above the exact K-th key T
        // plus, of the ties at T, the take_eq with the smallest ids (usually all of them: K - cgt == ceq)
        unsigned T = 0;
        int take_eq = 0, ceq = 0;
        int64_t U = INT64_MAX;      // ties at T with id < U are kept, of those with id == U the first take_u
        int take_u = 0;
        const bool all = Sn <= K;
        // the survivors' keys in registers (CAP / 256 per thread): the descent is ballots + popcounts, no LDS traffic
        unsigned mine[VPT_];
#pragma unroll
        for (int j = 0; j < VPT_; ++j) mine[j] = (j * NT_ + tid) < Sn ? sk[j * NT_ + tid] : 0u;
        // ... and their ids, every load of a thread in flight at once and under the descent below (fetched inside the output
        // loop each of its up to 32 rounds waited for its own scattered load: ~45 us of a ~170 us workgroup)
        int64_t idv[VPT_];
        if (tab_lds) {
            // the probe-table search of every slot first (a fixed number of branch-free steps, the slots interleaved), then all
            // the id loads together: through id_at() each slot's search loop and load sat behind the previous slot's
            int lo[VPT_], hi[VPT_], tg[VPT_], cl[VPT_];
#pragma unroll
            for (int j = 0; j < VPT_; ++j) {
                const int c = mine[j] != 0u ? (int)sid[j * NT_ + tid] : 0;
                tg[j] = c >> 6;
                cl[j] = c & 63;
                lo[j] = 0;
                hi[j] = nprobe;
            }
            int nsteps = 0;
            while ((1 << nsteps) < nprobe) ++nsteps;
            for (int st = 0; st < nsteps; ++st) {
#pragma unroll
                for (int j = 0; j < VPT_; ++j) {
                    const int mid = (lo[j] + hi[j]) >> 1;            // hi - lo <= 1: mid = lo, nothing moves
                    const bool le = s_pre[mid] <= tg[j];
                    lo[j] = le ? mid : lo[j];
                    hi[j] = le ? hi[j] : mid;
                }
            }
#pragma unroll
            for (int j = 0; j < VPT_; ++j) {
                const size_t at = (size_t)(s_goff[lo[j]] + (tg[j] - s_pre[lo[j]])) * 64 + cl[j];
                idv[j] = mine[j] != 0u ? list_ids[at] : (int64_t)0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < VPT_; ++j) idv[j] = mine[j] != 0u ? id_at((int)sid[j * NT_ + tid]) : (int64_t)0;
        }
        if (!all) {
            T = kth_largest(mine);
            int cg = 0, ce = 0;
#pragma unroll
            for (int j = 0; j < VPT_; ++j) {
                cg += __popcll(__ballot(mine[j] > T));
                ce += __popcll(__ballot(mine[j] == T && mine[j] != 0u));
            }
            const int cgt = block_sum(cg);
            ceq = block_sum(ce);
            take_eq = K - cgt;
            if (take_eq != ceq) {   // workgroup-uniform, rare: the cut falls inside a run of equal scores -- ids decide
                auto count_ties = [&](auto pred) -> int {
                    int c = 0;
                    for (int base = 0; base < Sn; base += NT_) {
                        const int e = base + tid;
                        c += __popcll(__ballot(e < Sn && sk[e] == T && pred(id_at((int)sid[min(e, Sn - 1)]))));
                    }
                    return block_sum(c);
                };
                U = 0;
                for (int bit = 62; bit >= 0; --bit) {
                    const int64_t t = U | ((int64_t)1 << bit);
                    if (count_ties([&](int64_t id) { return id < t; }) < take_eq) U = t;
                }
                take_u = take_eq - count_ties([&](int64_t id) { return id < U; });
            }
        }
        SELP_STAMP(4);
        if (tid == 0) {
            c_cnt = 0;
            c_eq = 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VPT_; ++j) {
            if (j * NT_ >= Sn) continue;                   // workgroup-uniform (no break: the loop must unroll, idv[] lives in registers)
            bool keep = mine[j] != 0u;
            const int64_t id = idv[j];
            if (keep && !all) {
                const unsigned kx = mine[j];
                keep = kx > T || (kx == T && (take_eq == ceq || id < U || (id == U && atomicAdd(&c_eq, 1) < take_u)));
            }
            const unsigned long long m = __ballot(keep);
            int o = 0;
            if (lane == 0 && m) o = atomicAdd(&c_cnt, __popcll(m));
            o = uniform_i(o);
            if (keep) I[row * ldo + o + lane_prefix_count(m)] = id;
        }
        __syncthreads();
        for (int e = c_cnt + tid; e < K; e += NT_) I[row * ldo + e] = (int64_t)-1;
        SELP_STAMP(5);
#ifdef MI_SELP_TS
        if (tid == 0 && blockIdx.x < 4096) selp_ts[blockIdx.x * 8 + 6] = (unsigned long long)Sn;
#endif
        return;
    } else {
    int P = 64;
    while (P < Sn) P <<= 1;
    for (int e = Sn + tid; e < P; e += NT_) {
        sk[e] = 0u;   // after every survivor
        sid[e] = INT64_MAX;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += NT_) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const unsigned ka = sk[lo], kb = sk[hi];
                const int64_t ia = sid[lo], ib = sid[hi];
                const bool b_first = kb > ka || (kb == ka && ib < ia);   // entry hi ranks before entry lo
                const bool desc = (lo & k2) == 0;
                if (b_first == desc && !(ka == kb && ia == ib)) {
                    sk[lo] = kb;
                    sid[lo] = ib;
                    sk[hi] = ka;
                    sid[hi] = ia;
                }
            }
            __syncthreads();
        }
    for (int e = tid; e < K; e += NT_) {
        const bool filled = e < Sn && sk[e] != 0u;
        if (D) D[row * ldo + e] = filled ? o2f(sk[e]) : -FLT_MAX;
        I[row * ldo + e] = filled ? (int64_t)sid[e] : (int64_t)-1;
    }
    }   // sorted mode
}

// =====================================================================
// Two-stage coarse quantiser for large batches (1024 queries x 65536 centroids: the exact
// f32 GEMM is 1.03 ms of a 1.43 ms search step).  Stage 1: f16 MFMA GEMM, approximate
// scores.  Stage 2 (select_refine_kernel): every centroid whose approximate score is
// within 2 eps of the K-th best approximate score, eps a proven bound on |approx - exact|,
// gets its exact score (the f32 fmaf chain, k ascending -- what the f32 MFMA GEMM and the
// oracle compute) and the K best of those under (exact score desc, index asc) are the
// result: bit-identical to the one-stage path, because the exact top K all lie inside the
// candidate set:  exact_c >= exact_(K) >= approx_(K) - eps  =>  approx_c >= approx_(K) - 2 eps.
// eps = eps_rel |q| max|c|, from  approx - exact = (approx - sum q~c~) + (sum q~c~ - sum qc)
// + (sum qc - exact)  and Cauchy-Schwarz, sum |q_i c_i| <= |q| |c|:
//   f16 rounding of both operands: relative 2 * 2^-11 + 2^-22 per product.  Every query row
//     and the centroid matrix are multiplied by a power of two first that puts their largest
//     magnitude in [2^13, 2^14) (no overflow; undone exactly on the scores), so an element
//     only leaves f16's normal range when it is below 2^-27 of that magnitude -- even if the
//     MFMA flushed such inputs to zero the loss is <= 2^-27 sqrt(d) |q| |c| per operand;
//   f32 accumulation of the d exact f16 x f16 products inside the MFMA: d 2^-22 relative to
//     sum |q~c~| -- two units in the last place per addition, whatever the summation order
//     or rounding mode of the unit;
//   the exact chain's own rounding: d 2^-24.
// The host adds 1 % for the second-order terms; |q| and max|c| are rounded up by 0.1 %.
// =====================================================================

// value of lane (CTRL & 3) of the lane's quad in every lane of the quad (CTRL = quad_perm [j, j, j, j] = j * 0x55)
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Power of two that brings a largest magnitude m into [2^13, 2^14): far from f16's overflow
// (65504), and an element then only leaves f16's normal range if it is below 2^-27 m.
__device__ __forceinline__ float f16_pow2_scale(float m) {
    if (!(m > 0.f) || !(m <= 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, min(max(14 - e, -100), 100));
}

// rows of x [rows][d] -> f16, each row times its own power-of-two scale (stored); one wave per row.
// scale_all != 0: one given scale for every row (the centroids: no per-column unscale later).
__global__ void __launch_bounds__(256) to_f16_rows_kernel(const float *__restrict__ x, int rows, int d, f16_t *__restrict__ y,
                                                          float *__restrict__ scale_out, float scale_all) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * d);
    float sc = scale_all;
    if (sc == 0.f) {
        float m = 0.f;
        for (int k = lane; k < (d >> 2); k += 64) {
            const float4 v = xr[k];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        sc = f16_pow2_scale(m);
        if (lane == 0) scale_out[row] = sc;
    }
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 *yr = reinterpret_cast<h4 *>(y + (size_t)row * d);
    for (int k = lane; k < (d >> 2); k += 64) {
        const float4 v = xr[k];
        h4 o = {(f16_t)(v.x * sc), (f16_t)(v.y * sc), (f16_t)(v.z * sc), (f16_t)(v.w * sc)};
        yr[k] = o;
    }
}

// out[0] = largest row norm of x [n][d] (rounded up), out[1] = largest |element|
// (float bits; non-negative floats order like ints)
__global__ void __launch_bounds__(256) max_row_norm_kernel(const float *__restrict__ x, int n, int d, unsigned *__restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    float acc = 0.f, m = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float v = x[(size_t)row * d + k];
        acc = __builtin_fmaf(v, v, acc);
        m = fmaxf(m, fabsf(v));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off);
        m = fmaxf(m, __shfl_xor(m, off));
    }
    if (lane == 0) {
        atomicMax(out, __float_as_uint(sqrtf(acc) * 1.001f));
        atomicMax(out + 1, __float_as_uint(m));
    }
}

// Stage 2: one 256-thread workgroup per query row of approximate scores Sa [rows][ldS].
//   cut = (K-th largest of 4096 group maxima of the approximate row) - 2 eps;
//   candidates = columns with approximate score >= cut;
//   exact score of each candidate: fmaf chain over d, k ascending (one lane per candidate,
//     the query row in LDS), = the f32 MFMA GEMM's value for that pair;
//   bitonic sort on (exact score key << 32 | ~column), first K out + the probe tables.
// A row with more than SELB_CAP candidates (or non-finite approximate scores: f16 overflow)
// is recomputed exactly in full -- slow, but only degenerate data gets there -- and selected
// with a zero margin.  Same outputs as select_kernel / select_big_kernel.
struct RefineArgs {
    const float *q;       // [rows][d]  f32 queries
    const float *cent;    // [n][d]     f32 centroids
    float *Sa;            // [rows][ldS] approximate scores (overwritten by exact ones on the fallback)
    int64_t ldS;
    int n, d, K;
    float eps_rel, cmax;
    const float *qscale;  // [rows] power-of-two scale of each f16 query row
    float cscale;         // power-of-two scale of the f16 centroids
    int32_t *out_i32;     // [rows][K]
    float *out_s;         // [rows][K]
    ProbeTables pt;
    unsigned *stats;      // null, or [2]: total candidates, rows that fell back
    int idx_off;          // added to every index written (a slice of a larger centroid table: mi_index_coarse_slice)
    const float *gmax;    // null, or [rows][ngroups]: maxima of the approximate row over 64-column groups, written by the
    int ngroups;          // f16 GEMM's epilogue (+inf marks a group with a non-finite score): the row itself is then read
                          // only where a group can hold a candidate (268 MB of scores at 1024 x 65536 otherwise)
    unsigned long long *ts;   // null, or [rows][8] s_memtime stamps of the workgroup's phases (MI_REFINE_TS=1, tools only)
};

// The exact fallback of select_refine_kernel (a row whose approximate scores are not finite, or with more candidates than slots:
// degenerate data only): exact scores of the WHOLE row, one lane per column.  Not inlined -- inside the kernel its two register sets
// of 8 x 16 B counted against the 128 VGPRs that four workgroups per CU allow, and the spills landed in the common path (eight
// scratch reloads in the descent region).  qs: the query row (LDS, through a generic pointer: this path may be slow).
__device__ __noinline__ void refine_exact_row(float *r, const float *cent, const float *qs, int n, int d) {
    for (int col = (int)threadIdx.x; col < n; col += 256) {
        // d % 128 == 0 (host).  The centroid row is streamed through two register sets of 8 x 16 B, named (not copied: a copy
        // would make hipcc wait for the newest loads), so the loads of one set are in flight while the chain -- d dependent
        // fmafs, k ascending -- runs on the other.
            const float4 *cp = reinterpret_cast<const float4 *>(cent + (size_t)col * d);
            const int n4 = d >> 2;
            float4 A[8], B[8];
    #pragma unroll
            for (int i = 0; i < 8; ++i) A[i] = cp[i];
            float acc = 0.f;
            for (int k4 = 0; k4 < n4; k4 += 16) {
    #pragma unroll
                for (int i = 0; i < 8; ++i) B[i] = cp[k4 + 8 + i];
    #pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qs + 4 * (k4 + i));
                    acc = __builtin_fmaf(qv.x, A[i].x, acc);
                    acc = __builtin_fmaf(qv.y, A[i].y, acc);
                    acc = __builtin_fmaf(qv.z, A[i].z, acc);
                    acc = __builtin_fmaf(qv.w, A[i].w, acc);
                }
                const int nx = min(k4 + 16, n4 - 8);   // last round: a harmless re-read
    #pragma unroll
                for (int i = 0; i < 8; ++i) A[i] = cp[nx + i];
    #pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qs + 4 * (k4 + 8 + i));
                    acc = __builtin_fmaf(qv.x, B[i].x, acc);
                    acc = __builtin_fmaf(qv.y, B[i].y, acc);
                    acc = __builtin_fmaf(qv.z, B[i].z, acc);
                    acc = __builtin_fmaf(qv.w, B[i].w, acc);
                }
            }
            r[col] = acc;
    }
}

// DCAP: the longest query row the instantiation holds in LDS (d <= DCAP).  <1024>: 39 KiB of LDS, FOUR workgroups per CU --
// 1024 rows are one round of workgroups on 256 CUs where the 51 KiB of <4096> made them a round of 768 and one of 256.
template <int DCAP>
__global__ void __launch_bounds__(256, DCAP <= 1024 ? 4 : 3) select_refine_kernel(RefineArgs a) {   // (HIP: min waves per SIMD = workgroups per CU here)
    __shared__ unsigned long long skey[SELB_CAP];
    __shared__ __attribute__((aligned(16))) float qs[DCAP];   // the query row (d <= DCAP), read as float4
    __shared__ int wcnt[2][4];
    __shared__ float wred[4];
    __shared__ int wtot[4];
    __shared__ int c_cnt;
    __shared__ int g_cnt;
    __shared__ unsigned t0_sh;
    __shared__ unsigned short glist[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = uniform_i(tid >> 6);
    const int64_t row = blockIdx.x;
    const int n = a.n, d = a.d, K = a.K;
    float *r = a.Sa + row * a.ldS;
    const float *qg = a.q + row * d;
    auto stamp = [&](int i) {   // no-op unless a profiling run asked for stamps
        if (a.ts && tid == 0) a.ts[(size_t)row * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    // the row's group maxima (first attempt of the loop below) requested in front of the query row: one round trip, not two
    const bool ext0 = a.gmax != nullptr && K <= 256 && a.ngroups <= 1024 && a.ngroups >= 4 * K && n == a.ngroups * 64;
    float graw[4] = {0.f, 0.f, 0.f, 0.f};
    if (ext0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) graw[g] = a.gmax[row * a.ngroups + min(g * 256 + tid, a.ngroups - 1)];
    }
    // query row -> LDS, its norm
    float nrm = 0.f;
    for (int k = tid; k < d; k += 256) {
        const float v = qg[k];
        qs[k] = v;
        nrm = __builtin_fmaf(v, v, nrm);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor(nrm, off);
    if (lane == 0) wred[w] = nrm;
    if (tid == 0) { c_cnt = 0; g_cnt = 0; }
    __syncthreads();
    const float qn = sqrtf(wred[0] + wred[1] + wred[2] + wred[3]) * 1.001f;   // an upper bound of |q| (f32 sum: d 2^-25 relative)
    float margin = 2.f * a.eps_rel * qn * a.cmax;
    const float inv = 1.f / (a.qscale[row] * a.cscale);   // exact: powers of two
    bool bad = !(margin < 3.0e38f);   // NaN / inf query
    // The same chain with the row LOADED by the four lanes of a quad: lane j of the quad requests the 16-byte pieces
    // 4 i + j of the row, so a quad reads 64 contiguous bytes per instruction (a lane on its own row: 16 B of a line per
    // instruction, eight instructions per line -- the one-lane chain was bound by its ~0.5 us load round trips, 128 B
    // in flight per row and set) and four times the bytes are in flight per row; every lane of the quad then runs the
    // whole chain on the pieces broadcast inside the quad (DPP quad_perm), k ascending: the same value in all four.
    auto exact_quad = [&](int col) -> float {
        const int j = lane & 3;
        const float4 *cp = reinterpret_cast<const float4 *>(a.cent + (size_t)col * d) + j;
        const int nset = d >> 7;   // 128 floats per set (d % 128 == 0)
        float4 A[8], B[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = cp[4 * i];
        float acc = 0.f;
        // the query's values for two pieces (8 x 16 B) are read from LDS together, ahead of the 32 fmacs that use them: read where
        // they are used, every ds_read_b128 was followed by s_waitcnt lgkmcnt(0) -- an LDS latency per four fmacs, ~40 k cycles a chain
#define MI_FMAC4_QUAD(QP, CX, CY, CZ, CW, QV)                                                                                      \
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %5 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                       \
        "v_fmac_f32_dpp %0, %2, %6 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                   \
        "v_fmac_f32_dpp %0, %3, %7 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                   \
        "v_fmac_f32_dpp %0, %4, %8 quad_perm:" QP " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                        \
        : "+v"(acc) : "v"(CX), "v"(CY), "v"(CZ), "v"(CW), "v"(QV.x), "v"(QV.y), "v"(QV.z), "v"(QV.w))
        // acc = fma(c of quad lane jj, q, acc), four times a statement: v_fmac_f32 with a DPP source (hipcc keeps a v_mov_b32_dpp
        // in front of every v_fma_f32: twice the VALU time of the chain).  The compiler does not see a DPP read in the asm:
        // s_nop 1 covers a VALU write of the first piece right in front (2 wait states).
        const float4 *q4 = reinterpret_cast<const float4 *>(qs);
        auto run = [&](const float4 (&S)[8], int set) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float4 qv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) qv[u] = q4[32 * set + 8 * h + u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = 2 * h + (u >> 2);
                    switch (u & 3) {
                    case 0: MI_FMAC4_QUAD("[0,0,0,0]", S[i].x, S[i].y, S[i].z, S[i].w, qv[u]); break;
                    case 1: MI_FMAC4_QUAD("[1,1,1,1]", S[i].x, S[i].y, S[i].z, S[i].w, qv[u]); break;
                    case 2: MI_FMAC4_QUAD("[2,2,2,2]", S[i].x, S[i].y, S[i].z, S[i].w, qv[u]); break;
                    default: MI_FMAC4_QUAD("[3,3,3,3]", S[i].x, S[i].y, S[i].z, S[i].w, qv[u]); break;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#undef MI_FMAC4_QUAD
        for (int set = 0; set < nset; set += 2) {
            const int nb = min(set + 1, nset - 1);   // (an odd number of sets: a harmless re-read, not used)
#pragma unroll
            for (int i = 0; i < 8; ++i) B[i] = cp[32 * nb + 4 * i];
            run(A, set);
            if (set + 1 >= nset) break;
            const int nx = min(set + 2, nset - 1);   // last round: a harmless re-read
#pragma unroll
            for (int i = 0; i < 8; ++i) A[i] = cp[32 * nx + 4 * i];
            run(B, set + 1);
        }
        return acc;
    };
    constexpr int VPT = 16, TILE = 256 * VPT;
    int ph = 0;
    auto block_sum = [&](int wave_total) -> int {
        if (lane == 0) wcnt[ph][w] = wave_total;
        __syncthreads();
        const int c = wcnt[ph][0] + wcnt[ph][1] + wcnt[ph][2] + wcnt[ph][3];
        ph ^= 1;
        return c;
    };
    bool exact_row = false;
    int Sn = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        unsigned key[VPT];
        unsigned gm[VPT];
        int nonfinite = 0;
        // group maxima from the GEMM epilogue: enough groups that the K-th largest of them is a useful cut
        const bool ext = ext0 && !exact_row;
        unsigned gq[4] = {0u, 0u, 0u, 0u};
        if (ext) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gi = g * 256 + tid;
                const float v = graw[g] * inv;
                const bool fin = fabsf(v) <= 3.0e38f;
                nonfinite |= (gi < a.ngroups && !fin);
                gq[g] = (gi < a.ngroups && fin) ? f2o(v) : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < VPT; ++j) gm[j] = 0u;
        if (!ext)
        for (int base = 0; base < n; base += TILE) {
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int c = base + j * 256 + tid;
                const float v = r[min(c, n - 1)] * (exact_row ? 1.f : inv);
                // approximate rows must be finite (else: exact fallback); exact rows follow
                // select_kernel: NaN never survives, +-inf are ordinary scores
                const bool fin = exact_row ? (v == v) : (fabsf(v) <= 3.0e38f);
                nonfinite |= (c < n && !fin);
                key[j] = (c < n && fin) ? f2o(v) : 0u;
                gm[j] = max(gm[j], key[j]);
            }
        }
        if (!exact_row && block_sum(__popcll(__ballot(nonfinite != 0))) > 0) bad = true;
        stamp(1);
        unsigned T0 = 0;
        if (K <= 256) {
            // barrier-free, as in select_kernel: 1024 maxima (4 per thread) in LDS, every wave
            // descends over all of them (K <= 16: over their 64 column maxima -- any K
            // distinct entries bound the K-th largest from below)
            unsigned *gk = reinterpret_cast<unsigned *>(skey);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gk[g * 256 + tid] = ext ? gq[g] : max(max(gm[g], gm[g + 4]), max(gm[g + 8], gm[g + 12]));
            __syncthreads();
            unsigned kk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) kk[i] = gk[i * 64 + ((lane + 4 * i) & 63)];
            if (K <= 16) {
                unsigned f = 0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) f = max(f, kk[i]);
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned t = T0 | (1u << bit);
                    const int c = __popcll(__ballot(f >= t));
                    if (c >= K) T0 = t;
                    if (c == K) break;
                }
            } else if (w == 0) {
                // ONE wave descends (16 ballots a step: with every wave of the CU's four workgroups at it the steps queued at
                // the CU's scalar unit -- 28-38 k cycles of the workgroup), from below the common prefix of the largest and the
                // smallest maximum (a query's scores share sign, exponent and a few mantissa bits: a third of the steps;
                // with at least K maxima set, every one bit of the prefix is decided, a zero bit of it can never be set)
                unsigned mx = 0u, mn = ~0u;
                int cn = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    mx = max(mx, kk[i]);
                    mn = min(mn, kk[i] != 0u ? kk[i] : ~0u);
                    cn += __popcll(__ballot(kk[i] != 0u));
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
                    mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
                }
                if (cn >= K) {
                    const unsigned diff = mx ^ mn;
                    int bit = diff ? 31 - __clz((int)diff) : -1;
                    T0 = diff ? (mx & ~((2u << bit) - 1u)) : mx;
                    if (cn > K)
                        for (; bit >= 0; --bit) {
                            const unsigned t = T0 | (1u << bit);
                            int c = 0;
#pragma unroll
                            for (int i = 0; i < 16; ++i) c += __popcll(__ballot(kk[i] >= t));
                            if (c >= K) T0 = t;
                            if (c == K) break;
                        }
                }
                if (lane == 0) t0_sh = T0;
            }
            __syncthreads();   // skey is reused for the candidates
            if (K > 16) T0 = t0_sh;
        } else {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned t = T0 | (1u << bit);
                int c = 0;
#pragma unroll
                for (int j = 0; j < VPT; ++j) c += __popcll(__ballot(gm[j] >= t));
                c = block_sum(c);
                if (c >= K) T0 = t;
                if (c == K) break;
            }
        }
        stamp(2);
        // cut in the score domain; T0 == 0: fewer than K finite entries, keep all
        const float cut = T0 ? o2f(T0) - (exact_row ? 0.f : margin) : -__builtin_inff();
        if (ext) {
            if (!bad) {
                // the groups that can hold a candidate (a few dozen of the 1024), then one wave per group: 256-byte reads
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (gq[g] != 0u && o2f(gq[g]) >= cut) glist[atomicAdd(&g_cnt, 1)] = (unsigned short)(g * 256 + tid);
                __syncthreads();
                stamp(3);
                // 16 bytes of a group per lane (4 scores; 16 groups a pass), eight passes' loads issued together: a wave per group
                // and pass was a dependent round trip per ~100 surviving groups / 4 waves
                const int ng = g_cnt;
                constexpr int lpg = 16, epl = 4, gpp = 256 / lpg, NPASS = 8;
                const int gsl = tid / lpg, part = tid % lpg;
                for (int i0 = 0; i0 < ng; i0 += NPASS * gpp) {
                    float4 raw[NPASS];
                    int c0[NPASS];
#pragma unroll
                    for (int u = 0; u < NPASS; ++u) {
                        const int gi = i0 + u * gpp + gsl;
                        c0[u] = gi < ng ? (int)glist[gi] * 64 + part * epl : -1;
                        if (gi < ng) raw[u] = *reinterpret_cast<const float4 *>(r + c0[u]);
                    }
#pragma unroll
                    for (int u = 0; u < NPASS; ++u) {
                        if (c0[u] < 0) continue;
                        const float wd[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
                        for (int e = 0; e < epl; ++e) {
                            const float v = wd[e] * inv;
                            if (v >= cut) {
                                const int pos = atomicAdd(&c_cnt, 1);
                                if (pos < SELB_CAP) skey[pos] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(c0[u] + e);
                            }
                        }
                    }
                }
            }
        } else if ((!bad || exact_row) && n > TILE) {
            // long rows: only the thread groups whose maximum reaches the cut hold candidates
            // (a few dozen of the 4096), so the second pass touches a few cache lines instead
            // of re-reading the row
            const float sc = exact_row ? 1.f : inv;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                if (gm[j] != 0u && o2f(gm[j]) >= cut) {
                    for (int c0 = j * 256 + tid; c0 < n; c0 += 16 * TILE) {   // 16 loads in flight
                        float vv[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) vv[i] = r[min(c0 + i * TILE, n - 1)];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int c = c0 + i * TILE;
                            const float v = vv[i] * sc;
                            if (c < n && v >= cut) {
                                const int pos = atomicAdd(&c_cnt, 1);
                                if (pos < SELB_CAP) skey[pos] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)c;
                            }
                        }
                    }
                }
            }
        } else if (!bad || exact_row) {
            for (int base = 0; base < n; base += TILE) {
                unsigned long long m[VPT];
                float sv[VPT];
                int tot = 0;
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const int c = base + j * 256 + tid;
                    sv[j] = r[min(c, n - 1)] * (exact_row ? 1.f : inv);
                    m[j] = __ballot(c < n && sv[j] >= cut);   // false for NaN
                    tot += __popcll(m[j]);
                }
                if (tot) {
                    int o = 0;
                    if (lane == 0) o = atomicAdd(&c_cnt, tot);
                    o = uniform_i(o);
#pragma unroll
                    for (int j = 0; j < VPT; ++j) {
                        if (m[j]) {
                            const int pos = o + lane_prefix_count(m[j]);
                            if (((m[j] >> lane) & 1ull) && pos < SELB_CAP)
                                skey[pos] = ((unsigned long long)__float_as_uint(sv[j]) << 32) |
                                            (unsigned)(base + j * 256 + tid);   // (score bits, column) for now
                            o += __popcll(m[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();
        stamp(4);
        Sn = c_cnt;
        if ((bad && !exact_row) || Sn > SELB_CAP) {
            if (exact_row) break;   // still too many after the exact pass: masses of exact ties
            // exact scores for the whole row, in place
            refine_exact_row(r, a.cent, qs, n, d);
            __threadfence_block();
            if (tid == 0) c_cnt = 0;
            if (a.stats && tid == 0) atomicAdd(a.stats + 1, 1u);
            __syncthreads();
            exact_row = true;
            continue;
        }
        break;
    }
    if (Sn > SELB_CAP) {
        __syncthreads();
        float *f = reinterpret_cast<float *>(skey);
        select_by_insertion(r, n, K, row, a.out_i32, nullptr, a.out_s, a.idx_off, f, reinterpret_cast<int *>(f + 256),
                            f + 512, reinterpret_cast<int *>(f + 768));
        if (a.pt.list_goff) emit_probe_tables(a.pt, row, K, a.out_i32 + (size_t)row * K, wtot);
        return;
    }
    if (a.stats && tid == 0) atomicAdd(a.stats, (unsigned)Sn);
    // exact score of every candidate, then the sort key
    __syncthreads();
    // candidate e goes to the quad e / 4 % 16 of wave e % 4: the chains spread over the four SIMDs, 64 candidates a round
    if (exact_row) {   // (scores already exact)
        for (int e0 = 0; e0 < Sn; e0 += 256) {
            const int e = e0 + (tid >> 6) + 4 * (tid & 63);
            if (e >= Sn) continue;
            const unsigned col = (unsigned)skey[e];
            const float s = __uint_as_float((unsigned)(skey[e] >> 32));
            skey[e] = s == s ? ((unsigned long long)f2o(s) << 32) | (unsigned)~col : 0ull;   // NaN never survives
        }
    } else {
        for (int e0 = 0; e0 < Sn; e0 += 64) {
            const int e = e0 + w + 4 * (lane >> 2);
            if (e0 + w + 4 * (lane >> 2) >= Sn) continue;
            const unsigned col = (unsigned)skey[e];
            const float s = exact_quad((int)col);
            if ((lane & 3) == 0) skey[e] = s == s ? ((unsigned long long)f2o(s) << 32) | (unsigned)~col : 0ull;
        }
    }
    stamp(5);
    if (Sn <= 256) {
        // up to a key per thread (the assignment step of add(): two or three candidates; nprobe 8: a dozen; nprobe 64: ~95): every
        // thread counts the keys above its own (broadcast reads; the keys carry their column: distinct but for NaN scores' zeros) and
        // stores it at its rank -- where the bitonic network of 64 / 128 slots is 21 / 28 block-wide barriers (9-12 k cycles)
        __syncthreads();
        const unsigned long long mine = tid < Sn ? skey[tid] : 0ull;
        int rk = 0;
        if (tid < Sn)
            for (int j = 0; j < Sn; ++j) {
                const unsigned long long kj = skey[j];
                rk += kj > mine || (kj == mine && j < tid);   // (equal keys: only the zeros of NaN scores -- slot order keeps the ranks distinct)
            }
        __syncthreads();   // every read in front of the writes
        if (tid < Sn) skey[rk] = mine;
        __syncthreads();
    } else {
    int P = 64;
    while (P < Sn) P <<= 1;
    for (int e = Sn + tid; e < P; e += 256) skey[e] = 0ull;
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const unsigned long long x = skey[lo], y = skey[hi];
                const bool desc = (lo & k2) == 0;
                if ((x < y) == desc) {
                    skey[lo] = y;
                    skey[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    stamp(6);
    constexpr int RPT = SELB_CAP / 256;
    unsigned long long res[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int e = tid + i * 256;
        res[i] = (e < K && e < Sn) ? skey[e] : 0ull;
    }
    __syncthreads();
    int32_t *sel = reinterpret_cast<int32_t *>(skey);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int e = tid + i * 256;
        if (e < K) {
            const bool filled = res[i] != 0ull;
            const int si = filled ? (int)~(unsigned)res[i] + a.idx_off : -1;
            const size_t o = (size_t)row * K + e;
            if (a.out_i32) a.out_i32[o] = si;
            if (a.out_s) a.out_s[o] = filled ? o2f((unsigned)(res[i] >> 32)) : -FLT_MAX;
            sel[e] = si;
        }
    }
    __syncthreads();
    if (a.pt.list_goff) emit_probe_tables(a.pt, row, K, sel, wtot);
    stamp(7);
}

template <int DSUB>
__global__ void __launch_bounds__(256) lut_kernel(LutArgs a) {
    lut_block<DSUB>(a, blockIdx.x);
}

// ---------------------------------------------------------------------
// PQ-code scan + per-slice top-k.
//
// One 512-thread workgroup per (query, slice); workgroup b runs on XCD b % 8 and the
// slices of one query are 8 workgroups apart (one XCD: the query's LUT is fetched
// from HBM once).  The slice's work items are the 64-code groups of the query's
// probed lists (concatenated, dealt round-robin to the 8 waves); lane j of a wave
// owns code j of the group: NCH coalesced 16-byte loads (+ one 8-byte id load), then
// M look-ups LUT[m][code[m]] from LDS added in ascending m (the oracle's order).
//
// Prologue: per-wave probe tables by LDS-DMA (nprobe <= 64: kept in registers), LUT
// rows by LDS-DMA behind them, the first two code groups of every wave requested
// before the one barrier that covers LUT and codes.
//
// Selection is threshold-filtered and latency-free on the common path:
//   - a score below the running threshold is dropped by one compare + ballot;
//   - survivors are appended to a 128-entry per-wave LDS buffer;
//   - when the buffer passes 64 entries and the wave has another group coming, it
//     is compressed to the wave's exact top-k by a bitwise descent on
//     order-preserving keys (ballot + popcount only: no shuffles, no serial
//     insertion), which also tightens the threshold; the threshold is shared across
//     the workgroup's waves through one LDS word (atomicMax);
//   - exact score ties at the cut are resolved by ascending id.
// Tail: two candidates per lane in registers -> block-wide lower bound of the k-th
// best from the folded per-thread maxima -> ~k survivors -> ranked by one wave and
// published write-through; the last slice of the query to take its ticket merges
// the nslice partial lists (one wave, registers) and writes D / I.
// ---------------------------------------------------------------------
struct ScanArgs {
    const float *lut;          // [nq][M*256]
    const float *coarse_dis;   // [nq][nprobe]
    const int32_t *p_goff;     // [nq][nprobe]   first group of each probed list
    const int32_t *p_len;      // [nq][nprobe]   its length in codes
    const int32_t *p_prefix;   // [nq][nprobe+1] exclusive prefix of group counts
    const uint8_t *codes;      // group-interleaved
    const int64_t *ids;        // [ngroups*64]
    const float *tnorm;        // [ngroups*64], METRIC_L2 only (scan_kernel<..., L2 = true>)
    float *part_s;             // [nq][nslice][k]
    int64_t *part_id;          // [nq][nslice][k]
    const float *bound_s;      // [nq] or null: only entries strictly after
    const int64_t *bound_id;   //      (bound_s, bound_id) are eligible
    int nq, nprobe, nslice, k, by_residual;
    int nw;     // waves per workgroup of this launch (8 or 16): host-side dispatch only
    int debug;  // ablation switches for tools/scan_ablate.py (0 in production)
    unsigned long long *ts;    // null, or [workgroups][SCAN_TS] s_memtime stamps (MI_SCAN_TS=1 profile replay)
    // fused final merge (the last slice of a query to finish merges all of the
    // query's partial lists; counters must be zero on entry and are left zero)
    unsigned *counters;        // [nq] or null = separate merge kernel
    float *D;                  // [nq][ldo]
    int64_t *I;                // [nq][ldo]
    float *next_bound_s;       // [nq] or null: last kept entry (next extraction pass)
    int64_t *next_bound_id;
    int64_t ldo;
    int out_off;
    // all-scores mode (scan_kernel<M, NW, true>; k > 64): no selection, every (score, id) of
    // the probed lists is stored at [q][group_in_query * 64 + lane] (score NaN in the padding
    // lanes of a list's last group) for select_pairs_kernel
    float *all_s;              // [nq][all_ld]
    int64_t *all_id;           // [nq][all_ld]
    int64_t all_ld;
    // exact early stop (top-k mode, nprobe <= 64, lists in descending coarse order): prune_A[q] = the chain of the query's
    // table-row maxima (lut_maxsum_kernel), so that lane p's U = dis0[p] + A bounds every code of probe p; a wave stops at the
    // first list whose U is below a threshold it already holds (see "Exact list pruning" above prune_tables_kernel)
    const float *prune_A;      // [nq] or null
    const int32_t *prune_sorted;   // null, or [nq]: 0 = this query's lists are NOT in descending coarse order (caller-assigned lists):
                                   // no early stop for it
    unsigned long long *prune_stats;   // null, or [PRUNE_SLOTS][3] += {groups processed, groups of all probes, queries}; one atomic
                                       // triple per WORKGROUP, spread over the slots (one per wave on one address: 10 k serialised
                                       // atomics, 50 us of a 245 us launch)
};
constexpr int PRUNE_SLOTS = 64;

// LDS bytes the fused final merge needs inside the LUT region
__host__ __device__ inline size_t scan_fused_merge_bytes(int nslice, int k) {
    return (size_t)nslice * k * 16 + 1024;
}

// waves per workgroup: template parameter NW of the kernel (8, or 16 for short slices)
constexpr int SCAN_TS = 24;       // phase stamps per workgroup (profiling replay only)
constexpr int SCAN_WBUF = 128;    // candidate slots per wave

// LDS carve: [ LUT M KiB (>= 2 KiB per wave, reused by the selection tail) | wave buffers
//              8 x 128 x (8+4) B | prefix | p_goff | p_len | p_dis | misc 16 B ]
__host__ __device__ inline size_t scan_lut_bytes(int M, int nw) {
    size_t b = (size_t)M * 1024, floor_b = (size_t)nw * 3072;   // selection tail: 3 x 64 nw entries x 16 B
    return b < floor_b ? floor_b : b;
}
// grid of the scan launch: the slices of a query share an XCD (see the kernel)
__host__ __device__ inline unsigned scan_grid(int nq, int nslice) {
    return 8u * (unsigned)((nq + 7) / 8) * (unsigned)nslice;
}
__host__ __device__ inline int scan_tab_stride(int nprobe) { return (nprobe + 1 + 3) & ~3; }
__host__ __device__ inline size_t scan_smem_bytes(int M, int nprobe, int nw) {
    return scan_lut_bytes(M, nw) + (size_t)nw * SCAN_WBUF * 12 + (size_t)scan_tab_stride(nprobe) * 4 * 4 + 16;
}

// (A lower bound of) the k-th largest of the wave's order-preserving keys, N per lane
// (0 = empty): exactly k keys are >= the result unless keys tie; 0 when fewer than k
// keys are set.  Bitwise descent with early exit -- as soon as exactly k keys are >= t,
// t itself separates the top k (typical scores: about a third of the 32 steps).
// (Two bits per step with three thresholds counted side by side measured slower.)
template <int N>
__device__ __forceinline__ unsigned wave_kth_largest_n(const unsigned (&key)[N], int k) {
    int nz = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) nz += __popcll(__ballot(key[i] != 0u));
    if (nz < k) return 0u;
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned t = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) c += __popcll(__ballot(key[i] >= t));
        if (c >= k) prefix = t;
        if (c == k) break;
    }
    return prefix;
}
__device__ __forceinline__ unsigned wave_kth_largest(unsigned ka, unsigned kb, int k) {
    const unsigned kk[2] = {ka, kb};
    return wave_kth_largest_n<2>(kk, k);
}

// Rank (number of better entries) of this lane's entry among the wave's first n
// (<= 64) lanes under (score desc, id asc, slot asc); empty entries and lanes >= n carry
// key 0 and count for nobody.  The keys go through a 64-word wave-private LDS scratch
// and every lane reads them all back four at a time (same address in all lanes: a
// broadcast, pipelined reads) -- a readlane loop costs a VALU -> SGPR -> VALU round
// trip (~50 cycles) per entry.  The full predicate only runs when two scores tie.
__device__ __forceinline__ int wave_rank(unsigned key, int64_t id, int n, int lane, unsigned *scratch) {
    scratch[lane] = key;
    int gt = 0, ge = 0;
    for (int j0 = 0; j0 < n; j0 += 4) {   // whole blocks of 4: j0 + 3 <= 63
        const uint4 kq = *reinterpret_cast<const uint4 *>(scratch + j0);
        gt += (kq.x > key) + (kq.y > key) + (kq.z > key) + (kq.w > key);
        ge += (kq.x >= key) + (kq.y >= key) + (kq.z >= key) + (kq.w >= key);
    }
    if (__ballot(key != 0u && ge - gt > 1)) {   // some score is held by two entries
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const unsigned jk = readlane_u(key, j);
            const int64_t jid = readlane_i64(id, j);
            r += (jk > key) || (jk == key && (jid < id || (jid == id && j < lane)));
        }
        return r;
    }
    return gt;
}

// Reduce a wave's candidate buffer (cnt <= 128 entries) to its exact top-k
// under (score desc, id asc); returns the new count and the k-th score.
// (ids: null, or the lists' id array when the buffer holds POSITIONS in it instead of ids -- scan_kernel's lazy mode: only the
// rare tie path below compares ids, and fetches them then)
__device__ __forceinline__ void wave_compress(float *buf_s, int64_t *buf_id, int lane, int k, int &cnt,
                                              float &thr, const int64_t *ids = nullptr) {
    const bool va = lane < cnt, vb = lane + 64 < cnt;
    const float sa = va ? buf_s[lane] : 0.f, sb = vb ? buf_s[lane + 64] : 0.f;
    const int64_t ia = va ? buf_id[lane] : 0, ib = vb ? buf_id[lane + 64] : 0;
    const unsigned ka = va ? f2o(sa) : 0u, kb = vb ? f2o(sb) : 0u;
    const unsigned T = wave_kth_largest(ka, kb, k);
    const bool ga = ka > T, gb = kb > T;
    const bool ea = (ka == T) && (T != 0u), eb = (kb == T) && (T != 0u);
    const int need = k - (__popcll(__ballot(ga)) + __popcll(__ballot(gb)));
    const int ties = __popcll(__ballot(ea)) + __popcll(__ballot(eb));
    bool keep_a = ga || ea, keep_b = gb || eb;
    if (ties > need) {  // rare: keep the `need` smallest ids among the tied entries
        const int64_t ra = (ids && ea) ? ids[ia] : ia, rb = (ids && eb) ? ids[ib] : ib;
        const unsigned long long ua = (unsigned long long)ra ^ (1ull << 63), ub = (unsigned long long)rb ^ (1ull << 63);
        unsigned long long pref = 0;
        for (int bit = 63; bit >= 0; --bit) {
            const unsigned long long t = pref | (1ull << bit);
            const int c = __popcll(__ballot(ea && ua < t)) + __popcll(__ballot(eb && ub < t));
            if (c < need) pref = t;
        }
        keep_a = ga || (ea && ua <= pref);
        keep_b = gb || (eb && ub <= pref);
    }
    const unsigned long long ma = __ballot(keep_a), mb = __ballot(keep_b);
    const int na = __popcll(ma);
    if (keep_a) {
        const int pa = lane_prefix_count(ma);
        buf_s[pa] = sa;
        buf_id[pa] = ia;
    }
    if (keep_b) {
        const int pb = na + lane_prefix_count(mb);
        buf_s[pb] = sb;
        buf_id[pb] = ib;
    }
    cnt = na + __popcll(mb);
    thr = (cnt >= k && T != 0u) ? o2f(T) : MI_NEG_INF;
}

template <int M, int NW, bool ALL = false, bool L2 = false>
__global__ void __launch_bounds__(NW * 64, 4) scan_kernel(ScanArgs a) {   // 4 waves per SIMD = two 512-thread workgroups per CU
    constexpr int NCH = (M + 15) / 16;
    constexpr int SCAN_NW = NW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lut_s = reinterpret_cast<float *>(smem);
    unsigned char *wb = smem + scan_lut_bytes(M, NW);
    const int TS = scan_tab_stride(a.nprobe);
    int *prefix = reinterpret_cast<int *>(wb + (size_t)SCAN_NW * SCAN_WBUF * 12);
    int *p_goff = prefix + TS;
    int *p_len = p_goff + TS;
    float *p_dis = reinterpret_cast<float *>(p_len + TS);
    unsigned *wg_thr = reinterpret_cast<unsigned *>(p_dis + TS);
    int *c_total = reinterpret_cast<int *>(wg_thr + 1);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = uniform_i(tid >> 6);
    // Workgroup b runs on XCD b % 8: the nslice slices of one query are placed on ONE
    // XCD (grid = 8 x ceil(nq/8) x nslice), so the query's LUT is fetched from HBM
    // once and re-staged from that XCD's L2 by the other slices.
    const int q = ((int)(blockIdx.x >> 3) / a.nslice) * 8 + (int)(blockIdx.x & 7);
    const int slice = (int)(blockIdx.x >> 3) % a.nslice;
    if (q >= a.nq) return;
    const int nprobe = a.nprobe, k = a.k;
    auto stamp = [&](int i) {   // wave 0 only; no-op unless a profiling replay asked for stamps
        if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * SCAN_TS + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    int64_t *buf_id = reinterpret_cast<int64_t *>(wb) + w * SCAN_WBUF;                          // [8][128]
    float *buf_s = reinterpret_cast<float *>(wb + (size_t)SCAN_NW * SCAN_WBUF * 8) + w * SCAN_WBUF;  // [8][128]

    // ---- probe tables.  nprobe <= 64: every wave keeps its own copy in registers
    // (lane p = probe p), so locating a group is one ballot + readlanes and the first
    // code groups are requested while the LUT is still in flight.  The copy travels
    // by LDS-DMA into the wave's (still unused) candidate buffer, issued BEFORE the LUT
    // rows: a counted wait then covers the tables but not the LUT.
    // Larger nprobe: tables staged in LDS by the whole workgroup.
    const bool reg_tab = nprobe <= 64;
    constexpr int LUT_ROWS = (M + SCAN_NW - 1) / SCAN_NW;   // LUT rows staged per wave
    if (reg_tab) {
        const int pl = min(lane, nprobe - 1);
        const size_t o = (size_t)q * nprobe + pl;
        unsigned char *wt = reinterpret_cast<unsigned char *>(buf_id);
        dma4_lds(a.p_prefix + (size_t)q * (nprobe + 1) + pl, wt);
        dma4_lds(a.p_goff + o, wt + 256);
        dma4_lds(a.p_len + o, wt + 512);
        dma4_lds(a.by_residual ? a.coarse_dis + o : reinterpret_cast<const float *>(a.p_len + o), wt + 768);
        dma4_lds(a.p_prefix + (size_t)q * (nprobe + 1) + pl + 1, buf_s);
    }
    // ---- stage the query's LUT: row m (1 KiB) by one global_load_lds_dwordx4
    // per wave (LDS-DMA: wave-uniform LDS base + lane*16, no VGPR round trip)
    if (!(a.debug & 8)) {
        const float *lg = a.lut + (size_t)q * M * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < LUT_ROWS; ++i) {
            const int m = w + i * SCAN_NW;
            if (M % SCAN_NW == 0 || m < M) dma16_lds(lg + m * 256, lut_s + m * 256);
        }
    }
    int r_pre0 = 0, r_pre1 = INT_MAX, r_goff = 0, r_len = 0;
    float r_dis = 0.f;
    if (reg_tab) {
        // tables landed once at most the (later issued) LUT rows are outstanding
        if (a.debug & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(M / SCAN_NW) : "memory");
        const int *wt = reinterpret_cast<const int *>(buf_id);
        r_pre0 = wt[lane];
        r_goff = wt[64 + lane];
        r_len = wt[128 + lane];
        if (a.by_residual) r_dis = __int_as_float(wt[192 + lane]);
        r_pre1 = reinterpret_cast<const int *>(buf_s)[lane];
    }
    if (!reg_tab) {
        for (int p = tid; p < nprobe; p += NT) {
            const size_t o = (size_t)q * nprobe + p;
            prefix[p] = a.p_prefix[(size_t)q * (nprobe + 1) + p];
            p_goff[p] = a.p_goff[o];
            p_len[p] = a.p_len[o];
            p_dis[p] = a.by_residual ? a.coarse_dis[o] : 0.0f;
        }
        if (tid == 0) prefix[nprobe] = a.p_prefix[(size_t)q * (nprobe + 1) + nprobe];
    }
    if (tid == 0) {
        *wg_thr = f2o(MI_NEG_INF);
        *c_total = 0;
        c_total[1] = 0;    // groups this workgroup processed (prune_stats)
    }
    if (!reg_tab) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // LDS tables visible (also drains the LUT DMA)
    }
    // exact early stop: lane p's upper bound of every score of probe p (+inf: never below a threshold)
    const bool early_stop = !ALL && !L2 && reg_tab && a.prune_A != nullptr && a.by_residual && (!a.prune_sorted || a.prune_sorted[q] != 0);
    float r_U = __builtin_inff();
    if (early_stop) r_U = r_dis + a.prune_A[q];
    // 64 < nprobe <= 256: the group -> probe search still runs on registers (lane p keeps
    // prefix[p+1], [p+65], [p+129], [p+193]: four ballots), only the four table values of
    // the located probe come from LDS, as independent reads.  The sequential walk with one
    // dependent LDS round trip per step is left for nprobe > 256.
    const bool wide_tab = !reg_tab && nprobe <= 256;
    int w_pre1[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
    if (wide_tab) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < nprobe) w_pre1[i] = prefix[lane + 64 * i + 1];
    }

    const int G = reg_tab ? __builtin_amdgcn_readlane(r_pre1, nprobe - 1) : prefix[nprobe];
    const int beg = (int)(((int64_t)G * slice) / a.nslice);
    const int end = (int)(((int64_t)G * (slice + 1)) / a.nslice);

    const bool has_bound = a.bound_s != nullptr;
    const bool lazy_ids = !ALL && !has_bound && !(a.debug & 16);   // (debug 16: the eager id loads, for A/B runs)
    float bs = 0.f;
    int64_t bid = 0;
    if (has_bound) {
        bs = a.bound_s[q];
        bid = a.bound_id[q];
    }

    int cnt = 0;
    float thr = MI_NEG_INF;
    float wthr_seen = MI_NEG_INF;   // the workgroup's threshold as this wave last read it
    int my_end = end;               // early stop: the first group of a list that cannot hold a result (the lists' bounds descend)

    // work items = 64-code groups [beg, end), dealt round-robin to the waves.  (Taking them from a workgroup counter instead --
    // the waves of a slice finish up to 60 k cycles apart and the tail's barrier waits for the slowest -- removes that wait and
    // lengthens the main loop by as much: the workgroup is bound by what it streams, not by its slowest wave.  Measured in
    // round 6, not kept: profiles/r06_scan_lazy_ids_ab.txt.)
    int t = beg + w;
    int p = 0;
    if (!reg_tab && !wide_tab && t < end) {  // smallest p with prefix[p+1] > t
        int lo = 0, hi = nprobe - 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (uniform_i(prefix[mid + 1]) > t) hi = mid;
            else lo = mid + 1;
        }
        p = lo;
    }
    struct Group {
        uint4 c[NCH];
        int64_t id;
        int nvalid;
        float dis0;
        float t;      // METRIC_L2 only: the vector's term |r^|^2 + 2<c, r^>
    };
    auto locate_and_load = [&](int tt, Group &g) {
        int gi, gg;
        if (reg_tab) {
            // probes whose group range ends at or before tt: prefix is non-decreasing
            const int pp = __popcll(__ballot(lane < nprobe && r_pre1 <= tt));
            if (early_stop && readlane_f(r_U, pp) < fmaxf(thr, wthr_seen)) {   // (wave-uniform) k scores above this list's bound are held already
                my_end = tt;
                return;
            }
            gi = tt - __builtin_amdgcn_readlane(r_pre0, pp);
            gg = __builtin_amdgcn_readlane(r_goff, pp) + gi;
            g.nvalid = min(64, __builtin_amdgcn_readlane(r_len, pp) - gi * 64);
            g.dis0 = readlane_f(r_dis, pp);
        } else if (wide_tab) {
            int pp = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) pp += __popcll(__ballot(w_pre1[i] <= tt));
            const int v0 = prefix[pp], v1 = p_goff[pp], v2 = p_len[pp];
            const float v3 = p_dis[pp];
            gi = tt - uniform_i(v0);
            gg = uniform_i(v1) + gi;
            g.nvalid = min(64, uniform_i(v2) - gi * 64);
            g.dis0 = uniform_f(v3);
        } else {
            while (uniform_i(prefix[p + 1]) <= tt) ++p;
            gi = tt - uniform_i(prefix[p]);
            gg = uniform_i(p_goff[p]) + gi;
            g.nvalid = min(64, uniform_i(p_len[p]) - gi * 64);
            g.dis0 = uniform_f(p_dis[p]);
        }
        const uint4 *gp = reinterpret_cast<const uint4 *>(a.codes + (size_t)gg * (NCH * 1024)) + lane;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) g.c[ch] = gp[ch * 64];
        // lazy ids (top-k mode without an extraction bound): a code's id is only ever COMPARED when two scores tie at a cut, and
        // only the <= 3 k survivors of the workgroup are published -- so the candidates carry their POSITION in the id array
        // (group x 64 + lane: no load) and the ids of the survivors are fetched in the selection tail.  The eager load was 8 of
        // the 72 bytes the scan moved per code.
        if (lazy_ids) g.id = (int64_t)gg * 64 + lane;
        else if (!ALL || a.all_id) g.id = a.ids[(size_t)gg * 64 + lane];
        if constexpr (L2) g.t = a.tnorm[(size_t)gg * 64 + lane];
    };
    // Two groups per wave are requested before the LUT barrier (most of a cfg2-sized
    // slice is then in flight while the LUT is staged) and the loop keeps two in flight,
    // rotating three register sets by name (a copy would wait for the newest load).
    // A fourth set measured slower: 162 VGPRs, one workgroup per CU instead of two.
    int n_proc = 0;
    bool last_ok = false;
    float last_s = 0.f;
    int64_t last_id = 0;
    auto process = [&](const Group &g, bool more) {
        // M table look-ups, m ascending, f32 adds in that order (= the oracle)
        float acc = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const unsigned wd[4] = {g.c[ch].x, g.c[ch].y, g.c[ch].z, g.c[ch].w};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = ch * 16 + j;
                if (m < M) {
                    unsigned byte = (wd[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                    acc += lut_s[m * 256 + byte];
                }
            }
        }
        // METRIC_L2: s = -|q - x^|^2 through the expansion dis0 + (2 <q, r^> - t)  (oracle: search_l2)
        float s;
        if constexpr (L2) s = g.dis0 + (2.0f * acc - g.t);
        else s = g.dis0 + acc;
        if constexpr (ALL) {
            const size_t o = (size_t)q * a.all_ld + (size_t)t * 64 + lane;   // t: this group's index in the query
            a.all_s[o] = lane < g.nvalid ? s : __builtin_nanf("");
            if (a.all_id) a.all_id[o] = g.id;                 // (null: the selection reads the survivors' ids from the lists)
            return;
        }
        if (n_proc == 0) stamp(13);
        if (n_proc == 1) stamp(17);
        if (n_proc == 2) stamp(18);

        const float wthr = o2f(__hip_atomic_load(wg_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        wthr_seen = wthr;
        const float thr_eff = fmaxf(thr, wthr);
        bool pf = (lane < g.nvalid) && (s >= thr_eff);
        if (has_bound) pf = pf && (s < bs || (s == bs && g.id > bid));
        if (a.debug & 1) pf = false;
        if (!more) {
            // the wave's last group never goes through the LDS buffer: its candidates stay in
            // registers as the third entry per lane of the selection tail (a slice of up to
            // three groups per wave -- the bench configuration -- never compresses)
            last_ok = pf;
            last_s = s;
            last_id = g.id;
        } else {
            const unsigned long long mask = __ballot(pf);
            if (mask) {
                if (cnt > 64) {  // make room: reduce the buffer to the wave's top k, tighten the threshold
                    wave_compress(buf_s, buf_id, lane, k, cnt, thr, lazy_ids ? a.ids : nullptr);
                    if (thr > wthr && lane == 0) atomicMax(wg_thr, f2o(thr));
                }
                if (pf) {
                    const int o = cnt + lane_prefix_count(mask);
                    buf_s[o] = s;
                    buf_id[o] = g.id;
                }
                cnt += __popcll(mask);
            }
        }
        if (n_proc == 0) stamp(3);
        if (n_proc == 1) stamp(16);
        if (n_proc == 2) stamp(19);
        ++n_proc;
    };
    Group g0{}, g1{};
    if (t < my_end) locate_and_load(t, g0);
    if (t + SCAN_NW < my_end) locate_and_load(t + SCAN_NW, g1);
    stamp(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own LUT rows (DMA) and the first code groups
    __syncthreads();  // every wave's LUT rows are in LDS
    stamp(2);
    // (my_end only ever moves to a group that was not requested yet: `more` below is decided when the next group was requested)
    while (t < my_end) {
        process(g0, t + SCAN_NW < my_end);
        if (t + 2 * SCAN_NW < my_end) locate_and_load(t + 2 * SCAN_NW, g0);
        t += SCAN_NW;
        if (t >= my_end) break;
        process(g1, t + SCAN_NW < my_end);
        if (t + 2 * SCAN_NW < my_end) locate_and_load(t + 2 * SCAN_NW, g1);
        t += SCAN_NW;
    }
    int *c_groups = c_total + 1;    // (a spare word of the carve's last 16 bytes)
    if (!ALL && a.prune_stats && lane == 0 && n_proc) atomicAdd(c_groups, n_proc);

    if constexpr (ALL) return;
    stamp(4);
    const bool va = lane < cnt, vb = lane + 64 < cnt;
    const float sa = va ? buf_s[lane] : 0.f, sb = vb ? buf_s[lane + 64] : 0.f;
    const int64_t ia = va ? buf_id[lane] : 0, ib = vb ? buf_id[lane + 64] : 0;
    const unsigned ka = va ? f2o(sa) : 0u, kb = vb ? f2o(sb) : 0u;
    const unsigned kc3 = last_ok ? f2o(last_s) : 0u;   // the last group, straight from registers
    unsigned *tmax = reinterpret_cast<unsigned *>(wb + (size_t)SCAN_NW * SCAN_WBUF * 8);  // over buf_s
    tmax[w * SCAN_WBUF + lane] = max(max(ka, kb), kc3);
    __syncthreads();  // every wave is done with the LUT: its LDS is reused below
    stamp(5);
    if (a.prune_stats && tid == 0) {
        unsigned long long *ps3 = a.prune_stats + (size_t)(blockIdx.x & (PRUNE_SLOTS - 1)) * 3;
        atomicAdd(ps3, (unsigned long long)*c_groups);
        if (slice == 0) {
            atomicAdd(ps3 + 1, (unsigned long long)G);
            atomicAdd(ps3 + 2, 1ull);
        }
    }
    // Any k distinct candidates bound the k-th largest from below, so for small k the
    // 512 thread maxima are first folded to 64 (k <= 16) or 128 (k <= 64) column maxima:
    // the descent then costs one or two ballots per step instead of eight.
    unsigned T0 = 0;
    {
        unsigned kk[SCAN_NW];
#pragma unroll
        for (int ww = 0; ww < SCAN_NW; ++ww)   // rotated: a compressed wave keeps its k entries in lanes 0..k-1
            kk[ww] = tmax[ww * SCAN_WBUF + ((lane - ww * (64 / SCAN_NW)) & 63)];
        if (k <= 16) {
            unsigned f[1] = {0u};
#pragma unroll
            for (int ww = 0; ww < SCAN_NW; ++ww) f[0] = max(f[0], kk[ww]);
            T0 = wave_kth_largest_n<1>(f, k);
        } else if (k <= 64) {
            unsigned f[2] = {0u, 0u};
#pragma unroll
            for (int ww = 0; ww < SCAN_NW; ++ww) f[ww & 1] = max(f[ww & 1], kk[ww]);
            T0 = wave_kth_largest_n<2>(f, k);
        } else {
            T0 = wave_kth_largest_n<SCAN_NW>(kk, k);
        }
    }
    constexpr int CAP = 3 * NT;                                  // three candidates per thread at most
    int64_t *g_id = reinterpret_cast<int64_t *>(lut_s);          // [CAP]
    float *g_s = lut_s + 2 * CAP;                                // [CAP]
    int *g_rank = reinterpret_cast<int *>(lut_s) + 3 * CAP;      // [CAP]
    int64_t *o_id = reinterpret_cast<int64_t *>(wb);             // [64]  (over buf_id, dead by now)
    float *o_s = reinterpret_cast<float *>(wb + 512);            // [64]
    {
        const bool pa = ka != 0u && ka >= T0, pb = kb != 0u && kb >= T0, pc = kc3 != 0u && kc3 >= T0;
        const unsigned long long ma = __ballot(pa), mb = __ballot(pb), mc = __ballot(pc);
        const int na = __popcll(ma), nb2 = __popcll(mb), tot = na + nb2 + __popcll(mc);
        if (tot) {
            int o = 0;
            if (lane == 0) o = atomicAdd(c_total, tot);
            o = uniform_i(o);
            // (lazy ids: the survivors' ids now, all of a lane's loads in flight together)
            int64_t ra = ia, rb = ib, rc = last_id;
            if (lazy_ids) {
                if (pa) ra = a.ids[ia];
                if (pb) rb = a.ids[ib];
                if (pc) rc = a.ids[last_id];
            }
            if (pa) {
                const int x = o + lane_prefix_count(ma);
                g_s[x] = sa;
                g_id[x] = ra;
            }
            if (pb) {
                const int x = o + na + lane_prefix_count(mb);
                g_s[x] = sb;
                g_id[x] = rb;
            }
            if (pc) {
                const int x = o + na + nb2 + lane_prefix_count(mc);
                g_s[x] = last_s;
                g_id[x] = rc;
            }
        }
    }
    stamp(6);
    __syncthreads();
    stamp(7);
    const int C = uniform_i(*c_total);  // <= CAP
    if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * SCAN_TS + 14] = (unsigned long long)C + 1;
    const size_t part_o = ((size_t)q * a.nslice + slice) * k;
    // The slice's partial list is published WRITE-THROUGH (relaxed agent-scope
    // atomic stores lower to `global_store ... sc1`), so the fused final merge
    // needs no release / acquire fence: the last arriver reads the lists back with
    // sc1 loads that bypass its L1 (guide section 6 G16, "R1" form; correct for any
    // placement of the slices on CUs / XCDs).
    auto publish = [&](int pos, float ps, int64_t pid) {
        if (a.counters) {
            __hip_atomic_store(reinterpret_cast<unsigned *>(a.part_s) + part_o + pos, __float_as_uint(ps),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.part_id) + part_o + pos,
                               (unsigned long long)pid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            a.part_s[part_o + pos] = ps;
            a.part_id[part_o + pos] = pid;
        }
    };
    const bool small_rank = C <= 64;
    if (small_rank) {
        if (w == 0) {
            const bool v = lane < C;
            const float es = v ? g_s[lane] : 0.f;
            const int64_t eid = v ? g_id[lane] : 0;
            const int r = (a.debug & 2) ? 0 : wave_rank(v ? f2o(es) : 0u, eid, C, lane, reinterpret_cast<unsigned *>(g_rank));
            if (v) {
                if (r < k) publish(r, es, eid);
            } else if (lane < k) {
                publish(lane, MI_NEG_INF, EMPTY_ID);   // positions C .. k-1
            }
        }
    } else {
        for (int e = tid; e < C; e += NT) g_rank[e] = 0;
        if (tid < 64) {
            o_s[tid] = MI_NEG_INF;
            o_id[tid] = EMPTY_ID;
        }
        __syncthreads();
        if (!(a.debug & 2)) {
            const int P = C < NT ? max(1, NT / C) : 1;   // thread groups sharing the j range
            for (int e0 = 0; e0 < C; e0 += NT) {
                const int part = C < NT ? tid / C : 0;
                const int e = C < NT ? tid - part * C : e0 + tid;
                if (part < P && e < C) {
                    const float es = g_s[e];
                    const int64_t eid = g_id[e];
                    const int j0 = (C * part) / P, j1 = (C * (part + 1)) / P;
                    int r = 0;
#pragma unroll 4
                    for (int j = j0; j < j1; ++j) {
                        const float js = g_s[j];
                        const int64_t jid = g_id[j];
                        r += (js > es) || (js == es && (jid < eid || (jid == eid && j < e)));
                    }
                    if (r) atomicAdd(&g_rank[e], r);
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < C; e += NT) {
            const int r = g_rank[e];
            if (r < k) {
                o_s[r] = g_s[e];
                o_id[r] = g_id[e];
            }
        }
        __syncthreads();
        if (tid < k) publish(tid, o_s[tid], o_id[tid]);
    }
    stamp(8);
    if (!a.counters) return;

    // ---- fused final merge.  Small case (this slice ranked by wave 0 and the
    // query's nslice*k partial entries fit one wave): wave 0 alone drains its
    // stores, takes the ticket and, as the last arriver, merges in registers.
    if (small_rank && a.nslice * k <= 64) {
        if (w != 0) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(9);
        unsigned ticket = 0;
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(a.counters + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = (unsigned)uniform_i((int)ticket);
        stamp(10);
        if (ticket != (unsigned)(a.nslice - 1)) return;
        if (lane == 0) __hip_atomic_store(a.counters + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int n = a.nslice * k;
        int64_t eid = EMPTY_ID;
        float es = 0.f;
        if (lane < n) {
            const size_t o = (size_t)q * n + lane;
            eid = (int64_t)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.part_id) + o,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            es = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(a.part_s) + o,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        const bool v = eid != EMPTY_ID;
        const int nvalid = __popcll(__ballot(v));
        stamp(11);
        const int r = wave_rank(v ? f2o(es) : 0u, eid, n, lane, reinterpret_cast<unsigned *>(g_rank));
        stamp(15);
        float *Dq = a.D + (size_t)q * a.ldo + a.out_off;
        int64_t *Iq = a.I + (size_t)q * a.ldo + a.out_off;
        if (v && r < k) {
            Dq[r] = es;
            Iq[r] = eid;
            if (a.next_bound_s && r == k - 1) {
                a.next_bound_s[q] = es;
                a.next_bound_id[q] = eid;
            }
        }
        if (lane < k && lane >= nvalid) {
            Dq[lane] = -FLT_MAX;
            Iq[lane] = -1;
        }
        if (a.next_bound_s && nvalid < k && lane == 0) {
            a.next_bound_s[q] = MI_NEG_INF;
            a.next_bound_id[q] = EMPTY_ID;
        }
        stamp(12);
        return;
    }

    // General case: every storing wave drains its stores, one lane takes a ticket;
    // the last arriver merges the query's partial lists with the whole workgroup.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket =
            __hip_atomic_fetch_add(a.counters + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == (unsigned)(a.nslice - 1);
        if (last) __hip_atomic_store(a.counters + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *c_total = last;
    }
    __syncthreads();
    if (!*c_total) return;
    const int n = a.nslice * k;                                   // <= 32 * 64
    int64_t *e_id = reinterpret_cast<int64_t *>(lut_s);           // [n]
    float *e_s = reinterpret_cast<float *>(e_id + n);             // [n]
    int *e_rank = reinterpret_cast<int *>(e_s + n);               // [n]
    int64_t *f_id = reinterpret_cast<int64_t *>(smem + (size_t)n * 16);  // [64]
    float *f_s = reinterpret_cast<float *>(f_id + 64);            // [64]
    for (int e = tid; e < n; e += NT) {
        const size_t o = (size_t)q * n + e;
        e_id[e] = (int64_t)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.part_id) + o,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e_s[e] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(a.part_s) + o,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        e_rank[e] = 0;
    }
    if (tid < 64) {
        f_s[tid] = MI_NEG_INF;
        f_id[tid] = EMPTY_ID;
    }
    __syncthreads();
    {
        const int P = n < NT ? max(1, NT / n) : 1;
        for (int e0 = 0; e0 < n; e0 += NT) {
            const int part = n < NT ? tid / n : 0;
            const int e = n < NT ? tid - part * n : e0 + tid;
            if (part < P && e < n) {
                const int64_t eid = e_id[e];
                if (eid != EMPTY_ID) {
                    const float es = e_s[e];
                    const int j0 = (n * part) / P, j1 = (n * (part + 1)) / P;
                    int r = 0;
#pragma unroll 4
                    for (int j = j0; j < j1; ++j) {
                        const float js = e_s[j];
                        const int64_t jid = e_id[j];
                        r += (js > es) || (js == es && (jid < eid || (jid == eid && j < e)));
                    }
                    if (r) atomicAdd(&e_rank[e], r);
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < n; e += NT) {
        const int r = e_rank[e];
        if (e_id[e] != EMPTY_ID && r < k) {
            f_s[r] = e_s[e];
            f_id[r] = e_id[e];
        }
    }
    __syncthreads();
    if (tid < k) {
        const int64_t id = f_id[tid];
        a.D[(size_t)q * a.ldo + a.out_off + tid] = id == EMPTY_ID ? -FLT_MAX : f_s[tid];
        a.I[(size_t)q * a.ldo + a.out_off + tid] = id == EMPTY_ID ? (int64_t)-1 : id;
    }
    if (a.next_bound_s && tid == 0) {
        a.next_bound_s[q] = f_s[k - 1];
        a.next_bound_id[q] = f_id[k - 1];
    }
}

// ---------------------------------------------------------------------
// Merge `nparts` partial top-k lists per query into the final k under
// (score desc, id asc).  Entry (p, q, j) is at p*stride_p + q*stride_q + j.
// Empty entries: id == EMPTY_ID or id < 0.  Writes D/I at
// [q*ldo + out_off + rank]; unfilled -FLT_MAX / -1.  Optionally records the
// last kept entry per query (bound for the next extraction pass).
// One wave per query, QPB = blockDim/64 queries per workgroup; rank by
// counting over the n = nparts*k candidates staged in LDS.
// ---------------------------------------------------------------------
// Part p's ids may be local to the part: global = id * id_mul + id_add + p * id_step (the
// closed form of a round-robin or contiguous-range shard numbering; identity = {1, 0, 0}),
// applied on load so that the exchange step needs no translation pass.
struct IdMap {
    int64_t mul = 1, add = 0, step = 0;
};

// Candidate lists too long for the LDS of merge_kernel: [nparts][nq][k] parts -> one row of
// (score, id) pairs per query (empty slots: NaN score), which select_pairs_kernel reduces.
// prefix[q] = {0, row length / 64}: the one-"probe" table select_pairs_kernel reads its n from.
__global__ void __launch_bounds__(256)
    gather_parts_kernel(const float *__restrict__ ps, const int64_t *__restrict__ pid, int nparts, int64_t stride_p,
                        int64_t stride_p_id, int64_t stride_q, int k, int64_t ld, float *__restrict__ rs,
                        int64_t *__restrict__ rid, int32_t *__restrict__ prefix, IdMap im) {
    const int64_t q = blockIdx.x;
    const int n = nparts * k;
    for (int e = threadIdx.x; e < (int)ld; e += 256) {
        float s = __builtin_nanf("");
        int64_t id = -1;
        if (e < n) {
            const int p = e / k, j = e - p * k;
            const int64_t v = pid[(size_t)p * stride_p_id + (size_t)q * stride_q + j];
            if (v >= 0) {
                id = v * im.mul + im.add + p * im.step;
                s = ps[(size_t)p * stride_p + (size_t)q * stride_q + j];
            }
        }
        rs[q * ld + e] = s;
        rid[q * ld + e] = id;
    }
    if (threadIdx.x == 0) {
        prefix[q * 2] = 0;
        prefix[q * 2 + 1] = (int32_t)(ld / 64);
    }
}

// The k best (score desc, id asc) of each row of kc (score, id) candidates, negative ids and NaN scores skipped: the last step of
// a re-rank, k <= 32 of up to 8192 exact scores per query.  One 256-thread workgroup per row, the candidates in registers
// (kc / 256 per thread), k rounds of a block-wide arg-max on 96-bit (score key, id) keys -- where the general route
// (gather_parts_kernel + the sorting select_pairs_kernel) spent 36 + 33 us at 1024 x 5120, this is one pass over the rows.
template <int VPT_>
__global__ void __launch_bounds__(256) topk_rows_kernel(const float *__restrict__ S, const int64_t *__restrict__ IDS, int kc, int k,
                                                        float *__restrict__ D, int64_t *__restrict__ I, int64_t ldo) {
    __shared__ unsigned wk[2][4];
    __shared__ long long wi[2][4];
    __shared__ int wp[2][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t row = blockIdx.x;
    unsigned key[VPT_];
    long long id[VPT_];
#pragma unroll
    for (int j = 0; j < VPT_; ++j) {
        const int c = j * 256 + tid;
        const int cc = min(c, kc - 1);
        const float v = S[row * kc + cc];
        const long long t = IDS[row * kc + cc];
        id[j] = t;
        key[j] = (c < kc && t >= 0 && v == v) ? f2o(v) : 0u;
    }
    // a thread's best under (key desc, id asc, position asc) is kept between the rounds: only the thread whose entry left
    // looks at its VPT_ values again (every thread rescanning every round was 20 three-word compares x 10 rounds x 16 waves
    // a CU: the kernel was VALU-bound at 43 us for 1024 x 4 640, three times its bytes' time)
    unsigned lk;
    long long li;
    int lp;
    auto rescan = [&]() {
        lk = 0u;
        li = 0x7fffffffffffffffll;
        lp = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < VPT_; ++j) {
            const bool better = key[j] > lk || (key[j] == lk && key[j] != 0u && id[j] < li);   // (positions ascend with j)
            if (better) { lk = key[j]; li = id[j]; lp = j * 256 + tid; }
        }
    };
    rescan();
    for (int r = 0; r < k; ++r) {
        unsigned bk = lk;
        long long bi = li;
        int bp = lp;
        auto take = [&](unsigned ok, long long oi, int op) {
            if (ok > bk || (ok == bk && (oi < bi || (oi == bi && op < bp)))) { bk = ok; bi = oi; bp = op; }
        };
        // the wave's best: the largest key alone decides unless two lanes hold it (tied exact scores: rare) -- one word
        // through the butterfly, the winner's id and position read from its lane
        unsigned mk = lk;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mk = max(mk, (unsigned)__shfl_xor((int)mk, off));
        const unsigned long long tie = __ballot(lk == mk);
        if (__popcll(tie) == 1) {                                // (wave-uniform)
            const int src = (int)__ffsll((long long)tie) - 1;
            bk = mk;
            bi = ((long long)__builtin_amdgcn_readlane((int)(li >> 32), src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)li, src);
            bp = __builtin_amdgcn_readlane(lp, src);
        } else {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned ok = (unsigned)__shfl_xor((int)bk, off);
                const long long oi = ((long long)__shfl_xor((int)(bi >> 32), off) << 32) | (unsigned)__shfl_xor((int)bi, off);
                const int op = __shfl_xor(bp, off);
                take(ok, oi, op);
            }
        }
        if (lane == 0) { wk[r & 1][w] = bk; wi[r & 1][w] = bi; wp[r & 1][w] = bp; }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 4; ++x) take(wk[r & 1][x], wi[r & 1][x], wp[r & 1][x]);
        if (tid == 0) {
            D[row * ldo + r] = bk ? o2f(bk) : -FLT_MAX;
            I[row * ldo + r] = bk ? (int64_t)bi : (int64_t)-1;
        }
        if (bk == 0u) continue;                              // (workgroup-uniform) nothing left: the remaining rounds write the padding
        if ((bp & 255) == tid) {
#pragma unroll
            for (int j = 0; j < VPT_; ++j)
                if (j == (bp >> 8)) key[j] = 0u;             // the winner leaves
            rescan();
        }
    }
}

__host__ __device__ inline size_t merge_wave_bytes(int nparts, int k) {
    size_t n = (size_t)nparts * k;
    return ((n * 12 + 7) & ~(size_t)7) + (size_t)k * 16;   // e_id[n] e_s[n] | o_id[k] o_s[k](+pad)
}

__global__ void __launch_bounds__(256)
    merge_kernel(const float *__restrict__ ps, const int64_t *__restrict__ pid, int nparts,
                 int64_t stride_p, int64_t stride_p_id, int64_t stride_q, int64_t nq, int k, float *__restrict__ D,
                 int64_t *__restrict__ I, int64_t ldo, int out_off, float *__restrict__ bound_s,
                 int64_t *__restrict__ bound_id, IdMap im) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = nparts * k;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, qpb = blockDim.x >> 6;
    unsigned char *base = smem + (size_t)w * merge_wave_bytes(nparts, k);
    int64_t *e_id = reinterpret_cast<int64_t *>(base);                                   // [n]
    float *e_s = reinterpret_cast<float *>(e_id + n);                                    // [n]
    int64_t *o_id = reinterpret_cast<int64_t *>(base + (((size_t)n * 12 + 7) & ~(size_t)7)); // [k]
    float *o_s = reinterpret_cast<float *>(o_id + k);                                    // [k]
    const int64_t q = (int64_t)blockIdx.x * qpb + w;
    const bool live = q < nq;
    if (live) {
        for (int e = lane; e < n; e += 64) {
            int p = e / k, j = e - p * k;
            const size_t o = (size_t)p * stride_p + (size_t)q * stride_q + j;
            const int64_t id = pid[(size_t)p * stride_p_id + (size_t)q * stride_q + j];
            e_id[e] = id < 0 ? EMPTY_ID : id * im.mul + im.add + p * im.step;
            e_s[e] = id < 0 ? MI_NEG_INF : ps[o];
        }
        for (int j = lane; j < k; j += 64) {
            o_s[j] = MI_NEG_INF;
            o_id[j] = EMPTY_ID;
        }
    }
    // Ranking by counting is O(n^2): with many candidates (re-ranking 64 x k exact scores,
    // 32 slices x 64) first drop everything below the k-th largest score -- a bitwise
    // descent on order-preserving keys, ballots only -- and compact the survivors (k plus
    // ties) to the front, in place (an entry only ever moves down, and every lane has read
    // its entry before any lane writes).  Re-ranking 640 candidates per query (cfg4 shard,
    // k_factor 64): whole search 3.24 -> 2.95 ms per 1024 queries.
    int S = n;
    if (live && n > 64) {
        auto keyof = [&](int e) -> unsigned { return (e < n && e_id[e] != EMPTY_ID) ? f2o(e_s[e]) : 0u; };
        unsigned T = 0;
        if (n <= 1024) {
            unsigned kk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) kk[i] = keyof(i * 64 + lane);
            T = wave_kth_largest_n<16>(kk, k);
        } else {
            int nz = 0;
            for (int e0 = 0; e0 < n; e0 += 64) nz += __popcll(__ballot(keyof(e0 + lane) != 0u));
            if (nz >= k)
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned t = T | (1u << bit);
                    int c = 0;
                    for (int e0 = 0; e0 < n; e0 += 64) c += __popcll(__ballot(keyof(e0 + lane) >= t));
                    if (c >= k) T = t;
                    if (c == k) break;
                }
        }
        int cnt = 0;
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            const bool in = e < n;
            const float es = in ? e_s[e] : 0.f;
            const int64_t ei = in ? e_id[e] : EMPTY_ID;
            const bool keep = ei != EMPTY_ID && f2o(es) >= T;
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int pos = cnt + lane_prefix_count(m);
                e_s[pos] = es;
                e_id[pos] = ei;
            }
            cnt += __popcll(m);
        }
        S = cnt;
    }
    __syncthreads();
    if (live) {
        for (int e = lane; e < S; e += 64) {
            const int64_t mi_ = e_id[e];
            if (mi_ == EMPTY_ID) continue;
            const float ms = e_s[e];
            int rank = 0;
#pragma unroll 8
            for (int j = 0; j < S; ++j) {
                float js = e_s[j];
                int64_t ji = e_id[j];
                rank += (js > ms) || (js == ms && (ji < mi_ || (ji == mi_ && j < e)));
            }
            if (rank < k) {
                o_s[rank] = ms;
                o_id[rank] = mi_;
            }
        }
    }
    __syncthreads();
    if (live) {
        for (int j = lane; j < k; j += 64) {
            int64_t id = o_id[j];
            D[q * ldo + out_off + j] = id == EMPTY_ID ? -FLT_MAX : o_s[j];
            I[q * ldo + out_off + j] = id == EMPTY_ID ? (int64_t)-1 : id;
        }
        if (bound_s && lane == 0) {
            bound_s[q] = o_s[k - 1];
            bound_id[q] = o_id[k - 1];
        }
    }
}

// ---------------------------------------------------------------------
// Index.add arithmetic: r = x - centroid[assign] (if by_residual), then for
// sub-quantiser m the index of the L2-nearest codeword (ties: smallest).
// grid (ceil(n/256), M); the sub-codebook (256 x DSUB f32) sits in LDS and is
// read with wave-uniform (broadcast) addresses.
// ---------------------------------------------------------------------
template <int DSUB>
__global__ void __launch_bounds__(256)
    pq_encode_kernel(const float *__restrict__ x, int64_t n, int d, int M,
                     const float *__restrict__ codebook, const float *__restrict__ centroids,
                     const int32_t *__restrict__ assign, uint8_t *__restrict__ codes) {
    __shared__ float cb[256 * DSUB];
    const int m = blockIdx.y, tid = threadIdx.x;
    const float *cg = codebook + (size_t)m * 256 * DSUB;
    for (int i = tid; i < 256 * DSUB; i += 256) cb[i] = cg[i];
    __syncthreads();
    const int64_t v = (int64_t)blockIdx.x * 256 + tid;
    if (v >= n) return;
    float r[DSUB];
    const float *xp = x + (size_t)v * d + m * DSUB;
#pragma unroll
    for (int t = 0; t < DSUB; ++t) r[t] = xp[t];
    if (centroids) {
        const float *cp = centroids + (size_t)assign[v] * d + m * DSUB;
#pragma unroll
        for (int t = 0; t < DSUB; ++t) r[t] = r[t] - cp[t];
    }
    int best = 0;
    float bd = 0.f;
    for (int j = 0; j < 256; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < DSUB; ++t) {
            float df = r[t] - cb[j * DSUB + t];
            acc = __builtin_fmaf(df, df, acc);
        }
        if (j == 0 || acc < bd) {
            bd = acc;
            best = j;
        }
    }
    codes[(size_t)v * M + m] = (uint8_t)best;
}

// ---------------------------------------------------------------------
// Exact list pruning for a by-residual inner-product scan (large batches: ivfpq.hip, search_chunk).
//
// A code's score is s = dis0 + acc, acc = the f32 chain 0 + LUT[0][c_0] + LUT[1][c_1] + ... (m ascending), dis0 = <q, centroid>
// of its list.  With mx_m = max_c LUT[m][c] and A = the SAME chain over the maxima, rounded addition being monotone in both
// operands gives acc <= A and s <= U = fl(dis0 + A) for EVERY code of the list, in floating point, with no slack term.  So once
// k scores >= T are known for the query (phase 1: the scan of its P1 best lists), a list with U < T cannot hold a result --
// its codes are all strictly below the final k-th best (>= T), ties included -- and phase 2 scans only the lists that remain.
// Results are bit-identical to the exhaustive scan; how much goes depends on the data (clustered corpora: the residual
// tables are small against the spread of the coarse scores, and ~15 of 16 probed lists go).
//
// mode 0: the probe tables of phase 1 (probes [0, P1) keep their length, the others get length 0 and no groups);
// mode 1: those of phase 2 (probes >= P1 that survive U >= T; T = the k-th result of phase 1, -inf when it found fewer).
// One 256-thread workgroup per query; the group counts come from the full prefix table (a list's groups, not its length / 64).
// ---------------------------------------------------------------------
// s_mx[m] = max_c LUT[q][m][c]: a wave per row and round (one coalesced KiB per instruction), every load of a thread issued before
// the first is used; 256-thread workgroup, M <= 128.  Ends with a barrier.
__device__ __forceinline__ void lut_row_maxima(const float *__restrict__ lutq, int M, float *s_mx) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int R = 16;                                  // rounds of 4 rows kept in flight together
    for (int m0 = 0; m0 < M; m0 += 4 * R) {
        float4 v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int m = m0 + i * 4 + w;
            if (m < M) v[i] = reinterpret_cast<const float4 *>(lutq + (size_t)m * 256)[lane];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int m = m0 + i * 4 + w;
            if (m < M) {                                   // (wave-uniform)
                float x = fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w));
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
                if (lane == 0) s_mx[m] = x;
            }
        }
    }
    __syncthreads();
}

struct PruneArgs {
    const float *lut;          // [nq][M*256]           (mode 1)
    const float *coarse_dis;   // [nq][nprobe]          (mode 1)
    const int32_t *p_len;      // [nq][nprobe]   full tables, in
    const int32_t *p_prefix;   // [nq][nprobe+1]
    const float *t_s;          // [nq][ld_t]: phase 1's results (mode 1); T = t_s[q][k-1] if t_id[q][k-1] >= 0
    const int64_t *t_id;
    int64_t ld_t;
    int k, nprobe, P1, M, mode;
    int32_t *len_out;          // [nq][nprobe]
    int32_t *prefix_out;       // [nq][nprobe+1]
    unsigned long long *stats; // null, or [PRUNE_SLOTS][3] += {groups this phase scans, groups of all probes, queries} (slot = workgroup % PRUNE_SLOTS)
};

__global__ void __launch_bounds__(256) prune_tables_kernel(PruneArgs a) {
    __shared__ float s_mx[128];
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t q = blockIdx.x;
    const int K = a.nprobe;
    float A = 0.f, T = MI_NEG_INF;
    if (a.mode == 1) {
        lut_row_maxima(a.lut + (size_t)q * a.M * 256, a.M, s_mx);
        for (int m = 0; m < a.M; ++m) A += s_mx[m];          // the scan's chain (0 + ..., m ascending) over the maxima
        if (a.t_id[q * a.ld_t + a.k - 1] >= 0) T = a.t_s[q * a.ld_t + a.k - 1];
    }
    auto groups_kept = [&](int p) -> int {
        bool keep;
        if (a.mode == 0) keep = p < a.P1;
        else keep = p >= a.P1 && !(a.coarse_dis[q * K + p] + A < T);      // U < T: nothing in this list can be a result
        return keep ? a.p_prefix[q * (K + 1) + p + 1] - a.p_prefix[q * (K + 1) + p] : -1;
    };
    const int per = (K + 255) / 256, b = tid * per;
    int sum = 0;
    for (int i = 0; i < per; ++i)
        if (b + i < K) sum += max(groups_kept(b + i), 0);
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int ww = 0; ww < w; ++ww) run += wtot[ww];
    for (int i = 0; i < per; ++i)
        if (b + i < K) {
            const int p = b + i, g = groups_kept(p);
            a.len_out[q * K + p] = g >= 0 ? a.p_len[q * K + p] : 0;
            a.prefix_out[q * (K + 1) + p] = run;
            run += max(g, 0);
            if (p == K - 1) a.prefix_out[q * (K + 1) + K] = run;
        }
    if (a.stats && tid == 0) {
        unsigned long long *ps3 = a.stats + (size_t)(blockIdx.x & (PRUNE_SLOTS - 1)) * 3;
        atomicAdd(ps3, (unsigned long long)(wtot[0] + wtot[1] + wtot[2] + wtot[3]));
        atomicAdd(ps3 + 1, (unsigned long long)a.p_prefix[q * (K + 1) + K]);
        atomicAdd(ps3 + 2, 1ull);
    }
}

// A[q] = the scan's chain (0 + ..., m ascending) over the row maxima of the query's look-up tables: ScanArgs::prune_A
__global__ void __launch_bounds__(256) lut_maxsum_kernel(const float *__restrict__ lut, int M, float *__restrict__ A) {
    __shared__ float s_mx[128];
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    lut_row_maxima(lut + (size_t)q * M * 256, M, s_mx);
    if (tid == 0) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += s_mx[m];
        A[q] = acc;
    }
}

// sorted[q] = 1 when the coarse scores of row q never ascend (caller-assigned lists: mi_index_search_preassigned); NaN = not sorted
__global__ void __launch_bounds__(256) rows_descending_kernel(const float *__restrict__ dis, int64_t nq, int nprobe, int32_t *__restrict__ sorted) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float *r = dis + q * nprobe;
    int ok = 1;
    for (int p = 1; p < nprobe; ++p) ok &= (r[p] <= r[p - 1]) ? 1 : 0;
    if (nprobe > 0 && !(r[0] == r[0])) ok = 0;
    sorted[q] = ok;
}

// Σ of a table of list lengths (profiling only: the codes a scan launch reads)
__global__ void sum_len_kernel(const int32_t *__restrict__ len, int64_t n, unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = i < n ? (unsigned long long)max(len[i], 0) : 0ull;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}



// ---------------------------------------------------------------------
// Inverted-list maintenance (Index.add's append; reference Makefile:25 `index fill`).
// The master copy of the lists is an append log in HBM, insertion order:
// codes u8 [n][M] row-major, list number i32 [n], id i64 [n], and the slot of the
// entry inside its list, pos i32 [n] (= how many earlier entries went to the same
// list: faiss's insertion order).  The group-interleaved image the scan streams is
// derived from the log by one scatter pass; nothing of a 207 M-vector index ever
// lives on the host.
// ---------------------------------------------------------------------

// pos[i] = cnt[list[i]] + #{ j < i in this chunk : list[j] == list[i] } -- deterministic
// (no atomics decide an order).  Chunks are <= 65536 entries and stream-ordered; the
// in-chunk rank is a brute-force count over LDS tiles read as broadcasts.
__global__ void __launch_bounds__(256)
    list_rank_kernel(const int32_t *__restrict__ list_no, int n, const int32_t *__restrict__ cnt,
                     int32_t *__restrict__ pos) {
    __shared__ __attribute__((aligned(16))) int32_t tile[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int i = b * 256 + tid;
    const int mine = i < n ? list_no[i] : -1;
    int r = 0;
    for (int t = 0; t <= b; ++t) {
        __syncthreads();
        const int j = t * 256 + tid;
        tile[tid] = j < n ? list_no[j] : -2;
        __syncthreads();
        const int4 *t4 = reinterpret_cast<const int4 *>(tile);
        if (t < b) {
#pragma unroll 8
            for (int jj = 0; jj < 64; ++jj) {
                const int4 v = t4[jj];
                r += (v.x == mine) + (v.y == mine) + (v.z == mine) + (v.w == mine);
            }
        } else {  // own tile: earlier entries only
            for (int jj = 0; jj < tid; ++jj) r += tile[jj] == mine;
        }
    }
    if (i < n) pos[i] = cnt[mine] + r;
}

__global__ void __launch_bounds__(256)
    list_count_kernel(const int32_t *__restrict__ list_no, int n, int32_t *__restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[list_no[i]], 1);
}

// out[i][:] = base[ids[i]][:], rows of `row_bytes` bytes (mi_flat_get_rows: the stored bytes of a sample of a store).
// One workgroup per row; 16-byte pieces when the row size allows, bytes otherwise.
__global__ void __launch_bounds__(256) gather_rows_bytes_kernel(const unsigned char *__restrict__ base, size_t row_bytes,
                                                                const int64_t *__restrict__ ids, unsigned char *__restrict__ out, int wide) {
    const unsigned char *src = base + (size_t)ids[blockIdx.x] * row_bytes;
    unsigned char *dst = out + (size_t)blockIdx.x * row_bytes;
    if (wide) {                                          // row_bytes % 16 == 0 and an aligned destination (host-checked)
        for (size_t o = (size_t)threadIdx.x * 16; o < row_bytes; o += 256 * 16)
            *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
    } else {
        for (size_t o = threadIdx.x; o < row_bytes; o += 256) dst[o] = src[o];
    }
}

__global__ void __launch_bounds__(256) iota_ids_kernel(int64_t *__restrict__ ids, int64_t n, int64_t start) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids[i] = start + i;
}

// log -> group-interleaved image: one thread per (entry, 16-byte chunk).  The image is
// pre-filled (codes 0, ids -1) so that the padding lanes of a list's last group are inert.
__global__ void __launch_bounds__(256)
    build_image_kernel(const uint8_t *__restrict__ log_codes, const int32_t *__restrict__ log_list,
                       const int32_t *__restrict__ log_pos, const int64_t *__restrict__ log_ids, int64_t n,
                       const int32_t *__restrict__ goff, int M, int NCH, uint8_t *__restrict__ img,
                       int64_t *__restrict__ img_ids, const float *__restrict__ log_t = nullptr,
                       float *__restrict__ img_t = nullptr) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e = t / NCH;
    const int ch = (int)(t - e * NCH);
    if (e >= n) return;
    const int l = log_list[e], p = log_pos[e];
    const size_t grp = (size_t)goff[l] + (size_t)(p >> 6);
    const int lane = p & 63;
    uint8_t *dst = img + (grp * NCH + ch) * 1024 + (size_t)lane * 16;
    const uint8_t *src = log_codes + (size_t)e * M + ch * 16;
    if ((M & 15) == 0) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
    } else {
        const int nb = min(16, M - ch * 16);
        for (int i = 0; i < nb; ++i) dst[i] = src[i];
    }
    if (ch == 0) {
        img_ids[grp * 64 + lane] = log_ids[e];
        if (log_t) img_t[grp * 64 + lane] = log_t[e];
    }
}

// group-interleaved image -> log (mi_index_seal freed the log; an add / export / save needs it again): one thread per image
// slot.  Entry e = start[l] + s for slot s of list l -- list order instead of insertion order, which no consumer of the log
// depends on: the slot inside the list (log_pos) is what keeps a list's insertion order.
__global__ void __launch_bounds__(256)
    image_to_log_kernel(const uint8_t *__restrict__ img, const int64_t *__restrict__ img_ids, const float *__restrict__ img_t,
                        int64_t nslots, const int32_t *__restrict__ goff, const int32_t *__restrict__ len,
                        const int64_t *__restrict__ start, int nlist, int M, int NCH, uint8_t *__restrict__ log_codes,
                        int32_t *__restrict__ log_list, int32_t *__restrict__ log_pos, int64_t *__restrict__ log_ids,
                        float *__restrict__ log_t) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nslots) return;
    const int64_t grp = t >> 6;
    const int lane = (int)(t & 63);
    int lo = 0, hi = nlist;                                  // the list whose groups [goff[l], goff[l + 1]) hold grp
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)(unsigned)goff[mid] <= grp) lo = mid; else hi = mid;
    }
    // (empty lists share their goff with the next list: take the last list that starts at or before grp and has groups)
    const int l = lo;
    const int64_t s = (grp - (int64_t)(unsigned)goff[l]) * 64 + lane;
    if (s >= len[l]) return;
    const int64_t e = start[l] + s;
    log_list[e] = l;
    log_pos[e] = (int32_t)s;
    log_ids[e] = img_ids[t];
    if (log_t) log_t[e] = img_t[t];
    for (int ch = 0; ch < NCH; ++ch) {
        const uint8_t *src = img + ((size_t)grp * NCH + ch) * 1024 + (size_t)lane * 16;
        uint8_t *dst = log_codes + (size_t)e * M + ch * 16;
        if ((M & 15) == 0) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
        else {
            const int nb = min(16, M - ch * 16);
            for (int i = 0; i < nb; ++i) dst[i] = src[i];
        }
    }
}

// log -> the lists [list_lo, list_hi) concatenated, row-major codes, insertion order
// (InvertedLists::get_codes / get_ids; write_index).  start[l - list_lo] = first output row
// of list l.  One thread per (entry, 16-byte piece).
__global__ void __launch_bounds__(256)
    export_lists_kernel(const uint8_t *__restrict__ log_codes, const int32_t *__restrict__ log_list,
                        const int32_t *__restrict__ log_pos, const int64_t *__restrict__ log_ids, int64_t n,
                        int list_lo, int list_hi, const int64_t *__restrict__ start, int M, int NCH,
                        uint8_t *__restrict__ out_codes, int64_t *__restrict__ out_ids) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e = t / NCH;
    const int ch = (int)(t - e * NCH);
    if (e >= n) return;
    const int l = log_list[e];
    if (l < list_lo || l >= list_hi) return;
    const int64_t row = start[l - list_lo] + log_pos[e];
    const uint8_t *src = log_codes + (size_t)e * M + ch * 16;
    uint8_t *dst = out_codes + (size_t)row * M + ch * 16;
    const int nb = min(16, M - ch * 16);
    if ((M & 15) == 0) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
    else
        for (int i = 0; i < nb; ++i) dst[i] = src[i];
    if (ch == 0) out_ids[row] = log_ids[e];
}


// ---------------------------------------------------------------------
// k-means update step (IndexIVFPQ.train; reference Makefile:39 `index train`), deterministic:
// the members of a cluster are summed in ascending row order by sequential f32 adds -- an
// order the oracle restates with a plain loop -- instead of atomic scatter-adds whose order
// changes from run to run.  list_rank_kernel gives every row its slot inside its cluster.
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    cluster_scatter_kernel(const int32_t *__restrict__ assign, const int32_t *__restrict__ pos,
                           const int64_t *__restrict__ start, int64_t n, int32_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) perm[start[assign[i]] + pos[i]] = (int32_t)i;
}

// one workgroup per cluster, one thread per dimension (strided): reads are coalesced along d
__global__ void __launch_bounds__(256)
    cluster_mean_kernel(const float *__restrict__ x, int d, const int32_t *__restrict__ perm,
                        const int64_t *__restrict__ start, const int32_t *__restrict__ cnt, float *__restrict__ cent) {
    const int c = blockIdx.x;
    const int m = cnt[c];
    if (m == 0) return;                       // an empty cluster keeps its row (the caller re-seeds it)
    const int32_t *mem = perm + start[c];
    for (int t = threadIdx.x; t < d; t += 256) {
        float acc = 0.f;
        for (int j = 0; j < m; ++j) acc = acc + x[(size_t)mem[j] * d + t];
        cent[(size_t)c * d + t] = acc / (float)m;
    }
}

// out[r] = -0.5 * <x_r, x_r>, the dot an ascending-k fmaf chain from +0 (the oracle's dot):
// the augmenting column that turns arg max <x, c> into arg min |x - c|^2.  One thread per row.
__global__ void __launch_bounds__(256)
    neg_half_sqnorm_kernel(const float *__restrict__ x, int64_t n, int d, float *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float *p = x + (size_t)r * d;
    float acc = 0.f;
    for (int k = 0; k < d; ++k) acc = __builtin_fmaf(p[k], p[k], acc);
    out[r] = -0.5f * acc;
}


// half -> f32, exact (IndexScalarQuantizer.reconstruct_n of a QT_fp16 store)
__global__ void __launch_bounds__(256) f16_to_f32_kernel(const f16_t *__restrict__ x, int64_t n, float *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}


// ---------------------------------------------------------------------
// Re-ranking (IndexRefine: exact scores of k * k_factor candidate rows per query; faiss
// IndexRefineFlat::search / Refine(SQfp16)).  The gather mode of ip_gemm_kernel fetches each
// 64-candidate tile through a two-chunk-deep register pipeline built for L2-resident centroids;
// on rows scattered over a 50-100 GB store every chunk step waits out a DRAM latency (measured:
// the same 0.62 ms for 640 or 800 candidates, f32 or half rows, a 1 GB or a 106 GB store --
// tools/gather_bench.py).  This kernel is built around memory-level parallelism instead:
//   * one wave per (query, 64 candidates); lane r owns candidate r and runs its score as ONE
//     ascending-k fmaf chain on the VALU -- the oracle's dot, the same bits the f32 MFMA
//     produces -- so 64 chains advance per instruction instead of 16 per 40-cycle MFMA;
//   * the rows stream through an NST-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4:
//     8 lanes fetch one row's 128-byte piece, 8 instructions = one 8 KiB stage), NST-1 stages
//     requested ahead with counted s_waitcnt vmcnt -- no registers hold data in flight, no
//     barrier (one wave), and 4-5 waves per CU keep ~100 KB of row pieces in flight per CU;
//   * the 16-byte slots of a row piece are XOR-swizzled by (row >> 1) & 7 on the global side,
//     so that lane r's ds_read_b128 of "its" slot is bank-conflict-free across 16 lanes;
//   * the query row is wave-uniform: scalar loads, operands straight from SGPRs.
// TB = float (IndexFlat) or f16_t (IndexScalarQuantizer QT_fp16: widened exactly, same chain).
// Requires d * sizeof(TB) % 128 == 0.
// ---------------------------------------------------------------------
template <typename TB, int NST>
__global__ void __launch_bounds__(64)
    rerank_rows_kernel(const float *__restrict__ q, const TB *__restrict__ base, int64_t nb, int d,
                       const int64_t *__restrict__ cand, int kc, float *__restrict__ S, int64_t ldS, int tiles) {
    constexpr int EPC = 128 / (int)sizeof(TB);   // elements per 128-byte piece
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NST][8192];
    __shared__ const unsigned char *rowp[64];
    const int lane = threadIdx.x;
    const int qi = blockIdx.x / tiles, t = blockIdx.x - qi * tiles;
    const int c = t * 64 + lane;
    int64_t id = cand[(size_t)qi * kc + min(c, kc - 1)];
    id = max(min(id, nb - 1), (int64_t)0);       // an empty slot (negative) or a stray id must not fault; the caller ignores its score
    rowp[lane] = reinterpret_cast<const unsigned char *>(base + (size_t)id * d);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned char *src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * i + (lane >> 3);
        src[i] = rowp[row] + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const int nch = d / EPC;
    auto issue = [&](int ch) {
        unsigned char *st = ring[ch % NST];
#pragma unroll
        for (int i = 0; i < 8; ++i) dma16_lds(src[i] + (size_t)ch * 128, st + i * 1024);
    };
    for (int ch = 0; ch < NST - 1 && ch < nch; ++ch) issue(ch);
    const float *qrow = q + (size_t)qi * d;
    const int sw = (lane >> 1) & 7;
    float acc = 0.f;
    auto consume = [&](int ch) {
        const unsigned char *mine = ring[ch % NST] + lane * 128;
        const float *qk = qrow + ch * EPC;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const unsigned char *slot = mine + ((p ^ sw) << 4);
            if constexpr (sizeof(TB) == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(slot);
                acc = __builtin_fmaf(qk[p * 4 + 0], v.x, acc);
                acc = __builtin_fmaf(qk[p * 4 + 1], v.y, acc);
                acc = __builtin_fmaf(qk[p * 4 + 2], v.z, acc);
                acc = __builtin_fmaf(qk[p * 4 + 3], v.w, acc);
            } else {
                const f16x8 v = *reinterpret_cast<const f16x8 *>(slot);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qk[p * 8 + e], (float)v[e], acc);
            }
        }
    };
    int ch = 0;
    for (; ch + NST - 1 < nch; ++ch) {           // steady state: NST-1 stages stay in flight behind the one consumed
        issue(ch + NST - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 1)) : "memory");
        consume(ch);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; ch < nch; ++ch) consume(ch);
    if (c < kc) S[(size_t)qi * ldS + c] = acc;
}


// ---------------------------------------------------------------------
// ScalarQuantizer QT_8bit refine store (faiss ",Refine(SQ8)"; oracle: "ScalarQuantizer QT_8bit" section of
// ivfpq_oracle.c): one byte per component, per-dimension ranges `trained` = [vmin | vdiff].  It is what lets the
// recall >= 0.95 operating point keep ALL 207 M vectors beside the index in one GPU's HBM (212 GB).
// ---------------------------------------------------------------------
// train, pass 1: per-dimension min / max of rows [r0, r1) of a row chunk -> part[chunk][2][d] (min | max).  min / max
// are exact whatever the order, so chunking changes nothing.  grid (d / 256, chunks), a thread owns one column.
__global__ void __launch_bounds__(256)
    sq8_minmax_kernel(const float *__restrict__ x, int64_t n, int d, int64_t rows_per_chunk, float *__restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    float lo = HUGE_VALF, hi = -HUGE_VALF;
    for (int64_t r = r0; r < r1; ++r) {
        const float v = x[(size_t)r * d + c];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    part[((size_t)blockIdx.y * 2 + 0) * d + c] = lo;
    part[((size_t)blockIdx.y * 2 + 1) * d + c] = hi;
}
// train, pass 2: fold the chunk partials (and, `merge` != 0, the ranges already in `trained`, as lo | hi) -> lo | hi
__global__ void __launch_bounds__(256)
    sq8_minmax_fold_kernel(const float *__restrict__ part, int chunks, int d, float *__restrict__ lohi, int merge) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float lo = merge ? lohi[c] : HUGE_VALF, hi = merge ? lohi[d + c] : -HUGE_VALF;
    for (int k = 0; k < chunks; ++k) {
        lo = fminf(lo, part[((size_t)k * 2 + 0) * d + c]);
        hi = fmaxf(hi, part[((size_t)k * 2 + 1) * d + c]);
    }
    lohi[c] = lo;
    lohi[d + c] = hi;
}
// lo | hi -> vmin | vdiff
__global__ void __launch_bounds__(256) sq8_ranges_kernel(const float *__restrict__ lohi, int d, float *__restrict__ trained) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    trained[c] = lohi[c];
    trained[d + c] = lohi[d + c] - lohi[c];
}

// encode: code = (int)(255 * clip((x - vmin) / vdiff, 0, 1)), 0 where vdiff == 0 (IEEE division: hipcc's default).
// A thread encodes 4 consecutive components (one float4 in, 4 bytes out).
__global__ void __launch_bounds__(256)
    sq8_encode_kernel(const float *__restrict__ x, int64_t n4, int d, const float *__restrict__ trained, uint8_t *__restrict__ codes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index
    if (i >= n4) return;
    const int c = (int)((i * 4) % d);
    const float4 v = reinterpret_cast<const float4 *>(x)[i];
    const float4 lo = *reinterpret_cast<const float4 *>(trained + c), df = *reinterpret_cast<const float4 *>(trained + d + c);
    auto one = [](float xv, float vmin, float vdiff) -> unsigned {
        float xi = 0.f;
        if (vdiff != 0.f) {
            xi = (xv - vmin) / vdiff;
            if (xi < 0.f) xi = 0.f;
            if (xi > 1.f) xi = 1.f;
        }
        return (unsigned)(int)(255.f * xi);
    };
    reinterpret_cast<unsigned *>(codes)[i] = one(v.x, lo.x, df.x) | (one(v.y, lo.y, df.y) << 8) | (one(v.z, lo.z, df.z) << 16) |
                                             (one(v.w, lo.w, df.w) << 24);
}

__device__ __forceinline__ float sq8_component(unsigned c, float vmin, float vdiff) {
    const float t = __builtin_fmaf((float)c, 1.0f / 255.0f, 0.5f / 255.0f);
    return __builtin_fmaf(t, vdiff, vmin);
}

// decode (IndexScalarQuantizer.reconstruct_n)
__global__ void __launch_bounds__(256)
    sq8_decode_kernel(const uint8_t *__restrict__ codes, int64_t n, int d, const float *__restrict__ trained, float *__restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * d) return;
    const int c = (int)(i % d);
    x[i] = sq8_component(codes[i], trained[c], trained[d + c]);
}

// per-query table of the asymmetric SQ8 score (oracle_sq8_query_table): w[q][i] = q[i] * (vdiff[i] / 255) and
// A[q] = chain_i fmaf(q[i], vmin[i] + vdiff[i] 0.5/255, .).  One wave per query: the lanes compute a, b and w for
// their components and park the chain's operands (q, a) in LDS, then lane 0 runs the 1024-link chain out of LDS
// (a thread per query walking global memory took 217 us for 1024 queries: two dependent cache misses per link).
__global__ void __launch_bounds__(64)
    sq8_query_table_kernel(const float *__restrict__ q, int64_t nq, int d, const float *__restrict__ trained, float *__restrict__ w,
                           float *__restrict__ A) {
    extern __shared__ float2 sq8_qa[];                    // [d]: (q[i], a[i])
    const int64_t r = blockIdx.x;
    if (r >= nq) return;
    const float *qr = q + (size_t)r * d;
    float *wr = w + (size_t)r * d;
    for (int i = threadIdx.x; i < d; i += 64) {
        const float vmin = trained[i], vdiff = trained[d + i], qi = qr[i];
        const float a = __builtin_fmaf(vdiff, 0.5f / 255.0f, vmin), b = vdiff / 255.0f;
        wr[i] = qi * b;
        sq8_qa[i] = make_float2(qi, a);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        int i = 0;
        for (; i + 4 <= d; i += 4) {
            const float4 p0 = *reinterpret_cast<const float4 *>(&sq8_qa[i]), p1 = *reinterpret_cast<const float4 *>(&sq8_qa[i + 2]);
            acc = __builtin_fmaf(p0.x, p0.y, acc);
            acc = __builtin_fmaf(p0.z, p0.w, acc);
            acc = __builtin_fmaf(p1.x, p1.y, acc);
            acc = __builtin_fmaf(p1.z, p1.w, acc);
        }
        for (; i < d; ++i) acc = __builtin_fmaf(sq8_qa[i].x, sq8_qa[i].y, acc);
        A[r] = acc;
    }
}

// Re-ranking over the SQ8 store: rerank_rows_kernel's scheme (one wave per (query, 64 candidates), lane r owns
// candidate r, rows stream through an NST-stage LDS ring filled by LDS-DMA, one ascending-i fmaf chain per lane) with
// 128 components per 128-byte piece, scored in the asymmetric form score = A(q) + sum_i w(q)[i] code[i] (oracle: "ScalarQuantizer
// QT_8bit"): per component one byte -> float conversion and one fma whose multiplier w(q)[i] is wave-uniform (scalar
// loads) -- 2 VALU operations per byte where decoding every component took 5 and made the kernel VALU-bound (1.86 ms for
// 1024 x 5 120 rows; the rows are 5.4 GB).  Requires d % 128 == 0.
template <int NST>
__global__ void __launch_bounds__(64)
    rerank_sq8_kernel(const float *__restrict__ wq, const float *__restrict__ Aq, const uint8_t *__restrict__ base, int64_t nb, int d,
                      const int64_t *__restrict__ cand, int kc, float *__restrict__ S, int64_t ldS, int tiles) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NST][8192];
    __shared__ const unsigned char *rowp[64];
    const int lane = threadIdx.x;
    const int qi = blockIdx.x / tiles, t = blockIdx.x - qi * tiles;
    const int c = t * 64 + lane;
    int64_t id = cand[(size_t)qi * kc + min(c, kc - 1)];
    id = max(min(id, nb - 1), (int64_t)0);
    rowp[lane] = base + (size_t)id * d;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned char *src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * i + (lane >> 3);
        src[i] = rowp[row] + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const int nch = d / 128;
    auto issue = [&](int ch) {
        unsigned char *st = ring[ch % NST];
#pragma unroll
        for (int i = 0; i < 8; ++i) dma16_lds(src[i] + (size_t)ch * 128, st + i * 1024);
    };
    for (int ch = 0; ch < NST - 1 && ch < nch; ++ch) issue(ch);
    const float *wrow = wq + (size_t)qi * d;
    const int sw = (lane >> 1) & 7;
    float acc = Aq[qi];
    auto consume = [&](int ch) {
        const unsigned char *mine = ring[ch % NST] + lane * 128;
        const int k0 = ch * 128;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint4 v = *reinterpret_cast<const uint4 *>(mine + ((p ^ sw) << 4));
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = k0 + p * 16 + e * 4 + b;
                    acc = __builtin_fmaf(wrow[i], (float)((w[e] >> (8 * b)) & 0xffu), acc);
                }
            }
        }
    };
    int ch = 0;
    for (; ch + NST - 1 < nch; ++ch) {
        issue(ch + NST - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 1)) : "memory");
        consume(ch);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; ch < nch; ++ch) consume(ch);
    if (c < kc) S[(size_t)qi * ldS + c] = acc;
}

// the same scores without the streaming layout's d % 128 requirement (small test shapes): one thread per (query, candidate)
__global__ void __launch_bounds__(256)
    rerank_sq8_simple_kernel(const float *__restrict__ wq, const float *__restrict__ Aq, const uint8_t *__restrict__ base, int64_t nb, int d,
                             const int64_t *__restrict__ cand, int kc, int64_t total, float *__restrict__ S, int64_t ldS) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int64_t qi = g / kc;
    const int c = (int)(g - qi * kc);
    int64_t id = cand[g];
    id = max(min(id, nb - 1), (int64_t)0);
    const uint8_t *row = base + (size_t)id * d;
    const float *wrow = wq + (size_t)qi * d;
    float acc = Aq[qi];
    for (int i = 0; i < d; ++i) acc = __builtin_fmaf(wrow[i], (float)row[i], acc);
    S[(size_t)qi * ldS + c] = acc;
}

// ---------------------------------------------------------------------
// METRIC_L2 through the inner-product machinery (oracle: "METRIC_L2" section of ivfpq_oracle.c).
// arg min |x - c|^2 = arg max S, S = <x, c> - |c|^2/2, evaluated as ONE ascending-k fmaf chain
// over vectors augmented to `da` columns: [x, 1, 0..] . [c, -|c|^2/2, 0..] -- the exact-score
// GEMM, the selects and the two-stage coarse quantiser run unchanged on those.
// ---------------------------------------------------------------------
// mode 0: out[r] = [x_r, 1, 0..]   mode 1: out[r] = [x_r, -0.5 * chain(x_r . x_r), 0..]; one workgroup per row
__global__ void __launch_bounds__(256)
    augment_rows_kernel(const float *__restrict__ x, int64_t n, int d, int da, int mode, float *__restrict__ out) {
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    const float *xr = x + (size_t)r * d;
    float *o = out + (size_t)r * da;
    for (int k = threadIdx.x; k < d; k += 256) o[k] = xr[k];
    for (int k = d + 1 + threadIdx.x; k < da; k += 256) o[k] = 0.f;
    if (threadIdx.x == 0) {
        float v = 1.f;
        if (mode == 1) {
            float acc = 0.f;
            for (int k = 0; k < d; ++k) acc = __builtin_fmaf(xr[k], xr[k], acc);
            v = -0.5f * acc;
        }
        o[d] = v;
    }
}

// qn[r] = chain(x_r . x_r), one wave-less thread per row (rows are few: a query batch)
__global__ void __launch_bounds__(64) row_sqnorm_kernel(const float *__restrict__ x, int64_t n, int d, float *__restrict__ qn) {
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n) return;
    const float *p = x + (size_t)r * d;
    float acc = 0.f;
    for (int k = 0; k < d; ++k) acc = __builtin_fmaf(p[k], p[k], acc);
    qn[r] = acc;
}

// coarse scores S -> dis = 2 S - qn (= -|q - c|^2; unfilled probes keep -FLT_MAX); scan term = dis (by_residual) or -qn
__global__ void __launch_bounds__(256)
    l2_coarse_term_kernel(float *__restrict__ cdis, float *__restrict__ cscan, const int32_t *__restrict__ cidx,
                          const float *__restrict__ qn, int64_t nq, int nprobe, int by_residual) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * nprobe) return;
    const float q2 = qn[i / nprobe];
    const float dis = cidx[i] < 0 ? -FLT_MAX : 2.0f * cdis[i] - q2;
    cdis[i] = dis;
    cscan[i] = by_residual ? dis : -q2;
}

// per-vector term of the L2 expansion: t = chain(r^ . r^) + 2 chain(c . r^)  (k ascending over d),
// r^ the decoded code; !by_residual: t = chain(x^ . x^).  One thread per vector.
__global__ void __launch_bounds__(256)
    l2_term_kernel(const uint8_t *__restrict__ codes, const int32_t *__restrict__ list_no, int64_t n, int d, int M,
                   const float *__restrict__ codebook, const float *__restrict__ centroids, int by_residual,
                   float *__restrict__ t) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const int dsub = d / M;
    const uint8_t *c = codes + (size_t)v * M;
    const float *cen = centroids + (size_t)list_no[v] * d;
    float a1 = 0.f, a2 = 0.f;
    for (int m = 0; m < M; ++m) {
        const float *cw = codebook + ((size_t)m * 256 + c[m]) * dsub;
        for (int u = 0; u < dsub; ++u) {
            a1 = __builtin_fmaf(cw[u], cw[u], a1);
            if (by_residual) a2 = __builtin_fmaf(cen[m * dsub + u], cw[u], a2);
        }
    }
    t[v] = a1 + 2.0f * a2;
}

// scores (larger = better, -FLT_MAX in unfilled slots) -> squared distances, best first ascending,
// +FLT_MAX in unfilled slots (faiss CMax neutral)
__global__ void __launch_bounds__(256) l2_finish_kernel(float *__restrict__ D, const int64_t *__restrict__ I, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) D[i] = I[i] < 0 ? FLT_MAX : -D[i];
}
// IndexFlatL2: aug scores S -> D = -(2 S - qn)
__global__ void __launch_bounds__(256)
    l2_flat_finish_kernel(float *__restrict__ D, const int64_t *__restrict__ I, const float *__restrict__ qn, int64_t nq, int k) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nq * k) D[i] = I[i] < 0 ? FLT_MAX : -(2.0f * D[i] - qn[i / k]);
}
// strided rows -> packed (reconstruct_n of an augmented store)
__global__ void __launch_bounds__(256)
    unaugment_rows_kernel(const float *__restrict__ x, int64_t n, int d, int da, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n * d) out[i] = x[(i / d) * da + (i % d)];
}

// out[i][j] += bias[j] (the b of a LinearTransform x -> A x + b behind mi_ip_gemm)
__global__ void __launch_bounds__(256) add_row_bias_kernel(float *__restrict__ out, int64_t total, int nc, const float *__restrict__ bias) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] += bias[i % nc];
}

}  // namespace mi
