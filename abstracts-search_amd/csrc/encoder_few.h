// encoder_few.h -- the query-time encoder: a forward pass over a handful of tokens (one prompted query, T <= 48) is a
// pass over the 93.6 MB of bf16 weights of every layer and nothing else, so every kernel here is a WEIGHT STREAM with the
// arithmetic hanging off it (reference call site: README.md:28, the query-time app; arithmetic restated in
// oracle/encoder_oracle.py).  Five launches per layer for one sequence of <= 32 tokens (six otherwise) instead of the general
// path's ten, no atomics (bit-reproducible):
//
//   few_row_kernel<EMBED>          token ids -> f32 stream + the first RMSNorm as bf16 fragments
//   few_qkv8_kernel                QKV projection (H <= 1536: a wave keeps the fragments of its K range in registers;
//                                  else few_gemm_kernel<FEW_QKV>, fragments staged through LDS) -> +bias -> RoPE -> Q|K rows, V^T
//                                  -- or, in front of few_ao_kernel, the same values as the attention's MFMA-operand pieces
//   few_ao_kernel                  one sequence of <= 32 tokens: attention INSIDE the O projection's workgroups + residual add
//   few_attn_kernel + few_o_kernel otherwise: attention with its output written as fragments; O projection + residual add,
//                                  8 output features per workgroup over ALL of K: writes the stream, bf16(x g) fragments for
//                                  the next RMSNorm and the per-token partial sums of squares of its 8 columns
//   few_gu8_kernel                 gate/up projection on 8-feature units (H <= 1536; else few_gemm_kernel<FEW_GU> on 16-feature
//                                  pairs), 1/rms applied to the accumulators -> SwiGLU -> h fragments
//   few_d_kernel                   down projection, K split over workgroups -> partial planes (plain stores)
//   few_row_kernel<REDUCE>         stream += planes (fixed order), next layer's first RMSNorm as fragments
//
// Layout.  A weight matrix W[N][K] is stored as 1-KiB PIECES: piece (16-row block rb, K step ks) holds, at lane l's 16
// bytes, W'[16 rb + (l & 15)][32 ks + 8 (l >> 4) .. +8] -- the A operand of v_mfma_f32_16x16x32_bf16, so a wave's load
// of a piece is ONE contiguous 1 KiB and a row block's K run is one contiguous stream (few_tile_kernel builds the
// copies once; the O projection uses 8-row half pieces of 512 B).  Activations are FRAGMENTS in the same sense: piece
// (K step ks, token tile mt) holds at lane l the 8 values [token 16 mt + (l & 15)][32 ks + 8 (l >> 4) ..] -- the B
// operand; every producer writes its output directly in that form, so every consumer's staging is a straight copy of
// whole 128-byte lines.  The accumulator tile is D[feature 4 (l >> 4) + r][token l & 15]: a lane owns FOUR CONSECUTIVE
// FEATURES OF ONE TOKEN -- 8-byte bf16 stores, 16-byte f32 stores, and (with the Q/K rows of a block permuted to
// {8b..8b+7} u {hd/2 + 8b..}) the RoPE partner of a value sits in lane l ^ 32 of the same tile.  The columns of a tile
// are independent: what the padding tokens of the last tile hold (never written: anything) reaches no real token.
//
// What bounds it (tools/micro/stream_bw.hip -> profiles/r04_stream_bw.txt, and the first version of this file):
//   * a CU pulls ~45-50 GB/s from HBM whatever its waves have in flight, so a stream needs >= ~200 CUs to reach the
//     chip's ~5.2 TB/s: hence 8-feature workgroups for the O projection (192 of them) and a K split for the down
//     projection (whose activations are too long to give every workgroup all of K);
//   * the memory pipe of a CU charges ~3.5 cycles per (instruction, 128-byte line): a workgroup that reads the
//     T x H f32 stream for its own RMSNorm (16 rows x 64 B per instruction) spent 4.5 us there at 32 tokens -- the
//     first version's QKV kernel took 11.6 us for a 1.3 us stream -- so the norm is computed ONCE by whoever completes
//     the stream's rows and handed on as fragments (98 KB per consumer workgroup, whole lines);
//   * f32 atomics into the stream ran at ~70 per ns chip-wide: 491 k of them (32 tokens x 1536 x 10 K slices) were
//     7 us of a 10 us O projection.  No atomics anywhere now;
//   * a launch that streams 5-6 MB costs ~2.9 us whatever it does (boundary + ramp): the launch COUNT is the lever.
#pragma once
#include "encoder_kernels.h"

namespace mienc {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
enum { FEW_QKV = 0, FEW_GU = 1 };
constexpr int FEW_NW = 6;          // waves of a few_gemm workgroup: 1, 2 or 3 row units x 6, 3 or 2 K ranges
constexpr int FEW_OW = 8;          // waves of a few_o workgroup (K ranges)
constexpr int FEW_MAX_T = 48;
constexpr int FEW_SSQ_LD = 64;     // tokens per row of the sum-of-squares partials

struct FewArgs {
    int T, H;                      // real tokens (packed back to back), model width
    int nk;                        // K steps of 32 of this GEMM
    int nunits;                    // row units: 16-row blocks (QKV, D), gate/up pairs of blocks (GU), 8-row halves (O)
    int nslices, ks_per_slice;     // D: K slices over workgroups (reduce: planes to add)
    const bf16_t *W;               // pieces
    const bf16_t *afrag;           // activation fragments [nk][MT] KiB
    float *x;                      // residual stream [T_pad][H]
    const float *norm_w;           // [H]: the RMSNorm gains applied to what this kernel hands on (O, reduce, embed)
    float eps;
    const float *ssq;              // GU: [nparts][FEW_SSQ_LD] partial sums of squares of the stream's rows (nparts = 0: operand already normalised)
    float *ssq_out;                // O: the same, written
    int nparts;
    bf16_t *xfrag;                 // O / reduce / embed: bf16 fragments of the (gain-scaled / normalised) stream, [H / 32][MT] KiB
    // QKV epilogue
    const float *bias;             // [qk_cols + v_cols], original feature order
    bf16_t *qk;                    // [T_pad][ldqk]
    bf16_t *vt;                    // [v_cols][ldvt]
    int ldqk, ldvt, qk_cols, hd, rope_blocks;   // rope_blocks = (n_heads + n_kv) * hd / 16
    const int32_t *pos;
    const float *cos_t, *sin_t;    // [max_seq][hd / 2]
    bf16_t *hfrag;                 // GU epilogue: h as fragments (K step of the down projection, token tile)
    float *part;                   // D: [nslices][T_pad][H] partial planes; reduce: the same, read
    unsigned *ctr;                 // D (fused reduction): one arrival counter per unit group, zero between launches
    int T_pad;
    // embed
    const int32_t *ids;
    const bf16_t *table;
    unsigned long long *ts;        // MI_FEW_TS (profiling): [workgroup][8] s_memtime stamps of wave 0
    // few_ao_kernel (attention inside the O projection): qk / vt above are read
    int n_heads, n_kv, causal;
    float scale;
    // attn_pieces (QKV epilogue -> few_ao_kernel, one sequence of <= 32 tokens): qk / vt hold the MFMA operands of the attention
    // as 1-KiB pieces instead of rows --
    //   Q   qk + ((head (hd/32) + kd) MT + mt) KiB: lane (token % 16, lg) = dims 32 kd + 8 lg .. + 8 of token 16 mt + ..
    //   K   behind the n_heads (hd/32) MT pieces of Q, ((kvh 2 + j) (hd/32) + kd) KiB: lane (p, lg), tile position p of key
    //       tile j = key 8 (p / 4) + 4 j + p % 4 (few_attn_kernel's key order)
    //   V^T vt + (kvh (hd/16) + n) KiB: lane (dim % 16, lg) = keys 8 lg .. + 8 of dim 16 n + ..
    // so that every load of the attention is one contiguous KiB (a row-wise operand load touches 16 lines of which it uses half:
    // 4 800 line requests per workgroup, 7 us of a CU's memory pipe, against ~1 100)
    int attn_pieces;
};

__device__ __forceinline__ void few_stamp(const FewArgs &a, int slot) {
    if (a.ts && threadIdx.x == 0) a.ts[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
}

// W [N][ldw] row-major bf16 -> pieces of RB rows (16, or 8 for the O projection: lane l then carries row l & 7, so a
// half piece is 512 B).  rope_blocks > 0 (RB = 16): the first rope_blocks blocks are Q / K head rows, permuted inside each
// head so that block b of a head holds the features {8 b .. 8 b + 7} and {hd/2 + 8 b .. hd/2 + 8 b + 7}.
// KPERM (RB = 8, few_ao_kernel's copy of the O projection): inside every K step lane group lg carries the columns
// {4 lg .. 4 lg + 3} u {16 + 4 lg .. 16 + 4 lg + 3} -- the order in which the attention's output tiles leave a lane's registers.
// GU8 (RB = 16, few_gu8_kernel's copy of the gate/up matrix, whose rows come as 16 gate rows / 16 up rows per 16 features):
// piece rb = the gate rows (positions 0..7) and the up rows (8..15) of the 8 features 8 rb ..
template <int RB, bool KPERM = false, bool GU8 = false>
__global__ void __launch_bounds__(64) few_tile_kernel(const bf16_t *__restrict__ W, int N, int K, int ldw, int rope_blocks,
                                                      int hd, bf16_t *__restrict__ out) {
    const int nk = K / 32, piece = blockIdx.x, rb = piece / nk, ks = piece - rb * nk, l = threadIdx.x;
    const int i = l & 15, lg = l >> 4;
    if (RB == 8 && i >= 8) return;
    int row = rb * RB + i;
    if constexpr (GU8) row = 32 * (rb >> 1) + 8 * (rb & 1) + (i < 8 ? i : i + 8);
    if (RB == 16 && rb < rope_blocks) {
        const int bph = hd / 16, head = rb / bph, b = rb - head * bph;
        row = head * hd + (i < 8 ? 8 * b + i : hd / 2 + 8 * b + (i - 8));
    }
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < N) {
        if constexpr (KPERM) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(W + (size_t)row * ldw + ks * 32 + lg * 4);
            const uint2 hi = *reinterpret_cast<const uint2 *>(W + (size_t)row * ldw + ks * 32 + 16 + lg * 4);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            v = *reinterpret_cast<const uint4 *>(W + (size_t)row * ldw + ks * 32 + lg * 8);
        }
    }
    *reinterpret_cast<uint4 *>(out + (size_t)piece * (RB * 32) + (lg * RB + i) * 8) = v;
}

// One piece of the weight stream -> registers; non-temporal: a piece is read by one wave once per forward pass
// (MI355X_MICROARCH: nt-weights).  Plain loads the compiler tracks, with a scheduling barrier behind every one of them:
// left to itself hipcc (a) gathers the U loads of an unrolled round behind the round's U steps and (b) orders the first U
// loads by address, so that the wait in front of the round's first MFMA, which must hold for the entry AND the back edge,
// becomes vmcnt(1) -- a drained ring, one HBM latency per round (the first version: 14 us for the down projection).
// Pinned to ring order, its own counted waits are the right ones (vmcnt((U - 1) WN)).  (Loads issued from inline asm with
// hand-placed waits are NOT an option here: for a tied operand the compiler copies the destination registers of a load
// that has not landed and recycles them -- the late data then overwrote an address: memory access fault.)
__device__ __forceinline__ void few_wload(bf16x8 &dst, const bf16_t *p) {
    dst = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const i32x4_t *>(p)));
    __builtin_amdgcn_sched_barrier(0);
}

// The stream: `nsteps` K steps of one row unit (WN row blocks; consecutive steps STEP elements apart) against the
// activation fragments in LDS, U steps of pieces in flight per wave.  `ring` arrives requested (few_ring_start, before
// the prologue's work); every step issues exactly WN loads (a clamped step index past the end).
template <int WN, int U, int STEP>
__device__ __forceinline__ void few_ring_start(const bf16_t *wp, size_t blk_stride, int nsteps, bf16x8 (&ring)[U][WN]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < WN; ++j) few_wload(ring[u][j], wp + j * blk_stride + (size_t)min(u, max(nsteps - 1, 0)) * STEP);
}

template <int WN, int MT, int U, int STEP>
__device__ __forceinline__ void few_stream(const bf16_t *wp, size_t blk_stride, int nsteps, const uint4 *afr,
                                           bf16x8 (&ring)[U][WN], f32x4 (&acc)[WN][MT]) {
    bf16x8 an[MT];                                       // activation fragments one step ahead of their MFMAs
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) an[mt] = as_bf16x8(afr[mt * 64]);
    auto step = [&](int s, bf16x8(&b)[WN]) {
        bf16x8 a[MT];
        const int sn = min(s + 1, nsteps - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            a[mt] = an[mt];
            an[mt] = as_bf16x8(afr[(sn * MT + mt) * 64]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[mt], acc[j][mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int sl = min(s + U, nsteps - 1);
#pragma unroll
        for (int j = 0; j < WN; ++j) few_wload(b[j], wp + j * blk_stride + (size_t)sl * STEP);
    };
    int s0 = 0;
    for (; s0 + U <= nsteps; s0 += U) {                  // whole rounds: no branch between the steps
#pragma unroll
        for (int u = 0; u < U; ++u) step(s0 + u, ring[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (s0 + u < nsteps) step(s0 + u, ring[u]);
}

// `npieces` contiguous 1-KiB fragments global -> LDS by NWV waves: few_stage_load requests the first AB pieces of every
// wave (the caller requests its weight ring next: loads return in order, so what is requested before the ring is usable
// before the ring has landed), few_stage_store writes them and copies whatever is left piece by piece (shapes beyond the
// one-batch sizes: H > 1536).  Unconditional accesses -- a clamped piece index rewrites the last piece with its own data;
// a guarded store is a branch.  (Plain arrays and free functions: as members of a struct the registers went to scratch.)
template <int NWV, int AB>
__device__ __forceinline__ void few_stage_load(i32x4_t (&av)[AB], const bf16_t *src, int npieces, int w, int lane) {
#pragma unroll
    for (int i = 0; i < AB; ++i)
        av[i] = *reinterpret_cast<const i32x4_t *>(src + (size_t)min(w + NWV * i, npieces - 1) * 512 + lane * 8);
}
template <int NWV, int AB>
__device__ __forceinline__ void few_stage_store(i32x4_t (&av)[AB], const bf16_t *src, uint4 *lds, int npieces, int w, int lane) {
    i32x4_t *l4 = reinterpret_cast<i32x4_t *>(lds);
#pragma unroll
    for (int i = 0; i < AB; ++i) l4[min(w + NWV * i, npieces - 1) * 64 + lane] = av[i];
    for (int p = NWV * AB + w; p < npieces; p += NWV)
        l4[p * 64 + lane] = *reinterpret_cast<const i32x4_t *>(src + (size_t)p * 512 + lane * 8);
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM over the full K = H of (already gain-scaled or normalised) fragments -> epilogue (QKV: bias + RoPE + stores;
// GU: 1/rms, SwiGLU -> h fragments).
// grid: G workgroups of 6 waves; workgroup b owns the row units [b nunits / G, (b + 1) nunits / G) (1, 2 or 3 of them:
// the host sizes G so), unit j of it is streamed by waves [j wpu, (j + 1) wpu), wpu = 6 / count, each over a K range;
// the partial tiles meet in LDS (a fixed summation order).
// dynamic LDS: max(nk, 6 WN) x MT KiB (the fragments, later the partial tiles) | ssw[6][64] f32
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, int MT>
__global__ void __launch_bounds__(64 * FEW_NW) few_gemm_kernel(FewArgs a) {
    constexpr int WN = MODE == FEW_GU ? 2 : 1;
#ifndef FEW_U_GU
#define FEW_U_GU 16
#endif
    constexpr int U = MODE == FEW_GU ? (MT == 3 ? 12 : FEW_U_GU) : 8;   // K steps in flight per wave: ALL of a 2-unit workgroup's (the memory pipe of the CU must never idle: 47 GB/s is all it has)
    extern __shared__ __attribute__((aligned(16))) uint4 few_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, H = a.H, T = a.T;
    const int area = max(nk, FEW_NW * WN) * MT * 64;           // uint4 units
    float *ssw = reinterpret_cast<float *>(few_lds + area);

    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int u0 = (int)((long)b * a.nunits / G), u1 = (int)((long)(b + 1) * a.nunits / G);
    const int nu = u1 - u0;                                      // 1..3 (0: nothing to do)
    if (nu <= 0) return;
    const int wpu = FEW_NW / nu;
    const int ul = min(w / wpu, nu - 1), kp = w - ul * wpu;      // (nu = 1..3 divides 6: every wave has a unit)
    const int kper = (nk + wpu - 1) / wpu;
    const int k0 = min(nk, kp * kper), k1 = min(nk, k0 + kper);
    const int nsteps = k1 - k0;
    const int unit = u0 + ul;
    const bf16_t *wp = a.W + ((size_t)unit * WN * nk + min(k0, nk - 1)) * 512 + lane * 8;
    const size_t blk_stride = (size_t)nk * 512;
    few_stamp(a, 0);

    // ---- prologue: the fragments -> LDS (8 MT pieces per wave at H = 1536: one batch), the head of the weight stream
    // requested behind them; the sums of squares of the stream's rows from the producer's partials
    // (requested in the order they are needed: loads return in order, and the first thing behind the ring is an HBM latency away)
    // QKV: the epilogue operands of this wave's first item (bias, position -> rotary table rows) ride in front of everything
    f32x4 pre_b = (f32x4){0.f, 0.f, 0.f, 0.f}, pre_c = pre_b, pre_s = pre_b;
    if (MODE == FEW_QKV && w < nu * MT) {
        const int eu = w / MT, mt = w - eu * MT, eunit = u0 + eu;
        if (eunit < a.rope_blocks) {
            const int bph = a.hd / 16, head = eunit / bph, bb = eunit - head * bph, half = a.hd / 2;
            const int jf = 8 * bb + 4 * (lg & 1);
            const int ps = a.pos[min(16 * mt + li, T - 1)];
            pre_b = *reinterpret_cast<const f32x4 *>(a.bias + head * a.hd + (lg < 2 ? 0 : half) + jf);
            pre_c = *reinterpret_cast<const f32x4 *>(a.cos_t + (size_t)ps * half + jf);
            pre_s = *reinterpret_cast<const f32x4 *>(a.sin_t + (size_t)ps * half + jf);
        } else {
            pre_b = *reinterpret_cast<const f32x4 *>(a.bias + a.qk_cols + 16 * (eunit - a.rope_blocks) + 4 * lg);
        }
    }
    constexpr int SQ = MODE == FEW_GU ? 32 : 4;                 // partials per wave requested at once (GU: 192 = H / 8 from the O projection; QKV: 24 unit groups of the down projection)
    float sq[SQ];
    if (a.nparts > 0) {                                         // wave w adds the partials w, w + 6, ... of token `lane`
#pragma unroll
        for (int i = 0; i < SQ; ++i) sq[i] = a.ssq[min(w + FEW_NW * i, a.nparts - 1) * FEW_SSQ_LD + lane];
    }
    i32x4_t av[8 * MT];
    few_stage_load<FEW_NW, 8 * MT>(av, a.afrag, nk * MT, w, lane);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 ring[U][WN];
    few_ring_start<WN, U, 512>(wp, blk_stride, nsteps, ring);
    __builtin_amdgcn_sched_barrier(0);
    if (a.nparts > 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < SQ; ++i) s += w + FEW_NW * i < a.nparts ? sq[i] : 0.f;
        for (int p = w + FEW_NW * SQ; p < a.nparts; p += FEW_NW) s += a.ssq[p * FEW_SSQ_LD + lane];
        ssw[w * 64 + lane] = s;
    }
    few_stamp(a, 1);
    few_stage_store<FEW_NW, 8 * MT>(av, a.afrag, few_lds, nk * MT, w, lane);
    few_stamp(a, 2);
    __syncthreads();
    few_stamp(a, 3);

    // ---- the stream
    f32x4 acc[WN][MT];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) few_stream<WN, MT, U, 512>(wp, blk_stride, nsteps, few_lds + (size_t)k0 * MT * 64 + lane, ring, acc);
    few_stamp(a, 4);
    __syncthreads();                                           // every wave is done with the fragments: the area is reused
    few_stamp(a, 5);
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            reinterpret_cast<f32x4 *>(few_lds)[((w * WN + j) * MT + mt) * 64 + lane] = acc[j][mt];
    __syncthreads();

    // ---- epilogue: item = (unit of the workgroup, token tile); a wave takes whole tiles (the lane keeps its tile position)
    few_stamp(a, 6);
    const f32x4 *part = reinterpret_cast<const f32x4 *>(few_lds);
    for (int it = w; it < nu * MT; it += FEW_NW) {
        const int eu = it / MT, mt = it - eu * MT;
        const int m = 16 * mt + li;
        float inv = 1.f;
        if (a.nparts > 0) {
            float tot = 0.f;
#pragma unroll
            for (int ww = 0; ww < FEW_NW; ++ww) tot += ssw[ww * 64 + m];
            inv = rsqrtf(tot / (float)H + a.eps);
        }
        f32x4 v[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < wpu; ++c) {                     // ascending K ranges: a fixed order
                const f32x4 p = part[(((eu * wpu + c) * WN + j) * MT + mt) * 64 + lane];
                v[j][0] += p[0]; v[j][1] += p[1]; v[j][2] += p[2]; v[j][3] += p[3];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] *= inv;
        }
        const int eunit = u0 + eu;
        if constexpr (MODE == FEW_GU) {
            float hq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) hq[r] = v[0][r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[0][r])) * v[1][r];
            uint2 o;
            o.x = pack2(hq[0], hq[1]);
            o.y = pack2(hq[2], hq[3]);
            // feature 16 eunit + 4 lg + r of token m -> piece (K step eunit / 2, tile mt), lane (2 (eunit & 1) + lg / 2, li), half lg & 1
            bf16_t *dst = a.hfrag + ((size_t)((eunit >> 1) * MT + mt) * 64 + ((2 * (eunit & 1) + (lg >> 1)) * 16 + li)) * 8 + 4 * (lg & 1);
            *reinterpret_cast<uint2 *>(dst) = o;
        } else {
            if (eunit < a.rope_blocks) {
                const int bph = a.hd / 16, head = eunit / bph, bb = eunit - head * bph, half = a.hd / 2;
                const int jf = 8 * bb + 4 * (lg & 1);              // index inside the half
                const int fih = (lg < 2 ? 0 : half) + jf;          // feature inside the head (r added below)
                f32x4 bv = pre_b, c4 = pre_c, s4 = pre_s;
                if (it != w) {                                     // (only the wave's first item was prefetched)
                    const int ps = a.pos[min(m, T - 1)];
                    bv = *reinterpret_cast<const f32x4 *>(a.bias + head * a.hd + fih);
                    c4 = *reinterpret_cast<const f32x4 *>(a.cos_t + (size_t)ps * half + jf);
                    s4 = *reinterpret_cast<const f32x4 *>(a.sin_t + (size_t)ps * half + jf);
                }
                float q[4] = {v[0][0] + bv[0], v[0][1] + bv[1], v[0][2] + bv[2], v[0][3] + bv[3]};
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = __shfl_xor(q[r], 32);
                const float cs[4] = {c4[0], c4[1], c4[2], c4[3]}, sn[4] = {s4[0], s4[1], s4[2], s4[3]};
                float o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = lg < 2 ? q[r] * cs[r] - p[r] * sn[r] : q[r] * cs[r] + p[r] * sn[r];
                if (m < T) {
                    uint2 o;
                    o.x = pack2(o4[0], o4[1]);
                    o.y = pack2(o4[2], o4[3]);
                    bf16_t *dst = a.qk + (size_t)m * a.ldqk + head * a.hd + fih;
                    if (a.attn_pieces) {                           // features fih .. fih + 3 of the head: K step fih / 32, lane group (fih / 8) % 4, half (fih / 4) % 2
                        const int nkd = a.hd / 32, kd = fih >> 5, sub = (((fih >> 3) & 3) * 16) * 8 + (fih & 4);
                        if (head < a.n_heads)
                            dst = a.qk + ((size_t)((head * nkd + kd) * MT + mt) * 64 + li) * 8 + sub;
                        else                                       // key m: tile (m / 4) % 2, position 4 (m / 8) + m % 4
                            dst = a.qk + ((size_t)a.n_heads * nkd * MT + (size_t)(((head - a.n_heads) * 2 + ((m >> 2) & 1)) * nkd + kd)) * 512 +
                                  (4 * (m >> 3) + (m & 3)) * 8 + sub;
                    }
                    *reinterpret_cast<uint2 *>(dst) = o;
                }
            } else {
                const int vf = 16 * (eunit - a.rope_blocks) + 4 * lg;   // V feature (r added below)
                f32x4 bv = pre_b;
                if (it != w) bv = *reinterpret_cast<const f32x4 *>(a.bias + a.qk_cols + vf);
                const float bb[4] = {bv[0], bv[1], bv[2], bv[3]};
                if (m < T) {
                    if (a.attn_pieces) {                           // dim vf + r of key m: piece (kv head, (dim % hd) / 16), lane (dim % 16, m / 8), element m % 8
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int dim = vf + r, kvh = dim / a.hd, wi = dim - kvh * a.hd;
                            a.vt[((size_t)(kvh * (a.hd / 16) + (wi >> 4)) * 64 + ((m >> 3) * 16 + (wi & 15))) * 8 + (m & 7)] = f2bf(v[0][r] + bb[r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) a.vt[(size_t)(vf + r) * a.ldvt + m] = f2bf(v[0][r] + bb[r]);
                    }
                }
            }
        }
    }
    few_stamp(a, 7);
}

// ---------------------------------------------------------------------------------------------------------------
// Gate/up projection + SwiGLU on 8-FEATURE units (H <= 1536): a unit = one 16-row piece stream, positions 0..7 the gate rows
// and 8..15 the up rows of 8 features (few_tile_kernel<16, false, true>), so that the I / 8 = 1120 units deal out as 4 or 5 per
// workgroup where the 560 16-feature pairs of few_gemm_kernel<FEW_GU> deal out as 2 or 3 -- its launch lasts as long as the
// 3-unit workgroups (a third more stream than the mean: 13.4 us against 11.6 by the in-kernel stamps).  Wave w = K range w of
// EVERY unit of the workgroup: its B fragments (8 K steps x MT tiles) stay in registers for the whole launch -- no staging
// through LDS, no barrier in front of the stream -- and the ring holds the next unit's pieces of the same range.  The six
// ranges' partial tiles meet in LDS in ascending order; 1 / rms on the sums; the up value of a gate value sits in lane ^ 32.
// dynamic LDS: FEW_GW x FEW_GNU x MT KiB (partial tiles) | ssw[FEW_GW][64] f32
// ---------------------------------------------------------------------------------------------------------------
constexpr int FEW_GW = 6, FEW_GKP = 8, FEW_GNU = 6;     // waves = K ranges, K steps of a range at most, units of a workgroup at most
template <int MT>
__global__ void __launch_bounds__(64 * FEW_GW) few_gu8_kernel(FewArgs a) {
    constexpr int KP = FEW_GKP;
    extern __shared__ __attribute__((aligned(16))) uint4 few_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, H = a.H;
    f32x4 *part = reinterpret_cast<f32x4 *>(few_lds);
    float *ssw = reinterpret_cast<float *>(few_lds + (size_t)FEW_GW * FEW_GNU * MT * 64);
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int u0 = (int)((long)b * a.nunits / G), u1 = (int)((long)(b + 1) * a.nunits / G);
    const int nu = u1 - u0;                                      // <= FEW_GNU (the host sizes G so)
    if (nu <= 0) return;
    const int kper = (nk + FEW_GW - 1) / FEW_GW;                 // <= KP
    const int k0 = min(nk, w * kper), k1 = min(nk, k0 + kper), nsteps = k1 - k0;
    few_stamp(a, 0);
    // ---- requests, in the order they are needed: the rows' partial sums of squares, the fragments of this K range, unit 0's pieces
    constexpr int SQ = 32;
    float sq[SQ];
    if (a.nparts > 0) {
#pragma unroll
        for (int i = 0; i < SQ; ++i) sq[i] = a.ssq[min(w + FEW_GW * i, a.nparts - 1) * FEW_SSQ_LD + lane];
    }
    bf16x8 bf[KP][MT];
#pragma unroll
    for (int s = 0; s < KP; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            bf[s][mt] = *reinterpret_cast<const bf16x8 *>(a.afrag + ((size_t)(min(k0 + s, nk - 1) * MT + mt) * 64 + lane) * 8);
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t *wbase = a.W + (size_t)lane * 8;
    auto piece = [&](int unit, int s) { return wbase + ((size_t)unit * nk + min(k0 + s, nk - 1)) * 512; };
    bf16x8 ring[KP];
#pragma unroll
    for (int s = 0; s < KP; ++s) few_wload(ring[s], piece(u0, s));
    if (a.nparts > 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < SQ; ++i) t += w + FEW_GW * i < a.nparts ? sq[i] : 0.f;
        for (int p = w + FEW_GW * SQ; p < a.nparts; p += FEW_GW) t += a.ssq[p * FEW_SSQ_LD + lane];
        ssw[w * 64 + lane] = t;
    }
#pragma unroll
    for (int s = 0; s < KP; ++s)
        if (s >= nsteps) {                                       // (past the range: the step multiplies zeros)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) bf[s][mt] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    few_stamp(a, 1);
    // ---- the stream: unit by unit over this wave's K range
    for (int u = 0; u < nu; ++u) {
        if (u == 1) few_stamp(a, 2);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int un = u0 + min(u + 1, nu - 1);
#pragma unroll
        for (int s = 0; s < KP; ++s) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[s], bf[s][mt], acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            few_wload(ring[s], piece(un, s));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) part[((w * FEW_GNU + u) * MT + mt) * 64 + lane] = acc[mt];
    }
    few_stamp(a, 4);
    __syncthreads();
    few_stamp(a, 5);
    // ---- epilogue: item = (unit of the workgroup, token tile)
    for (int it = w; it < nu * MT; it += FEW_GW) {
        const int u = it / MT, mt = it - u * MT;
        const int m = 16 * mt + li;
        float inv = 1.f;
        if (a.nparts > 0) {
            float tot = 0.f;
#pragma unroll
            for (int ww = 0; ww < FEW_GW; ++ww) tot += ssw[ww * 64 + m];
            inv = rsqrtf(tot / (float)H + a.eps);
        }
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < FEW_GW; ++c) {                        // ascending K ranges: a fixed order
            const f32x4 p = part[((c * FEW_GNU + u) * MT + mt) * 64 + lane];
            v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
        }
        float hq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float g = v[r] * inv;
            const float up = __shfl_xor(g, 32);                   // lanes lg < 2: gate rows 4 lg + r; their up rows sit in lg + 2
            hq[r] = g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)) * up;
        }
        if (lg < 2) {
            uint2 o;
            o.x = pack2(hq[0], hq[1]);
            o.y = pack2(hq[2], hq[3]);
            // feature 8 unit + 4 lg + r of token m -> piece (K step unit / 4, tile mt), lane (unit & 3, li), half lg
            const int unit = u0 + u;
            bf16_t *dst = a.hfrag + ((size_t)((unit >> 2) * MT + mt) * 64 + ((unit & 3) * 16 + li)) * 8 + 4 * lg;
            *reinterpret_cast<uint2 *>(dst) = o;
        }
    }
    few_stamp(a, 7);
}

// ---------------------------------------------------------------------------------------------------------------
// QKV projection with the fragments in registers (H <= 1536): workgroup = one 16-row unit, wave w = K range w, like
// few_gemm_kernel<FEW_QKV> -- but a wave loads the fragments of ITS range (8 K steps x MT tiles) straight into its B operands
// instead of the workgroup staging all of them through LDS first: the first MFMA waits for one wave's 24 KB, not for the
// workgroup's 144 KB and a barrier (few_gemm_kernel's stamps: barrier at 5.5 k cycles of a 9.5 k-cycle workgroup).  The six
// partial tiles meet in LDS in ascending order; epilogue = few_gemm_kernel<FEW_QKV>'s (1 / rms, bias, RoPE, Q | K rows or
// pieces, V^T).   grid: nunits;  static LDS: FEW_GW x MT KiB | ssw[FEW_GW][64]
// ---------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ void __launch_bounds__(64 * FEW_GW) few_qkv8_kernel(FewArgs a) {
    constexpr int KP = FEW_GKP;
    __shared__ f32x4 part[FEW_GW * MT * 64];
    __shared__ float ssw[FEW_GW * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, H = a.H, T = a.T;
    const int unit = (int)blockIdx.x;
    const int kper = (nk + FEW_GW - 1) / FEW_GW;                 // <= KP
    const int k0 = min(nk, w * kper), k1 = min(nk, k0 + kper), nsteps = k1 - k0;
    // the epilogue operands of the tile this wave will finish (wave mt: bias, position -> rotary table rows) ride in front
    f32x4 pre_b = (f32x4){0.f, 0.f, 0.f, 0.f}, pre_c = pre_b, pre_s = pre_b;
    if (w < MT) {
        if (unit < a.rope_blocks) {
            const int bph = a.hd / 16, head = unit / bph, bb = unit - head * bph, half = a.hd / 2;
            const int jf = 8 * bb + 4 * (lg & 1);
            const int ps = a.pos[min(16 * w + li, T - 1)];
            pre_b = *reinterpret_cast<const f32x4 *>(a.bias + head * a.hd + (lg < 2 ? 0 : half) + jf);
            pre_c = *reinterpret_cast<const f32x4 *>(a.cos_t + (size_t)ps * half + jf);
            pre_s = *reinterpret_cast<const f32x4 *>(a.sin_t + (size_t)ps * half + jf);
        } else {
            pre_b = *reinterpret_cast<const f32x4 *>(a.bias + a.qk_cols + 16 * (unit - a.rope_blocks) + 4 * lg);
        }
    }
    constexpr int SQ = 4;
    float sq[SQ];
    if (a.nparts > 0) {
#pragma unroll
        for (int i = 0; i < SQ; ++i) sq[i] = a.ssq[min(w + FEW_GW * i, a.nparts - 1) * FEW_SSQ_LD + lane];
    }
    bf16x8 bf[KP][MT], ring[KP];
#pragma unroll
    for (int s = 0; s < KP; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            bf[s][mt] = *reinterpret_cast<const bf16x8 *>(a.afrag + ((size_t)(min(k0 + s, nk - 1) * MT + mt) * 64 + lane) * 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KP; ++s) few_wload(ring[s], a.W + ((size_t)unit * nk + min(k0 + s, nk - 1)) * 512 + lane * 8);
    if (a.nparts > 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < SQ; ++i) t += w + FEW_GW * i < a.nparts ? sq[i] : 0.f;
        for (int p = w + FEW_GW * SQ; p < a.nparts; p += FEW_GW) t += a.ssq[p * FEW_SSQ_LD + lane];
        ssw[w * 64 + lane] = t;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KP; ++s)
        if (s < nsteps) {                                        // (uniform)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[s], bf[s][mt], acc[mt], 0, 0, 0);
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) part[(w * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    if (w >= MT) return;
    // ---- epilogue: wave mt finishes tile mt
    const int mt = w, m = 16 * mt + li;
    float inv = 1.f;
    if (a.nparts > 0) {
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < FEW_GW; ++ww) tot += ssw[ww * 64 + m];
        inv = rsqrtf(tot / (float)H + a.eps);
    }
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < FEW_GW; ++c) {                            // ascending K ranges: a fixed order
        const f32x4 p = part[(c * MT + mt) * 64 + lane];
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= inv;
    if (unit < a.rope_blocks) {
        const int bph = a.hd / 16, head = unit / bph, bb = unit - head * bph, half = a.hd / 2;
        const int jf = 8 * bb + 4 * (lg & 1);                     // index inside the half
        const int fih = (lg < 2 ? 0 : half) + jf;                 // feature inside the head (r added below)
        float q[4] = {v[0] + pre_b[0], v[1] + pre_b[1], v[2] + pre_b[2], v[3] + pre_b[3]};
        float pp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[r] = __shfl_xor(q[r], 32);
        float o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = lg < 2 ? q[r] * pre_c[r] - pp[r] * pre_s[r] : q[r] * pre_c[r] + pp[r] * pre_s[r];
        if (m < T) {
            uint2 o;
            o.x = pack2(o4[0], o4[1]);
            o.y = pack2(o4[2], o4[3]);
            bf16_t *dst = a.qk + (size_t)m * a.ldqk + head * a.hd + fih;
            if (a.attn_pieces) {                                  // (few_gemm_kernel<FEW_QKV>'s piece addresses)
                const int nkd = a.hd / 32, kd = fih >> 5, sub = (((fih >> 3) & 3) * 16) * 8 + (fih & 4);
                if (head < a.n_heads)
                    dst = a.qk + ((size_t)((head * nkd + kd) * MT + mt) * 64 + li) * 8 + sub;
                else
                    dst = a.qk + ((size_t)a.n_heads * nkd * MT + (size_t)(((head - a.n_heads) * 2 + ((m >> 2) & 1)) * nkd + kd)) * 512 +
                          (4 * (m >> 3) + (m & 3)) * 8 + sub;
            }
            *reinterpret_cast<uint2 *>(dst) = o;
        }
    } else {
        const int vf = 16 * (unit - a.rope_blocks) + 4 * lg;      // V feature (r added below)
        if (m < T) {
            if (a.attn_pieces) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dim = vf + r, kvh = dim / a.hd, wi = dim - kvh * a.hd;
                    a.vt[((size_t)(kvh * (a.hd / 16) + (wi >> 4)) * 64 + ((m >> 3) * 16 + (wi & 15))) * 8 + (m & 7)] = f2bf(v[r] + pre_b[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a.vt[(size_t)(vf + r) * a.ldvt + m] = f2bf(v[r] + pre_b[r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// O projection + residual add: workgroup = 8 output features (a half unit) over ALL of K -- 192 workgroups at H = 1536,
// every one of which finishes its columns of the stream (no cross-workgroup sum).  8 waves = 8 K ranges, half pieces of
// 512 B (lanes l and l + 8 of a 16-lane group fetch the same row: the MFMA's rows 8..15 repeat rows 0..7).  Epilogue:
// x += sum; bf16(x g) fragments and this workgroup's part of the rows' sums of squares for the next RMSNorm.
// dynamic LDS: max(nk, 8) x MT KiB
// ---------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ void __launch_bounds__(64 * FEW_OW) few_o_kernel(FewArgs a) {
    constexpr int U = 6;
    extern __shared__ __attribute__((aligned(16))) uint4 few_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, T = a.T, H = a.H;
    const int hu = (int)blockIdx.x;
    const int kper = (nk + FEW_OW - 1) / FEW_OW;
    const int k0 = min(nk, w * kper), k1 = min(nk, k0 + kper), nsteps = k1 - k0;
    const bf16_t *wp = a.W + ((size_t)hu * nk + min(k0, nk - 1)) * 256 + (lg * 8 + (li & 7)) * 8;

    // the lanes that will finish a tile (wave mt, lg < 2) request their 4 columns of the stream and the gains first
    const int em = 16 * w + li, ef = 8 * hu + 4 * (lg & 1);
    const bool eown = w < MT && lg < 2 && em < T;
    float *px = a.x + (size_t)min(em, T - 1) * H + ef;
    f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f}, gv = xv;
    if (eown) xv = *reinterpret_cast<const f32x4 *>(px);
    if (w < MT) gv = *reinterpret_cast<const f32x4 *>(a.norm_w + ef);
    i32x4_t av[6 * MT];
    few_stage_load<FEW_OW, 6 * MT>(av, a.afrag, nk * MT, w, lane);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 ring[U][1];
    few_ring_start<1, U, 256>(wp, 0, nsteps, ring);
    __builtin_amdgcn_sched_barrier(0);
    few_stage_store<FEW_OW, 6 * MT>(av, a.afrag, few_lds, nk * MT, w, lane);
    __syncthreads();
    f32x4 acc[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[0][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) few_stream<1, MT, U, 256>(wp, 0, nsteps, few_lds + (size_t)k0 * MT * 64 + lane, ring, acc);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) reinterpret_cast<f32x4 *>(few_lds)[(w * MT + mt) * 64 + lane] = acc[0][mt];
    __syncthreads();
    // tile mt is finished by wave mt; lanes lg < 2 hold the 8 real features (4 lg + r), token li
    if (w < MT) {
        const int mt = w, m = 16 * mt + li;
        const f32x4 *part = reinterpret_cast<const f32x4 *>(few_lds);
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < FEW_OW; ++c) {
            const f32x4 p = part[(c * MT + mt) * 64 + lane];
            v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
        }
        float ss = 0.f;                                         // (lanes lg >= 2 repeat lg - 2; they store nothing)
        if (eown) {
            xv[0] += v[0]; xv[1] += v[1]; xv[2] += v[2]; xv[3] += v[3];
            *reinterpret_cast<f32x4 *>(px) = xv;
            ss = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3];
        }
        ss += __shfl_xor(ss, 16);                               // the two halves of the 8 features
        if (lg == 0) a.ssq_out[hu * FEW_SSQ_LD + m] = ss;       // (padding tokens: 0)
        if (lg < 2) {                                           // fragments: every token of the tile (padding tokens: zeros)
            uint2 o;
            o.x = pack2(xv[0] * gv[0], xv[1] * gv[1]);
            o.y = pack2(xv[2] * gv[2], xv[3] * gv[3]);
            // feature ef + r of token m -> piece (K step hu / 4, tile mt), lane (hu & 3, li), half lg
            bf16_t *dst = a.xfrag + ((size_t)((hu >> 2) * MT + mt) * 64 + ((hu & 3) * 16 + li)) * 8 + 4 * lg;
            *reinterpret_cast<uint2 *>(dst) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// few_o_kernel with the fragments in registers (q_cols <= 1536): the 8 waves' K ranges are disjoint, so a wave loads the
// attention fragments of ITS range (6 K steps x MT tiles) straight into its B operands -- no staging of all of them through
// LDS, no barrier in front of the stream (few_qkv8_kernel's reasoning).  Same sums in the same order as few_o_kernel.
// static LDS: 8 x MT KiB
// ---------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ void __launch_bounds__(64 * FEW_OW) few_o8_kernel(FewArgs a) {
    constexpr int KP = 6;
    __shared__ f32x4 part[FEW_OW * MT * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, T = a.T, H = a.H;
    const int hu = (int)blockIdx.x;
    const int kper = (nk + FEW_OW - 1) / FEW_OW;                 // <= KP (the host checks)
    const int k0 = min(nk, w * kper), k1 = min(nk, k0 + kper), nsteps = k1 - k0;
    // the lanes that will finish a tile (wave mt, lg < 2) request their 4 columns of the stream and the gains first
    const int em = 16 * w + li, ef = 8 * hu + 4 * (lg & 1);
    const bool eown = w < MT && lg < 2 && em < T;
    float *px = a.x + (size_t)min(em, T - 1) * H + ef;
    f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f}, gv = xv;
    if (eown) xv = *reinterpret_cast<const f32x4 *>(px);
    if (w < MT) gv = *reinterpret_cast<const f32x4 *>(a.norm_w + ef);
    bf16x8 bf[KP][MT], ring[KP];
#pragma unroll
    for (int s = 0; s < KP; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            bf[s][mt] = *reinterpret_cast<const bf16x8 *>(a.afrag + ((size_t)(min(k0 + s, nk - 1) * MT + mt) * 64 + lane) * 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KP; ++s) few_wload(ring[s], a.W + ((size_t)hu * nk + min(k0 + s, nk - 1)) * 256 + (lg * 8 + (li & 7)) * 8);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KP; ++s)
        if (s < nsteps) {                                        // (uniform)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[s], bf[s][mt], acc[mt], 0, 0, 0);
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) part[(w * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    // tile mt is finished by wave mt; lanes lg < 2 hold the 8 real features (4 lg + r), token li
    if (w < MT) {
        const int mt = w, m = 16 * mt + li;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < FEW_OW; ++c) {
            const f32x4 p = part[(c * MT + mt) * 64 + lane];
            v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
        }
        float ss = 0.f;                                         // (lanes lg >= 2 repeat lg - 2; they store nothing)
        if (eown) {
            xv[0] += v[0]; xv[1] += v[1]; xv[2] += v[2]; xv[3] += v[3];
            *reinterpret_cast<f32x4 *>(px) = xv;
            ss = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3];
        }
        ss += __shfl_xor(ss, 16);                               // the two halves of the 8 features
        if (lg == 0) a.ssq_out[hu * FEW_SSQ_LD + m] = ss;       // (padding tokens: 0)
        if (lg < 2) {                                           // fragments: every token of the tile (padding tokens: zeros)
            uint2 o;
            o.x = pack2(xv[0] * gv[0], xv[1] * gv[1]);
            o.y = pack2(xv[2] * gv[2], xv[3] * gv[3]);
            bf16_t *dst = a.xfrag + ((size_t)((hu >> 2) * MT + mt) * 64 + ((hu & 3) * 16 + li)) * 8 + 4 * lg;
            *reinterpret_cast<uint2 *>(dst) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Down projection: h (T x I) is too long to give every workgroup all of K, so K is split over workgroups and slice s
// writes its partial tile into plane s (plain 16-byte stores; few_row_kernel adds the planes in order).
// grid: (ceil(nunits / 4) unit groups) x nslices, 4 waves; wave w streams unit 4 ug + w over the slice, whose activation
// fragments are staged in LDS once.   dynamic LDS: ks_per_slice x MT KiB
// ---------------------------------------------------------------------------------------------------------------
template <int MT, bool FUSE>
__global__ void __launch_bounds__(256) few_d_kernel(FewArgs a) {
#ifndef FEW_U_D
#define FEW_U_D 14
#endif
    constexpr int U = FEW_U_D;
    extern __shared__ __attribute__((aligned(16))) uint4 few_lds[];   // ks_per_slice x MT KiB of fragments | 16 B: the arrival ticket
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, T = a.T, H = a.H;
    const int sl = (int)blockIdx.x % a.nslices, ug = (int)blockIdx.x / a.nslices;
    const int k0 = sl * a.ks_per_slice, k1 = min(nk, k0 + a.ks_per_slice), nsteps = max(k1 - k0, 1);   // (the host never leaves a slice empty)
    const int unit = ug * 4 + w;
    const bool live = unit < a.nunits;
    unsigned &s_ticket = *reinterpret_cast<unsigned *>(few_lds + (size_t)a.ks_per_slice * MT * 64);   // (all LDS in the dynamic region: Guideline 17)
    const bf16_t *wp = a.W + ((size_t)min(unit, a.nunits - 1) * nk + k0) * 512 + lane * 8;
    const bf16_t *asrc = a.afrag + (size_t)k0 * MT * 512;
    i32x4_t av[7 * MT];                                          // the default slice (28 K steps) in one batch
    few_stage_load<4, 7 * MT>(av, asrc, nsteps * MT, w, lane);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 ring[U][1];
    few_ring_start<1, U, 512>(wp, 0, nsteps, ring);
    __builtin_amdgcn_sched_barrier(0);
    few_stage_store<4, 7 * MT>(av, asrc, few_lds, nsteps * MT, w, lane);
    __syncthreads();
    f32x4 acc[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[0][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) few_stream<1, MT, U, 512>(wp, 0, nsteps, few_lds + lane, ring, acc);
    float *plane = a.part + (size_t)sl * a.T_pad * H;
    if (live) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * mt + li;
            if (m < T) {
                float *pp = plane + (size_t)m * H + unit * 16 + lg * 4;
                if constexpr (FUSE) {
                    // write-through (sc1): the last arriver of this unit group reads the planes back past its L1, no fence on
                    // either side (cdna_hip_programming.md Guideline 16, form R1; the index library's scan merges its slices so)
                    typedef unsigned long long u64;
                    __hip_atomic_store(reinterpret_cast<u64 *>(pp), ((u64)__float_as_uint(acc[0][mt][1]) << 32) | __float_as_uint(acc[0][mt][0]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(reinterpret_cast<u64 *>(pp + 2), ((u64)__float_as_uint(acc[0][mt][3]) << 32) | __float_as_uint(acc[0][mt][2]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *reinterpret_cast<float4 *>(pp) = make_float4(acc[0][mt][0], acc[0][mt][1], acc[0][mt][2], acc[0][mt][3]);
                }
            }
        }
    }
    if constexpr (!FUSE) return;
    // ---- the unit group's LAST slice to arrive completes the group's 64 columns of the stream: planes added in ascending
    // slice order (whoever arrives last: the sum does not depend on it), residual add, and what the next QKV projection
    // needs -- bf16(x g) fragments and this group's part of the rows' sums of squares (1 / rms rides on its accumulators)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(a.ctr + ug, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)(a.nslices - 1)) return;
    if (tid == 0) __hip_atomic_store(a.ctr + ug, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    const int c4 = tid & 15, col = ug * 64 + c4 * 4;            // 16 threads per row: 4 columns each
    const bool colok = col < H;
    const size_t plane_sz = (size_t)a.T_pad * H;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (tid >> 4) + 16 * i;
        const bool own = m < T && colok;
        const size_t o = (size_t)min(m, T - 1) * H + min(col, H - 4);
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < a.nslices; s0 += 5) {             // five planes' loads in flight, added in ascending order
            unsigned long long q[5][2];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const unsigned long long *pp = reinterpret_cast<const unsigned long long *>(a.part + (size_t)min(s0 + u, a.nslices - 1) * plane_sz + o);
                q[u][0] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q[u][1] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 5; ++u)
                if (s0 + u < a.nslices) {
                    s4[0] += __uint_as_float((unsigned)q[u][0]); s4[1] += __uint_as_float((unsigned)(q[u][0] >> 32));
                    s4[2] += __uint_as_float((unsigned)q[u][1]); s4[3] += __uint_as_float((unsigned)(q[u][1] >> 32));
                }
        }
        f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f};
        float ss = 0.f;
        if (own) {
            xv = *reinterpret_cast<const f32x4 *>(a.x + o);
            xv[0] += s4[0]; xv[1] += s4[1]; xv[2] += s4[2]; xv[3] += s4[3];
            *reinterpret_cast<f32x4 *>(a.x + o) = xv;
            ss = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3];
        }
        ss = row16_sum(ss);                                      // the row's 16 threads are one DPP row
        if (a.norm_w) {
            if (c4 == 0 && m < 16 * MT) a.ssq_out[ug * FEW_SSQ_LD + m] = ss;   // (padding tokens: 0)
            if (colok && m < 16 * MT) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(a.norm_w + col);
                uint2 ov;
                ov.x = pack2(xv[0] * g[0], xv[1] * g[1]);
                ov.y = pack2(xv[2] * g[2], xv[3] * g[3]);
                // columns col .. col + 3 of token m -> piece (K step col / 32, tile m / 16), lane ((col % 32) / 8, m % 16), half (col / 4) & 1
                bf16_t *dst = a.xfrag + ((size_t)((col >> 5) * MT + (m >> 4)) * 64 + (((col & 31) >> 3) * 16 + (m & 15))) * 8 + 4 * ((col >> 2) & 1);
                *reinterpret_cast<uint2 *>(dst) = ov;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One workgroup per token: the row of the stream is completed here (embedding lookup, or += the down projection's
// planes in ascending order), so its RMSNorm is too: bf16(x / rms * g) fragments for the next QKV projection.
// norm_w == null (after the last layer): the stream only.   H <= 4096.
// ---------------------------------------------------------------------------------------------------------------
template <bool EMBED, int MT>
__global__ void __launch_bounds__(256) few_row_kernel(FewArgs a) {
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int H = a.H, nch = H / 8;                              // chunks of 8 columns; a thread owns chunks tid, tid + 256
    float v[2][8];
    float ss = 0.f;
    float *xr = a.x + (size_t)t * H;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] = 0.f;
        if (c < nch) {
            if constexpr (EMBED) {
                const uint4 e = *reinterpret_cast<const uint4 *>(a.table + (size_t)a.ids[t] * H + c * 8);
                const unsigned u[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[i][2 * q] = __uint_as_float(u[q] << 16);
                    v[i][2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u);
                }
            } else {
                const float4 x0 = *reinterpret_cast<const float4 *>(xr + c * 8), x1 = *reinterpret_cast<const float4 *>(xr + c * 8 + 4);
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float *p = a.part + (size_t)t * H + c * 8;
                const size_t plane = (size_t)a.T_pad * H;
                for (int s0 = 0; s0 < a.nslices; s0 += 8) {       // eight planes' loads in flight, added in ascending order
                    f32x4 q0[8], q1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float *pp = p + (size_t)min(s0 + u, a.nslices - 1) * plane;
                        q0[u] = *reinterpret_cast<const f32x4 *>(pp);
                        q1[u] = *reinterpret_cast<const f32x4 *>(pp + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (s0 + u < a.nslices) {
                            acc[0] += q0[u][0]; acc[1] += q0[u][1]; acc[2] += q0[u][2]; acc[3] += q0[u][3];
                            acc[4] += q1[u][0]; acc[5] += q1[u][1]; acc[6] += q1[u][2]; acc[7] += q1[u][3];
                        }
                }
                v[i][0] = x0.x + acc[0]; v[i][1] = x0.y + acc[1]; v[i][2] = x0.z + acc[2]; v[i][3] = x0.w + acc[3];
                v[i][4] = x1.x + acc[4]; v[i][5] = x1.y + acc[5]; v[i][6] = x1.z + acc[6]; v[i][7] = x1.w + acc[7];
            }
            *reinterpret_cast<float4 *>(xr + c * 8) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
            *reinterpret_cast<float4 *>(xr + c * 8 + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
#pragma unroll
            for (int q = 0; q < 8; ++q) ss += v[i][q] * v[i][q];
        }
    }
    if (!a.norm_w) return;
    ss = wave_sum(ss);
    if (lane == 0) red[w] = ss;
    __syncthreads();
    const float inv = rsqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)H + a.eps);
    const int mt = t >> 4, li = t & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        if (c < nch) {
            const float4 g0 = *reinterpret_cast<const float4 *>(a.norm_w + c * 8), g1 = *reinterpret_cast<const float4 *>(a.norm_w + c * 8 + 4);
            uint4 o;
            o.x = pack2(v[i][0] * inv * g0.x, v[i][1] * inv * g0.y); o.y = pack2(v[i][2] * inv * g0.z, v[i][3] * inv * g0.w);
            o.z = pack2(v[i][4] * inv * g1.x, v[i][5] * inv * g1.y); o.w = pack2(v[i][6] * inv * g1.z, v[i][7] * inv * g1.w);
            // columns 8 c .. of token t -> piece (K step c / 4, tile mt), lane (c & 3, li)
            *reinterpret_cast<uint4 *>(a.xfrag + ((size_t)((c >> 2) * MT + mt) * 64 + ((c & 3) * 16 + li)) * 8) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention of a few short sequences (every sequence <= 48 tokens: the whole key axis is ONE pass, no online softmax):
// one wave per (head, 16-query tile of a sequence).  Computed transposed like attn_kernel (encoder_kernels.h): S^T = K Q^T
// with MFMA row position p of key tile j standing for key 32 (j / 2) + 8 (p / 4) + 4 (j % 2) + p % 4, so that the lane's
// probabilities of tiles 2 J, 2 J + 1 are the B fragment of the P V product as they stand; O^T = V^T P^T leaves four
// consecutive dims of one query per lane, written as fragments of the O projection's operand (the query-time path) or as
// rows of O (a batch of short sequences on the general path: 16 queries 7.8 -> ~5 us a layer against the persistent kernel).  Q, K rows and V^T rows come
// straight from global memory (all 36 loads of a lane requested at once: the kernel is one latency chain).  The key axis
// starts at the sequence's first token rounded down to 8 (V^T rows are fetched in 16-byte pieces); the < 8 foreign keys in
// front and the keys past the end are masked.
// grid: (n_heads, nwork * 3): blockIdx.y = work item (a sequence) x 16-query tile
// ---------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(64) few_attn_kernel(AttnArgs a) {
    constexpr int NKD = HD / 32, NDT = HD / 16, NJ = 4;            // K steps over the head dim, output tiles, key tiles (64 keys)
    const int h = blockIdx.x, wi = blockIdx.y / 3, qt = blockIdx.y % 3;
    const int seq = a.work_seq[wi], s0 = a.seq_start[seq], L = a.seq_len[seq];
    if (16 * qt >= L) return;
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    const int sa = s0 & ~7, off = s0 - sa, Le = off + L;            // the key axis and the sequence's span on it
    const int kvh = h / (a.n_heads / a.n_kv);
    const int qidx = 16 * qt + li;                                   // this lane's query (inside the sequence)
    // ---- every load of the wave, requested together
    bf16x8 qf[NKD], kf[NJ][NKD], vf[NDT][2];
    const bf16_t *qp = a.QK + (size_t)(s0 + min(qidx, L - 1)) * a.ldqk + h * HD + lg * 8;
#pragma unroll
    for (int kd = 0; kd < NKD; ++kd) qf[kd] = *reinterpret_cast<const bf16x8 *>(qp + kd * 32);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int key = 32 * (j >> 1) + 8 * (li >> 2) + 4 * (j & 1) + (li & 3);
        const bf16_t *kp = a.QK + (size_t)(sa + min(key, Le - 1)) * a.ldqk + (a.n_heads + kvh) * HD + lg * 8;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) kf[j][kd] = *reinterpret_cast<const bf16x8 *>(kp + kd * 32);
    }
#pragma unroll
    for (int n = 0; n < NDT; ++n) {
        const bf16_t *vp = a.Vt + (size_t)(kvh * HD + 16 * n + li) * a.ldvt + sa + lg * 8;
#pragma unroll
        for (int J = 0; J < 2; ++J) vf[n][J] = *reinterpret_cast<const bf16x8 *>(vp + 32 * J);
    }
    // ---- S^T = K Q^T
    f32x4 sc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][kd], qf[kd], sc[j], 0, 0, 0);
    }
    // ---- softmax over the keys of query li: the lane holds keys 32 (j / 2) + 8 lg + 4 (j % 2) + r
    const float scale2 = a.scale * 1.4426950408889634f;
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kidx = 32 * (j >> 1) + 8 * lg + 4 * (j & 1) + r - off;      // key index inside the sequence
            if (kidx < 0 || kidx >= L || (a.causal && kidx > qidx)) sc[j][r] = -__builtin_huge_valf();
            mx = fmaxf(mx, sc[j][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float msub = mx == -__builtin_huge_valf() ? 0.f : mx * scale2;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j][r], scale2, -msub));
            sum += p;
            sc[j][r] = p;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // ---- O^T = V^T P^T
    f32x4 o[NDT];
#pragma unroll
    for (int n = 0; n < NDT; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int J = 0; J < 2; ++J) {
        union { bf16x8 v; unsigned u[4]; } pf;
        pf.u[0] = pack2(sc[2 * J][0], sc[2 * J][1]);
        pf.u[1] = pack2(sc[2 * J][2], sc[2 * J][3]);
        pf.u[2] = pack2(sc[2 * J + 1][0], sc[2 * J + 1][1]);
        pf.u[3] = pack2(sc[2 * J + 1][2], sc[2 * J + 1][3]);
#pragma unroll
        for (int n = 0; n < NDT; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[n][J], pf.v, o[n], 0, 0, 0);
    }
    if (qidx < L) {
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        const int tok = s0 + qidx;
        // dims h HD + 16 n + 4 lg + r -> piece (step (h HD + 16 n) / 32, tile tok / 16), lane (2 (n & 1) + lg / 2, tok % 16), half lg & 1;
        // or (Ofrag null: a batch of short sequences on the general path) row tok of O, 4 consecutive dims per lane and tile
        bf16_t *op = a.Ofrag ? a.Ofrag + (((size_t)(h * (HD / 32)) * a.frag_mt + (tok >> 4)) * 64 + (lg >> 1) * 16 + (tok & 15)) * 8 + 4 * (lg & 1)
                             : a.O + (size_t)tok * (a.n_heads * HD) + h * HD + 4 * lg;
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
            uint2 pk;
            pk.x = pack2(o[n][0] * inv, o[n][1] * inv);
            pk.y = pack2(o[n][2] * inv, o[n][3] * inv);
            const size_t off = a.Ofrag ? ((size_t)(n >> 1) * a.frag_mt * 64 + (n & 1) * 32) * 8 : (size_t)n * 16;
            *reinterpret_cast<uint2 *>(op + off) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention INSIDE the O projection, for ONE sequence of <= 32 tokens (the prompted query of README.md:28): the launch that
// few_attn_kernel was (4.6 us of a 43 us layer for ~6 MFLOP) and the round trip of its output through memory disappear.
// Workgroup = 8 output features over all of K like few_o_kernel; wave w = head w (FEW_AW waves; more heads: w, w + FEW_AW, ..):
// it computes its head's attention for every query tile exactly as few_attn_kernel does (S^T = K Q^T, one pass over <= 32
// keys, O^T = V^T P^T) -- redundantly in every workgroup: Q | K | V^T are 4 KB per token, less than the fragments few_o_kernel
// stages -- and its lane's output registers (four dims of tiles 2 J, 2 J + 1 of one query) ARE the B operand of the O
// projection's K step (head, J) once the weight pieces carry their K columns in that order (few_tile_kernel<8, true>).
// The heads' partial tiles meet in LDS in ascending order; the epilogue is few_o_kernel's.  (33 .. 48 tokens keep the two launches:
// 64 keys and three query tiles are ~180 operand registers -- on 12 waves the build spilled (one query 1.26 -> 1.69 ms), on 6 waves
// of two heads each it measured 1.31 against 1.26.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int FEW_AW = 12;
template <int HD, int MT>
__global__ void __launch_bounds__(64 * FEW_AW) few_ao_kernel(FewArgs a) {
    constexpr int NKD = HD / 32, NDT = HD / 16, NJ = 2;            // K steps of a head, output tiles, key tiles (32 keys)
    static_assert(MT <= 2, "one pass over 32 keys");
    extern __shared__ __attribute__((aligned(16))) uint4 few_lds[];   // n_kv (2 NKD + NDT) KiB: the K and V^T pieces | FEW_AW MT KiB: the heads' partial tiles
    const int PK = a.n_kv * NJ * NKD, PKV = PK + a.n_kv * NDT;
    f32x4 *part = reinterpret_cast<f32x4 *>(few_lds + (size_t)PKV * 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nk = a.nk, T = a.T, H = a.H;
    const int hu = (int)blockIdx.x;
    // the lanes that will finish a tile (wave mt, lg < 2) request their 4 columns of the stream and the gains first
    const int em = 16 * w + li, ef = 8 * hu + 4 * (lg & 1);
    const bool eown = w < MT && lg < 2 && em < T;
    float *px = a.x + (size_t)min(em, T - 1) * H + ef;
    f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f}, gv = xv;
    if (eown) xv = *reinterpret_cast<const f32x4 *>(px);
    if (w < MT) gv = *reinterpret_cast<const f32x4 *>(a.norm_w + ef);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float scale2 = a.scale * 1.4426950408889634f;
    // ---- the K and V^T pieces of every KV head -> LDS, once per workgroup (a head group shares them); the first head's Q and
    // weight pieces requested behind them
    {
        const bf16_t *kbase = a.qk + (size_t)a.n_heads * NKD * MT * 512;
        constexpr int KVB = 3;                                      // pieces per wave requested at once (32 pieces over 12 waves)
        i32x4_t kv[KVB];
#pragma unroll
        for (int i = 0; i < KVB; ++i) {
            const int p = min(w + FEW_AW * i, PKV - 1);
            kv[i] = *reinterpret_cast<const i32x4_t *>((p < PK ? kbase + (size_t)p * 512 : a.vt + (size_t)(p - PK) * 512) + lane * 8);
        }
#pragma unroll
        for (int i = 0; i < KVB; ++i) reinterpret_cast<i32x4_t *>(few_lds)[min(w + FEW_AW * i, PKV - 1) * 64 + lane] = kv[i];
        for (int p = w + FEW_AW * KVB; p < PKV; p += FEW_AW)
            reinterpret_cast<i32x4_t *>(few_lds)[p * 64 + lane] =
                *reinterpret_cast<const i32x4_t *>((p < PK ? kbase + (size_t)p * 512 : a.vt + (size_t)(p - PK) * 512) + lane * 8);
    }
    bf16x8 qf[MT][NKD], wf[NKD];
    auto head_loads = [&](int h) {                                 // the head's Q pieces and its K steps of the weight stream
        h = min(h, a.n_heads - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kd = 0; kd < NKD; ++kd)
                qf[mt][kd] = *reinterpret_cast<const bf16x8 *>(a.qk + ((size_t)((h * NKD + kd) * MT + mt) * 64 + lane) * 8);
        const bf16_t *wp = a.W + ((size_t)hu * nk + h * NKD) * 256 + (lg * 8 + (li & 7)) * 8;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) few_wload(wf[kd], wp + kd * 256);
    };
    head_loads(w);
    __syncthreads();
    for (int h = w; h < a.n_heads; h += FEW_AW) {
        const int kvh = h / (a.n_heads / a.n_kv);
        bf16x8 kf[NJ][NKD], vf[NDT];
        if (h != w) head_loads(h);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int kd = 0; kd < NKD; ++kd) kf[j][kd] = as_bf16x8(few_lds[((kvh * NJ + j) * NKD + kd) * 64 + lane]);
#pragma unroll
        for (int n = 0; n < NDT; ++n) vf[n] = as_bf16x8(few_lds[(PK + kvh * NDT + n) * 64 + lane]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int qidx = 16 * mt + li;
            // ---- S^T = K Q^T; softmax over the keys of query li: the lane holds keys 8 lg + 4 j + r
            f32x4 sc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kd = 0; kd < NKD; ++kd) sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][kd], qf[mt][kd], sc[j], 0, 0, 0);
            }
            float mx = -__builtin_huge_valf();
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kidx = 8 * lg + 4 * j + r;
                    if (kidx >= T || (a.causal && kidx > qidx)) sc[j][r] = -__builtin_huge_valf();
                    mx = fmaxf(mx, sc[j][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float msub = mx == -__builtin_huge_valf() ? 0.f : mx * scale2;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j][r], scale2, -msub));
                    sum += p;
                    sc[j][r] = p;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = sum > 0.f ? 1.0f / sum : 0.f;
            union { bf16x8 v; unsigned u[4]; } pf;
            pf.u[0] = pack2(sc[0][0], sc[0][1]);
            pf.u[1] = pack2(sc[0][2], sc[0][3]);
            pf.u[2] = pack2(sc[1][0], sc[1][1]);
            pf.u[3] = pack2(sc[1][2], sc[1][3]);
            // ---- O^T = V^T P^T, two output tiles at a time: they are one K step of the O projection
#pragma unroll
            for (int kd = 0; kd < NKD; ++kd) {
                f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[2 * kd], pf.v, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[2 * kd + 1], pf.v, o1, 0, 0, 0);
                union { bf16x8 v; unsigned u[4]; } of;
                of.u[0] = pack2(o0[0] * inv, o0[1] * inv);
                of.u[1] = pack2(o0[2] * inv, o0[3] * inv);
                of.u[2] = pack2(o1[0] * inv, o1[1] * inv);
                of.u[3] = pack2(o1[2] * inv, o1[3] * inv);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kd], of.v, acc[mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) part[(w * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    // tile mt is finished by wave mt; lanes lg < 2 hold the 8 real features (4 lg + r), token li
    if (w < MT) {
        const int mt = w, m = 16 * mt + li;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < FEW_AW; ++c) {
            const f32x4 p = part[(c * MT + mt) * 64 + lane];
            v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
        }
        float ss = 0.f;                                         // (lanes lg >= 2 repeat lg - 2; they store nothing)
        if (eown) {
            xv[0] += v[0]; xv[1] += v[1]; xv[2] += v[2]; xv[3] += v[3];
            *reinterpret_cast<f32x4 *>(px) = xv;
            ss = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3];
        }
        ss += __shfl_xor(ss, 16);                               // the two halves of the 8 features
        if (lg == 0) a.ssq_out[hu * FEW_SSQ_LD + m] = ss;       // (padding tokens: 0)
        if (lg < 2) {                                           // fragments: every token of the tile (padding tokens: zeros)
            uint2 o;
            o.x = pack2(xv[0] * gv[0], xv[1] * gv[1]);
            o.y = pack2(xv[2] * gv[2], xv[3] * gv[3]);
            bf16_t *dst = a.xfrag + ((size_t)((hu >> 2) * MT + mt) * 64 + ((hu & 3) * 16 + li)) * 8 + 4 * lg;
            *reinterpret_cast<uint2 *>(dst) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The tail of a query-time pass: mean pooling of the final-norm fragments, Dense, and the row's L2 normalisation.
// few_pool_kernel, grid (nseq, parts): every workgroup pools its sequence (the bf16 normalised hidden states arrive as
// fragments from the last few_row_kernel: T x H x 2 bytes instead of two passes over the f32 stream; four accumulators by
// token mod 4 and a bf16-rounded mean, the rounding points of pool_kernel / meanpool_kernel) and multiplies its share of
// the Dense rows; few_finish_kernel normalises the row and writes it where the caller wants it.   H <= 2048.
// ---------------------------------------------------------------------------------------------------------------
struct FewPoolArgs {
    const bf16_t *xfrag;           // [H / 32][MT] KiB: bf16(x / rms * g_final)
    int MT, H, out_dim, parts;
    const int32_t *seq_start, *seq_len;
    const bf16_t *dense_w;         // [out_dim][H]
    const float *dense_b;          // [out_dim] or null
    float *raw;                    // [nseq][out_dim]
};

__global__ void __launch_bounds__(256) few_pool_kernel(FewPoolArgs a) {
    __shared__ float pooled[2048];
    const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s0 = a.seq_start[seq], L = a.seq_len[seq], H = a.H;
    if (tid < H / 8) {                                           // thread = 8 columns 8 tid ..: piece tid / 4, lane group tid & 3
        float acc[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[u][q] = 0.f;
        const bf16_t *base = a.xfrag + ((size_t)(tid >> 2) * a.MT * 64 + (tid & 3) * 16) * 8;
        for (int t0 = 0; t0 < L; t0 += 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = s0 + min(t0 + u, L - 1);
                v[u] = *reinterpret_cast<const uint4 *>(base + ((size_t)(t >> 4) * 64 + (t & 15)) * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < L) {
                    const unsigned x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[u][2 * q] += __uint_as_float(x[q] << 16);
                        acc[u][2 * q + 1] += __uint_as_float(x[q] & 0xffff0000u);
                    }
                }
        }
        const float invL = 1.0f / (float)max(L, 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) pooled[tid * 8 + q] = bf2f(f2bf(((acc[0][q] + acc[1][q]) + (acc[2][q] + acc[3][q])) * invL));
    }
    __syncthreads();
    const int per = (a.out_dim + a.parts - 1) / a.parts;
    const int j0 = blockIdx.y * per, j1 = min(a.out_dim, j0 + per);
    for (int j = j0 + w * 4; j < j1; j += 16) {                 // four Dense rows per wave step: four weight streams in flight
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 8; c < H; c += 512) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint4 v = *reinterpret_cast<const uint4 *>(a.dense_w + (size_t)min(j + r, j1 - 1) * H + c);
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[r] += pooled[c + 2 * q] * __uint_as_float(u[q] << 16);
                    acc[r] += pooled[c + 2 * q + 1] * __uint_as_float(u[q] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = wave_sum(acc[r]);
            if (lane == 0 && j + r < j1) a.raw[(size_t)seq * a.out_dim + j + r] = t + (a.dense_b ? a.dense_b[j + r] : 0.f);
        }
    }
}

// out[rows[seq]][:] = raw[seq][:] (/ its L2 norm)
__global__ void __launch_bounds__(256) few_finish_kernel(const float *__restrict__ raw, int n, int normalize,
                                                         const int32_t *__restrict__ rows, float *__restrict__ out) {
    __shared__ float red[4];
    const float *r = raw + (size_t)blockIdx.x * n;
    float *d = out + (size_t)(rows ? rows[blockIdx.x] : (int)blockIdx.x) * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float scale = 1.f;
    if (normalize) {
        float ss = 0.f;
        for (int c = tid; c < n; c += 256) ss += r[c] * r[c];
        ss = wave_sum(ss);
        if (lane == 0) red[w] = ss;
        __syncthreads();
        scale = 1.0f / fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);   // torch.nn.functional.normalize eps
    }
    for (int c = tid; c < n; c += 256) d[c] = r[c] * scale;
}

}  // namespace mienc
