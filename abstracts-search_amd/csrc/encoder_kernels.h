// encoder_kernels.h -- gfx950 (CDNA4, wave64) kernels of the encode hot path:
// the Qwen2-style decoder stack of stella_en_1.5B_v5, mean pooling, Dense and
// L2 normalisation (reference call sites: Makefile:65 `sidecar-search build`,
// README.md:28 query-time app; arithmetic restated in oracle/encoder_oracle.py).
//
//   embed_kernel        token ids -> f32 residual stream
//   rmsnorm_kernel      f32 residual -> bf16 GEMM operand (f32 statistics)
//   gemm_bf16_ring_kernel C = A . W^T on v_mfma_f32_16x16x32_bf16, 256x256 (or
//                       128x32) tiles staged by LDS-DMA (global_load_lds_dwordx4)
//                       through a multi-stage XOR-swizzled LDS ring, f32
//                       accumulation, fused epilogues:
//                       QKV (+bias, V written transposed), residual add into the
//                       f32 stream, SwiGLU (gate/up interleaved weight rows)
//   rope_kernel         rotary embedding on Q and K (host-built f32 tables)
//   attn_kernel         varlen flash attention (bidirectional or causal, GQA),
//                       QK^T and PV on MFMA, online softmax in registers
//   pool_kernel         final RMSNorm + mean pooling + Dense + L2 normalise
//
// Tokens are packed; every sequence starts at a multiple of 8 tokens so that
// rows of the transposed V buffer are 16-byte aligned, and T_pad is a multiple
// of 128 (GEMM tile).  Padding tokens compute values no real token uses --
// attention masks keys >= the sequence length, pooling walks only the real
// tokens -- but those values must stay FINITE: a masked key's probability is an
// exact zero that still multiplies its V row in the P.V MFMA (0 x NaN = NaN).
// Hence padding ids embed token 0, the V^T slack and the attention-output
// buffer (whose padding rows no kernel writes) are cleared when allocated.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace mienc {

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two f32 -> packed bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction where the integer
// formulation above takes nine -- the epilogue of a 256x256 tile converts 65 536 values)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return *reinterpret_cast<bf16x8 *>(&v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---------------------------------------------------------------------
// x[t][:] = embed[ids[t]][:]   (bf16 table -> f32 residual stream)
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    embed_kernel(const int32_t *__restrict__ ids, const bf16_t *__restrict__ table, int H, int T,
                 float *__restrict__ x) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const bf16_t *src = table + (size_t)ids[t] * H;
    float *dst = x + (size_t)t * H;
    for (int c = lane * 8; c < H; c += 512) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + c);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float4 a, b;
        a.x = __uint_as_float(w[0] << 16); a.y = __uint_as_float(w[0] & 0xffff0000u);
        a.z = __uint_as_float(w[1] << 16); a.w = __uint_as_float(w[1] & 0xffff0000u);
        b.x = __uint_as_float(w[2] << 16); b.y = __uint_as_float(w[2] & 0xffff0000u);
        b.z = __uint_as_float(w[3] << 16); b.w = __uint_as_float(w[3] & 0xffff0000u);
        *reinterpret_cast<float4 *>(dst + c) = a;
        *reinterpret_cast<float4 *>(dst + c + 4) = b;
    }
}

// ---------------------------------------------------------------------
// The same with the first layer's RMSNorm in the same pass (the wave holds the whole row): x as above and
// y[t][:] = bf16( x[t][:] * rsqrt(mean(x^2) + eps) * w[:] ) -- exactly what rmsnorm_kernel would then compute from x (the
// f32 values are the table's bf16 values widened, the sums run in the same order), without reading the stream back.
// H <= 2048 (8 chunks of 256 columns per lane in registers); the launcher falls back to the two kernels otherwise.
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    embed_norm_kernel(const int32_t *__restrict__ ids, const bf16_t *__restrict__ table, int H, int T, float *__restrict__ x,
                      const float *__restrict__ w, float eps, bf16_t *__restrict__ y) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const bf16_t *src = table + (size_t)ids[t] * H;
    float *dst = x + (size_t)t * H;
    bf16_t *yd = y + (size_t)t * H;
    constexpr int MAXC = 8;
    float4 v[MAXC], g[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = min(lane * 4 + 256 * i, H - 4);         // clamped: unconditional loads (rmsnorm_kernel's column map)
        const uint2 raw = *reinterpret_cast<const uint2 *>(src + c);
        v[i].x = __uint_as_float(raw.x << 16); v[i].y = __uint_as_float(raw.x & 0xffff0000u);
        v[i].z = __uint_as_float(raw.y << 16); v[i].w = __uint_as_float(raw.y & 0xffff0000u);
        g[i] = *reinterpret_cast<const float4 *>(w + c);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
        if (lane * 4 + 256 * i < H) ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            *reinterpret_cast<float4 *>(dst + c) = v[i];
            uint2 o;
            o.x = pack2(v[i].x * inv * g[i].x, v[i].y * inv * g[i].y);
            o.y = pack2(v[i].z * inv * g[i].z, v[i].w * inv * g[i].w);
            *reinterpret_cast<uint2 *>(yd + c) = o;
        }
    }
}

// ---------------------------------------------------------------------
// y[t][:] = bf16( x[t][:] * rsqrt(mean(x^2) + eps) * w[:] ), one wave per token
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    rmsnorm_kernel(const float *__restrict__ x, const float *__restrict__ w, int H, int T, float eps,
                   bf16_t *__restrict__ y) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const float *src = x + (size_t)t * H;
    bf16_t *dst = y + (size_t)t * H;
    constexpr int MAXC = 8;   // rows up to 2048 floats stay in registers: one pass over x, every
                              // load of the row (and of the gains) in flight at once
    if (H <= 256 * MAXC) {
        float4 v[MAXC], g[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = min(lane * 4 + 256 * i, H - 4);   // clamped: unconditional loads
            v[i] = *reinterpret_cast<const float4 *>(src + c);
            g[i] = *reinterpret_cast<const float4 *>(w + c);
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (lane * 4 + 256 * i < H) ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        ss = wave_sum(ss);
        const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < H) {
                uint2 o;
                o.x = pack2(v[i].x * inv * g[i].x, v[i].y * inv * g[i].y);
                o.y = pack2(v[i].z * inv * g[i].z, v[i].w * inv * g[i].w);
                *reinterpret_cast<uint2 *>(dst + c) = o;
            }
        }
        return;
    }
    float ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(src + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)H + eps);
    for (int c = lane * 4; c < H; c += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(src + c);
        const float4 g = *reinterpret_cast<const float4 *>(w + c);
        uint2 o;
        o.x = pack2(v.x * inv * g.x, v.y * inv * g.y);
        o.y = pack2(v.z * inv * g.z, v.w * inv * g.w);
        *reinterpret_cast<uint2 *>(dst + c) = o;
    }
}

// ---------------------------------------------------------------------
// bf16 GEMM, C[M][N] = A[M][K] . W[N][K]^T, f32 accumulation.
// 128x128 workgroup tile, 4 waves (2x2) of 64x64, K step 64, double-buffered
// LDS filled by LDS-DMA; LDS rows are 128 B with the 16-byte slot index XORed
// with (row & 7) (applied on the global source address, the DMA itself is
// lane-linear), which makes the ds_read_b128 fragment reads conflict-free.
// Requires K % 64 == 0, lda/ldw % 8 == 0, 16-byte aligned bases.
// ---------------------------------------------------------------------
// LDS-DMA of 16 bytes per lane: LDS address = wave-uniform base + lane*16.
// Issued through inline asm on purpose: hipcc tracks the builtin form as an LDS
// write and inserts `s_waitcnt vmcnt(0)` before the next ds_read of the same
// array, which drains the whole prefetch ring every K step (measured: every
// variant of the GEMM stuck at ~700 TFLOP/s).  With the DMA invisible to the
// compiler the waits are ours: counted `s_waitcnt vmcnt(N)` + s_barrier.
__device__ __forceinline__ void dma16(const void *gptr, void *lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n" ::"s"(m0v), "v"(gptr) : "memory");
}

// Tile order.  Block b runs on XCD b % 8 (each XCD has its own 4 MiB L2), so the
// 8 XCDs each take a contiguous eighth of the tile sequence, and the sequence
// itself is "grouped": GM consecutive M tiles are walked together along N, so
// the ~32 tiles an XCD has in flight form a GM x 4 patch that shares GM A row
// blocks and 4 W strips in L2 instead of streaming a different W strip per tile.
//
// order 1 ("chip patches", big tiles): what the 256 CUs have in flight at one time is ONE 16 x 16 patch of tiles, of
// which XCD x holds the 8 (M) x 4 (N) sub-patch (x & 1, x >> 1) -- block b of a full patch is tile j = b % 256 with
// XCD j % 8 and slot j / 8 inside the sub-patch.  The per-XCD L2 sharing is that of order 0, but the eight XCDs now
// fetch the SAME 16 A row blocks and 16 W strips at about the same time instead of eight unrelated sets, and
// consecutive patches walk down M under the same W strips: what misses an L2 is then mostly in the memory-side cache.
// Full patches first (256 tiles each: XCD alignment holds), then the right edge (N remainder) and the bottom edge
// (M remainder) in grouped order.
__device__ __forceinline__ bool tile_coords_patch(int tiles_m, int tiles_n, int &tm, int &tn, int b) {
    const int ntiles = tiles_m * tiles_n;
    if (b >= ntiles) return false;
    const int fm = tiles_m >> 4, fn = tiles_n >> 4, rm = tiles_m & 15, rn = tiles_n & 15;
    const int nfull = fm * fn * 256;
    if (b < nfull) {
        const int p = b >> 8, j = b & 255, x = j & 7, i = j >> 3;
        const int pn = p / fm, pm = p - pn * fm;
        tm = pm * 16 + (x & 1) * 8 + (i & 7);
        tn = pn * 16 + (x >> 1) * 4 + (i >> 3);
        return true;
    }
    int r = b - nfull;
    const int right = fm * 16 * rn;                      // rows [0, 16 fm) x columns [16 fn, tiles_n)
    if (r < right) {
        const int g = r / (16 * rn), within = r - g * (16 * rn);
        tn = fn * 16 + within / 16;
        tm = g * 16 + (within & 15);
        return true;
    }
    r -= right;                                          // rows [16 fm, tiles_m) x all columns
    tn = r / rm;
    tm = fm * 16 + (r - tn * rm);
    return true;
}

__device__ __forceinline__ bool tile_coords(int tiles_m, int tiles_n, int &tm, int &tn, int vb = -1, int order = 0) {
    constexpr int GM = 8;
    const int ntiles = tiles_m * tiles_n;
    const int bid = vb < 0 ? (int)blockIdx.x : vb;
    if (order == 1) return tile_coords_patch(tiles_m, tiles_n, tm, tn, bid);
    const int per = (ntiles + 7) / 8;
    const int t = (bid & 7) * per + (bid >> 3);
    if (t >= ntiles || (bid >> 3) >= per) return false;
    const int group = t / (GM * tiles_n), within = t - group * (GM * tiles_n);
    const int gm = min(GM, tiles_m - group * GM);
    tn = within / gm;
    tm = group * GM + (within - tn * gm);
    return true;
}

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_SWIGLU = 3,
       EPI_F32H = 4,     // f16 (not bf16) operands, plain f32 store to X: the index library's approximate score GEMM
       EPI_PART = 5 };   // K split over workgroups, slice ks stores its partial tile (plain f32) into plane ks of `part`; a
                         // reduction kernel adds the planes in order and finishes the epilogue (splitk_reduce_norm_kernel: residual
                         // add + next RMSNorm; splitk_reduce_qkv_kernel: bias + RoPE + Q|K rows + V^T) -- the few-hundred-token path

struct GemmArgs {
    const bf16_t *A;   // [M][lda]
    const bf16_t *W;   // [N][ldw]
    const bf16_t *Wt;  // fragment-major copy of W (tile_weights_kernel) or null: piece (16-row block nb,
                       // K step ks) is the 1 KiB at ((nb * K/32) + ks) * 512 elements, already in the
                       // lane order the LDS-DMA wants -- a 16-row block's K run is one contiguous stream
    int lda, ldw, M, N, K;
    const float *bias; // [N] or null
    bf16_t *C;         // STORE / SWIGLU / QKV(q,k part): bf16 [M][ldc]
    float *X;          // RESID: f32 residual stream [M][ldc], updated in place
    bf16_t *Vt;        // QKV: V^T bf16 [N - qk_cols][ldvt]
    int ldc, ldvt, qk_cols;
    int tiles_m, tiles_n;
    int ksplit;        // > 1 (EPI_RESID only): K range split over workgroups, f32 atomic adds into X
    // EPI_RESID, big tiles: the last, partly filled round of workgroups (blockIdx >=
    // tail_first) is split tail_split ways along K so that it occupies the whole chip
    // for 1/tail_split of a tile time instead of a few CUs for a full one
    int tail_first, tail_split;
    int order;         // tile order of the 256x256 slab kernel: 0 = per-XCD grouped eighths, 1 = chip patches (tile_coords), 2 = every tile K-split, slice-major eighths per XCD
    size_t part_bytes; // capacity of `part`
    float *part;       // EPI_RESID slab kernel with tail_first == 0 (EVERY tile split tail_split ways along K: too few tiles to fill
                       // the chip): slice ks writes its partial tile to part[ks][M][N] (plain whole-line stores, no atomics) and
                       // splitk_reduce_kernel adds the slices to X in ascending order -- bit-reproducible, unlike atomics
    unsigned long long *ts;   // null, or [ts_rows][8] s_memtime stamps (MI_GEMM_TS=1 profile launch): row = workgroup; a persistent
    int ts_rows;              // workgroup's i-th unit writes row blockIdx + i x gridDim
    float *gmax;       // EPI_F32H on the 8-wave slab kernel: null, or [M][ld_gmax] maxima of every 64-column group of a
    int ld_gmax;       // score row (+inf if the group holds a non-finite score) -- what select_refine_kernel's cut needs
    // host side only (EPI_RESID): the RMSNorm that reads the updated stream next -- y = bf16(X * rsqrt(mean(X^2) + eps) * w).
    // A launch that ends in the split-K reduction pass folds it into that pass (splitk_reduce_norm_kernel) and says so
    // (launch_gemm returns true); any other launch leaves it to the caller's rmsnorm_kernel.
    const float *norm_w;
    bf16_t *norm_y;
    float norm_eps;
    // host side only (EPI_QKV): what the reduction pass of a K-split QKV projection needs to finish the epilogue
    // (splitk_reduce_qkv_kernel: RoPE in the same pass) -- null pos: the caller runs rope_kernel itself
    const int32_t *rope_pos;
    const float *rope_cos, *rope_sin;
    int rope_hd;
    // 256 x 256 slab kernel, many-token path: the RMSNorm between a residual GEMM and the GEMM that reads the stream is split
    // over the two epilogues, and the rotary embedding rides in the QKV epilogue (no rmsnorm_kernel / rope_kernel launches):
    //   EPI_RESID, ssq_out set: besides X, the epilogue writes norm_y = bf16(X * norm_w) (NOT yet scaled by 1/rms) and, per
    //     row, the sum of squares of its 16*WNT columns into ssq_out[slot][M], slot = (first column) / (16*WNT) < SSQ_LD;
    //     row_rms_kernel turns the slots into 1/rms per row (summing them inside the consumer cost its every tile a second
    //     memory round trip before the first K step: +47 us on the gate/up GEMM of 28 k tokens);
    //   EPI_QKV / EPI_SWIGLU, row_scale set: A is that unscaled row; the accumulator rows are multiplied by row_scale[row]
    //     (= rsqrt(mean(X^2) + eps)) before bias / SwiGLU -- a scale per row commutes with the GEMM;
    //   EPI_QKV, rope_cs set: W's Q and K rows are interleaved inside each head (row 2f + h = rotary pair f, half h) so that a
    //     lane's 4 consecutive columns hold two whole pairs; rope_cs[pos][f] = (cos, sin).  Q and K leave in the interleaved
    //     order -- the attention scores are dot products over the head, indifferent to a permutation applied to both.
    float *ssq_out;
    int krot;              // one-round launches (slab kernel, encoder_mid.h): row tile tm walks its K tiles from tm x krot round the end
                           // instead of from 0 (-1: from tm x tiles / tiles_m).  The row tiles of a column strip run side by side on one
                           // XCD and otherwise ask for the same W line at the same moment, each waiting out the miss; a few K tiles apart,
                           // one takes the miss and the others find the line in L2.  Worth 1.6 % of a 563-token forward pass -- a CU's
                           // stream is capped elsewhere: ~32 KB of lines in flight whatever the ring holds (tools/micro/w_stream.hip:
                           // 36 GB/s per CU from HBM, 57 from the Infinity Cache, 64 or 128 KiB requested ahead alike)
    float *rms_out;        // encoder_mid.h, residual epilogue with ssq_out: the last tile of a row block to arrive writes 1/rms of
    unsigned *arrive;      // the block's rows here (arrive[row block] counts the tiles; zero before and after a launch)
    const float *row_scale;
    const float2 *rope_cs;
    // PERSIST launches of the slab kernel (one workgroup per CU walks its tiles; see gemm_bf16_slab_kernel): the tiles that do not
    // fill a last round of workgroups are dealt out as K ranges (stream-K).  A range that does not start at K = 0 leaves its
    // partial accumulators in sk_part[blockIdx.x][BM x BN] (register-major, write-through) and counts itself in on
    // sk_ctr[XCD x workgroups-per-XCD + tile]; the workgroup that holds the tile's head adds the partials in range order and runs
    // the epilogue.  *sk_err counts polls that gave up (a bounded spin: never a hang; one word per device, read by tests through
    // mi_enc_debug_counter("sk_giveups")); null sk_part: no stream-K, the remainder is a partly filled round.
    float *sk_part;
    unsigned *sk_ctr, *sk_err;
};
constexpr int SK_CTR_WORDS = 2048;   // GemmArgs::sk_ctr: 8 x 256 tile counters
constexpr int SSQ_LD = 24;   // partial sums of squares per row (1536 / 64)

// W [N][ldw] (N % 16 == 0, K % 32 == 0) -> fragment-major Wt: one 64-thread workgroup per
// (16-row block, K step); lane i carries what the ring kernel's DMA lane i would fetch:
// row 16 nb + (i >> 2), 16-byte slot (i & 3) ^ ((-(i >> 4)) & 3) of the 64-byte K step.
__global__ void __launch_bounds__(64) tile_weights_kernel(const bf16_t *__restrict__ W, int N, int K, int ldw,
                                                          bf16_t *__restrict__ Wt) {
    const int nk = K / 32, piece = blockIdx.x, nb = piece / nk, ks = piece - nb * nk, i = threadIdx.x;
    const int row = nb * 16 + (i >> 2), slot = (i & 3) ^ ((0 - (i >> 4)) & 3);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < N) v = *reinterpret_cast<const uint4 *>(W + (size_t)row * ldw + ks * 32 + slot * 8);
    *reinterpret_cast<uint4 *>(Wt + (size_t)piece * 512 + i * 8) = v;
}

// inv[row] = rsqrt(sum_s ssq[s][row] / dim + eps), s ascending: the second half of the RMSNorm a residual slab epilogue began
// (GemmArgs::ssq_out -> row_scale).  One thread per row; a slot's loads are consecutive rows.
__global__ void __launch_bounds__(256) row_rms_kernel(const float *__restrict__ ssq, int nslots, int M, int dim, float eps, float *__restrict__ inv) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    float v[SSQ_LD];
#pragma unroll
    for (int q = 0; q < SSQ_LD; ++q) v[q] = ssq[(size_t)min(q, nslots - 1) * M + r];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < SSQ_LD; ++q)
        if (q < nslots) s += v[q];
    inv[r] = rsqrtf(s / (float)dim + eps);
}

// Wr / br = W / bias with the Q and K rows interleaved inside each head: row 2 f + h of a head <- row h * hd/2 + f (rotary
// pair f, half h); the V rows (>= qk_rows) copied.  One 16-byte chunk per thread.  (GemmArgs::rope_cs)
__global__ void __launch_bounds__(256) interleave_qk_rows_kernel(const bf16_t *__restrict__ W, const float *__restrict__ bias, int qk_rows,
                                                                 int nrows, int K, int hd, bf16_t *__restrict__ Wr, float *__restrict__ br) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = K / 8;
    if (gi >= (int64_t)nrows * cpr) return;
    const int n = (int)(gi / cpr), ch = (int)(gi - (int64_t)n * cpr);
    int src = n;
    if (n < qk_rows) {
        const int head = n / hd, p = n - head * hd;
        src = head * hd + (p & 1) * (hd / 2) + (p >> 1);
    }
    *reinterpret_cast<uint4 *>(Wr + (size_t)n * K + ch * 8) = *reinterpret_cast<const uint4 *>(W + (size_t)src * K + ch * 8);
    if (ch == 0) br[n] = bias[src];
}

// 4x4 transpose across the 4 lanes of a quad (DPP quad_perm, no LDS): before,
// lane b of the quad holds C[4*lg + r][4a + b] in v[r]; after, it holds
// C[4*lg + b][4a + c] in v[c] -- four consecutive columns of one row, so the
// epilogue stores 8 or 16 bytes per lane instead of four scattered elements.
__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
}
// All-reduce over the 16 lanes of a DPP row in four VALU steps (no LDS crossbar):
// quad butterflies, then row_half_mirror (lane i <- 7 - i within each 8: the other quad's
// result) and row_mirror (lane i <- 15 - i: the other half's result).
__device__ __forceinline__ float dpp_row_half_mirror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_row_mirror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_quad_xor1(v));
    v = fmaxf(v, dpp_quad_xor2(v));
    v = fmaxf(v, dpp_row_half_mirror(v));
    return fmaxf(v, dpp_row_mirror(v));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_quad_xor1(v);
    v += dpp_quad_xor2(v);
    v += dpp_row_half_mirror(v);
    return v + dpp_row_mirror(v);
}
__device__ __forceinline__ f32x4 quad_transpose(f32x4 v, int lane) {
    const bool odd = lane & 1, hi = lane & 2;
    float r0 = dpp_quad_xor1(odd ? v[0] : v[1]);
    float r1 = dpp_quad_xor1(odd ? v[2] : v[3]);
    if (odd) { v[0] = r0; v[2] = r1; } else { v[1] = r0; v[3] = r1; }
    r0 = dpp_quad_xor2(hi ? v[0] : v[2]);
    r1 = dpp_quad_xor2(hi ? v[1] : v[3]);
    if (hi) { v[0] = r0; v[1] = r1; } else { v[2] = r0; v[3] = r1; }
    return v;
}

// Epilogue of 4 consecutive rows x 1 column held by a lane (the lanes of a quad
// hold 4 consecutive columns of the same rows); two values for SwiGLU (gate, up).
// row0 = first of the lane's 4 rows, col = the lane's column in C's column space.
template <int EPI>
__device__ __forceinline__ void store_rows4(const GemmArgs &g, f32x4 v, f32x4 v2, int row0, int col, int lane) {
    if constexpr (EPI == EPI_QKV) {
        if (col >= g.qk_cols) {  // V: 4 consecutive tokens of one channel -> one 8-byte store (V^T layout)
            if (col < g.N && row0 < g.M) {
                const float bv = g.bias ? g.bias[col] : 0.f;
                uint2 o;
                o.x = pack2(v[0] + bv, v[1] + bv);
                o.y = pack2(v[2] + bv, v[3] + bv);
                *reinterpret_cast<uint2 *>(g.Vt + (size_t)(col - g.qk_cols) * g.ldvt + row0) = o;
            }
            return;
        }
    }
    const int row = row0 + (lane & 3), col0 = col & ~3;
    v = quad_transpose(v, lane);
    if constexpr (EPI == EPI_SWIGLU) {
        v2 = quad_transpose(v2, lane);
        if (row < g.M && col0 < g.ldc) {
            float h[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) h[c] = v[c] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[c])) * v2[c];   // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the result is rounded to bf16
            uint2 o;
            o.x = pack2(h[0], h[1]);
            o.y = pack2(h[2], h[3]);
            *reinterpret_cast<uint2 *>(g.C + (size_t)row * g.ldc + col0) = o;
        }
    } else {
        if (row >= g.M || col0 >= g.N) return;
        if constexpr (EPI == EPI_F32H) {
            *reinterpret_cast<float4 *>(g.X + (size_t)row * g.ldc + col0) = make_float4(v[0], v[1], v[2], v[3]);
            return;
        }
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) b = *reinterpret_cast<const float4 *>(g.bias + col0);
        if constexpr (EPI == EPI_RESID) {
            if (g.ksplit > 1) {  // several workgroups add their K slice into the same element
                float *px = g.X + (size_t)row * g.ldc + col0;
                unsafeAtomicAdd(px + 0, v[0] + b.x);
                unsafeAtomicAdd(px + 1, v[1] + b.y);
                unsafeAtomicAdd(px + 2, v[2] + b.z);
                unsafeAtomicAdd(px + 3, v[3] + b.w);
            } else {
                float4 *px = reinterpret_cast<float4 *>(g.X + (size_t)row * g.ldc + col0);
                float4 x = *px;
                x.x += v[0] + b.x; x.y += v[1] + b.y; x.z += v[2] + b.z; x.w += v[3] + b.w;
                *px = x;
            }
        } else {
            uint2 o;
            o.x = pack2(v[0] + b.x, v[1] + b.y);
            o.y = pack2(v[2] + b.z, v[3] + b.w);
            *reinterpret_cast<uint2 *>(g.C + (size_t)row * g.ldc + col0) = o;
        }
    }
}

// 16x16 accumulator tile: lane holds rows (lane>>4)*4 + r of column lane&15.
// trow / tcol = first row / column of the tile (tcol in C's column space).
template <int EPI>
__device__ __forceinline__ void store_tile(const GemmArgs &g, f32x4 v, f32x4 v2, int trow, int tcol, int lane) {
    store_rows4<EPI>(g, v, v2, trow + (lane >> 4) * 4, tcol + (lane & 15), lane);
}

// The same epilogues for a tile accumulated with the MFMA operands swapped (acc = W_tile . A_tile^T,
// the transpose of the output tile): lane (li = lane & 15, lg = lane >> 4) then holds
// C[trow + li][tcol + 4 lg + r], r = 0..3 -- four consecutive columns of one row, so the 8- or
// 16-byte store needs no cross-lane transpose at all (the quad transposes were ~40 % of a 6 us
// epilogue: s_memtime stamps, tools/gemm_stamps.py).  Not for the V part of the QKV projection,
// whose output is column-major: that epilogue keeps the untransposed tile.
template <int EPI>
__device__ __forceinline__ void store_tile_t(const GemmArgs &g, f32x4 v, f32x4 v2, int trow, int tcol, int lane) {
    static_assert(EPI != EPI_QKV, "the QKV epilogue keeps the untransposed accumulator layout");
    const int row = trow + (lane & 15), col0 = tcol + (lane >> 4) * 4;
    if constexpr (EPI == EPI_SWIGLU) {
        if (row < g.M && col0 < g.ldc) {
            float h[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) h[c] = v[c] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[c])) * v2[c];   // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the result is rounded to bf16
            uint2 o;
            o.x = pack2(h[0], h[1]);
            o.y = pack2(h[2], h[3]);
            *reinterpret_cast<uint2 *>(g.C + (size_t)row * g.ldc + col0) = o;
        }
    } else {
        if (row >= g.M || col0 >= g.N) return;
        if constexpr (EPI == EPI_F32H || EPI == EPI_PART) {
            *reinterpret_cast<float4 *>(g.X + (size_t)row * g.ldc + col0) = make_float4(v[0], v[1], v[2], v[3]);
            return;
        }
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) b = *reinterpret_cast<const float4 *>(g.bias + col0);
        if constexpr (EPI == EPI_RESID) {
            if (g.ksplit > 1) {  // several workgroups add their K slice into the same element
                float *px = g.X + (size_t)row * g.ldc + col0;
                unsafeAtomicAdd(px + 0, v[0] + b.x);
                unsafeAtomicAdd(px + 1, v[1] + b.y);
                unsafeAtomicAdd(px + 2, v[2] + b.z);
                unsafeAtomicAdd(px + 3, v[3] + b.w);
            } else {
                float4 *px = reinterpret_cast<float4 *>(g.X + (size_t)row * g.ldc + col0);
                float4 x = *px;
                x.x += v[0] + b.x; x.y += v[1] + b.y; x.z += v[2] + b.z; x.w += v[3] + b.w;
                *px = x;
            }
        } else {
            uint2 o;
            o.x = pack2(v[0] + b.x, v[1] + b.y);
            o.y = pack2(v[2] + b.z, v[3] + b.w);
            *reinterpret_cast<uint2 *>(g.C + (size_t)row * g.ldc + col0) = o;
        }
    }
}

template <int EPI>
__global__ void __launch_bounds__(256) gemm_bf16_nt_kernel(GemmArgs g) {
    constexpr int BM = 128, BN = 128, BK = 64;
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BM + BN) * BK];  // 64 KiB
    bf16_t *As = smem;                    // [2][BM][BK]
    bf16_t *Bs = smem + 2 * BM * BK;      // [2][BN][BK]

    int tm, tn;
    if (!tile_coords(g.tiles_m, g.tiles_n, tm, tn)) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int li = lane & 15, lg = lane >> 4;

    // staging: per operand a wave issues 4 DMA instructions of 8 rows x 128 B
    const int srow = lane >> 3;                       // row inside the 8-row piece
    const int scol = ((lane & 7) ^ srow) * 8;         // swizzled source column (elements)
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r0 = (w * 4 + i) * 8;
            const int ra = min(m0 + r0 + srow, g.M - 1);
            const int rb = min(n0 + r0 + srow, g.N - 1);
            dma16(g.A + (size_t)ra * g.lda + k0 + scol, As + buf * BM * BK + r0 * BK);
            dma16(g.W + (size_t)rb * g.ldw + k0 + scol, Bs + buf * BN * BK + r0 * BK);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
        const bf16_t *ab = As + (kt & 1) * BM * BK + (wm * 64 + li) * BK;
        const bf16_t *bb = Bs + (kt & 1) * BN * BK + (wn * 64 + li) * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = ((kk * 4 + lg) ^ (li & 7)) * 8;  // row & 7 == li & 7 (tile rows are multiples of 16)
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = as_bf16x8(*reinterpret_cast<const uint4 *>(ab + i * 16 * BK + slot));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = as_bf16x8(*reinterpret_cast<const uint4 *>(bb + j * 16 * BK + slot));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's DMA (issued above) has landed
        __syncthreads();
    }

    // epilogue: 16x16 tiles, quad-transposed so that a lane stores 4 consecutive columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int trow = m0 + wm * 64 + i * 16;
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < 4; j += 2)
                store_tile<EPI>(g, acc[i][j], acc[i][j + 1], trow, (n0 + wn * 64) / 2 + (j / 2) * 16, lane);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) store_tile<EPI>(g, acc[i][j], acc[i][j], trow, n0 + wn * 64 + j * 16, lane);
        }
    }
}

// ---------------------------------------------------------------------
// Ring-pipelined GEMM (the one the encoder runs).  Workgroup tile
// BM x BN = (16*WMT*WAVES_M) x (16*WNT*WAVES_N), K step 32, an ST-stage LDS ring
// filled by LDS-DMA ST-1 tiles ahead; fragments of tile t+1 are read into a
// second register set while tile t is multiplied.  Waits are counted
// (`s_waitcnt vmcnt(N)`: only the tile needed next must have landed) and the
// barrier is the raw s_barrier, so the DMA of later tiles stays in flight across
// it.  LDS rows are 64 B; 16-byte slot index XORed with (-(row >> 2)) & 3.
//   <EPI, 8, 4, 2, 4, 4>  256x256, 8 waves, 128 KiB ring: large token counts
//   <EPI, 4, 2, 2, 1, 8>  128x32,  2 waves,  80 KiB ring, 7 tiles in flight:
//                         small batches (a single query is weight-streaming
//                         bound: many narrow workgroups, deep prefetch)
// Requires K % 32 == 0; rows are clamped, stores guarded.
// ---------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `tiles` tiles (PPW DMA instructions each) are still in flight
template <int PPW, int MAXT>
__device__ __forceinline__ void wait_tiles(int tiles, bool lgkm) {
    static_assert(MAXT * PPW <= 63, "vmcnt immediate is 6 bits");
#define MIENC_CASE(T)                                       \
    if constexpr (T <= MAXT)                                \
        if (tiles == T) {                                   \
            if (lgkm) wait_vm_lgkm0<T * PPW>();             \
            else wait_vm<T * PPW>();                        \
            return;                                         \
        }
    MIENC_CASE(0) MIENC_CASE(1) MIENC_CASE(2) MIENC_CASE(3) MIENC_CASE(4) MIENC_CASE(5) MIENC_CASE(6)
#undef MIENC_CASE
    if (lgkm) wait_vm_lgkm0<0>();
    else wait_vm<0>();
}

// (4-wave workgroups ask for two waves per SIMD: without it hipcc spreads the accumulators over
// arch + acc registers -- 316 for the 128x256 tile -- and a second workgroup never fits the CU)
// PERSIST: the grid is one workgroup per CU and a workgroup walks its tiles (virtual block ids
// blockIdx.x + i * gridDim.x: the same XCD and the same tile window per round as the one-tile-
// per-workgroup launch).  The first D K tiles of the NEXT tile are requested before the
// epilogue of the current one, so workgroup launch, pipeline fill and the epilogue's stores --
// ~15 us of a ~50 us K = 1536 tile when every tile is its own workgroup -- overlap.
template <int EPI, int WMT, int WNT, int WAVES_M, int WAVES_N, int ST, bool PERSIST = false>
__global__ void __launch_bounds__(64 * WAVES_M *WAVES_N, (WAVES_M * WAVES_N == 4 ? 2 : 1)) gemm_bf16_ring_kernel(GemmArgs g) {
    constexpr int BM = 16 * WMT * WAVES_M, BN = 16 * WNT * WAVES_N, BK = 32;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int PA = BM / 16, PB = BN / 16;       // 1-KiB DMA pieces (16 rows x 64 B) per stage
    static_assert((PA + PB) % NW == 0, "DMA pieces must divide evenly over the waves");
    constexpr int PPW = (PA + PB) / NW;             // pieces per wave per stage
    // prefetch distance in tiles.  The fragments of tile t are in registers a step before they are
    // multiplied, so after the barrier at the top of step t the stage of tile t is already free
    // and tile t + ST may be requested into it (D = ST).  The power-of-two rings keep the distance
    // ST - 1 they were tuned with; the 3-stage ring (72 KiB: two workgroups per CU) uses ST.
    constexpr int D = ((ST & (ST - 1)) == 0) ? ST - 1 : ST;
    static_assert(ST >= 3, "ring size must be >= 3");
    static_assert(EPI != EPI_SWIGLU || WNT % 2 == 0, "SwiGLU pairs gate/up tiles inside a wave");
    constexpr bool SWAP = EPI != EPI_QKV;           // accumulate the transposed tile (store_tile_t)
    __shared__ __attribute__((aligned(16))) bf16_t smem[ST * (BM + BN) * BK];

    // split-K (small batches, residual GEMMs): workgroup = (tile, K slice)
    int ksplit = 1, ks = 0, vb = -1;
    if constexpr (EPI == EPI_RESID || EPI == EPI_PART) {
        if (g.ksplit > 1 || EPI == EPI_PART) {
            ksplit = g.ksplit;
            ks = (int)blockIdx.x % ksplit;
            vb = (int)blockIdx.x / ksplit;
        } else if (g.tail_split > 1 && (int)blockIdx.x >= g.tail_first) {
            const int j = (int)blockIdx.x - g.tail_first;
            ksplit = g.tail_split;
            ks = j % ksplit;
            vb = g.tail_first + j / ksplit;
        }
    }
    int tm, tn;
    int vbid = PERSIST ? (int)blockIdx.x : vb;
    const unsigned long long ts0 = g.ts ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int slot) {   // wave 0 of every workgroup; profile launches only
        if (g.ts && threadIdx.x == 0) g.ts[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime() - ts0 + 1;
    };
    if (!tile_coords(g.tiles_m, g.tiles_n, tm, tn, vbid)) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w % WAVES_M, wn = w / WAVES_M;
    const int li = lane & 15, lg = lane >> 4;

    // DMA sources of this wave's pieces (A pieces first, then B pieces)
    const int srow = lane >> 2;
    const int scol = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 8;   // row>>2 & 3 == lane>>4 (pieces are 16 rows)
    const bf16_t *src[PPW];
    int dst[PPW], kstride[PPW];   // elements a piece's source advances per K step
    auto setup = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = w * PPW + i;
        if (q < PA) {
            src[i] = g.A + (size_t)min(m0 + q * 16 + srow, g.M - 1) * g.lda + scol;
            dst[i] = q * 16 * BK;
            kstride[i] = BK;
        } else if (g.Wt) {   // fragment-major weights: the piece is 1 KiB contiguous, pre-swizzled
            const int nblk = min((n0 + (q - PA) * 16) / 16, g.N / 16 - 1);
            src[i] = g.Wt + (size_t)nblk * (g.K / BK) * 512 + lane * 8;
            dst[i] = BM * BK + (q - PA) * 16 * BK;
            kstride[i] = 512;
        } else {
            src[i] = g.W + (size_t)min(n0 + (q - PA) * 16 + srow, g.N - 1) * g.ldw + scol;
            dst[i] = BM * BK + (q - PA) * 16 * BK;
            kstride[i] = BK;
        }
    }
    };
    setup(tm * BM, tn * BN);
    const int nk_all = g.K / BK;
    const int kt0 = (nk_all * ks) / ksplit;              // first K tile of this slice
    const int nk = (nk_all * (ks + 1)) / ksplit - kt0;   // tiles in this slice
    auto issue = [&](int tile) {
        bf16_t *base = smem + (tile % ST) * (BM + BN) * BK;
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma16(src[i] + (size_t)(kt0 + tile) * kstride[i], base + dst[i]);
    };

    f32x4 acc[WMT][WNT];

    // f(q) = (-q) & 3 with q = (row >> 2) & 3 makes each hardware 16-lane group of a
    // ds_read_b128 hit 16 distinct 16-byte slots (the groups mix lg values)
    const int fslot = (lg ^ ((0 - (li >> 2)) & 3)) * 8;
    const int a_off = (wm * WMT * 16 + li) * BK + fslot, b_off = BM * BK + (wn * WNT * 16 + li) * BK + fslot;
    bf16x8 a0[WMT], b0[WNT], a1[WMT], b1[WNT];
    auto read_frags = [&](int tile, bf16x8(&a)[WMT], bf16x8(&b)[WNT]) {
        const bf16_t *base = smem + (tile % ST) * (BM + BN) * BK;
#pragma unroll
        for (int j = 0; j < WNT; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(base + b_off + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < WMT; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(base + a_off + i * 16 * BK);
    };
    auto mma = [&](const bf16x8(&a)[WMT], const bf16x8(&b)[WNT]) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                // SWAP: acc = W_tile . A_tile^T (operand order exchanged; the fragment layouts of the two
                // operands are the same), i.e. the output tile transposed -- see store_tile_t
                if constexpr (EPI == EPI_F32H) {
                    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b[j]), __builtin_bit_cast(h8, a[i]),
                                                                       acc[i][j], 0, 0, 0);
                } else if constexpr (SWAP) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
    };
    // tile `next` must have landed in every wave's view before it is read: own DMA
    // pieces by counted vmcnt (up to D-2 later tiles stay in flight), the other
    // waves' pieces by the barrier; lgkmcnt(0) first so that no ds_read of the stage
    // about to be refilled is still pending when its DMA is issued.
    auto arrive = [&](int next) {
        wait_tiles<PPW, D - 2>(min(D - 2, nk - 1 - next), true);
        __builtin_amdgcn_s_barrier();
    };

#pragma unroll
    for (int s_ = 0; s_ < D; ++s_)
        if (s_ < nk) issue(s_);
    bool first_tile = true;
    for (;;) {   // one pass per output tile (PERSIST: several)
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!PERSIST || first_tile) wait_tiles<PPW, D - 1>(min(D - 1, nk - 1), false);
    else wait_vm<0>();   // the next tile's first K tiles were requested before the last epilogue, whose stores count too
    first_tile = false;
    stamp(1);                      // first K tile landed (own pieces)
    __builtin_amdgcn_s_barrier();
    stamp(2);
    read_frags(0, a0, b0);
    // steady state, branch-free (so that hipcc counts the prefetch reads as
    // allowed-outstanding in front of the MFMAs instead of waiting lgkmcnt(0))
    int t = 0;
    for (; t + D + 1 < nk; t += 2) {
        wait_vm_lgkm0<(D - 2) * PPW>();
        __builtin_amdgcn_s_barrier();
        read_frags(t + 1, a1, b1);
        issue(t + D);
        mma(a0, b0);
        wait_vm_lgkm0<(D - 2) * PPW>();
        __builtin_amdgcn_s_barrier();
        read_frags(t + 2, a0, b0);
        issue(t + D + 1);
        mma(a1, b1);
    }
    for (; t < nk; t += 2) {  // tail: same schedule with guards
        if (t + 1 < nk) {
            arrive(t + 1);
            read_frags(t + 1, a1, b1);
        }
        if (t + D < nk) issue(t + D);
        mma(a0, b0);
        if (t + 1 < nk) {
            if (t + 2 < nk) {
                arrive(t + 2);
                read_frags(t + 2, a0, b0);
            }
            if (t + D + 1 < nk) issue(t + D + 1);
            mma(a1, b1);
        }
    }

    stamp(3);                      // K loop issued
    bool more = false;
    int tm2 = 0, tn2 = 0;
    if constexpr (PERSIST) {
        vbid += (int)gridDim.x;
        more = tile_coords(g.tiles_m, g.tiles_n, tm2, tn2, vbid);
        if (more) {
            // every wave has read its last fragments (they were in registers a step before their
            // MFMAs): the ring is free -- request the next tile's first K tiles now
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            setup(tm2 * BM, tn2 * BN);
#pragma unroll
            for (int s_ = 0; s_ < D; ++s_)
                if (s_ < nk) issue(s_);
        }
    }
    GemmArgs ge = g;
    ge.ksplit = ksplit;   // this workgroup's share: > 1 = atomic adds
    if constexpr (EPI == EPI_PART) {   // this slice's plane of the workspace
        ge.X = g.part + (size_t)ks * g.M * g.N;
        ge.ldc = g.N;
    }
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int trow = m0 + (wm * WMT + i) * 16;
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < WNT; j += 2)
                store_tile_t<EPI>(ge, acc[i][j], acc[i][j + 1], trow, (n0 + wn * WNT * 16) / 2 + (j / 2) * 16, lane);
        } else if constexpr (SWAP) {
#pragma unroll
            for (int j = 0; j < WNT; ++j)
                store_tile_t<EPI>(ge, acc[i][j], acc[i][j], trow, n0 + (wn * WNT + j) * 16, lane);
        } else {
#pragma unroll
            for (int j = 0; j < WNT; ++j)
                store_tile<EPI>(ge, acc[i][j], acc[i][j], trow, n0 + (wn * WNT + j) * 16, lane);
        }
    }
    stamp(4);                      // epilogue issued
    if (g.ts) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);                  // output stores acknowledged
    }
    if (!more) break;
    tm = tm2;
    tn = tn2;
    }   // tile loop
}

// ---------------------------------------------------------------------
// 256x256 tile, hand-ordered K loop ("slab" kernel).  Two differences to gemm_bf16_ring_kernel:
//
//  * LDS holds a ring of five 32 KiB SLABS, alternately A and W: slab 2T = rows 0..255 of A at
//    K columns [64 T, 64 T + 64), slab 2T+1 the same of W.  A DMA piece is 8 rows x 128 B -- whole
//    128-byte cache lines.  (The ring kernel's pieces are 16 rows x 64 B: every line is fetched
//    into the CU's L1 twice, half used each time; a timing build with contiguous pieces measured
//    +11 % on the GEMM.)  A K step is still 32 columns: step u multiplies half u & 1 of tile u >> 1.
//  * The K loop is an ordered sequence of `asm volatile` statements (MFMA, ds_read_b128, LDS-DMA,
//    s_waitcnt, s_barrier).  The compiler allocates registers and computes addresses; the issue
//    order and every wait are ours: per step the MFMAs of the fragment set read one step earlier,
//    with the ds_reads of the next step's set and the DMA pieces of slab u + 4 spread between
//    them (an LDS-DMA piece costs its wave 60-180 cycles of issue time; a burst of them in front
//    of the MFMAs is what held the ring kernel at ~78 % MFMA-busy inside its K loop).  The
//    accumulators are pinned to AGPRs ("+a"): left to hipcc, 256 of them get shuffled through
//    v_accvgpr_read/write around every MFMA.
//
// WN_ = 4: 8 waves (2 x 4) of 128 x 64, two per SIMD;  WN_ = 2: 4 waves (2 x 2) of 128 x 128, one
// per SIMD, all 256 AGPRs.  Ordering rules (the compiler knows nothing about the asm loads):
//   - fragments read during step u are first used in step u+1, behind `s_waitcnt lgkmcnt(0)`;
//   - slab q is requested during step q - 4 and first read during step (q | 1) - 2: the wait in
//     front of an odd step u leaves at most one slab's pieces in flight (slabs <= u + 2 have landed:
//     the next tile's pair), and the s_barrier behind it publishes the other waves' pieces; an even
//     step reads the second half of slabs that landed a step earlier and may leave two in flight;
//   - slab q's last read is issued during step q | 1 and retired by the lgkmcnt(0) in front of the
//     next barrier, so its slot (q mod 5) is rewritten from step (q | 1) + 1 at the earliest:
//     slab q + 5, requested during step q + 1.  Both parities hold.
// Past the last slab the DMA pieces are skipped behind a uniform branch; the reads stay
// unconditional (stale LDS into a set nobody multiplies): the accumulators never meet a merge of
// different instruction sequences, where a failed coalesce would spill AGPRs to scratch.
// Requires K % 64 == 0.
// ---------------------------------------------------------------------
// An accumulator tuple read through asm with 'a'-class inputs (the slab kernel's epilogues): the compiler then never sees the
// accumulators in VALU code and cannot decide to move them into VGPRs -- and from there to scratch -- ahead of the epilogue.
__device__ __forceinline__ f32x4 acc_read(const f32x4 &a) {
    float x0, x1, x2, x3;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(a[0]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(a[1]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x2) : "a"(a[2]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x3) : "a"(a[3]));
    return (f32x4){x0, x1, x2, x3};
}

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void dma16_off(const void *gptr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n" ::"s"(lds_off), "v"(gptr) : "memory");
}

// Which W row feeds position p of a wave's MFMA tile j is free -- it only decides which output column a lane ends
// up holding.  PERM 0: tile j = W rows 16 j + p (a lane holds 4 consecutive columns of tile j).  bf16 outputs
// choose rows so that a lane's values of two tiles are 8 CONSECUTIVE output columns: one 16-byte store, 64-byte
// row segments per instruction instead of 32 (tools/micro/store_pattern.hip: 5.3 vs 8.2 us per 128 KiB tile; the
// 32-byte version was ~10 of the ~52 us of a K = 1536 tile).
//   PERM 1 (plain store):  row = 32 (j/2) + 8 (p/4) + 4 (j%2) + p%4        -> tiles 2J, 2J+1: columns 32 J + 8 lg + 0..7
//   PERM 2 (SwiGLU; W rows are 16 gate rows / 16 up rows per 16 outputs; tiles 4Q..4Q+3 = gate, up, gate, up):
//          row = 64 Q + 32 (p/8) + 8 (p/4 % 2) + 16 (j%2) + 4 (j/2 % 2) + p%4 -> outputs 32 Q + 8 lg + 0..7
// Returns the byte offset of tile j's rows relative to the lane's base row.
template <int PERM>
__host__ __device__ constexpr int slab_w_tile_off(int j) {
    return (PERM == 1 ? 32 * (j / 2) + 4 * (j % 2) : PERM == 2 ? 64 * (j / 4) + 16 * (j % 2) + 4 * ((j / 2) % 2) : 16 * j) * 128;
}

// WNT_ (16-column MFMA tiles per wave along N): 16 / WN_ gives the 256-column tile; narrower tiles (residual epilogue only):
// WN_ = 2, WNT_ = 6 is a 256 x 192 tile for the token counts at which N = 1536 yields 130-190 tiles of 256 columns on 256 CUs and
// exactly <= 256 tiles of 192; WNT_ = 4 / 2 are 128- / 64-column tiles for the K-split down projection of a few hundred tokens
// (fewer, longer K slices: fewer f32 planes to write and add -- and a workgroup's W strip, the operand that misses L2, is the
// narrow side).  WMT_ = 6: 192-row tiles for the token counts at which 256-row tiles multiply mostly padding (513 .. 576 tokens
// are three row tiles either way: a quarter fewer MFMAs per step).  The slabs keep their 32 KiB ring slots whatever the tile; a
// slab REQUESTS the tile's rows only (PA / PPW pieces per wave), and the waits count those.
template <int EPI, int WN_, bool PERSIST = false, int WNT_ = 16 / WN_, int WMT_ = 8>
__global__ void __launch_bounds__(128 * WN_, 1) gemm_bf16_slab_kernel(GemmArgs g) {
    constexpr int WMT = WMT_, BM = 32 * WMT, WNT = WNT_, BN = WN_ * WNT * 16, NW = 2 * WN_;
    static_assert(WMT_ == 8 || WMT_ == 6, "row tiles of 256 or 192");
    static_assert(BN <= 256 && (WNT_ == 16 / WN_ || EPI == EPI_RESID), "narrower tiles: residual epilogue only");
    static_assert(WMT_ == 8 || !PERSIST, "192-row tiles: one unit per workgroup");
    constexpr int PPW = (BN / 8) / NW;                       // DMA pieces (1 KiB: 8 rows x 128 B) per wave per W slab: the tile's rows, not the 32 KiB slot's
    constexpr int PA = (BM / 8) / NW;                        // ... per A slab (a 192-row tile requests 24 pieces)
    constexpr int PMAX = PA > PPW ? PA : PPW, PMIN = PA < PPW ? PA : PPW;
    static_assert(PA * NW * 8 == BM && PPW * NW * 8 == BN, "pieces");
    constexpr int NS = 5, DQ = 4;                            // slabs in the ring, request distance in slabs
    constexpr unsigned SLAB_B = 256 * 128;                   // 32 KiB
    constexpr int NMF = WMT * WNT, NRD = WMT + WNT;          // MFMAs / fragment reads per wave per step
    constexpr int RSTEP = (NMF * 3 / 4) / NRD;               // one read every RSTEP MFMAs, all inside the first 3/4 of the step (placement
                                                             // of reads and DMA pieces inside the step measured +-2 %: noise)
    constexpr int DSTEP = NMF / PMAX;                        // one DMA piece every DSTEP MFMAs
    constexpr bool SWAP = EPI != EPI_QKV;
    constexpr int PERM = EPI == EPI_STORE ? 1 : EPI == EPI_SWIGLU ? 2 : 0;
    static_assert(RSTEP >= 1 && DSTEP >= 1 && PPW % 2 == 0, "schedule");
    __shared__ __attribute__((aligned(16))) bf16_t smem[NS * SLAB_B / 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;
    const int nt_all = g.K / 64;                             // 64-column tiles

    // A work unit = an output tile, or (residual epilogue, wave-quantisation tail) a K slice of one: units >=
    // tail_first are (tile, slice) pairs whose partial sums meet in f32 atomics.  PERSIST: one workgroup per CU walks
    // units blockIdx.x + i * gridDim.x (the same XCD, the same tile window per round as one-unit workgroups).
    int ks_cur = 0;                                          // K slice of the current unit
    auto decode = [&](int unit, int &tm_, int &tn_, int &ksplit_, int &kt0_, int &nk_) -> bool {
        int ks = 0, vb = unit;
        ksplit_ = 1;
        if constexpr (EPI == EPI_RESID) {
            if (g.order == 2) {
                // EVERY tile split tail_split ways (a few hundred tokens): the (slice, tile) pairs in slice-major order, XCD x
                // (= unit % 8: where the hardware puts the workgroup) takes the x-th eighth of them -- the workgroups of an
                // XCD then share one or two K slices of A and W in their L2 instead of touching all of them (PMC at 576
                // tokens: the down projection fetched 124 MB past L2 for 38 MB of operands)
                const int ntiles = g.tiles_m * g.tiles_n, P = ntiles * g.tail_split;
                const int x = unit & 7, p = (int)(((long)x * P) >> 3) + (unit >> 3);
                if (p >= (int)(((long)(x + 1) * P) >> 3)) return false;
                ks = p / ntiles;
                const int t = p - ks * ntiles;
                tn_ = t / g.tiles_m;
                tm_ = t - tn_ * g.tiles_m;
                ksplit_ = g.tail_split;
                ks_cur = ks;
                kt0_ = (nt_all * ks) / ksplit_;
                nk_ = 2 * ((nt_all * (ks + 1)) / ksplit_ - kt0_);
                return true;
            }
            if (g.tail_split > 1 && unit >= g.tail_first) {
                const int j = unit - g.tail_first;
                ksplit_ = g.tail_split;
                ks = j % ksplit_;
                vb = g.tail_first + j / ksplit_;
            }
        }
        if (!tile_coords(g.tiles_m, g.tiles_n, tm_, tn_, vb, g.order)) return false;
        ks_cur = ks;
        kt0_ = (nt_all * ks) / ksplit_;                      // first 64-column tile of this K slice
        nk_ = 2 * ((nt_all * (ks + 1)) / ksplit_ - kt0_);    // K steps = slabs of this slice (even)
        return true;
    };

    // DMA sources of this wave's pieces w*PPW .. of every A slab and every W slab: lane -> row lane >> 3 of the
    // piece, 16-byte column slot (lane & 7) ^ key(row) (the XOR swizzle lives on the source side: the DMA itself
    // is lane-linear).  A slabs: key = row & 7.  W slabs: key = (row & 3) | (bit 3 of row) << 2 -- the 16 rows a
    // ds_read_b128 lane group touches are {0-3, 8-11, 16-19, 24-27} (+ multiples of 4) under the permuted tile
    // maps above and {0..15} under the plain one; this key keeps all of them on 16 distinct 16-byte LDS granules.
    // (PERSIST: what depends on the lane is derived again for every unit from an opaque copy of the lane number -- values that
    // stay live across the epilogue are spilled there, and a reload that is first used inside the K loop brings the compiler's
    // own s_waitcnt vmcnt(n) into the loop, which drains the hand-counted DMA pipeline: the first persistent build ran its K
    // loop 22 % slower and its epilogue 5 x slower for exactly that)
    auto opaque_lane = [&]() -> int {
        int l_ = lane;
        if constexpr (PERSIST) asm volatile("" : "+v"(l_));
        return l_;
    };
    const unsigned dma_dst = lds0 + (unsigned)w * PPW * 1024u;       // + slot * SLAB_B + p * 1024
    const unsigned dma_dstA = lds0 + (unsigned)w * PA * 1024u;
    const bf16_t *srcA[PA], *srcW[PPW];
    auto set_sources = [&](int tm_, int tn_, int kt0_) {
        const int l_ = opaque_lane();
        const int prow = l_ >> 3, scol = ((l_ & 7) ^ prow) * 8;
        const int scolW[2] = {((l_ & 7) ^ (prow & 3)) * 8, ((l_ & 7) ^ ((prow & 3) | 4)) * 8};   // even / odd piece (bit 3 of the row)
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const int r = (w * PPW + p) * 8 + prow;
            srcW[p] = g.W + (size_t)min(tn_ * BN + r, g.N - 1) * g.ldw + scolW[p & 1] + (size_t)kt0_ * 64;   // PPW is even: piece parity = p & 1
        }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int r = (w * PA + p) * 8 + prow;
            srcA[p] = g.A + (size_t)min(tm_ * BM + r, g.M - 1) * g.lda + scol + (size_t)kt0_ * 64;
        }
    };
    int krot = 0, kn = 1;                                    // GemmArgs::krot: the unit's K tiles are walked from krot round the end
    auto kmap = [&](int t) -> size_t { const int r = t + krot; return (size_t)(r >= kn ? r - kn : r) * 64; };
    // PERSIST: a unit's slab q lives in ring slot (C0 + q) % NS with C0 = 3, so that slabs 0 and 1 of the NEXT unit -- requested
    // before the current unit's epilogue -- land in [96 KiB, 160 KiB), clear of every epilogue's staging blocks (< 96 KiB; the
    // V^T tiles of the QKV projection, whose staging takes most of the ring, request theirs after the epilogue)
    constexpr unsigned C0 = PERSIST ? 3u : 0u;
    auto request_slabs = [&](int nk_, int q0, int q1) {      // slabs q0 .. q1-1 (of the first DQ) into ring slots (C0 + q) % NS
#pragma unroll
        for (int q = 0; q < DQ; ++q)
            if (q >= q0 && q < q1 && q < nk_) {
                const unsigned slot = (C0 + q) % NS;
                if (q & 1) {
#pragma unroll
                    for (int p = 0; p < PPW; ++p) dma16_off(srcW[p] + kmap(q >> 1), dma_dst + slot * SLAB_B + p * 1024u);
                } else {
#pragma unroll
                    for (int p = 0; p < PA; ++p) dma16_off(srcA[p] + kmap(q >> 1), dma_dstA + slot * SLAB_B + p * 1024u);
                }
            }
    };
    auto request_first = [&](int nk_) { request_slabs(nk_, 0, DQ); };

    // fragment (16 rows x 32 k) of K half kk: lane (li, lg) reads row li, global slot 4 kk + lg -> LDS slot ^ (row & 7)
    unsigned rdA = 0, rdB = 0;
    auto set_read_addresses = [&] {
        const int l_ = opaque_lane(), li_ = l_ & 15, lg_ = l_ >> 4;
        rdA = lds0 + (unsigned)(wm * (WMT * 16) + li_) * 128u + (unsigned)((lg_ ^ (li_ & 7)) * 16);   // kk = 1: ^ 64
        const int bsub = li_ >> 2, bc = li_ & 3;
        const int brow = PERM == 1 ? 8 * bsub + bc : PERM == 2 ? 32 * (bsub >> 1) + 8 * (bsub & 1) + bc : li_;   // W row of tile position li
        const int bkey = (brow & 3) | (((brow >> 3) & 1) << 2);
        rdB = lds0 + (unsigned)(wn * WNT * 16 + brow) * 128u + (unsigned)((lg_ ^ bkey) * 16);
    };
    set_read_addresses();
    f32x4 acc[WMT][WNT];
    bf16x8 a0[WMT], b0[WNT], a1[WMT], b1[WNT];
    int nk = 0;                                              // K steps of the current unit

    auto mfma = [&](f32x4 &c, const bf16x8 &a, const bf16x8 &b) {
        if constexpr (EPI == EPI_F32H) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
        else if constexpr (SWAP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    };
    auto wrap = [&](unsigned s) { return s >= (unsigned)NS ? s - NS : s; };
    // step u (parity PAR): multiply (ac, bc); read the fragments of step u+1 into (an, bn); request slab u + DQ.
    // c = ring slot of slab 2 (u >> 1), the A slab of the tile step u multiplies.
    auto step = [&](bf16x8(&ac)[WMT], bf16x8(&bc)[WNT], bf16x8(&an)[WMT], bf16x8(&bn)[WNT], auto PAR, auto STEADY, int u, unsigned c) {
        constexpr int par = decltype(PAR)::value;
        // step u+1 reads: odd u+1 -> same tile, half 1; even u+1 -> next tile, half 0
        const unsigned sa = par == 0 ? c : wrap(c + 2), sb = wrap(sa + 1);
        const unsigned ra = (rdA ^ (par == 0 ? 64u : 0u)) + sa * SLAB_B, rb = (rdB ^ (par == 0 ? 64u : 0u)) + sb * SLAB_B;
        // slab u + DQ: DQ even -> an A slab in even steps, a W slab in odd steps; tile (u + DQ) >> 1
        const unsigned sd = wrap(c + par + DQ);
        const bool dma_on = decltype(STEADY)::value || u + DQ < nk;
        const size_t koff = kmap((u + DQ) >> 1);
        static_for<NMF>([&](auto M_) {
            constexpr int m = decltype(M_)::value, i = m / WNT, j = m % WNT;
            mfma(acc[i][j], ac[i], bc[j]);
            if constexpr (m % RSTEP == RSTEP - 1 && m / RSTEP < NRD) {
                constexpr int r = m / RSTEP;                          // first the W fragments, then the A fragments
                if constexpr (r < WNT) lds_read16<slab_w_tile_off<PERM>(r)>(bn[r], rb);
                else lds_read16<(r - WNT) * 2048>(an[r - WNT], ra);
            }
            if constexpr (m % DSTEP == DSTEP - 1) {
                constexpr int p = m / DSTEP;
                if constexpr (par == 0) {
                    if constexpr (p < PA) {
                        if (dma_on) dma16_off(srcA[p] + koff, dma_dstA + sd * SLAB_B + p * 1024u);
                    }
                } else {
                    if constexpr (p < PPW) {
                        if (dma_on) dma16_off(srcW[p] + koff, dma_dst + sd * SLAB_B + p * 1024u);
                    }
                }
            }
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    int unit = (int)blockIdx.x;
    int tm, tn, ksplit, kt0;
    // profile launches (MI_GEMM_TS=1): slot 0 = absolute start, 1..4 = ticks since the start at: first slabs landed, K loop
    // issued, epilogue issued, stores acknowledged; 5 = which CU (HW_ID | XCC_ID << 32)
    unsigned long long ts0 = g.ts ? __builtin_amdgcn_s_memtime() : 0ull;
    int ts_row = (int)blockIdx.x;
    auto stamp = [&](int slot) {
        if (g.ts && threadIdx.x == 0 && ts_row < g.ts_rows) g.ts[(size_t)ts_row * 8 + slot] = __builtin_amdgcn_s_memtime() - ts0 + 1;
    };
    auto stamp_row = [&] {
        if (g.ts && threadIdx.x == 0 && ts_row < g.ts_rows) {
            g.ts[(size_t)ts_row * 8] = ts0;
            g.ts[(size_t)ts_row * 8 + 5] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |   // HW_REG_HW_ID
                                           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);   // HW_REG_XCC_ID
        }
    };
    stamp_row();
    // ---- PERSIST: this workgroup's schedule.  XCD x = blockIdx % 8 (where the hardware puts the block: locality only) owns the
    // tiles tile_coords(vb = 8 s + x), s = 0 .. n_x - 1 -- the tiles the one-tile-per-workgroup launch runs there, in the same
    // order.  Its G8 = gridDim / 8 workgroups take whole tiles round by round (s = l + i G8) while full rounds last; the R < G8
    // tiles left over are ONE range of R x KT K tiles cut into P equal pieces (stream-K): workgroup l works [bnd(l), bnd(l+1)),
    // which touches at most a tile's tail and the next tile's head.  A piece that does not start at K = 0 is a PARTIAL: its
    // accumulators leave for GemmArgs::sk_part; the holder of a tile's head (K = 0) runs the tile's epilogue after adding the
    // partials in piece order -- a fixed order: the result does not depend on timing.  Nobody waits for a later piece of their
    // own: every workgroup runs its partial (if any) first, so a head's partials were published long before it asks.  Cuts
    // closer than SK_MINK K tiles to a tile edge snap to the edge; pieces are >= SK_MINSEG K tiles.
    constexpr int SK_MINK = 4, SK_MINSEG = 6;
    [[maybe_unused]] int ps_x = 0, ps_l = 0, ps_G8 = 1, ps_nx = 0, ps_F = 0, ps_base = 0, ps_P = 0, ps_S = 0, ps_i = 0, ps_u = 0, ps_u1 = 0;
    [[maybe_unused]] int sk_kind = 0, sk_nc = 0, sk_tile = 0;    // current unit: 0 whole tile / 1 partial / 2 head with sk_nc partials to add
    auto sk_bnd = [&](int p) -> int {
        int raw = (int)(((long)p * ps_S) / max(ps_P, 1));
        const int o = raw % nt_all;
        if (o < SK_MINK) raw -= o;
        else if (o > nt_all - SK_MINK) raw += nt_all - o;
        return raw;
    };
    auto pinit = [&] {
        ps_x = (int)blockIdx.x & 7; ps_l = (int)blockIdx.x >> 3; ps_G8 = max((int)gridDim.x >> 3, 1);
        const int ntiles = g.tiles_m * g.tiles_n;
        if (g.order == 1) ps_nx = ntiles > ps_x ? (ntiles - ps_x + 7) >> 3 : 0;
        else { const int per = (ntiles + 7) >> 3; ps_nx = max(0, min(per, ntiles - ps_x * per)); }
        ps_F = ps_nx / ps_G8;
        const int R = ps_nx - ps_F * ps_G8;
        ps_base = ps_F * ps_G8;
        const bool sk_on = g.sk_part && g.sk_ctr && R > 0 && R * 5 <= ps_G8 * 4 && nt_all >= 16;
        if (!sk_on) { if (R > 0) ++ps_F; return; }          // the remainder as a partly filled round of whole tiles
        ps_S = R * nt_all;
        ps_P = min(ps_G8, ps_S / SK_MINSEG);
        if (ps_l < ps_P) { ps_u = sk_bnd(ps_l); ps_u1 = sk_bnd(ps_l + 1); }
    };
    auto pnext = [&](int &tm_, int &tn_, int &kt0_, int &nk_, int &kind_, int &nc_, int &tile_) -> bool {
        kind_ = 0; nc_ = 0; tile_ = 0;
        if (ps_i < ps_F) {
            const int sidx = ps_l + ps_i * ps_G8;
            ++ps_i;
            if (sidx < ps_nx && tile_coords(g.tiles_m, g.tiles_n, tm_, tn_, sidx * 8 + ps_x, g.order)) {
                kt0_ = 0; nk_ = 2 * nt_all;
                return true;
            }
            ps_i = ps_F;                                     // (a partly filled last round: nothing for this workgroup)
        }
        if (ps_u < ps_u1) {
            const int t = ps_u / nt_all, o = ps_u - t * nt_all, e = min(ps_u1 - t * nt_all, nt_all);
            if (!tile_coords(g.tiles_m, g.tiles_n, tm_, tn_, (ps_base + t) * 8 + ps_x, g.order)) { ps_u = ps_u1; return false; }
            kt0_ = o; nk_ = 2 * (e - o);
            tile_ = ps_x * ps_G8 + t;
            if (o > 0) kind_ = 1;
            else if (e < nt_all) {
                kind_ = 2;
                for (int q = ps_l + 1; q < ps_P && sk_bnd(q) < (t + 1) * nt_all; ++q) ++nc_;
            }
            ps_u = t * nt_all + e;
            return true;
        }
        return false;
    };
    // fused RMSNorm (consumer side) / rotary positions: lane l keeps the scale (position) of rows l and 64 + l of the wave's
    // 128 rows; the epilogue fetches its rows' values with ds_bpermute.  Requested right behind the first slabs, so the
    // latency is the pipeline fill's.
    [[maybe_unused]] float inv_lo = 1.f, inv_hi = 1.f;
    [[maybe_unused]] int pos_lo = 0, pos_hi = 0;
    auto load_row_consts = [&](int tm_) {
        if constexpr (EPI == EPI_QKV || EPI == EPI_SWIGLU) {
            const int r0 = min(tm_ * BM + wm * (WMT * 16) + lane, g.M - 1), r1 = min(tm_ * BM + wm * (WMT * 16) + 64 + lane, g.M - 1);
            if (g.row_scale) { inv_lo = g.row_scale[r0]; inv_hi = g.row_scale[r1]; }
            if constexpr (EPI == EPI_QKV) {
                if (g.rope_cs) { pos_lo = g.rope_pos[r0]; pos_hi = g.rope_pos[r1]; }
            }
        }
    };
    if constexpr (PERSIST) {
        pinit();
        if (!pnext(tm, tn, kt0, nk, sk_kind, sk_nc, sk_tile)) return;
        ksplit = 1;
        set_sources(tm, tn, kt0);
        kn = max(nk >> 1, 1);
        krot = 0;
    } else {
        if (!decode(unit, tm, tn, ksplit, kt0, nk)) return;
        set_sources(tm, tn, kt0);
        kn = max(nk >> 1, 1);
        krot = g.krot < 0 ? (int)(((long)tm * kn) / g.tiles_m) : (tm * g.krot) % kn;
    }
    request_first(nk);
    load_row_consts(tm);
    auto row_f = [&](int i, int r16) -> float {      // the scale of row 16 i + r16 of the wave's rows
        return __int_as_float(__builtin_amdgcn_ds_bpermute(((i & 3) * 16 + r16) * 4, __float_as_int(i < 4 ? inv_lo : inv_hi)));
    };
    auto row_i = [&](int i, int r16) -> int { return __builtin_amdgcn_ds_bpermute(((i & 3) * 16 + r16) * 4, i < 4 ? pos_lo : pos_hi); };
    for (;;) {   // one pass per work unit
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // slabs 0 and 1 landed (after a previous unit's epilogue its stores count as well: more conservative, never less --
    // loads return in order among themselves); fragments of step 0 read
    wait_tiles<PMIN, DQ - 2>(max(0, min(DQ - 2, nk - 2)), false);   // (the smaller of the two slab sizes: never fewer landed than needed)
    asm volatile("s_barrier" ::: "memory");
    stamp(1);
    static_for<WNT>([&](auto R) { lds_read16<slab_w_tile_off<PERM>(decltype(R)::value)>(b0[decltype(R)::value], rdB + ((C0 + 1) % NS) * SLAB_B); });
    static_for<WMT>([&](auto R) { lds_read16<decltype(R)::value * 2048>(a0[decltype(R)::value], rdA + C0 * SLAB_B); });

    unsigned c = C0;                                         // ring slot of the current tile's A slab
    int u = 0;
    for (; u + DQ + 1 < nk; u += 2) {                        // steady state, branch-free
        static_assert(DQ == 4, "the two slabs in flight at an even step are one A and one W slab");
        wait_vm_lgkm0<PA + PPW>();                           // even step: reads the second half of slabs that landed a step ago
        asm volatile("s_barrier" ::: "memory");
        step(a0, b0, a1, b1, P0{}, T_{}, u, c);
        wait_vm_lgkm0<PA>();                                 // odd step: only the A slab requested a step ago may be in flight
        asm volatile("s_barrier" ::: "memory");
        step(a1, b1, a0, b0, P1{}, T_{}, u + 1, c);
        c = wrap(c + 2);
    }
    for (; u < nk; u += 2) {                                 // the last steps: fewer slabs in flight, no new requests
        wait_tiles<PMIN, DQ - 3>(max(0, min(DQ - 3, nk - 3 - u)), true);
        asm volatile("s_barrier" ::: "memory");
        step(a0, b0, a1, b1, P0{}, F_{}, u, c);
        wait_tiles<PMIN, DQ - 3>(max(0, min(DQ - 3, nk - 4 - u)), true);
        asm volatile("s_barrier" ::: "memory");
        step(a1, b1, a0, b0, P1{}, F_{}, u + 1, c);
        c = wrap(c + 2);
    }
    // the inline-asm MFMAs are opaque to the hazard recogniser: results -> v_accvgpr_read needs wait states; the last
    // step's (unused) fragment reads are retired here too, so nothing of the asm stream is in flight past this point
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    stamp(2);

    const int m0 = tm * BM, n0 = tn * BN;
    GemmArgs ge = g;
    ge.ksplit = ksplit;
    constexpr int VROW = WMT * 16 + 8;                       // bf16 per staged V^T channel row (128 tokens + 16 B pad)
    static_assert(NW * WNT * 16 * VROW * 2 <= NS * (int)SLAB_B, "the V^T staging blocks fit the ring");
    // QKV: a tile that lies wholly inside the V columns (and whole 8-token chunks: M % 8 == 0, ldvt % 8 == 0) leaves through LDS
    [[maybe_unused]] const bool vt_staged = EPI == EPI_QKV && n0 >= g.qk_cols && n0 + BN <= g.N && (g.M & 7) == 0 && (g.ldvt & 7) == 0;

    // PERSIST: the next unit's slabs 0 and 1 are requested before this unit's epilogue, into the ring slots the staging blocks
    // do not reach -- the pipeline fill (their trip from HBM / L2) runs under the epilogue, and the epilogue's stores drain under
    // the next K loop's first steps.  (The source pointers are recomputed behind the epilogue: held across it they cost the
    // epilogue 16-32 registers it does not have.)
    bool more = false;
    [[maybe_unused]] bool early = false;
    int tm2 = 0, tn2 = 0, ksplit2 = 1, kt02 = 0, nk2 = 0;
    [[maybe_unused]] int kind2 = 0, nc2 = 0, tile2 = 0;
    if constexpr (PERSIST) {
        more = pnext(tm2, tn2, kt02, nk2, kind2, nc2, tile2);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last step's (unused) fragment reads: the ring is free
            asm volatile("s_barrier" ::: "memory");              // ... in every wave
            early = !vt_staged;
            if (early) {
                // (its own temporaries, not srcA / srcW: those are written once per unit, behind the epilogue)
                const int l_ = opaque_lane();
                const int prow = l_ >> 3, scol = ((l_ & 7) ^ prow) * 8;
                const int scolW[2] = {((l_ & 7) ^ (prow & 3)) * 8, ((l_ & 7) ^ ((prow & 3) | 4)) * 8};
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int r = (w * PA + p) * 8 + prow;
                    dma16_off(g.A + (size_t)min(tm2 * BM + r, g.M - 1) * g.lda + scol + (size_t)kt02 * 64, dma_dstA + C0 * SLAB_B + p * 1024u);
                }
                if (nk2 > 1) {
#pragma unroll
                    for (int p = 0; p < PPW; ++p) {
                        const int r = (w * PPW + p) * 8 + prow;
                        dma16_off(g.W + (size_t)min(tn2 * BN + r, g.N - 1) * g.ldw + scolW[p & 1] + (size_t)kt02 * 64,
                                  dma_dst + ((C0 + 1) % NS) * SLAB_B + p * 1024u);
                    }
                }
            }
        }
    }
    // stream-K roles (PERSIST): a partial leaves its accumulators in the workspace and skips the epilogue; a head waits for its
    // tile's partials (published long ago: every workgroup runs its partial first) and adds them in piece order.  Both sides
    // touch the accumulators through 'a'-class asm operands only, like the K loop: with every AGPR taken, accumulators that
    // the compiler sees in VALU code migrate to VGPRs and spill (the first version: 1.9 KiB of scratch per lane).
    //   store: global_store_dwordx4 straight from the AGPR tuple, register-major (1 KiB per wave and instruction), write-through;
    //   add:   D += T as four v_mfma_f32_16x16x4_f32 per 16 x 16 tile -- MFMA s multiplies the selector A_s[i][k] = (i == 4 s + k)
    //          by B_s[k][j] = T[4 s + k][j], which in the register-major block is the dword at 256 s + 16 li + 4 lg of the tile's
    //          KiB: the products are exact (1 x T, 0 x T), so D + T is rounded once, as a v_add_f32 would.
    bool run_epilogue = true;
    if constexpr (PERSIST) {
        static_assert(NMF % 4 == 0, "four accumulator tiles per base address");
        // (control flow kept to what the K loop already has -- a counted loop over asm that updates the accumulators in place, zero
        // trips for a whole tile -- and ONE two-way branch, store or epilogue, behind which the accumulators are dead: a
        // three-way switch with the accumulators live across its merge cost 250 spilled registers)
        if (sk_nc > 0) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(g.sk_ctr + sk_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)sk_nc) {
                    if (++spins > (1 << 22)) {                   // ~1 s: never a hang -- the tile is wrong and the give-up counter says so
                        if (g.sk_err) __hip_atomic_fetch_add(g.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                __hip_atomic_store(g.sk_ctr + sk_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left at zero for the next launch
            }
            asm volatile("s_barrier" ::: "memory");
        }
        float sel[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sel[q] = li == 4 * q + lg ? 1.f : 0.f;
        constexpr int NG = NMF / 4, DEPTH = NG < 4 ? NG : 4;  // groups of four tiles (16 dword loads each); groups in flight
        for (int cidx = 0; cidx < sk_nc; ++cidx) {
            // piece cidx of the tile belongs to workgroup l + 1 + cidx of this XCD = block blockIdx.x + 8 (1 + cidx)
            const float *pp = g.sk_part + (size_t)(blockIdx.x + 8u * (1 + cidx)) * (BM * BN) + (size_t)w * NMF * 256 + li * 4 + lg;
            float t[DEPTH][16];
            auto issue = [&](int slot) {
                asm volatile("" : "+v"(pp));
                asm volatile(
                    "global_load_dword %0, %16, off sc0 sc1\n\tglobal_load_dword %1, %16, off offset:256 sc0 sc1\n\t"
                    "global_load_dword %2, %16, off offset:512 sc0 sc1\n\tglobal_load_dword %3, %16, off offset:768 sc0 sc1\n\t"
                    "global_load_dword %4, %16, off offset:1024 sc0 sc1\n\tglobal_load_dword %5, %16, off offset:1280 sc0 sc1\n\t"
                    "global_load_dword %6, %16, off offset:1536 sc0 sc1\n\tglobal_load_dword %7, %16, off offset:1792 sc0 sc1\n\t"
                    "global_load_dword %8, %16, off offset:2048 sc0 sc1\n\tglobal_load_dword %9, %16, off offset:2304 sc0 sc1\n\t"
                    "global_load_dword %10, %16, off offset:2560 sc0 sc1\n\tglobal_load_dword %11, %16, off offset:2816 sc0 sc1\n\t"
                    "global_load_dword %12, %16, off offset:3072 sc0 sc1\n\tglobal_load_dword %13, %16, off offset:3328 sc0 sc1\n\t"
                    "global_load_dword %14, %16, off offset:3584 sc0 sc1\n\tglobal_load_dword %15, %16, off offset:3840 sc0 sc1"
                    : "=&v"(t[slot][0]), "=&v"(t[slot][1]), "=&v"(t[slot][2]), "=&v"(t[slot][3]), "=&v"(t[slot][4]), "=&v"(t[slot][5]),
                      "=&v"(t[slot][6]), "=&v"(t[slot][7]), "=&v"(t[slot][8]), "=&v"(t[slot][9]), "=&v"(t[slot][10]), "=&v"(t[slot][11]),
                      "=&v"(t[slot][12]), "=&v"(t[slot][13]), "=&v"(t[slot][14]), "=&v"(t[slot][15])
                    : "v"(pp) : "memory");
                pp += 1024;
            };
#pragma unroll
            for (int gq = 0; gq < DEPTH; ++gq) issue(gq);
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                const int left = NG - 1 - gq < DEPTH - 1 ? NG - 1 - gq : DEPTH - 1;   // groups requested behind this one
                if (left == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
                else if (left == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else if (left == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = gq * 4 + q / 4;
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m / WNT][m % WNT]) : "v"(sel[q & 3]), "v"(t[gq % DEPTH][q]));
                }
                if (gq + DEPTH < NG) issue(gq % DEPTH);      // (the MFMAs read their operands at issue: the returning loads cannot overtake them)
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // MFMA results -> v_accvgpr_read: the asm is opaque to the hazard recogniser
        if (sk_kind == 1) {
            // (as EXEC-masked straight-line code instead of this branch the stores were tried too: the compiler schedules the
            // epilogue's own VALU work between the masked asm statements -- wrong embeddings -- and the spills stayed)
            run_epilogue = false;
            float *pp = g.sk_part + (size_t)blockIdx.x * (BM * BN) + ((size_t)w * NMF * 64 + lane) * 4;
#pragma unroll
            for (int m = 0; m < NMF; m += 4) {
                asm volatile("" : "+v"(pp));                     // (opaque: one base per four stores, not 64 addresses computed up front)
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc0 sc1\n\t"
                             "global_store_dwordx4 %0, %3, off offset:2048 sc0 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:3072 sc0 sc1"
                             ::"v"(pp), "a"(acc[m / WNT][m % WNT]), "a"(acc[(m + 1) / WNT][(m + 1) % WNT]), "a"(acc[(m + 2) / WNT][(m + 2) % WNT]),
                               "a"(acc[(m + 3) / WNT][(m + 3) % WNT]) : "memory");
                pp += 1024;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // acknowledged (device scope) ...
            asm volatile("s_barrier" ::: "memory");              // ... by every wave, before the piece counts itself in
            if (tid == 0) __hip_atomic_fetch_add(g.sk_ctr + sk_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (run_epilogue) {
    // (PERSIST: the epilogue's lane-dependent values start from an opaque copy of the lane number -- shadowing the kernel's --
    // so that nothing the epilogue derives from it is computed before the K loop, spilled across it and reloaded store by
    // store behind an s_waitcnt vmcnt(0) each: that made a 7 k-cycle epilogue 33 k)
    const int lane = opaque_lane();
    const int li = lane & 15, lg = lane >> 4;
    // every accumulator read of the epilogues goes through acc_read (asm, 'a'-class inputs): read in plain C++ the residual
    // epilogues carried 148-164 bytes of scratch per lane (reloads behind s_waitcnt vmcnt(0) between their stores), with it none
    // of the slab kernel's instantiations spills and the 8-wave ones need 104-109 VGPRs instead of 128: +1.3 % on the bulk
    // encode (1 848 -> 1 873 abstracts/s, alternating on one box: profiles/r06_persist_gemm_ab.txt)
    auto AC = [&](int i_, int j_) -> f32x4 { return acc_read(acc[i_][j_]); };
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int trow = m0 + (wm * WMT + i) * 16;
        const int row = trow + li;
        if constexpr (PERM == 1) {
            // plain bf16 store: the lane's 8 consecutive columns (one 16-byte value per tile pair) go through a wave-private LDS
            // block of 16 rows x OW columns and leave as whole rows -- 8 rows x 128 B (or 4 x 256 B) per store instruction
            // where the MFMA layout gives 16 rows x 64 B: the memory pipe charges per (instruction, line); +1.3 ... 2.5 % on
            // the K = 1536 shapes.  (The SwiGLU epilogue measured 7 200 cycles staged against 6 300 direct: its 128 outputs
            // per lane are v_exp / v_rcp time, not store issue -- it keeps the direct stores below.)
            constexpr int OW = PERM == 2 ? WNT * 8 : WNT * 16, ROWH = OW + 8, LPR = OW / 8, RPI = 64 / LPR;
            bf16_t *stg = smem + w * 16 * ROWH;
            if (i == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done with the ring
                asm volatile("s_barrier" ::: "memory");
            }
#pragma unroll
            for (int q = 0; q < OW / 32; ++q) {
                uint4 o;
                if constexpr (PERM == 2) {  // SwiGLU outputs from tiles (gate, up, gate, up)
                    float h[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float ga = AC(i, 4 * q)[c], gb = AC(i, 4 * q + 2)[c];
                        h[c] = ga * __builtin_amdgcn_rcpf(1.0f + __expf(-ga)) * AC(i, 4 * q + 1)[c];
                        h[4 + c] = gb * __builtin_amdgcn_rcpf(1.0f + __expf(-gb)) * AC(i, 4 * q + 3)[c];
                    }
                    o.x = pack2(h[0], h[1]); o.y = pack2(h[2], h[3]); o.z = pack2(h[4], h[5]); o.w = pack2(h[6], h[7]);
                } else {                    // plain store (+ bias) from tiles 2q, 2q+1
                    const int col0 = n0 + wn * WNT * 16 + 32 * q + 8 * lg;
                    float4 b0v = make_float4(0.f, 0.f, 0.f, 0.f), b1v = b0v;
                    if (g.bias && col0 < g.N) {
                        b0v = *reinterpret_cast<const float4 *>(g.bias + col0);
                        b1v = *reinterpret_cast<const float4 *>(g.bias + col0 + 4);
                    }
                    const f32x4 v0 = AC(i, 2 * q), v1 = AC(i, 2 * q + 1);
                    o.x = pack2(v0[0] + b0v.x, v0[1] + b0v.y); o.y = pack2(v0[2] + b0v.z, v0[3] + b0v.w);
                    o.z = pack2(v1[0] + b1v.x, v1[1] + b1v.y); o.w = pack2(v1[2] + b1v.z, v1[3] + b1v.w);
                }
                *reinterpret_cast<uint4 *>(stg + li * ROWH + 32 * q + 8 * lg) = o;
            }
            const int c8 = lane % LPR;
            const int col = (PERM == 2 ? (n0 + wn * WNT * 16) / 2 : n0 + wn * WNT * 16) + c8 * 8;
            const int ncols = PERM == 2 ? g.ldc : g.N;
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int r = it * RPI + lane / LPR;
                const uint4 o = *reinterpret_cast<const uint4 *>(stg + r * ROWH + c8 * 8);
                if (trow + r < g.M && col < ncols) *reinterpret_cast<uint4 *>(g.C + (size_t)(trow + r) * g.ldc + col) = o;
            }
        } else if constexpr (PERM == 2) {   // 8 consecutive SwiGLU outputs per lane from tiles (gate, up, gate, up)
            const float sc = row_f(i, li);   // (fused RMSNorm: 1/rms of the lane's row; 1 without)
#pragma unroll
            for (int q = 0; q < WNT / 4; ++q) {
                const int col0 = (n0 + wn * WNT * 16) / 2 + 32 * q + 8 * lg;
                if (row < g.M && col0 < g.ldc) {
                    float h[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float ga = AC(i, 4 * q)[c] * sc, gb = AC(i, 4 * q + 2)[c] * sc;
                        h[c] = ga * __builtin_amdgcn_rcpf(1.0f + __expf(-ga)) * (AC(i, 4 * q + 1)[c] * sc);
                        h[4 + c] = gb * __builtin_amdgcn_rcpf(1.0f + __expf(-gb)) * (AC(i, 4 * q + 3)[c] * sc);
                    }
                    uint4 o;
                    o.x = pack2(h[0], h[1]); o.y = pack2(h[2], h[3]); o.z = pack2(h[4], h[5]); o.w = pack2(h[6], h[7]);
                    *reinterpret_cast<uint4 *>(g.C + (size_t)row * g.ldc + col0) = o;
                }
            }
        } else if constexpr (PERM == 1) {   // 8 consecutive columns per lane from tiles 2J, 2J+1
#pragma unroll
            for (int J = 0; J < WNT / 2; ++J) {
                const int col0 = n0 + wn * WNT * 16 + 32 * J + 8 * lg;
                if (row < g.M && col0 < g.N) {
                    float4 b0v = make_float4(0.f, 0.f, 0.f, 0.f), b1v = b0v;
                    if (g.bias) {
                        b0v = *reinterpret_cast<const float4 *>(g.bias + col0);
                        b1v = *reinterpret_cast<const float4 *>(g.bias + col0 + 4);
                    }
                    const f32x4 v0 = AC(i, 2 * J), v1 = AC(i, 2 * J + 1);
                    uint4 o;
                    o.x = pack2(v0[0] + b0v.x, v0[1] + b0v.y); o.y = pack2(v0[2] + b0v.z, v0[3] + b0v.w);
                    o.z = pack2(v1[0] + b1v.x, v1[1] + b1v.y); o.w = pack2(v1[2] + b1v.z, v1[3] + b1v.w);
                    *reinterpret_cast<uint4 *>(g.C + (size_t)row * g.ldc + col0) = o;
                }
            }
        } else if constexpr (EPI == EPI_RESID || EPI == EPI_F32H) {
            if (ksplit == 1 || (EPI == EPI_RESID && g.part)) {
                // X += acc through a wave-private LDS staging block: straight from the MFMA layout an instruction
                // touches 16 rows x 64 B, and the memory pipe charges ~3.5 cycles per (instruction, 128-byte line)
                // whatever the bytes (tools/micro/store_pattern.hip: 128 KiB in 4.3 us as 64-byte segments, 1.6 us
                // as whole lines) -- the read-modify-write of a 256 KiB f32 tile was ~17 us of a 55 us O-projection
                // tile.  Staged, a wave's 16 x CW block goes out (and X comes in) as 256 / CW rows x CW*4 B per
                // instruction.  Rows are padded by 16 B: the 8 lanes of a ds_write_b128 group sit on 8 rows.
                constexpr int CW = WNT * 16, ROWF = CW + 4, LPR = CW / 4, RPI = 64 / LPR;   // floats per staged row, lanes per row, rows per instruction
                const bool lane_on = lane < RPI * LPR;               // CW = 96: 2 rows x 24 lanes per instruction, 16 lanes idle
                float *stg = reinterpret_cast<float *>(smem) + w * 16 * ROWF;
                if (i == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done with the ring
                    asm volatile("s_barrier" ::: "memory");
                }
#pragma unroll
                for (int j = 0; j < WNT; ++j)
                    *reinterpret_cast<f32x4 *>(stg + li * ROWF + j * 16 + 4 * lg) = AC(i, j);
                const int c4 = lane % LPR, col = n0 + wn * CW + c4 * 4;
                if (EPI == EPI_RESID && ksplit > 1) {     // a K slice's partial tile -> its plane of the workspace (whole lines)
                    float *pp = g.part + (size_t)ks_cur * g.M * g.N;
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int r = it * RPI + lane / LPR;
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(stg + r * ROWF + c4 * 4);
                        if (lane_on && trow + r < g.M && col < g.N)
                            *reinterpret_cast<float4 *>(pp + (size_t)(trow + r) * g.N + col) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                    continue;
                }
                if constexpr (EPI == EPI_F32H) {          // plain f32 store (the index library's approximate scores)
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int r = it * RPI + lane / LPR;
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(stg + r * ROWF + c4 * 4);
                        if (trow + r < g.M && col < g.N)
                            *reinterpret_cast<float4 *>(g.X + (size_t)(trow + r) * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                        if constexpr (CW == 64) {
                            // the staged row of this wave is one 64-column group: its maximum rides along (16 lanes per row)
                            if (g.gmax) {
                                const bool okc = col < g.N;
                                const bool fin = fabsf(v[0]) <= 3.0e38f && fabsf(v[1]) <= 3.0e38f && fabsf(v[2]) <= 3.0e38f && fabsf(v[3]) <= 3.0e38f;
                                float m = !okc ? -__builtin_inff() : !fin ? __builtin_inff() : fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                                m = row16_max(m);
                                const int grp = (n0 + wn * CW) >> 6;
                                if (c4 == 0 && trow + r < g.M && grp < g.ld_gmax) g.gmax[(size_t)(trow + r) * g.ld_gmax + grp] = m;
                            }
                        }
                    }
                } else {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g.bias && col < g.N) bv = *reinterpret_cast<const float4 *>(g.bias + col);
                // fused RMSNorm, producer side (GemmArgs::ssq_out): the unscaled normalised row and the columns' sum of squares
                const bool fuse = EPI == EPI_RESID && g.ssq_out != nullptr;
                float4 nw = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fuse && col < g.N) nw = *reinterpret_cast<const float4 *>(g.norm_w + col);
                float *sred = reinterpret_cast<float *>(smem) + 10240 + w * 512;   // [16 / RPI][64] behind the staging blocks
                // (requesting X two or four 16-row blocks ahead, so that the eight blocks of a wave are not eight dependent round
                // trips, measured no better: the in-kernel stamps put this epilogue at ~45 000 cycles of a ~110 000-cycle
                // O-projection tile because all 256 tiles of a round move 128 MiB at once -- HBM-bound, not latency-bound)
                float4 xs[16 / RPI];
#pragma unroll
                for (int it = 0; it < 16 / RPI; ++it) {
                    const int r = it * RPI + lane / LPR;
                    if (lane_on && trow + r < g.M && col < g.N) xs[it] = *reinterpret_cast<const float4 *>(g.X + (size_t)(trow + r) * g.ldc + col);
                }
#pragma unroll
                for (int it = 0; it < 16 / RPI; ++it) {
                    const int r = it * RPI + lane / LPR;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(stg + r * ROWF + c4 * 4);
                    float ss = 0.f;
                    if (lane_on && trow + r < g.M && col < g.N) {
                        float4 x = xs[it];
                        x.x += v[0] + bv.x; x.y += v[1] + bv.y; x.z += v[2] + bv.z; x.w += v[3] + bv.w;
                        *reinterpret_cast<float4 *>(g.X + (size_t)(trow + r) * g.ldc + col) = x;
                        if (fuse) {
                            uint2 o;
                            o.x = pack2(x.x * nw.x, x.y * nw.y);
                            o.y = pack2(x.z * nw.z, x.w * nw.w);
                            *reinterpret_cast<uint2 *>(g.norm_y + (size_t)(trow + r) * g.ldc + col) = o;
                            ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
                        }
                    }
                    if (fuse) sred[it * 64 + lane] = ss;
                }
                if (fuse && lane < 16) {
                    // row `lane` of the block: its LPR lanes' sums, in lane order (a fixed order: bit-reproducible)
                    const float4 *sp = reinterpret_cast<const float4 *>(sred + (lane / RPI) * 64 + (lane % RPI) * LPR);
                    float t = 0.f;
#pragma unroll
                    for (int q = 0; q < LPR / 4; ++q) {
                        const float4 a = sp[q];
                        t += a.x; t += a.y; t += a.z; t += a.w;
                    }
                    if (trow + lane < g.M) g.ssq_out[(size_t)((n0 + wn * CW) / CW) * g.M + trow + lane] = t;   // slot-major: one 64-byte segment
                }
                }
            } else {
#pragma unroll
                for (int j = 0; j < WNT; ++j) store_tile_t<EPI>(ge, AC(i, j), AC(i, j), trow, n0 + (wn * WNT + j) * 16, lane);
            }
        } else if constexpr (SWAP) {
#pragma unroll
            for (int j = 0; j < WNT; ++j) store_tile_t<EPI>(ge, AC(i, j), AC(i, j), trow, n0 + (wn * WNT + j) * 16, lane);
        } else if constexpr (EPI == EPI_QKV) {
            if (n0 + BN <= g.qk_cols) {
                // Q / K columns (row-major bf16): the untransposed accumulator tile gives 32-byte row segments per store;
                // staged like the residual epilogue, a wave's 16 x CW block leaves as whole 128-byte lines.  (The V
                // columns keep the direct path: their output is V^T, 4 consecutive tokens per lane.)
                constexpr int CW = WNT * 16, ROWH = CW + 8, LPR = CW / 8, RPI = 64 / LPR;   // bf16 per staged row (+16 B pad), lanes per row, rows per instruction
                bf16_t *stg = smem + w * 16 * ROWH;
                if (i == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done with the ring
                    asm volatile("s_barrier" ::: "memory");
                }
                const int srow = 4 * lg + (li & 3), scol0 = 4 * (li >> 2);
                const float sc = row_f(i, srow);             // fused RMSNorm: 1/rms of the lane's row (1 without)
                float4 cs[WNT];
                if (g.rope_cs) {
                    // (cos, sin) of the wave's 16 rows x CW/2 rotary pairs (columns 2f, 2f+1 = pair f): 256 contiguous bytes per
                    // row, fetched as whole lines (4 rows per instruction) into a wave-private LDS block and read back per
                    // lane -- straight from the table a lane's float4 is 16 rows x 64 B per instruction, twice the lines, and
                    // the CU's memory pipe charges per (instruction, line): the first version cost what rope_kernel takes
                    char *csb = reinterpret_cast<char *>(smem) + 24576 + w * 5120;   // [16 rows][320 B] behind the staging blocks
                    const int fb = ((n0 + wn * CW) & (g.rope_hd - 1)) >> 1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rr = 4 * k + (lane >> 4);
                        const int pp = row_i(i, rr);         // rotary position of row rr
                        const float4 v = *reinterpret_cast<const float4 *>(g.rope_cs + (size_t)pp * (g.rope_hd >> 1) + fb + 2 * (lane & 15));
                        *reinterpret_cast<float4 *>(csb + rr * 320 + (lane & 15) * 16) = v;
                    }
#pragma unroll
                    for (int j = 0; j < WNT; ++j) cs[j] = *reinterpret_cast<const float4 *>(csb + srow * 320 + 64 * j + 16 * (li >> 2));
                }
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    const f32x4 v = quad_transpose(AC(i, j), lane);
                    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g.bias) b = *reinterpret_cast<const float4 *>(g.bias + n0 + wn * CW + j * 16 + scol0);
                    float x0 = v[0] * sc + b.x, x1 = v[1] * sc + b.y, x2 = v[2] * sc + b.z, x3 = v[3] * sc + b.w;
                    if (g.rope_cs) {
                        const float4 c = cs[j];
                        const float y0 = x0 * c.x - x1 * c.y, y1 = x1 * c.x + x0 * c.y;
                        const float y2 = x2 * c.z - x3 * c.w, y3 = x3 * c.z + x2 * c.w;
                        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
                    }
                    uint2 o;
                    o.x = pack2(x0, x1);
                    o.y = pack2(x2, x3);
                    *reinterpret_cast<uint2 *>(stg + srow * ROWH + j * 16 + scol0) = o;
                }
                const int c8 = lane % LPR, col = n0 + wn * CW + c8 * 8;
#pragma unroll
                for (int it = 0; it < 16 / RPI; ++it) {
                    const int r = it * RPI + lane / LPR;
                    const uint4 o = *reinterpret_cast<const uint4 *>(stg + r * ROWH + c8 * 8);
                    if (trow + r < g.M) *reinterpret_cast<uint4 *>(g.C + (size_t)(trow + r) * g.ldc + col) = o;
                }
            } else if (vt_staged) {
                // V columns, a whole tile of them: the lane's 4 consecutive tokens of one channel go to the wave's LDS block
                // [channel][128 tokens]; the block leaves below as whole 256-byte channel rows of V^T
                if (i == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done with the ring
                    asm volatile("s_barrier" ::: "memory");
                }
                bf16_t *stg = smem + w * (WNT * 16) * VROW;
                const float s0 = row_f(i, 4 * lg), s1 = row_f(i, 4 * lg + 1), s2 = row_f(i, 4 * lg + 2), s3 = row_f(i, 4 * lg + 3);
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    const float bv = g.bias ? g.bias[n0 + (wn * WNT + j) * 16 + li] : 0.f;
                    const f32x4 v = AC(i, j);
                    uint2 o;
                    o.x = pack2(v[0] * s0 + bv, v[1] * s1 + bv);
                    o.y = pack2(v[2] * s2 + bv, v[3] * s3 + bv);
                    *reinterpret_cast<uint2 *>(stg + (j * 16 + li) * VROW + i * 16 + 4 * lg) = o;
                }
            } else {
                const float s0 = row_f(i, 4 * lg), s1 = row_f(i, 4 * lg + 1), s2 = row_f(i, 4 * lg + 2), s3 = row_f(i, 4 * lg + 3);
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    f32x4 v = AC(i, j);
                    v[0] *= s0; v[1] *= s1; v[2] *= s2; v[3] *= s3;
                    store_tile<EPI>(ge, v, v, trow, n0 + (wn * WNT + j) * 16, lane);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WNT; ++j) store_tile<EPI>(ge, AC(i, j), AC(i, j), trow, n0 + (wn * WNT + j) * 16, lane);
        }
    }
    if constexpr (EPI == EPI_QKV) {
        if (vt_staged) {
            // straight from the accumulators a store is 16 channels x 32 bytes (the V tiles' epilogue measured ~50 000 cycles
            // against the Q / K tiles' ~12 000: in-kernel stamps, profiles/r03_encoder_gemm_stamps.txt); from the block, 4 channels x 256 bytes
            const bf16_t *stg = smem + w * (WNT * 16) * VROW;
            const int chunk = lane & 15, tok0 = m0 + wm * WMT * 16 + chunk * 8;
#pragma unroll
            for (int it = 0; it < WNT * 4; ++it) {
                const int ch = it * 4 + (lane >> 4);
                const uint4 o = *reinterpret_cast<const uint4 *>(stg + ch * VROW + chunk * 8);
                if (tok0 < g.M)
                    *reinterpret_cast<uint4 *>(g.Vt + (size_t)(n0 - g.qk_cols + wn * WNT * 16 + ch) * g.ldvt + tok0) = o;
            }
        }
    }
    }   // run_epilogue
    stamp(3);
    if (g.ts) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(4);
    }
    if (!more) break;
    tm = tm2; tn = tn2; ksplit = ksplit2; kt0 = kt02; nk = nk2;
    if constexpr (PERSIST) {
        sk_kind = kind2; sk_nc = nc2; sk_tile = tile2;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the staging blocks are read: the rest of the ring is free ...
        asm volatile("s_barrier" ::: "memory");                  // ... in every wave
        asm volatile("" : "+s"(tm), "+s"(tn), "+s"(kt0));        // (opaque: the unit's source pointers are computed HERE, not before the epilogue)
        set_sources(tm, tn, kt0);
        set_read_addresses();
        kn = max(nk >> 1, 1);
        request_slabs(nk, early ? 2 : 0, DQ);
        load_row_consts(tm);
        if (g.ts) {                                              // the next unit's stamps: its own row
            ts_row += (int)gridDim.x;
            ts0 = __builtin_amdgcn_s_memtime();
            stamp_row();
        }
    }
    }   // unit loop
}

// X[m][n] += bias[n] + sum_s part[s][m][n], s ascending (a fixed order: the split-K result is bit-reproducible).  One
// float4 per thread; the planes are read once, X read and written once.
__global__ void __launch_bounds__(256)
    splitk_reduce_kernel(float *__restrict__ X, int64_t ldx, const float *__restrict__ part, int S, int M, int N,
                         const float *__restrict__ bias) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index inside [M][N]
    const int n4 = N >> 2;
    if (i >= (int64_t)M * n4) return;
    const int m = (int)(i / n4), c = (int)(i - (int64_t)m * n4) * 4;
    const size_t plane = (size_t)M * N;
    const float *p = part + (size_t)m * N + c;
    float4 acc = *reinterpret_cast<const float4 *>(p);
    for (int sidx = 1; sidx < S; ++sidx) {
        const float4 v = *reinterpret_cast<const float4 *>(p + sidx * plane);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + c);
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    float4 *px = reinterpret_cast<float4 *>(X + (size_t)m * ldx + c);
    float4 x = *px;
    x.x += acc.x; x.y += acc.y; x.z += acc.z; x.w += acc.w;
    *px = x;
}

// The same reduction with the RMSNorm that reads the updated row folded in (N <= 2048, ldx == N): one workgroup per row,
// thread t owns the float4 chunks at columns 4 t and 4 t + 1024; the row's sum of squares goes wave -> LDS -> a fixed-order
// sum, so the result does not depend on anything but the row.  Saves the separate pass over X (and its launch) for the
// few-hundred-token batches the split-K path serves.
__global__ void __launch_bounds__(256)
    splitk_reduce_norm_kernel(float *__restrict__ X, const float *__restrict__ part, int S, int M, int N,
                              const float *__restrict__ bias, const float *__restrict__ w, float eps, bf16_t *__restrict__ y) {
    __shared__ float wsum[4];
    const int m = blockIdx.x, t = threadIdx.x;
    const size_t plane = (size_t)M * N;
    float4 v[2];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t * 4 + 1024 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < N) {
            const float *p = part + (size_t)m * N + c;
            float4 *px = reinterpret_cast<float4 *>(X + (size_t)m * N + c);
            float4 x = *px;
            float4 acc = *reinterpret_cast<const float4 *>(p);
            for (int s0 = 1; s0 < S; s0 += 8) {          // eight planes' loads in flight, added in ascending order
                float4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const float4 *>(p + (size_t)min(s0 + u, S - 1) * plane);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + u < S) { acc.x += q[u].x; acc.y += q[u].y; acc.z += q[u].z; acc.w += q[u].w; }
            }
            if (bias) {
                const float4 b = *reinterpret_cast<const float4 *>(bias + c);
                acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            }
            x.x += acc.x; x.y += acc.y; x.z += acc.z; x.w += acc.w;
            *px = x;
            v[j] = x;
            ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
    }
    ss = wave_sum(ss);
    if ((t & 63) == 0) wsum[t >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) / (float)N + eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t * 4 + 1024 * j;
        if (c < N) {
            const float4 g = *reinterpret_cast<const float4 *>(w + c);
            uint2 o;
            o.x = pack2(v[j].x * inv * g.x, v[j].y * inv * g.y);
            o.y = pack2(v[j].z * inv * g.z, v[j].w * inv * g.w);
            *reinterpret_cast<uint2 *>(y + (size_t)m * N + c) = o;
        }
    }
}

// The QKV projection's reduction pass: one workgroup per token sums the S planes of its row (ascending: a fixed order),
// adds the bias, rotates the Q / K heads (the pair (j, j + hd/2) of a head lives in the same row: f32, no second pass over
// bf16 values) and writes the Q|K row and the token's column of V^T.  N <= 4096 (a thread owns columns 4 t .. and + 1024 ..).
__global__ void __launch_bounds__(256)
    splitk_reduce_qkv_kernel(const float *__restrict__ part, int S, int M, int N, const float *__restrict__ bias,
                             bf16_t *__restrict__ qk, int ldqk, int qk_cols, bf16_t *__restrict__ vt, int ldvt, int hd,
                             const int32_t *__restrict__ pos, const float *__restrict__ cos_t, const float *__restrict__ sin_t) {
    extern __shared__ __attribute__((aligned(16))) float row[];      // [N] the summed row (+ bias)
    const int m = blockIdx.x, t = threadIdx.x;
    const size_t plane = (size_t)M * N;
    for (int c = t * 4; c < N; c += 1024) {
        const float *p = part + (size_t)m * N + c;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(p);
        for (int s0 = 1; s0 < S; s0 += 4) {                       // four planes' loads in flight, added in ascending order
            f32x4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)min(s0 + u, S - 1) * plane);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s0 + u < S) { acc[0] += q[u][0]; acc[1] += q[u][1]; acc[2] += q[u][2]; acc[3] += q[u][3]; }
        }
        if (bias) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + c);
            acc[0] += b[0]; acc[1] += b[1]; acc[2] += b[2]; acc[3] += b[3];
        }
        *reinterpret_cast<f32x4 *>(row + c) = acc;
    }
    __syncthreads();
    const int half = hd / 2, ps = pos[m];
    for (int c = t * 4; c < N; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
        if (c < qk_cols) {
            const int j = c % hd, jj = j % half;                  // 4 consecutive features never straddle a half (hd % 8 == 0)
            const f32x4 pv = *reinterpret_cast<const f32x4 *>(row + (j < half ? c + half : c - half));
            const f32x4 cs = *reinterpret_cast<const f32x4 *>(cos_t + (size_t)ps * half + jj);
            const f32x4 sn = *reinterpret_cast<const f32x4 *>(sin_t + (size_t)ps * half + jj);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = j < half ? v[r] * cs[r] - pv[r] * sn[r] : v[r] * cs[r] + pv[r] * sn[r];
            uint2 w;
            w.x = pack2(o[0], o[1]);
            w.y = pack2(o[2], o[3]);
            *reinterpret_cast<uint2 *>(qk + (size_t)m * ldqk + c) = w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) vt[(size_t)(c - qk_cols + r) * ldvt + m] = f2bf(v[r]);
        }
    }
}

// ---------------------------------------------------------------------
// Skinny GEMM for a handful of tokens (M <= 32: one query, or a few) over the
// fragment-major weight copy.  The job is weight streaming: W never touches LDS --
// each wave reads its B fragments (16 W rows x 32 k = one contiguous 1 KiB piece of
// Wt, 16 bytes per lane) straight into registers, UNR K steps ahead, and multiplies
// them with the A fragments of the (at most two) 16-token tiles, which sit in LDS for
// the whole kernel (padded rows: conflict-free ds_read_b128, read one step ahead).
// No barrier, no DMA and no LDS traffic for W inside the K loop.  A wave owns WN
// 16-row blocks of W (WN = 2 for SwiGLU: one gate/up pair); a workgroup is 1-4
// independent waves; K is split over workgroups in slices of at most SKINNY_KS_MAX
// (the A slice must fit LDS), with f32 atomics for the residual epilogue (the only
// one that may be split).
// ---------------------------------------------------------------------
constexpr int SKINNY_KS_MAX = 1536;
template <int EPI, int WN>
__global__ void __launch_bounds__(256) gemm_bf16_skinny_kernel(GemmArgs g, int kslice) {
    constexpr int UNR = 8;   // W fragments in flight per owned row block
    extern __shared__ __attribute__((aligned(16))) bf16_t a_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ksplit = g.ksplit > 1 ? g.ksplit : 1;
    const int ks = (int)blockIdx.x % ksplit, nb = (int)blockIdx.x / ksplit;
    const int k_lo = ks * kslice, k_hi = min(g.K, k_lo + kslice), klen = k_hi - k_lo;
    if (klen <= 0) return;
    const int mtiles = (g.M + 15) / 16;                  // 1 or 2
    const int stride = klen + 8;                         // elements: one 16-byte slot of padding per row

    const int n0 = (nb * (int)(blockDim.x >> 6) + w) * WN * 16;   // first W row of this wave
    // B fragment of (row block, K step): lane (n = li, k group lg) wants W[row li][8 lg ..]; in the
    // piece that element group sits at DMA-lane position 4 li + (lg ^ ((-(li >> 2)) & 3))
    const int jpos = li * 4 + (lg ^ ((0 - (li >> 2)) & 3));
    const bf16_t *wpiece[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int nblk = min(n0 / 16 + j, g.N / 16 - 1);
        wpiece[j] = g.Wt + ((size_t)nblk * (g.K / 32) + k_lo / 32) * 512 + jpos * 8;
    }
    const int nkk = klen / 32;
    bf16x8 bq[UNR][WN];
    auto wload = [&](int kk, bf16x8(&b)[WN]) {
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(wpiece[j] + (size_t)kk * 512);
    };
    // the first UNR weight fragments are requested before anything else: their latency
    // overlaps the staging of A
#pragma unroll
    for (int u = 0; u < UNR; ++u) wload(min(u, nkk - 1), bq[u]);

    // A[0 .. 16 mtiles)[k_lo .. k_hi) -> LDS (rows clamped to M - 1; coalesced 16-byte pieces,
    // eight loads per thread in flight; unconditional stores -- a clamped index rewrites the
    // last piece with its own data: a guarded store is a branch, and behind a branch hipcc
    // waits vmcnt(0) per load)
    {
        const int slots = klen / 8, total = mtiles * 16 * slots, nthr = (int)blockDim.x;
        for (int base = tid; base < total; base += 8 * nthr) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(base + u * nthr, total - 1), r = i / slots, c = i - r * slots;
                v[u] = *reinterpret_cast<const uint4 *>(g.A + (size_t)min(r, g.M - 1) * g.lda + k_lo + c * 8);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(base + u * nthr, total - 1), r = i / slots, c = i - r * slots;
                *reinterpret_cast<uint4 *>(a_lds + r * stride + c * 8) = v[u];
            }
        }
    }
    __syncthreads();
    if (n0 >= g.N) return;
    const bf16_t *arow = a_lds + li * stride + lg * 8;

    f32x4 acc[2][WN];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A fragments one step ahead of their MFMAs
    const int a1off = mtiles > 1 ? 16 * stride : 0;
    bf16x8 an0 = *reinterpret_cast<const bf16x8 *>(arow);
    bf16x8 an1 = *reinterpret_cast<const bf16x8 *>(arow + a1off);
    auto step = [&](int kk, const bf16x8(&b)[WN]) {
        const bf16x8 a0 = an0, a1 = an1;
        const int kn = min(kk + 1, nkk - 1);
        an0 = *reinterpret_cast<const bf16x8 *>(arow + kn * 32);
        an1 = *reinterpret_cast<const bf16x8 *>(arow + a1off + kn * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[j], acc[0][j], 0, 0, 0);
        if (mtiles > 1) {
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[j], acc[1][j], 0, 0, 0);
        }
    };
    // ring of UNR fragment sets: unconditional loads in the steady state, clamped K index in the tail
    int kk0 = 0;
    for (; kk0 + 2 * UNR <= nkk; kk0 += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            step(kk0 + u, bq[u]);
            wload(kk0 + u + UNR, bq[u]);
        }
    }
    for (; kk0 < nkk; kk0 += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (kk0 + u < nkk) {
                step(kk0 + u, bq[u]);
                wload(min(kk0 + u + UNR, nkk - 1), bq[u]);
            }
        }
    }

    GemmArgs ge = g;
    ge.ksplit = ksplit;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m < mtiles) {
            if constexpr (EPI == EPI_SWIGLU) {
                static_assert(EPI != EPI_SWIGLU || WN == 2, "SwiGLU: a wave owns one gate/up pair");
                store_tile<EPI>(ge, acc[m][0], acc[m][WN - 1], m * 16, n0 / 2, lane);
            } else {
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    if (n0 + j * 16 < g.N) store_tile<EPI>(ge, acc[m][j], acc[m][j], m * 16, n0 + j * 16, lane);
            }
        }
    }
}

// ---------------------------------------------------------------------
// Rotary embedding in place on the q/k part of the QKV output:
// X[t][h*hd + i], X[t][h*hd + i + hd/2]  (HF rotate_half pairing),
// cos/sin f32 tables [max_seq][hd/2]; thread = (token, head, 8 pairs).
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    rope_kernel(bf16_t *__restrict__ X, int ldx, int nheads_qk, int hd, const int32_t *__restrict__ pos,
                const float *__restrict__ cos_t, const float *__restrict__ sin_t, int T) {
    const int per_tok = nheads_qk * (hd / 16);
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (int64_t)T * per_tok) return;
    const int t = (int)(gi / per_tok), rem = (int)(gi - (int64_t)t * per_tok);
    const int h = rem / (hd / 16), i0 = (rem - h * (hd / 16)) * 8;
    bf16_t *p1 = X + (size_t)t * ldx + h * hd + i0;
    bf16_t *p2 = p1 + hd / 2;
    const float *c = cos_t + (size_t)pos[t] * (hd / 2) + i0;
    const float *s = sin_t + (size_t)pos[t] * (hd / 2) + i0;
    const uint4 v1 = *reinterpret_cast<const uint4 *>(p1), v2 = *reinterpret_cast<const uint4 *>(p2);
    const unsigned a[4] = {v1.x, v1.y, v1.z, v1.w}, b[4] = {v2.x, v2.y, v2.z, v2.w};
    unsigned o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x1l = __uint_as_float(a[j] << 16), x1h = __uint_as_float(a[j] & 0xffff0000u);
        const float x2l = __uint_as_float(b[j] << 16), x2h = __uint_as_float(b[j] & 0xffff0000u);
        const float cl = c[2 * j], ch = c[2 * j + 1], sl = s[2 * j], sh = s[2 * j + 1];
        o1[j] = pack2(x1l * cl - x2l * sl, x1h * ch - x2h * sh);
        o2[j] = pack2(x2l * cl + x1l * sl, x2h * ch + x1h * sh);
    }
    *reinterpret_cast<uint4 *>(p1) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    *reinterpret_cast<uint4 *>(p2) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

// ---------------------------------------------------------------------
// Varlen attention.  A work item is a 64-row query block of one sequence x a group of HPW heads; a head takes
// 4 waves x 16 query rows; the workgroups are persistent (see the kernel).  Everything is computed TRANSPOSED so that a lane owns
// one query row: per 64-key chunk S^T = K Q^T (A operand = K rows, 16 MFMA per wave), the lane holds 16
// scores of its query (row statistics: in-lane + two cross-lane steps over the 4 lanes that share the row),
// online softmax in registers, and O^T += V^T P^T (A operand = V^T rows, B operand = P) -- 16 MFMA per wave.
// Which key an MFMA row position of S^T stands for is free: tile j, position p <-> key
// 32 (j/2) + 8 (p/4) + 4 (j%2) + p%4, so that the lane's values of tiles 2J, 2J+1 are the 8 consecutive keys
// 32 J + 8 lg .. -- exactly the B fragment of the PV MFMA: P never leaves the registers (no LDS round trip,
// no cross-lane transposes; the untransposed version spent 70 % of its issue slots on VALU work).
// K rows sit in LDS with their 16-byte slots XORed by a key of row bits {0, 1, 3, 4} (HD = 64: {0, 1, 3}): the
// 16 rows a ds_read_b128 lane group touches under that map fall on 16 distinct slots.
// HPW heads per workgroup (4 waves each) share one K/V head (GQA): the chunk a workgroup stages serves all
// of them (stella: 12 query heads on 2 K/V heads; 64 KiB of LDS, two workgroups per CU).
// ---------------------------------------------------------------------
struct AttnArgs {
    const bf16_t *QK;   // [T_pad][ldqk]: q heads then k heads (RoPE applied)
    const bf16_t *Vt;   // [n_kv*hd][ldvt]
    bf16_t *O;          // [T_pad][n_heads*hd]
    const int32_t *work_seq, *work_q0;   // [nwork]
    const int32_t *seq_start, *seq_len;  // [nseq] (padded-packed token offsets)
    int ldqk, ldvt, n_heads, n_kv, causal;
    float scale;
    int nwork;          // work items (64-row query blocks); the launch is 1-D and persistent: min(nwork * n_heads / HPW, 2 per CU) workgroups
    bf16_t *Ofrag;      // non-null (the query-time path, encoder_few.h): the output goes here as B-operand fragments instead of O --
    int frag_mt;        // piece (K step d / 32, token tile t / 16) of frag_mt tiles per step: lane ((d % 32) / 8, t % 16) holds 8 dims
};

template <int HD, int HPW = 1>
__global__ void __launch_bounds__(256 * HPW, 4) attn_kernel(AttnArgs a) {   // (HIP: min waves per SIMD) four waves per SIMD = two 8-wave workgroups per CU: at most 128 VGPRs
    constexpr int KC = 64;          // keys per chunk
    constexpr int NKK = HD / 32;    // MFMA k-steps over the head dimension
    constexpr int NDT = HD / 16;    // 16-wide output tiles over the head dimension
    constexpr int KSL = HD / 8;     // 16-byte slots per K row
    constexpr int KRPP = 64 / KSL;  // K rows per 1-KiB DMA piece
    constexpr int STG = 2 * KC * HD;  // elements per stage: K tile + V^T tile
    // one LDS array: [2 stages][K: key x HD, slot ^= kswz(key) | V^T: d x 64 keys, slot ^= d&7]
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STG];
    static_assert((HD / 32) % HPW == 0, "DMA pieces must divide over the waves");
    auto kswz = [](int key) { return (key & 3) | (((key >> 3) & (KSL / 4 - 1)) << 2); };

    // PERSISTENT: the launch is gridDim.x workgroups (two per CU) that walk the items it, it + gridDim.x, ...; an item =
    // (64-row query block of a sequence, group of HPW heads), head group fastest (neighbouring workgroups share the
    // sequence's K / V in L2).  While the LAST key chunk of an item is multiplied, the first chunk and the Q fragments of the
    // next item are already in flight: a 220-token abstract is 4 chunks per item, and with one item per workgroup the first of
    // them was fetched with nothing to overlap it (SQ counters: 46 % of the wave cycles waiting, profiles/r03_attn_pmc.txt).
    const int nhp = a.n_heads / HPW, nitems = a.nwork * nhp;
    int it = blockIdx.x;
    if (it >= nitems) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: head w / 4 of this workgroup, query rows 16 (w % 4) ..
    const int wq = w & 3;
    const int li = lane & 15, lg = lane >> 4;

    // Sequences are packed back to back (no alignment padding between them), but the V^T rows are fetched in 16-byte
    // pieces: the key axis of a sequence therefore starts at the aligned-down token sa; its first `off` (< 8) keys
    // belong to the previous sequence and are masked out like the keys past the end.  Le = keys on that axis.
    struct Item { int s0, L, sa, off, Le, q0, h, kvh, kend; };
    auto decode = [&](int it_) {
        Item t;
        const int wi = it_ / nhp, hp = it_ - wi * nhp;
        const int seq = a.work_seq[wi];
        t.q0 = a.work_q0[wi];
        t.s0 = a.seq_start[seq]; t.L = a.seq_len[seq];
        t.sa = t.s0 & ~7; t.off = t.s0 - t.sa; t.Le = t.off + t.L;
        t.h = hp * HPW + (w >> 2);
        t.kvh = (hp * HPW) / (a.n_heads / a.n_kv);               // the same for every head of the workgroup (host checks)
        t.kend = a.causal ? min(t.Le, t.off + t.q0 + 64) : t.Le;
        return t;
    };

    // K and V^T chunks arrive by LDS-DMA (1 KiB per wave instruction), double
    // buffered: chunk c+1 is in flight while chunk c is multiplied
    auto issue = [&](const Item &t, int stage, int kc) {
        bf16_t *Ks = smem + stage * STG, *Vs = Ks + KC * HD;
#pragma unroll
        for (int i = 0; i < HD / 32 / HPW; ++i) {   // HD/8 K pieces over the 4 HPW waves
            const int p = w * (HD / 32 / HPW) + i;
            const int key = p * KRPP + lane / KSL, sl = lane % KSL;
            const int krow = min(kc + key, t.Le - 1);
            dma16(a.QK + (size_t)(t.sa + krow) * a.ldqk + (a.n_heads + t.kvh) * HD + ((sl ^ kswz(key)) * 8),
                  Ks + p * 512);
        }
#pragma unroll
        for (int i = 0; i < HD / 32 / HPW; ++i) {   // HD/8 V^T pieces of 8 rows x 128 B
            const int p = w * (HD / 32 / HPW) + i;
            const int d = p * 8 + (lane >> 3), sl = lane & 7;
            dma16(a.Vt + (size_t)(t.kvh * HD + d) * a.ldvt + t.sa + kc + ((sl ^ (d & 7)) * 8), Vs + p * 512);
        }
    };
    // Q fragments of this wave's 16 rows (B operand of S^T: query row li, 8 dims at 32*kk + 8*lg)
    auto load_q = [&](const Item &t, bf16x8(&q)[NKK]) {
        const int qrow = min(t.q0 + wq * 16 + li, t.L - 1);
        const bf16_t *qp = a.QK + (size_t)(t.s0 + qrow) * a.ldqk + t.h * HD + lg * 8;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) q[kk] = *reinterpret_cast<const bf16x8 *>(qp + kk * 32);
    };

    const float scale2 = a.scale * 1.4426950408889634f;
    const int x16 = (lane ^ 16) << 2, x32 = (lane ^ 32) << 2;   // ds_bpermute addresses of the lanes that share the row
    // LDS row of tile position li (tile j adds 32 (j/2) + 4 (j%2)), and the slot key of those rows
    const int krow0 = 8 * (li >> 2) + (li & 3);
    const int kkey = kswz(krow0);                   // 32 (j/2) and 4 (j%2) leave bits {0,1,3,4} of the row alone... (4 (j%2) sets bit 2 only)

    Item cur = decode(it);
    bf16x8 qf[NKK];
    load_q(cur, qf);
    // Pin the wait for these loads HERE.  Left to the compiler it sits in front of the first MFMA inside the loop as
    // `s_waitcnt vmcnt(3..0)` -- counted without the LDS-DMA pieces (issued from asm, invisible to it), so every iteration
    // it would drain the chunk that was just requested and the double buffering would be gone.
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[kk]));
    int c = 0;                                      // chunks so far, across items: chunk c sits in stage c & 1
    issue(cur, 0, 0);
    bool prewaited = false;                         // the first chunk of the current item was waited for at the end of the previous one
    for (;;) {
        const int itn = it + (int)gridDim.x;
        const bool more = itn < nitems;
        Item nxt = cur;
        if (more) nxt = decode(itn);
        bf16x8 qn[NKK];
        const int L = cur.L, off = cur.off, Le = cur.Le, kend = cur.kend;
        const int qidx = cur.q0 + wq * 16 + li;     // the query this lane owns
        f32x4 o[NDT];                               // O^T tiles: dims 16 n + 4 lg + r of query li
#pragma unroll
        for (int n = 0; n < NDT; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float mrow = -__builtin_huge_valf(), lrow = 0.f;   // running maximum (log2 units) and denominator of query li

        // one key chunk: wait for its pieces, request what comes next, multiply
        auto chunk = [&](int kc, auto LAST) {
            if (!(kc == 0 && prewaited)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA pieces of chunk c
            __syncthreads();                                   // everyone's pieces; chunk c-1 fully consumed
            if constexpr (!decltype(LAST)::value) issue(cur, (c + 1) & 1, kc + KC);
            else if (more) {                                   // last chunk of this item: the next item's first chunk and Q rows
                issue(nxt, (c + 1) & 1, 0);
                load_q(nxt, qn);                               // (requested from inline asm instead, so that no compiler wait sits
                                                               // between here and the stores: measured the same, 118 us)
            }
            const bf16_t *Ks = smem + (c & 1) * STG, *Vs = Ks + KC * HD;

            // S^T = K Q^T for 4 tiles of 16 keys
            f32x4 s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bf16_t *kb = Ks + (32 * (j >> 1) + 4 * (j & 1) + krow0) * HD;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kb + (((kk * 4 + lg) ^ kkey) * 8));
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[j], 0, 0, 0);
                }
            }
            // mask, online softmax: the lane holds keys kc + 32 (j/2) + 8 lg + 4 (j%2) + r of query li
            const bool need_mask = a.causal || kc + KC > Le || (kc == 0 && off != 0);   // wave-uniform: the first / last (partial) chunk, or causal
            float pmax = -__builtin_huge_valf();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[j][r];            // raw scores; the scale to log2 units rides in the fma of the exponent below
                    if (need_mask) {
                        const int kidx = kc + 32 * (j >> 1) + 8 * lg + 4 * (j & 1) + r - off;   // key index inside the sequence
                        if (kidx < 0 || kidx >= L || (a.causal && kidx > qidx)) v = -__builtin_huge_valf();
                    }
                    s[j][r] = v;
                    pmax = fmaxf(pmax, v);
                }
            pmax = fmaxf(pmax, __int_as_float(__builtin_amdgcn_ds_bpermute(x16, __float_as_int(pmax))));
            pmax = fmaxf(pmax, __int_as_float(__builtin_amdgcn_ds_bpermute(x32, __float_as_int(pmax))));
            const float mnew = fmaxf(mrow, pmax * scale2);          // log2 units (scale2 > 0: the maximum commutes with the scale)
            const float alpha = (mrow == -__builtin_huge_valf()) ? 0.f : __builtin_amdgcn_exp2f(mrow - mnew);
            mrow = mnew;
            const float msub = (mrow == -__builtin_huge_valf()) ? 0.f : mrow;   // a row that has seen no key yet: exp2(-inf - 0) = 0
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // one fma per score (scale and subtract) instead of a multiply, a subtract and a select: the softmax of a
                    // 64-key chunk is ~800 VALU cycles per wave against 512 of MFMA (SQ counters: profiles/r03_attn_pmc.txt)
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], scale2, -msub));
                    psum += p;
                    s[j][r] = p;
                }
            psum += __int_as_float(__builtin_amdgcn_ds_bpermute(x16, __float_as_int(psum)));
            psum += __int_as_float(__builtin_amdgcn_ds_bpermute(x32, __float_as_int(psum)));
            lrow = lrow * alpha + psum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {   // the running maxima settle after the first chunks
#pragma unroll
                for (int n = 0; n < NDT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[n][r] *= alpha;
            }
            // O^T += V^T P^T  (A: V^T[d = 16n+li][keys 32 J + 8 lg ..], B: P[query li][the same keys] = tiles 2J, 2J+1 of this lane)
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                union { bf16x8 v; unsigned u[4]; } pf;
                pf.u[0] = pack2(s[2 * J][0], s[2 * J][1]);
                pf.u[1] = pack2(s[2 * J][2], s[2 * J][3]);
                pf.u[2] = pack2(s[2 * J + 1][0], s[2 * J + 1][1]);
                pf.u[3] = pack2(s[2 * J + 1][2], s[2 * J + 1][3]);
#pragma unroll
                for (int n = 0; n < NDT; ++n) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8 *>(
                        Vs + (n * 16 + li) * KC + (((J * 4 + lg) ^ (li & 7)) * 8));
                    o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf.v, o[n], 0, 0, 0);
                }
            }
            ++c;
        };
        int kc = 0;
        for (; kc + KC < kend; kc += KC) chunk(kc, std::false_type{});
        chunk(kc, std::true_type{});
        // normalise and write row qidx: 4 consecutive dims per lane and tile
        const bool stores_behind = __builtin_amdgcn_ballot_w64(qidx < L) != 0;   // wave-uniform: NDT stores follow, or none
        if (qidx < L) {
            const float inv = lrow > 0.f ? 1.0f / lrow : 0.f;
            const int tok = cur.s0 + qidx;
            // fragments: dims h HD + 16 n + 4 lg + r -> piece (step (h HD + 16 n) / 32, tile tok / 16), lane (2 (n & 1) + lg / 2, tok % 16), half lg & 1
            bf16_t *op = a.Ofrag ? a.Ofrag + (((size_t)(cur.h * (HD / 32)) * a.frag_mt + (tok >> 4)) * 64 + (lg >> 1) * 16 + (tok & 15)) * 8 + 4 * (lg & 1)
                                 : a.O + (size_t)tok * (a.n_heads * HD) + cur.h * HD + 4 * lg;
            const size_t nstride = a.Ofrag ? 0 : 16;
#pragma unroll
            for (int n = 0; n < NDT; ++n) {
                uint2 pk;
                pk.x = pack2(o[n][0] * inv, o[n][1] * inv);
                pk.y = pack2(o[n][2] * inv, o[n][3] * inv);
                const size_t foff = a.Ofrag ? ((size_t)(n >> 1) * a.frag_mt * 64 + (n & 1) * 32) * 8 : 0;
                *reinterpret_cast<uint2 *>(op + n * nstride + foff) = pk;
            }
        }
        if (!more) break;
        cur = nxt;
        it = itn;
        // The next item's first chunk and Q rows were requested during the last chunk, in front of the stores above; vmcnt
        // counts in issue order, so the NDT stores may stay in flight.
        if (stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prewaited = true;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            asm volatile("" : "+v"(qn[kk]));
            qf[kk] = qn[kk];
        }
    }
}

// ---------------------------------------------------------------------
// Final RMSNorm of every real token -> f32 [T_real][H] (parity hook).
// tok_map[t_real] = padded-packed row.
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    final_norm_kernel(const float *__restrict__ x, const float *__restrict__ w, int H, int Treal,
                      const int32_t *__restrict__ tok_map, float eps, float *__restrict__ out) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= Treal) return;
    const float *src = x + (size_t)tok_map[t] * H;
    float ss = 0.f;
    for (int c = lane; c < H; c += 64) ss += src[c] * src[c];
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)H + eps);
    for (int c = lane; c < H; c += 64) out[(size_t)t * H + c] = src[c] * inv * w[c];
}

// ---------------------------------------------------------------------
// Per sequence: final RMSNorm, mean over the tokens, Dense (+bias), optional
// L2 normalisation.  One 256-thread workgroup per sequence.
// dynamic LDS: inv[Lmax] | pooled[H] | outv[out_dim] | red[8]
// ---------------------------------------------------------------------
struct PoolArgs {
    const float *x;          // [T_pad][H] residual stream
    const float *norm_w;     // [H]
    const bf16_t *dense_w;   // [out_dim][H] or null (no Dense: out = pooled)
    const float *dense_b;    // [out_dim] or null
    const int32_t *seq_start, *seq_len;
    float *out;              // [nseq][out_dim]
    int H, out_dim, Lmax, normalize;
    float eps;
    int parts;               // gridDim.y: the Dense rows are split over `parts` workgroups per sequence
                             // (few sequences: one workgroup streaming the whole Dense matrix is
                             // latency-bound, 536 us for a single query); > 1 => l2norm_rows_kernel follows
};

__global__ void __launch_bounds__(256) pool_kernel(PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *inv = reinterpret_cast<float *>(smem_raw);
    float *pooled = inv + a.Lmax;
    float *outv = pooled + a.H;
    float *red = outv + a.out_dim;
    const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s0 = a.seq_start[seq], L = a.seq_len[seq];
    const int H = a.H;
    for (int t = w; t < L; t += 4) {
        const float *src = a.x + (size_t)(s0 + t) * H;
        float ss = 0.f;
        for (int c = lane * 4; c < H; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(src + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss = wave_sum(ss);
        if (lane == 0) inv[t] = rsqrtf(ss / (float)H + a.eps);
    }
    __syncthreads();
    // the same rounding points and summation order as the many-sequence path (rmsnorm_kernel -> meanpool_kernel): the
    // normalised hidden state of a token is a bf16 value, a column is summed with four accumulators by token mod 4, the
    // pooled vector is rounded to bf16 (it is the Dense GEMM's A operand there) -- an embedding then depends on how many
    // sequences share a pass only through the f32 summation order of the Dense dot products (~1e-7)
    // (a thread takes 4 adjacent columns: the four tokens of a step are four independent 16-byte loads -- one column per
    // thread was a chain of 4-byte strided loads, 43 us for one 31-token query)
    const float invL = 1.0f / (float)max(L, 1);
    const int Hv = (H & 3) == 0 ? H : 0;                   // rows are 16-byte aligned only when H % 4 == 0
    for (int c = tid * 4; c < Hv; c += 1024) {
        const float4 g = *reinterpret_cast<const float4 *>(a.norm_w + c);
        const float *col = a.x + (size_t)s0 * H + c;
        float a4[4][4] = {};
        int t = 0;
        for (; t + 4 <= L; t += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(col + (size_t)(t + u) * H);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float it = inv[t + u];
                a4[u][0] += bf2f(f2bf(v[u].x * it * g.x)); a4[u][1] += bf2f(f2bf(v[u].y * it * g.y));
                a4[u][2] += bf2f(f2bf(v[u].z * it * g.z)); a4[u][3] += bf2f(f2bf(v[u].w * it * g.w));
            }
        }
        for (; t < L; ++t) {
            const float4 v = *reinterpret_cast<const float4 *>(col + (size_t)t * H);
            const float it = inv[t];
            a4[0][0] += bf2f(f2bf(v.x * it * g.x)); a4[0][1] += bf2f(f2bf(v.y * it * g.y));
            a4[0][2] += bf2f(f2bf(v.z * it * g.z)); a4[0][3] += bf2f(f2bf(v.w * it * g.w));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) pooled[c + k] = bf2f(f2bf(((a4[0][k] + a4[1][k]) + (a4[2][k] + a4[3][k])) * invL));
    }
    for (int c = Hv + tid; c < H; c += 256) {               // (H % 4 != 0: one column per thread)
        const float g = a.norm_w[c];
        const float *col = a.x + (size_t)s0 * H + c;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        int t = 0;
        for (; t + 4 <= L; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += bf2f(f2bf(col[(size_t)(t + u) * H] * inv[t + u] * g));
        }
        for (; t < L; ++t) a4[0] += bf2f(f2bf(col[(size_t)t * H] * inv[t] * g));
        pooled[c] = bf2f(f2bf(((a4[0] + a4[1]) + (a4[2] + a4[3])) * invL));
    }
    __syncthreads();
    const int parts = max(1, a.parts), part = blockIdx.y;
    const int per = (a.out_dim + parts - 1) / parts;
    const int j0 = part * per, j1 = min(a.out_dim, j0 + per);
    if (a.dense_w) {
        // four Dense rows per wave step: four independent weight streams in flight
        for (int j = j0 + w * 4; j < j1; j += 16) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane * 8; c < H; c += 512) {
                float pv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) pv[q] = pooled[c + q];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jr = min(j + r, j1 - 1);
                    const uint4 v = *reinterpret_cast<const uint4 *>(a.dense_w + (size_t)jr * H + c);
                    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[r] += pv[2 * q] * __uint_as_float(u[q] << 16);
                        acc[r] += pv[2 * q + 1] * __uint_as_float(u[q] & 0xffff0000u);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = wave_sum(acc[r]);
                if (lane == 0 && j + r < j1) outv[j + r] = t + (a.dense_b ? a.dense_b[j + r] : 0.f);
            }
        }
    } else {
        for (int c = j0 + tid; c < j1; c += 256) outv[c] = pooled[c];
    }
    __syncthreads();
    float scale = 1.f;
    if (a.normalize && parts == 1) {
        float ss = 0.f;
        for (int c = tid; c < a.out_dim; c += 256) ss += outv[c] * outv[c];
        ss = wave_sum(ss);
        if (lane == 0) red[w] = ss;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        scale = 1.0f / fmaxf(sqrtf(tot), 1e-12f);  // torch.nn.functional.normalize eps
    }
    for (int c = j0 + tid; c < j1; c += 256) a.out[(size_t)seq * a.out_dim + c] = outv[c] * scale;
}

// Mean over the real tokens of every sequence of the (final-norm) bf16 rows xn [T_pad][H] -> bf16 [nseq][H]: the
// pooling step when there are many sequences (grid: nseq x H / 512; a thread owns two columns and walks the tokens
// with four independent accumulators per column, summed in a fixed order).
__global__ void __launch_bounds__(256) meanpool_kernel(const bf16_t *__restrict__ xn, const int32_t *__restrict__ seq_start,
                                                       const int32_t *__restrict__ seq_len, int H, bf16_t *__restrict__ out) {
    const int seq = blockIdx.x, c = (blockIdx.y * 256 + threadIdx.x) * 2;
    if (c >= H) return;
    const int s0 = seq_start[seq], L = seq_len[seq];
    const bf16_t *p = xn + (size_t)s0 * H + c;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    int t = 0;
    for (; t + 4 <= L; t += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned v = *reinterpret_cast<const unsigned *>(p + (size_t)(t + u) * H);
            a0[u] += __uint_as_float(v << 16);
            a1[u] += __uint_as_float(v & 0xffff0000u);
        }
    }
    for (; t < L; ++t) {
        const unsigned v = *reinterpret_cast<const unsigned *>(p + (size_t)t * H);
        a0[0] += __uint_as_float(v << 16);
        a1[0] += __uint_as_float(v & 0xffff0000u);
    }
    const float inv = 1.0f / (float)max(L, 1);
    *reinterpret_cast<unsigned *>(out + (size_t)seq * H + c) =
        pack2(((a0[0] + a0[1]) + (a0[2] + a0[3])) * inv, ((a1[0] + a1[1]) + (a1[2] + a1[3])) * inv);
}

// L2 normalisation of the rows of out[nseq][n] (second step of a split pool_kernel)
__global__ void __launch_bounds__(256) l2norm_rows_kernel(float *__restrict__ out, int n) {
    __shared__ float red[4];
    float *row = out + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float ss = 0.f;
    for (int c = tid; c < n; c += 256) ss += row[c] * row[c];
    ss = wave_sum(ss);
    if (lane == 0) red[w] = ss;
    __syncthreads();
    const float scale = 1.0f / fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int c = tid; c < n; c += 256) row[c] *= scale;
}

// dst[rows[i]][:] = src[i][:] (the un-sort of a length-sorted pass, one workgroup per row)
__global__ void __launch_bounds__(256) scatter_rows_kernel(const float *__restrict__ src, int n, const int32_t *__restrict__ rows,
                                                           float *__restrict__ dst) {
    const float *s = src + (size_t)blockIdx.x * n;
    float *d = dst + (size_t)rows[blockIdx.x] * n;
    for (int c = threadIdx.x; c < n; c += 256) d[c] = s[c];
}

// ---------------------------------------------------------------------
// weight import: dst_bf16[map(r)][c] = convert(src[r][c]),
// map(r) = (r / blk) * stride + off + r % blk
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    import_rows_kernel(const void *__restrict__ src, int dtype, int64_t rows, int64_t cols, int64_t blk,
                       int64_t stride, int64_t off, bf16_t *__restrict__ dst, float *__restrict__ dst_f32) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    float v;
    if (dtype == 0) v = static_cast<const float *>(src)[i];
    else if (dtype == 1) v = bf2f(static_cast<const bf16_t *>(src)[i]);
    else v = __half2float(static_cast<const __half *>(src)[i]);
    const int64_t dr = (r / blk) * stride + off + r % blk;
    if (dst) dst[dr * cols + c] = f2bf(v);
    else dst_f32[dr * cols + c] = v;
}

}  // namespace mienc
