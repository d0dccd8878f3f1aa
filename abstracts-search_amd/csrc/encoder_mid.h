// encoder_mid.h -- the K = hidden projections (QKV, O) of a forward pass of ~100 .. ~4000 tokens (a batch of queries:
// reference README.md:28 served in batches, BASELINE.json configs[4] batch 16 / 64; arithmetic restated in
// oracle/encoder_oracle.py): ONE launch per projection, whole K inside the workgroup, the elementwise work in the epilogue.
//
// What it replaces, and why.  At 576 tokens the 128 x 128 tiles of these GEMMs are 80 / 60 workgroups, so round 4 split K
// four ways into f32 planes and finished with a reduction pass (bias + RoPE, or residual + RMSNorm): 14 + 7 us per
// projection, of which the GEMM's 3.6 GFLOP are 1.5 us at the MFMA peak -- the rest is the planes' round trip (19 MB
// written and read back), two launches, and DMA pieces of 16 rows x 64 B that touch every operand line twice.  What bounds
// such a GEMM is the CU's memory pipe (~3.5 cycles per 128-byte line: ~36 B/clk = ~75 GB/s per CU), i.e. the
// bytes a workgroup ingests, (BM + BN) x K x 2 -- so: tiles small enough that one round of <= 256 workgroups covers the
// output (64 x 64 at 576 tokens: 393 KB each, ~5 us), operands staged in whole 128-byte lines (a DMA piece = 8 rows x 128 B,
// K slabs of 64 columns), a ring of 3-4 slabs per workgroup and two workgroups per CU so that one's epilogue and pipeline
// fill sit under the other's stream.
//
//   mid_gemm_kernel<MID_QKV>   A = the normalised stream (or bf16(x g) with 1/rms per row: row_scale) x the QKV weights whose
//                              Q / K rows are interleaved by rotary pair (interleave_qk_rows_kernel: a lane's four
//                              consecutive columns are two whole pairs) -> + bias -> RoPE -> Q|K rows; V tiles are
//                              accumulated untransposed (a lane = 4 consecutive tokens of one channel) -> V^T
//   mid_gemm_kernel<MID_O>     attention output x W_o -> x += ...; optionally bf16(x g2) and the tile's part of the rows'
//                              sums of squares (slot = tile column); the last tile of a row block to arrive turns the block's
//                              slots into 1/rms (GemmArgs::rms_out / arrive: no row_rms_kernel launch) and the gate/up slab
//                              GEMM applies it to its accumulators
//
// LDS: slab s of the ring = BM rows of A then BN rows of W, 128 B per row, the 16-byte slot index XORed with (row & 7) on
// the SOURCE side (the DMA itself is lane-linear): the 16 rows a ds_read_b128 lane group touches fall on 16 distinct
// granules.  Waits are counted (own pieces of the slab needed next; D - 1 slabs stay in flight across the barrier).
#pragma once
#include "encoder_kernels.h"

namespace mienc {

enum { MID_QKV = 0, MID_O = 1 };

template <int EPI, int WMT, int WNT, int NS>
__global__ void __launch_bounds__(256, (NS * (WMT + WNT) * 4096 <= 80 * 1024 ? 2 : 1)) mid_gemm_kernel(GemmArgs g) {
    constexpr int BM = 32 * WMT, BN = 32 * WNT;
    constexpr int PA = BM / 8, PB = BN / 8;                  // 1-KiB pieces (8 rows x 128 B) of a slab
    static_assert((PA + PB) % 4 == 0, "pieces divide over the four waves");
    constexpr int PPW = (PA + PB) / 4;
    constexpr int D = NS - 1;                                // slabs requested ahead
    static_assert(NS >= 2 && (D - 1) * PPW <= 63, "vmcnt is 6 bits");
    constexpr unsigned STAGE_B = (unsigned)(BM + BN) * 128u;
    extern __shared__ __attribute__((aligned(16))) unsigned char mid_lds[];

    int tm, tn;
    if (!tile_coords(g.tiles_m, g.tiles_n, tm, tn)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1;
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = tm * BM, n0 = tn * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)mid_lds;

    // DMA sources of this wave's pieces w PPW .. (A pieces first): lane -> row lane >> 3 of the piece, slot (lane & 7) ^ row
    const int prow = lane >> 3, scol = ((lane & 7) ^ prow) * 8;
    const bf16_t *src[PPW];
    unsigned dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = w * PPW + i;
        if (q < PA) {
            src[i] = g.A + (size_t)min(m0 + q * 8 + prow, g.M - 1) * g.lda + scol;
            dst[i] = (unsigned)q * 1024u;
        } else {
            src[i] = g.W + (size_t)min(n0 + (q - PA) * 8 + prow, g.N - 1) * g.ldw + scol;
            dst[i] = (unsigned)BM * 128u + (unsigned)(q - PA) * 1024u;
        }
    }
    const int nslab = g.K / 64;
    const int krot = g.krot < 0 ? (int)(((long)tm * nslab) / g.tiles_m) : (tm * g.krot) % nslab;   // GemmArgs::krot: the row tiles of a column strip start at different K
    auto issue = [&](int s, unsigned slot) {
        const int r = s + krot;
        const size_t ko = (size_t)(r >= nslab ? r - nslab : r) * 64;
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma16_off(src[i] + ko, lds0 + slot * STAGE_B + dst[i]);
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nslab) issue(s, (unsigned)s);

    // what the epilogue needs per row, requested behind the first slabs (the latency is the pipeline fill's)
    const bool is_v = EPI == MID_QKV && n0 >= g.qk_cols;     // workgroup-uniform: qk_cols % BN == 0 (host)
    [[maybe_unused]] int pos[WMT];
    [[maybe_unused]] float rsc[WMT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int row = min(m0 + (wm * WMT + i) * 16 + li, g.M - 1);
        if constexpr (EPI == MID_QKV) {
            pos[i] = g.rope_pos[row];
            rsc[i] = g.row_scale ? g.row_scale[row] : 1.f;
        }
    }

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment (16 rows x 32 k) of K half kk: lane (li, lg) reads row li, global slot 4 kk + lg -> LDS slot ^ (row & 7)
    const unsigned fa = (unsigned)(wm * WMT * 16 + li) * 128u, fb = (unsigned)BM * 128u + (unsigned)(wn * WNT * 16 + li) * 128u;
    const unsigned fs0 = (unsigned)((lg ^ (li & 7)) * 16), fs1 = (unsigned)(((4 + lg) ^ (li & 7)) * 16);
    auto compute = [&](unsigned slot, auto SWAP_) {
        constexpr bool swap = decltype(SWAP_)::value;
        const unsigned char *stage = mid_lds + slot * STAGE_B;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned fs = kk ? fs1 : fs0;
            bf16x8 a[WMT], b[WNT];
#pragma unroll
            for (int j = 0; j < WNT; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(stage + fb + j * 2048 + fs);
#pragma unroll
            for (int i = 0; i < WMT; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(stage + fa + i * 2048 + fs);
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    if constexpr (swap) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    auto k_loop = [&](auto SWAP_) {
        unsigned slot = 0;                                   // ring slot of slab s
        int s = 0;
        for (; s + D < nslab; ++s) {                         // steady state, branch-free
            wait_vm_lgkm0<(D - 1) * PPW>();                  // own pieces of slab s landed; no read of the slot refilled below is pending
            asm volatile("s_barrier" ::: "memory");
            issue(s + D, slot == 0 ? (unsigned)(NS - 1) : slot - 1);
            compute(slot, SWAP_);
            slot = slot + 1 == (unsigned)NS ? 0u : slot + 1;
        }
        for (; s < nslab; ++s) {
            wait_tiles<PPW, D - 1>(min(D - 1, nslab - 1 - s), true);
            asm volatile("s_barrier" ::: "memory");
            compute(slot, SWAP_);
            slot = slot + 1 == (unsigned)NS ? 0u : slot + 1;
        }
    };
    if (is_v) k_loop(std::false_type{});
    else k_loop(std::true_type{});

    if constexpr (EPI == MID_QKV) {
        if (is_v) {
            // V^T: the lane holds tokens trow + 4 lg + r of channel li of tile j
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int col = n0 + (wn * WNT + j) * 16 + li;
                const float bv = g.bias && col < g.N ? g.bias[col] : 0.f;
#pragma unroll
                for (int i = 0; i < WMT; ++i) {
                    const int t0 = m0 + (wm * WMT + i) * 16 + 4 * lg;
                    float s4[4] = {1.f, 1.f, 1.f, 1.f};
                    if (g.row_scale) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s4[r] = __shfl(rsc[i], 4 * lg + r);   // lane li' = row li' of the tile
                    }
                    if (t0 < g.M && col < g.N) {
                        const f32x4 v = acc[i][j];
                        uint2 o;
                        o.x = pack2(v[0] * s4[0] + bv, v[1] * s4[1] + bv);
                        o.y = pack2(v[2] * s4[2] + bv, v[3] * s4[3] + bv);
                        *reinterpret_cast<uint2 *>(g.Vt + (size_t)(col - g.qk_cols) * g.ldvt + t0) = o;
                    }
                }
            }
            return;
        }
        // Q | K: the lane holds columns col0 .. col0 + 3 of token row -- two rotary pairs (interleaved rows)
        const int half = g.rope_hd >> 1;
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            const int row = m0 + (wm * WMT + i) * 16 + li;
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int col0 = n0 + (wn * WNT + j) * 16 + 4 * lg;
                if (row >= g.M || col0 >= g.N) continue;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g.bias) b = *reinterpret_cast<const float4 *>(g.bias + col0);
                const float4 c = *reinterpret_cast<const float4 *>(g.rope_cs + (size_t)pos[i] * half + ((col0 % g.rope_hd) >> 1));
                const f32x4 v = acc[i][j];
                const float x0 = v[0] * rsc[i] + b.x, x1 = v[1] * rsc[i] + b.y, x2 = v[2] * rsc[i] + b.z, x3 = v[3] * rsc[i] + b.w;
                uint2 o;
                o.x = pack2(x0 * c.x - x1 * c.y, x1 * c.x + x0 * c.y);
                o.y = pack2(x2 * c.z - x3 * c.w, x3 * c.z + x2 * c.w);
                *reinterpret_cast<uint2 *>(g.C + (size_t)row * g.ldc + col0) = o;
            }
        }
    } else {
        // x += acc (+ bias); fused RMSNorm, producer side: bf16(x g) and the tile's part of the rows' sums of squares
        const bool fuse = g.ssq_out != nullptr;
        float *sred = reinterpret_cast<float *>(mid_lds);    // [2][BM] (the ring is free behind the barrier below)
        if (fuse) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            const int row = m0 + (wm * WMT + i) * 16 + li;
            float ss = 0.f;
            float4 xv[WNT];
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int col0 = n0 + (wn * WNT + j) * 16 + 4 * lg;
                if (row < g.M && col0 < g.N) xv[j] = *reinterpret_cast<const float4 *>(g.X + (size_t)row * g.ldc + col0);
            }
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int col0 = n0 + (wn * WNT + j) * 16 + 4 * lg;
                if (row >= g.M || col0 >= g.N) continue;
                float4 x = xv[j];
                const f32x4 v = acc[i][j];
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g.bias) b = *reinterpret_cast<const float4 *>(g.bias + col0);
                x.x += v[0] + b.x; x.y += v[1] + b.y; x.z += v[2] + b.z; x.w += v[3] + b.w;
                *reinterpret_cast<float4 *>(g.X + (size_t)row * g.ldc + col0) = x;
                if (fuse) {
                    const float4 nw = *reinterpret_cast<const float4 *>(g.norm_w + col0);
                    uint2 o;
                    o.x = pack2(x.x * nw.x, x.y * nw.y);
                    o.y = pack2(x.z * nw.z, x.w * nw.w);
                    *reinterpret_cast<uint2 *>(g.norm_y + (size_t)row * g.ldc + col0) = o;
                    ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
                }
            }
            if (fuse) {                                      // the four lane groups of a row, a fixed order
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                if (lg == 0) sred[wn * BM + (wm * WMT + i) * 16 + li] = ss;
            }
        }
        if (fuse) {
            __syncthreads();
            const bool mine = tid < BM && m0 + tid < g.M;
            if (!g.rms_out) {
                if (mine) g.ssq_out[(size_t)tn * g.M + m0 + tid] = sred[tid] + sred[BM + tid];
                return;
            }
            // The RMSNorm finished here, by the LAST tile of the row block to arrive (no row_rms_kernel launch between this
            // projection and gate/up): the slots leave as write-through stores (device scope: past the XCD's L2) that are
            // acknowledged before the tile counts itself in on the row block's counter; the last arriver reads the block's slots
            // past its own L2 and adds them in slot order -- the order row_rms_kernel adds them in -- and puts the counter back
            // to zero for the next launch.  No release / acquire fence: a device-scope release is a write-back of the XCD's
            // whole L2 (buffer_wbl2: it made this launch 13.4 -> 22.6 us), and the only data that has to cross are these slots.
            if (mine) __hip_atomic_store(g.ssq_out + (size_t)tn * g.M + m0 + tid, sred[tid] + sred[BM + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __shared__ unsigned s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) s_last = __hip_atomic_fetch_add(g.arrive + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)g.tiles_n - 1u;
            __syncthreads();
            if (!s_last) return;
            if (mine) {
                float v[SSQ_LD];                             // every load in flight before the first add (one round trip, not 24)
#pragma unroll
                for (int q = 0; q < SSQ_LD; ++q)
                    v[q] = __hip_atomic_load(g.ssq_out + (size_t)min(q, g.tiles_n - 1) * g.M + m0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float ssum = 0.f;
#pragma unroll
                for (int q = 0; q < SSQ_LD; ++q)
                    if (q < g.tiles_n) ssum += v[q];
                g.rms_out[m0 + tid] = rsqrtf(ssum / (float)g.N + g.norm_eps);
            }
            if (tid == 0) __hip_atomic_store(g.arrive + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace mienc
