// encoder.hip -- C ABI (include/mi_encoder.h) over the gfx950 kernels of
// encoder_kernels.h.  Host orchestration only; no CPU fallback.
#include "../../include/mi_encoder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "encoder_kernels.h"
#include "encoder_few.h"
#include "encoder_mid.h"

using namespace mi;
using namespace mienc;

namespace {

hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Environment knobs.  Read ONCE -- at the first call into the library, or again by mi_encoder_reload_env() (tests and tools:
// not beside a running encode) -- into this struct: no encode call path calls getenv (not safe against a concurrent setenv).
// Every knob is exercised by a test or a committed tool; what each overrides is a measured dispatch rule.
struct Knobs {
    std::string gemm_tile;   // MI_GEMM_TILE=big|slab8|slab4|mid|mid64|small|tiny|128: one tile configuration for every GEMM
    bool gemm_ring;          // MI_GEMM_RING=1: round 1's 256 x 256 ring kernel instead of the slab kernel
    bool tail_split_force;   // MI_TAIL_SPLIT_FORCE=1: split a short last round along K whatever the cost model says
    bool no_bulk_fuse, no_rope_fuse, no_norm_fuse;   // MI_NO_BULK_FUSE / MI_NO_ROPE_FUSE / MI_NO_NORM_FUSE: standalone rmsnorm / rope kernels
    bool no_mid_gemm;        // MI_NO_MID_GEMM=1: round 4's K-split planes for the QKV / O projections of a few hundred tokens
    std::string mid_tile;    // MI_MID_TILE=128x128|128x64|96x64|64x64: one tile shape for encoder_mid.h
    bool no_short_attn;      // MI_NO_SHORT_ATTN=1: the persistent flash-attention kernel for batches of short sequences too
    int krot;                // MI_KROT=n: K tiles between the starts of consecutive row tiles (default 3; -1: K / row tiles)
    bool no_krot;            // MI_NO_KROT=1: one-round GEMMs walk K from 0 in every row tile (GemmArgs::krot off)
    int down_bn;             // MI_DOWN_BN=128|192|256: columns of the down projection's K-split tiles (0: by the cost model)
    bool no_m192;            // MI_NO_M192=1: 256-row slab tiles where 192-row ones would pay
    int splitk;              // MI_SPLITK=S: K slices of the all-tiles split (-1: by shape)
    int pool_gemm;           // MI_POOL_GEMM=0: the per-sequence pooling kernel for every batch size
    bool no_few;             // MI_NO_FEW=1: a handful of tokens through the general path (fragment-major tiles)
    bool few_d_fuse, few_sync, few_ts;   // MI_FEW_D_FUSE (in-launch reduction of the down projection), MI_FEW_SYNC, MI_FEW_TS
    bool no_few_gu8;         // MI_NO_FEW_GU8=1: the query-time gate/up projection on 16-feature unit pairs (few_gemm_kernel<FEW_GU>) instead of 8-feature units
    bool no_few_qkv8;        // MI_NO_FEW_QKV8=1: the query-time QKV / O projections stage their fragments through LDS (few_gemm_kernel<FEW_QKV>, few_o_kernel)
    bool no_few_ao;          // MI_NO_FEW_AO=1: one sequence of <= 32 tokens keeps few_attn_kernel + few_o_kernel (no attention inside the O projection)
    int gemm_persist;        // MI_GEMM_PERSIST=mask: which many-token slab GEMMs run as one persistent workgroup per CU with a stream-K tail --
                             // 1 plain store, 2 residual, 4 QKV, 8 SwiGLU.  Default 0 (one tile per workgroup): built, parity-tested
                             // and measured in round 6, the persistent launches LOSE on every shape (profiles/r06_persist_gemm_ab.txt,
                             // DESIGN 6.1) -- the GEMMs are bound by the 1 400 W cap, not by idle CUs
    int enc_ts;              // MI_ENC_TS=1: in-kernel stamps of the first four slab GEMMs of > 4096 tokens; =n (n > 1): of > n tokens
    bool gemm_ts;            // MI_GEMM_TS=1: the same for mi_enc_gemm_bf16
    void load() {
        auto num = [](const char *n, int dflt) { const char *e = std::getenv(n); return e && *e ? std::atoi(e) : dflt; };
        auto set = [](const char *n) { return std::getenv(n) != nullptr; };
        auto str = [](const char *n) { const char *e = std::getenv(n); return std::string(e ? e : ""); };
        gemm_tile = str("MI_GEMM_TILE");
        gemm_ring = set("MI_GEMM_RING");
        tail_split_force = set("MI_TAIL_SPLIT_FORCE");
        no_bulk_fuse = set("MI_NO_BULK_FUSE"); no_rope_fuse = set("MI_NO_ROPE_FUSE"); no_norm_fuse = set("MI_NO_NORM_FUSE");
        no_mid_gemm = set("MI_NO_MID_GEMM");
        mid_tile = str("MI_MID_TILE");
        no_m192 = set("MI_NO_M192");
        down_bn = num("MI_DOWN_BN", 0);
        no_krot = set("MI_NO_KROT");
        krot = num("MI_KROT", 3);      // 563 tokens, forward pass: off 3.192 ms, 1 / 2 / 3 / 4 K tiles 3.171 / 3.158 / 3.142 / 3.147, K / row tiles 3.157
        no_short_attn = set("MI_NO_SHORT_ATTN");
        splitk = num("MI_SPLITK", -1);
        pool_gemm = num("MI_POOL_GEMM", -1);
        no_few = set("MI_NO_FEW");
        few_d_fuse = set("MI_FEW_D_FUSE"); few_sync = set("MI_FEW_SYNC"); few_ts = set("MI_FEW_TS");
        no_few_ao = set("MI_NO_FEW_AO"); no_few_gu8 = set("MI_NO_FEW_GU8"); no_few_qkv8 = set("MI_NO_FEW_QKV8");
        enc_ts = num("MI_ENC_TS", 0); gemm_ts = set("MI_GEMM_TS");
        gemm_persist = num("MI_GEMM_PERSIST", 0);
    }
};
Knobs &knobs_mut() {
    static Knobs k = [] { Knobs x{}; x.load(); return x; }();
    return k;
}
// bumped by mi_encoder_reload_env(): state a handle derived from the knobs (the query-time path's weight pieces: which
// gate/up unit size they are laid out for) is rebuilt when the handle's generation is behind
std::atomic<int> g_knob_gen{0};
inline const Knobs &knobs() { return knobs_mut(); }

// launches that took the K-split tail (f32 atomics into the residual stream): tests assert the path ran
std::atomic<int64_t> g_tail_split_launches{0};
std::atomic<int64_t> g_splitk_launches{0};
std::atomic<int64_t> g_reduce_norm_launches{0};
std::atomic<int64_t> g_n192_launches{0};
std::atomic<int64_t> g_m192_launches{0};      // slab GEMMs on 192-row tiles
// one word per device: polls of a stream-K head that gave up waiting for its tile's partials (GemmArgs::sk_err)
unsigned *sk_err_word(int device) {
    static std::mutex mu;
    static unsigned *words[16] = {};                         // (never freed: process-lifetime, a few bytes per device)
    std::lock_guard<std::mutex> lk(mu);
    unsigned *&b = words[device & 15];
    if (!b) {
        MI_HIP(hipMalloc(reinterpret_cast<void **>(&b), 256));
        MI_HIP(hipMemset(b, 0, 256));
    }
    return b;
}
std::atomic<int64_t> g_persist_launches{0};      // slab GEMMs launched as one persistent workgroup per CU (stream-K tail)
std::atomic<int64_t> g_fused_norm_launches{0};   // residual slab GEMMs whose epilogue carried the next RMSNorm (GEMM_RAWNORM)
std::atomic<int64_t> g_fused_rope_launches{0};   // QKV slab GEMMs whose epilogue rotated Q and K
std::atomic<int64_t> g_short_attn_launches{0};   // attention launches of the general path that took few_attn_kernel (every sequence <= 48 tokens)
std::atomic<int64_t> g_mid_launches{0};   // QKV / O projections on the one-launch whole-K tiles of encoder_mid.h
std::atomic<int64_t> g_few_passes{0};     // forward passes that took the query-time path (encoder_few.h)
std::atomic<int64_t> g_few_qkv8_passes{0}; // ... with the QKV projection's fragments in registers (few_qkv8_kernel)
std::atomic<int64_t> g_few_gu8_passes{0}; // ... with the gate/up projection on 8-feature units (few_gu8_kernel)
std::atomic<int64_t> g_few_ao_passes{0};  // ... of them, with the attention inside the O projection (few_ao_kernel)

struct LayerW {
    DevBuf wqkv, bqkv, wo, wgu, wd, ln1, ln2;
    DevBuf wqkv_t, wo_t, wgu_t, wd_t;   // fragment-major copies for the general path's few-token tiles (MI_NO_FEW=1; built lazily)
    DevBuf few_qkv, few_o, few_gu, few_d;   // 1-KiB pieces of the query-time path (encoder_few.h; built lazily)
    DevBuf few_op;                          // the O projection's half pieces in few_ao_kernel's K order
    DevBuf wqkv_r, bqkv_r;                  // Q / K rows interleaved by rotary pair: the many-token path's QKV epilogue rotates in place (built lazily)
};

// what a GEMM launch did besides the GEMM (launch_gemm's return value)
enum { GEMM_RMS_DONE = 8,    // ... and already turned them into 1/rms per row (GemmArgs::rms_out): no row_rms_kernel launch
       GEMM_NORMED = 1,      // a split-K reduction pass also wrote the RMSNorm of the updated stream (GemmArgs::norm_w / norm_y)
       GEMM_ROPED = 2,       // the QKV epilogue (or its reduction pass) rotated Q and K
       GEMM_RAWNORM = 4 };   // the residual epilogue wrote bf16(X * norm_w) and the rows' partial sums of squares
                             // (GemmArgs::ssq_out; the number of slots per row in bits 8..15)

template <int WMT, int WNT, int WAVES_M, int WAVES_N, int ST>
void launch_ring(int epi, GemmArgs g, hipStream_t st) {
    constexpr int BM = 16 * WMT * WAVES_M, BN = 16 * WNT * WAVES_N;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int per = (ntiles + 7) / 8;
    if (epi != EPI_RESID || g.bias) g.ksplit = 1;
    g.ksplit = std::max(1, g.ksplit);
    unsigned nblocks = 8u * per * g.ksplit;
    g.tail_first = 0;
    g.tail_split = 1;
    if (epi == EPI_RESID && !g.bias && g.ksplit == 1 && BM * BN >= 256 * 256) {
        // wave quantisation: 780 tiles on 256 CUs are 3 full rounds + 12 tiles that would
        // hold the kernel for a 4th tile time; split those along K (f32 atomics into the
        // residual stream, as the small-batch split-K path does)
        const int ncu = 256, nb = 8 * per;
        const int main_b = nb / ncu * ncu, rem = nb - main_b;
        const int nk = g.K / 32;
        if (main_b > 0 && rem > 0 && rem <= ncu * 5 / 8) {
            const int sp = std::min({ncu / rem, nk / 8, 16});
            if (sp >= 2) {
                g.tail_first = main_b;
                g.tail_split = sp;
                nblocks = (unsigned)(main_b + rem * sp);
                ++g_tail_split_launches;
            }
        }
    }
    dim3 grid(nblocks), block(64 * WAVES_M * WAVES_N);
    // (a persistent variant -- one workgroup per CU walking its tiles, the next tile's first slabs requested before the
    //  epilogue -- measured 5 % slower here and no faster on the slab kernel: DESIGN_HISTORY.md)
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_STORE, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_RESID, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        case EPI_QKV: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_QKV, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_SWIGLU, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        default: throw Error("bad epilogue");
    }
    MI_HIP(hipGetLastError());
}

// 256x256 tiles, slab ring + hand-ordered K loop (gemm_bf16_slab_kernel; WN_ = 4: 8 waves, 2: 4 waves): whole-K workgroups, the same wave-quantisation tail split
template <int WN_, int WMT_ = 8, int WNT_ = 16 / WN_>
int launch_slab(int epi, GemmArgs g, hipStream_t st) {
    constexpr int BM = 32 * WMT_, BN = WN_ * WNT_ * 16;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const int per = (g.tiles_m * g.tiles_n + 7) / 8;
    g.ksplit = 1;
    unsigned nblocks = 8u * per;
    g.tail_first = 0;
    g.tail_split = 1;
    float *part = g.part;
    g.part = nullptr;                                     // set again below if every tile is split
    // measured at 29 312 tokens (tools/gemm_bench.py, profiles/r03_gemm_tile_order.txt): chip patches +3 / +5 / +5 % on the
    // QKV / O / down shapes (N <= 2048), -1.5 % on gate-up (N = 17 920: 70 tile columns) -- so by the width of the GEMM
    g.order = g.tiles_n <= 16 ? 1 : 0;
    // (a persistent launch deals the short last round out as K ranges itself: no K-split tail through atomics)
    const int persist_bit = epi == EPI_STORE ? 1 : epi == EPI_RESID ? 2 : epi == EPI_QKV ? 4 : epi == EPI_SWIGLU ? 8 : 0;
    const bool persist = (knobs().gemm_persist & persist_bit) && WMT_ == 8 && WNT_ == 16 / WN_ && g.tiles_m * g.tiles_n >= 2 * 256;
    if (epi == EPI_RESID && !g.bias && !persist) {
        const int ncu = 256, nb = 8 * per;
        const int main_b = nb / ncu * ncu, rem = nb - main_b;
        const int nk = g.K / 32;
        if (main_b > 0 && rem > 0 && rem <= ncu * 5 / 8) {
            // split the tail only where it pays: a slice saves tile_time (1 - 1/sp) of the last round, and its partial
            // tile goes through 65 536 f32 atomics -- the chip retires ~115 G of those per second (measured with a
            // stream-K tail experiment: 22 M atomics per GEMM cost 0.2 ms), i.e. ~0.57 us per slice
            const int sp_max = std::min({ncu / rem, nk / 8, 16});
            const double tile_us = 0.77 * nk + 12.0;
            int best = 1;
            double best_gain = 10.0;                                   // a split must buy more than 10 us
            for (int sp = 2; sp <= sp_max; ++sp) {
                const double gain = tile_us * (1.0 - 1.0 / sp) - 0.57 * rem * sp - 6.0;
                if (gain > best_gain) { best_gain = gain; best = sp; }
            }
            if (knobs().tail_split_force && sp_max >= 2) best = sp_max;   // tests: the split whatever the model says
            // the split tail meets in f32 atomics: no workgroup sees the finished rows, so the fused RMSNorm (ssq_out) and the
            // split exclude each other -- the fusion saves ~30 us a launch, the split has to buy more than that
            if (best >= 2 && g.ssq_out && best_gain < 40.0 && !knobs().tail_split_force) best = 1;
            if (best >= 2) {
                g.ssq_out = nullptr;
                g.tail_first = main_b;
                g.tail_split = best;
                nblocks = (unsigned)(main_b + rem * best);
                ++g_tail_split_launches;
            }
        }
    }
    // Too few tiles to fill the chip (a few hundred tokens through a residual GEMM: 18 tiles at 576 x 1536): split EVERY
    // tile along K so that ~256 workgroups run, the slices' partial tiles through a workspace (whole-line f32 stores) and one
    // reduction pass -- no atomics (576 x 1536 x 14 of them would take ~110 us at the chip's 115 G atomic adds per second).
    int split_all = 1;
    if (epi == EPI_RESID && part && g.tail_split == 1 && nblocks < 200) {
        const int ntiles = g.tiles_m * g.tiles_n, nt64 = g.K / 64;
        // (>= 4 K tiles (8 steps) per slice; at most 10 slices: every slice writes an M x N f32 plane that the reduction pass
        // reads back -- 14 slices of 576 x 1536 are 49 MB each way; 16 queries 3.85 -> 3.71 ms with 10: profiles/r04 notes)
        int S = std::min({10, 256 / std::max(1, ntiles), nt64 / 4});
        if (knobs().splitk >= 0) S = std::min(knobs().splitk, nt64);
        if (S >= 2 && (size_t)S * g.M * g.N * 4 <= g.part_bytes) {
            split_all = S;
            ++g_splitk_launches;
            g.part = part;
            g.tail_first = 0;
            g.tail_split = S;
            nblocks = (unsigned)(ntiles * S);
            g.order = 2;                                  // slices of K to XCDs (gemm_bf16_slab_kernel::decode): the down projection's
            nblocks = 8u * (unsigned)((ntiles * S + 7) / 8);   // fetch 124 -> ~40 MB at 576 tokens
        }
    }
    // one round of workgroups: the row tiles of a column strip run side by side -- each walks K from its own offset (GemmArgs::krot)
    g.krot = (!knobs().no_krot && nblocks <= 256 && g.tiles_m > 1 && g.tail_split == (split_all > 1 ? split_all : 1)) ? knobs().krot : 0;
    dim3 grid(nblocks), block(128 * WN_);
    constexpr int CW = 16 * WNT_;                         // columns per wave = per slot of sums of squares
    const int nslots = (g.N + CW - 1) / CW;
    if (split_all > 1 || nslots > SSQ_LD) g.ssq_out = nullptr;
    const int rawnorm = epi == EPI_RESID && g.ssq_out ? (GEMM_RAWNORM | nslots << 8) : 0;
    bool normed = false;                                  // the reduction pass also wrote the RMSNorm the caller asked for
    auto finish_split = [&] {
        if (split_all == 1) return;
        if (g.norm_w && g.norm_y && g.ldc == g.N && g.N <= 2048) {
            hipLaunchKernelGGL(splitk_reduce_norm_kernel, dim3((unsigned)g.M), dim3(256), 0, st, g.X, g.part, split_all, g.M, g.N, g.bias,
                               g.norm_w, g.norm_eps, g.norm_y);
            normed = true;
            ++g_reduce_norm_launches;
        } else {
            const int64_t n4 = (int64_t)g.M * (g.N / 4);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g.X, (int64_t)g.ldc, g.part,
                               split_all, g.M, g.N, g.bias);
        }
        MI_HIP(hipGetLastError());
    };
    if constexpr (WNT_ != 16 / WN_) {                     // narrow tiles: the K-split down projection of a few hundred tokens
        MI_REQUIRE(epi == EPI_RESID && (split_all > 1 || WMT_ == 8), "narrow slab tiles: residual epilogue, K split or 256-row tiles");
        if (WMT_ == 6) ++g_m192_launches;
        hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_RESID, WN_, false, WNT_, WMT_>), grid, block, 0, st, g);
    } else if constexpr (WMT_ == 6) {                     // 192-row tiles: the two epilogues the few-hundred-token passes send here
        ++g_m192_launches;
        switch (epi) {
            case EPI_RESID: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_RESID, WN_, false, 16 / WN_, 6>), grid, block, 0, st, g); break;
            case EPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_SWIGLU, WN_, false, 16 / WN_, 6>), grid, block, 0, st, g); break;
            default: throw Error("192-row slab tiles: epilogue not instantiated");
        }
    } else if (persist && g.tail_split == 1 && split_all == 1) {
        // >= two rounds of tiles: ONE workgroup per CU walks them (the next tile's first slabs in flight under the epilogue) and
        // the tiles that do not fill a last round are dealt out as K ranges (stream-K; GemmArgs::sk_part)
        g.krot = 0;
        ++g_persist_launches;
        const dim3 pgrid(256);
        switch (epi) {
            case EPI_STORE: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_STORE, WN_, true>), pgrid, block, 0, st, g); break;
            case EPI_RESID: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_RESID, WN_, true>), pgrid, block, 0, st, g); break;
            case EPI_QKV: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_QKV, WN_, true>), pgrid, block, 0, st, g); break;
            case EPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_SWIGLU, WN_, true>), pgrid, block, 0, st, g); break;
            default: throw Error("bad epilogue");
        }
    } else {
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_STORE, WN_>), grid, block, 0, st, g); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_RESID, WN_>), grid, block, 0, st, g); break;
        case EPI_QKV: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_QKV, WN_>), grid, block, 0, st, g); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_SWIGLU, WN_>), grid, block, 0, st, g); break;
        default: throw Error("bad epilogue");
    }
    }
    MI_HIP(hipGetLastError());
    finish_split();
    if (rawnorm) ++g_fused_norm_launches;
    if (epi == EPI_QKV && g.rope_cs) ++g_fused_rope_launches;
    return (normed ? GEMM_NORMED : 0) | rawnorm | (epi == EPI_QKV && g.rope_cs ? GEMM_ROPED : 0);
}

// 256 x 192 tiles of the slab kernel (residual epilogue): N = 1536 is 8 tile columns instead of 6 -- at 5 400 .. 8 192
// tokens that is one full round of <= 256 workgroups where 256-column tiles leave a quarter of the CUs idle
int launch_slab_n192(GemmArgs g, hipStream_t st) {
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 191) / 192;
    const int per = (g.tiles_m * g.tiles_n + 7) / 8;
    g.ksplit = 1;
    g.tail_first = 0;
    g.tail_split = 1;
    g.part = nullptr;
    g.order = g.tiles_n <= 16 ? 1 : 0;
    const int nslots = (g.N + 95) / 96;
    if (nslots > SSQ_LD) g.ssq_out = nullptr;
    const int rawnorm = g.ssq_out ? (GEMM_RAWNORM | nslots << 8) : 0;
    hipLaunchKernelGGL((gemm_bf16_slab_kernel<EPI_RESID, 2, false, 6>), dim3(8u * per), dim3(256), 0, st, g);
    MI_HIP(hipGetLastError());
    ++g_n192_launches;
    if (rawnorm) ++g_fused_norm_launches;
    return rawnorm;
}

// whether 192-column tiles beat 256-column tiles for this residual GEMM: rounds of workgroups on 256 CUs, a 192-column
// tile costing ~0.8 of a 256-column one (48 MFMAs per step against 64; the A side of the step is unchanged)
bool n192_pays(const GemmArgs &g) {
    const long tm = (g.M + 255) / 256;
    const long t256 = tm * ((g.N + 255) / 256);
    const long r256 = (t256 + 255) / 256, r192 = (tm * ((g.N + 191) / 192) + 255) / 256;
    constexpr double rel = 0.8;
    // a short last round of 256-column tiles is split along K by launch_slab (nearly free): measured at 11 290 tokens
    // (270 tiles) the 256-column tiles win by 28 us per layer, at 13 205 (312 tiles) the 192-column ones by 66
    if (r256 >= 2 && t256 % 256 != 0 && t256 % 256 <= 48) return false;
    return g.N % 8 == 0 && (double)r192 * rel < (double)r256 - 0.05;
}

// narrow tiles (WNT = 1): no SwiGLU instantiation (it pairs two N tiles inside a wave)
template <int WMT, int WNT, int WAVES_M, int WAVES_N, int ST>
void launch_ring_narrow(int epi, GemmArgs g, hipStream_t st) {
    constexpr int BM = 16 * WMT * WAVES_M, BN = 16 * WNT * WAVES_N;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const int per = (g.tiles_m * g.tiles_n + 7) / 8;
    if (epi != EPI_RESID || g.bias) g.ksplit = 1;
    g.ksplit = std::max(1, g.ksplit);
    g.tail_first = 0;
    g.tail_split = 1;
    dim3 grid(8 * per * g.ksplit), block(64 * WAVES_M * WAVES_N);
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_STORE, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_RESID, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        case EPI_QKV: hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_QKV, WMT, WNT, WAVES_M, WAVES_N, ST>), grid, block, 0, st, g); break;
        default: throw Error("narrow GEMM tiles: epilogue not implemented");
    }
    MI_HIP(hipGetLastError());
}

// skinny GEMM (M <= 32, fragment-major weights): W streamed global -> registers, A slice in LDS
template <int EPI, int WN>
void launch_skinny_e(GemmArgs g, int wpb, hipStream_t st) {
    set_max_dynamic_lds(reinterpret_cast<const void *>(&gemm_bf16_skinny_kernel<EPI, WN>), 128 * 1024);
    // K slices: as few as the LDS-resident A slice allows, more (residual epilogue only) when
    // the N dimension alone does not give every CU a wave
    const int waves_n = (g.N + WN * 16 - 1) / (WN * 16);
    int ksplit = (g.K + SKINNY_KS_MAX - 1) / SKINNY_KS_MAX;
    if (EPI == EPI_RESID && !g.bias) ksplit = std::max(ksplit, std::min(8, 1024 / std::max(1, waves_n)));
    int kslice = ((g.K / 32 + ksplit - 1) / ksplit) * 32;
    ksplit = (g.K + kslice - 1) / kslice;
    g.ksplit = ksplit;
    const int mtiles = (g.M + 15) / 16;
    const size_t smem = (size_t)mtiles * 16 * (kslice + 8) * sizeof(bf16_t);
    const int nblocks_n = (waves_n + wpb - 1) / wpb;
    hipLaunchKernelGGL((gemm_bf16_skinny_kernel<EPI, WN>), dim3((unsigned)(nblocks_n * ksplit)), dim3(64 * wpb), smem, st,
                       g, kslice);
    MI_HIP(hipGetLastError());
}
void launch_skinny(int epi, GemmArgs g, hipStream_t st) {
    switch (epi) {
        case EPI_STORE: launch_skinny_e<EPI_STORE, 1>(g, 2, st); break;
        case EPI_RESID: launch_skinny_e<EPI_RESID, 1>(g, 2, st); break;
        case EPI_QKV: launch_skinny_e<EPI_QKV, 1>(g, 2, st); break;
        case EPI_SWIGLU: launch_skinny_e<EPI_SWIGLU, 2>(g, 4, st); break;
        default: throw Error("bad epilogue");
    }
}

// whether 192-row tiles beat 256-row tiles for this GEMM: at least a tenth fewer padded rows and no more rounds of workgroups
// (513 .. 576 tokens: three row tiles either way, 576 rows instead of 768)
bool m192_pays(const GemmArgs &g) {
    const bool off = knobs().no_m192;
    const long t192 = (g.M + 191) / 192, t256 = (g.M + 255) / 256, tn = (g.N + 255) / 256;
    return !off && t192 * 192 * 10 <= t256 * 256 * 9 && (t192 * tn + 255) / 256 <= (t256 * tn + 255) / 256;
}

// whether splitting every 256x256 tile of a residual GEMM along K (launch_slab's workspace path) fills the chip
bool mid_split_pays(const GemmArgs &g) {
    const int ntiles = ((g.M + 255) / 256) * ((g.N + 255) / 256), nt64 = g.K / 64;
    const int S = std::min({16, 256 / std::max(1, ntiles), nt64 / 4});
    constexpr int min_wg = 96;
    // long K only (the down projection): at K = 1536 a slice is 4 K tiles and the tile's fixed cost plus the reduction pass
    // (17.2 + 7.5 us at 576 tokens) lose to the 128-row ring tiles (20 us) -- profiles/r03_encode_nq16_kernel_stats_v1.csv
    constexpr int min_k = 4096;
    return g.M > 64 && g.K >= min_k && S >= 2 && ntiles * S >= min_wg && (size_t)S * g.M * g.N * 4 <= g.part_bytes;
}


// whether launch_gemm sends this GEMM to the 256 x 256 slab kernel with whole-K tiles (the many-token path): the shapes whose
// epilogues carry the fused RMSNorm / rotary embedding
bool slab_shape(int epi, const GemmArgs &g) {               // by the shape alone: the token count at which the 256 x 256 tiles take over
    const long tiles_big = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
    return tiles_big >= 100 && (epi == EPI_SWIGLU ? g.ldc % 8 == 0 : g.N % 8 == 0);
}
bool slab_whole_k(int epi, const GemmArgs &g) {
    if (!slab_shape(epi, g)) return false;
    return knobs().gemm_tile.empty() && !knobs().gemm_ring && !knobs().no_bulk_fuse;
}

// A few hundred tokens through the K = 1536 projections (QKV, O): 128 x 128 tiles are too few to fill the chip (80 / 60 at
// 576 tokens) and a CU's memory pipe bounds what one workgroup can pull (~590 KB per 128 x 64 x 1536 tile: 17-19 us a launch,
// profiles/r04_encode_nq16_kernel_stats_v1.csv).  K split S ways over ~240 workgroups, partial tiles as plain f32 planes,
// and ONE reduction pass that also finishes the epilogue: bias + RoPE + Q|K rows + V^T (QKV: rope_kernel disappears), or
// residual add + the next RMSNorm (O: rmsnorm_kernel disappears).  Returns the GEMM_* flags of what the pass did; 0 with
// nothing launched when the shape does not fit.
int launch_mid_part(int epi, GemmArgs g, hipStream_t st) {
    if (!g.part || (epi != EPI_QKV && epi != EPI_RESID) || g.N % 8 != 0 || g.N > 4096) return 0;
    if (epi == EPI_QKV && !(g.rope_pos && g.rope_cos && g.rope_sin && g.rope_hd % 8 == 0 && g.qk_cols % g.rope_hd == 0)) return 0;
    if (epi == EPI_RESID && (g.ldc != g.N)) return 0;
    constexpr int BM = 128, BN = 128;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = g.tiles_m * g.tiles_n, nk = g.K / 32;
    const int S = std::min({8, 256 / std::max(1, ntiles), nk / 8});       // >= 8 K steps per slice
    if (S < 2 || (size_t)S * g.M * g.N * 4 > g.part_bytes) return 0;
    g.ksplit = S;
    g.Wt = nullptr;
    g.tail_first = 0; g.tail_split = 1;
    const int per = (ntiles + 7) / 8;                                    // tile_coords: eight per-XCD eighths
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI_PART, 4, 4, 2, 2, 4>), dim3((unsigned)(8 * per * S)), dim3(256), 0, st, g);
    MI_HIP(hipGetLastError());
    ++g_splitk_launches;
    if (epi == EPI_QKV) {
        hipLaunchKernelGGL(splitk_reduce_qkv_kernel, dim3((unsigned)g.M), dim3(256), (size_t)g.N * 4, st, g.part, S, g.M, g.N, g.bias, g.C, g.ldc,
                           g.qk_cols, g.Vt, g.ldvt, g.rope_hd, g.rope_pos, g.rope_cos, g.rope_sin);
        MI_HIP(hipGetLastError());
        return GEMM_ROPED;
    }
    if (g.norm_w && g.norm_y && g.N <= 2048) {
        hipLaunchKernelGGL(splitk_reduce_norm_kernel, dim3((unsigned)g.M), dim3(256), 0, st, g.X, g.part, S, g.M, g.N, g.bias,
                           g.norm_w, g.norm_eps, g.norm_y);
        MI_HIP(hipGetLastError());
        ++g_reduce_norm_launches;
        return GEMM_NORMED;
    }
    const int64_t n4 = (int64_t)g.M * (g.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g.X, (int64_t)g.ldc, g.part, S, g.M, g.N, g.bias);
    MI_HIP(hipGetLastError());
    return 0x100;                                                        // launched, nothing extra folded in
}

// ---- encoder_mid.h: the K = hidden projections of ~100 .. ~4000 tokens, one launch each --------------------------------
struct MidTile { int bm, bn, ns; };
// whether launch_gemm sends this GEMM to mid_gemm_kernel.  QKV: the caller must hand over the rotary-interleaved weights and
// the (cos, sin) table (GemmArgs::rope_cs) -- `assume_rope`: the caller is asking in order to decide whether to build them.
bool mid_takes(int epi, const GemmArgs &g, bool assume_rope = false) {
    const bool off = knobs().no_mid_gemm || !knobs().gemm_tile.empty();   // (a forced tile configuration is for every GEMM)
    if (off || g.M <= 64 || g.K % 64 != 0 || g.K >= 4096 || g.N % 64 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0) return false;
    if (slab_shape(epi, g)) return false;                               // enough tokens for whole rounds of 256 x 256 tiles (whatever the knobs say)
    if (epi == EPI_QKV)
        return (assume_rope || (g.rope_cs && g.rope_pos)) && g.qk_cols % 64 == 0 && g.rope_hd % 4 == 0 && g.qk_cols % std::max(g.rope_hd, 1) == 0 &&
               g.ldc % 4 == 0 && g.M % 4 == 0;
    return epi == EPI_RESID && g.ldc == g.N;
}

MidTile mid_pick_tile(const GemmArgs &g, int qk_cols) {
    // what a workgroup costs is what it ingests through its CU's memory pipe, (BM + BN) x K x 2 bytes at ~75 GB/s; the launch
    // lasts as long as the busiest CU: ceil(workgroups / 256) of them.  Ties: the larger tile (fewer bytes chip-wide).
    // ring depth: what bounds a workgroup is its bytes IN FLIGHT (an LDS-DMA piece lands ~1.1-1.5 us after its issue: three
    // slabs of a 96 x 64 tile were 60 KB = ~45 GB/s, 14 us for the QKV projection of 576 tokens) -- one workgroup per CU with
    // most of the CU's LDS as its ring
    static const MidTile cand[] = {{128, 128, 4}, {128, 64, 6}, {96, 64, 7}, {64, 64, 8}};
    const std::string &env = knobs().mid_tile;                          // tools: "128x128" | "128x64" | "96x64" | "64x64"
    MidTile best = cand[3];
    double best_cost = 1e30;
    for (const MidTile &c : cand) {
        if (qk_cols % c.bn != 0 || g.N % c.bn != 0) continue;
        if (!env.empty() && env != std::to_string(c.bm) + "x" + std::to_string(c.bn)) continue;
        const long n = (long)((g.M + c.bm - 1) / c.bm) * (g.N / c.bn);
        const double cost = (double)((n + 255) / 256) * ((double)(c.bm + c.bn) * g.K * 2.0 / 75e3 + 1.0);
        if (cost < best_cost * 0.999) { best_cost = cost; best = c; }
    }
    return best;
}

template <int EPI, int WMT, int WNT, int NS>
void launch_mid_t(GemmArgs g, hipStream_t st) {
    constexpr int BM = 32 * WMT, BN = 32 * WNT;
    constexpr size_t lds = (size_t)NS * (BM + BN) * 128;
    static std::once_flag once[16];                                     // per device
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    std::call_once(once[dev & 15], [] {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mid_gemm_kernel<EPI, WMT, WNT, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const int per = (g.tiles_m * g.tiles_n + 7) / 8;
    g.krot = (!knobs().no_krot && 8 * per <= 512 && g.tiles_m > 1) ? knobs().krot : 0;
    hipLaunchKernelGGL((mid_gemm_kernel<EPI, WMT, WNT, NS>), dim3((unsigned)(8 * per)), dim3(256), lds, st, g);
    MI_HIP(hipGetLastError());
}

int launch_mid(int epi, GemmArgs g, hipStream_t st) {
    const MidTile t = mid_pick_tile(g, epi == EPI_QKV ? g.qk_cols : g.N);
    int flags = 0;
    if (epi == EPI_QKV) {
        flags = GEMM_ROPED;
        ++g_fused_rope_launches;
    } else {
        const int nslots = g.N / t.bn;
        if (!(g.ssq_out && g.norm_w && g.norm_y && nslots <= SSQ_LD)) g.ssq_out = nullptr;
        if (!(g.ssq_out && g.arrive)) g.rms_out = nullptr;
        if (g.ssq_out) { flags = GEMM_RAWNORM | (g.rms_out ? GEMM_RMS_DONE : 0) | nslots << 8; ++g_fused_norm_launches; }
    }
    g.part = nullptr;
#define MI_MID_CASE(BM_, BN_, NS_)                                                                              \
    if (t.bm == BM_ && t.bn == BN_) {                                                                           \
        if (epi == EPI_QKV) launch_mid_t<MID_QKV, BM_ / 32, BN_ / 32, NS_>(g, st);                              \
        else launch_mid_t<MID_O, BM_ / 32, BN_ / 32, NS_>(g, st);                                               \
    }
    MI_MID_CASE(128, 128, 4) MI_MID_CASE(128, 64, 6) MI_MID_CASE(96, 64, 7) MI_MID_CASE(64, 64, 8)
#undef MI_MID_CASE
    ++g_mid_launches;
    return flags;
}

// returns the GEMM_* flags: GEMM_NORMED when the launch also wrote the RMSNorm of the updated stream the caller asked for
// (GemmArgs::norm_w / norm_y), GEMM_ROPED when the QKV epilogue already rotated Q and K
int launch_gemm(int epi, GemmArgs g, hipStream_t st) {
    MI_REQUIRE(g.K % 64 == 0, "encoder GEMM: K must be a multiple of 64");
    MI_REQUIRE(g.lda % 8 == 0 && g.ldw % 8 == 0, "encoder GEMM: leading dimensions must be multiples of 8");
    MI_REQUIRE(g.N % 4 == 0 && g.ldc % 4 == 0, "encoder GEMM: N and ldc must be multiples of 4");
    const bool force = !knobs().gemm_tile.empty();    // tuning knob: big | mid | small | tiny | 128 (legacy kernel)
    const bool legacy = knobs().gemm_tile == "128";
    if (!legacy) {
        // tile by how many workgroups the problem yields (256 CUs to fill):
        //   big   256x256 (8 waves)  needs >= ~100 tiles to pay off
        //   mid   128x128 (4 waves)
        //   small 128x32  (2 waves, 7 K tiles in flight): skinny / few-row GEMMs
        //   tiny  32x64 + split-K on the residual GEMMs: a handful of tokens (one query)
        const long tiles_big = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        const long tiles_mid = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
        std::string cfg = knobs().gemm_tile;
        if (cfg.empty()) cfg = g.M <= 64 ? "tiny" : tiles_big >= 100 ? "big" : tiles_mid >= 150 ? "mid" : "small";
        g.ksplit = 1;
        if (mid_takes(epi, g)) return launch_mid(epi, g, st);
        // the fused RMSNorm / rotary epilogues live in the whole-K slab kernel only: a consumer that asks for them anywhere
        // else is a caller's error (its A operand is not normalised); a producer's request is just dropped
        MI_REQUIRE((!g.row_scale && !g.rope_cs) || (slab_whole_k(epi, g) && (epi == EPI_QKV || epi == EPI_SWIGLU)),
                   "encoder GEMM: fused RMSNorm / rotary epilogue requested on a shape the slab kernel does not take");
        if (!(epi == EPI_RESID && slab_whole_k(epi, g) && g.norm_w && g.norm_y && g.N % 64 == 0 && g.N <= 64 * SSQ_LD && g.ldc == g.N && !g.bias))
            g.ssq_out = nullptr;
        const bool split_k = !force && epi == EPI_RESID && g.part && g.N % 8 == 0 && mid_split_pays(g);
        if (!force && cfg == "small" && g.M > 64 && g.K < 4096 && g.part) {
            const int fl = launch_mid_part(epi, g, st);
            if (fl) return fl & 0xff;
        }
        if (!split_k) g.part = nullptr;
        if (cfg != "tiny") g.Wt = nullptr;   // only the few-token path streams fragment-major weights
        const bool skinny_ok = g.Wt && g.M <= 32 && g.N % 16 == 0 && (g.K <= SKINNY_KS_MAX || (epi == EPI_RESID && !g.bias));
        // measured per launch for one query: gate/up 14.5 vs 17.3 us (ring tiles), but QKV 10.7 vs 8.1
        // and the residual GEMMs 13.0 vs 10.2 -- so only the SwiGLU GEMM takes it
        if (skinny_ok && cfg == "tiny" && epi == EPI_SWIGLU) {
            launch_skinny(epi, g, st);
        } else if ((cfg == "slab8" || cfg == "slab4") && !(epi == EPI_SWIGLU ? g.ldc % 8 == 0 : g.N % 8 == 0)) {
            launch_ring<8, 4, 2, 4, 4>(epi, g, st);          // the slab kernel stores 8 bf16 columns per lane
        } else if (cfg == "slab8") {
            launch_slab<4>(epi, g, st);
        } else if (cfg == "slab4") {
            launch_slab<2>(epi, g, st);
        } else if (cfg == "half") {
            launch_ring<8, 4, 1, 4, 3>(epi, g, st);   // 128x256, 4 waves, 72 KiB ring: two workgroups per CU
        } else if (split_k && (cfg == "small" || cfg == "mid" || cfg == "big")) {
            // ~100 to ~5000 tokens through the down projection: the 256x256 slab kernel with EVERY tile split along K through the
            // workspace (launch_slab) -- 576 x 1536 x 8960: 18 tiles x 14 slices = 252 workgroups of 20 K steps + one reduction
            // pass, where 128x128 tiles with K split three ways by f32 atomics took 76 us; at 1558 / 2097 tokens the forward
            // pass went 7.91 -> 6.46 / 9.21 -> 7.82 ms against the 128x128 ring tiles
            // Columns of the K-split tiles.  What a slice costs is its operand stream, (BM + BN) x K / S x 2 bytes at the ~50 GB/s
            // a CU pulls, and its share of the S planes leaving and coming back (S x M x N x 4 bytes at ~3.2 and ~3.6
            // TB/s chip-wide): narrower tiles are more tiles, hence fewer, longer slices -- fewer planes.  563 tokens, forward
            // pass: 256 columns x 10 slices 3.246 ms, 128 x 7 3.09, 128 x 6 / 5 3.11 / 3.17, 64 x 3 3.33 (the model: 36.7 / 30.9 /
            // 31.5 / 33.2 / 36.8 us a launch + pass).  MI_DOWN_BN forces one width.
            const bool m192 = m192_pays(g);
            const int bm = m192 ? 192 : 256, tm_ = (g.M + bm - 1) / bm, nt64 = g.K / 64;
            int bn = 256;
            double best = 1e30;
            for (int c : {256, 192, 128}) {
                if (knobs().down_bn && knobs().down_bn != c) continue;
                const int ntiles = tm_ * ((g.N + c - 1) / c), per = (ntiles + 7) / 8;
                const int S = std::min({10, 256 / std::max(1, ntiles), nt64 / 4});
                if (8 * per >= 200 || S < 2 || (size_t)S * g.M * g.N * 4 > g.part_bytes) continue;
                const double cost = (double)(bm + c) * ((double)g.K / S) * 2.0 / 50e3 + (double)S * g.M * g.N * 4.0 * 0.59e-6;
                if (cost < best) { best = cost; bn = c; }
            }
            if (bn == 128) return m192 ? launch_slab<2, 6, 4>(epi, g, st) : launch_slab<2, 8, 4>(epi, g, st);
            if (bn == 192) return m192 ? launch_slab<2, 6, 6>(epi, g, st) : launch_slab<2, 8, 6>(epi, g, st);
            return m192 ? launch_slab<2, 6>(epi, g, st) : launch_slab<2>(epi, g, st);
        } else if (cfg == "big" && !force && epi == EPI_RESID && n192_pays(g)) {
            return launch_slab_n192(g, st);
        } else if (cfg == "big" && !knobs().gemm_ring && (epi == EPI_SWIGLU ? g.ldc % 8 == 0 : g.N % 8 == 0)) {   // the slab kernel stores 8 bf16 columns per lane
            // measured (tools/gemm_bench.py, 32768 tokens): 8 waves 1051 / 1060 TF on QKV / O, 4 waves 1106 / 1303 on
            // gate-up / down (ring kernel: 968 / 952 / 1006 / 1166)
            if ((epi == EPI_SWIGLU || (epi == EPI_RESID && g.K >= 4096)) && !force && m192_pays(g)) return launch_slab<2, 6>(epi, g, st);
            if (epi == EPI_SWIGLU || g.K >= 4096) return launch_slab<2>(epi, g, st);
            return launch_slab<4>(epi, g, st);
        } else if (cfg == "big") {
            launch_ring<8, 4, 2, 4, 4>(epi, g, st);
        } else if (cfg == "mid") {
            launch_ring<4, 4, 2, 2, 4>(epi, g, st);
        } else if (cfg == "mid64" || (cfg == "small" && !force && epi != EPI_SWIGLU && g.M > 256 && g.K < 4096)) {
            // a few hundred to ~1500 tokens through the QKV / O projections: 128x64 tiles, 4 waves of 64x32 -- 3 DMA pieces per
            // wave per K step where the 128x32 / 2-wave tiles issue 5 (they are DMA-issue-bound): QKV 16.0 -> 13.5 us at 576
            // tokens, 35.8 -> 24.6 at 1152; O 21.8 -> 14.8 at 1152 (profiles/r03_gemm_mid_bench_v2.txt)
            launch_ring<4, 2, 2, 2, 6>(epi, g, st);
        } else if (cfg == "small") {
            if (epi == EPI_RESID && !g.bias && g.K >= 4096 && !force) {
                // a few hundred tokens through the down projection (K = 8960): 128x32 tiles are one
                // long DMA-issue-bound K loop per workgroup (82 us at 512 tokens); 128x128 tiles
                // with K split three ways (f32 atomics) measured best (batch-16 encode 5.07 -> 4.80 ms;
                // 2 ways 5.11, 4 ways 4.84, 6 ways 5.38)
                g.ksplit = 3;
                launch_ring<4, 4, 2, 2, 4>(epi, g, st);
            } else {
                launch_ring<4, 2, 2, 1, 8>(epi, g, st);
            }
        } else {
            const int tiles = ((g.M + 31) / 32) * ((g.N + 63) / 64);
            const int nk = g.K / 32;
            if (epi != EPI_RESID && epi != EPI_SWIGLU && tiles < 128 && cfg != "tiny64") {
                // no split-K for a bf16 output: a 32x64 tiling of the QKV projection of one query
                // is 32 workgroups on 256 CUs -- 32x16 tiles, one wave each, give 128
                launch_ring_narrow<2, 1, 1, 1, 8>(epi, g, st);
            } else if (epi == EPI_SWIGLU) {
                // gate/up of one query streams 55 MB of weights: 32x32 one-wave tiles (560
                // workgroups, two per CU, twice the tiles in flight) measured 18.8 vs 22.4 us
                g.ksplit = 1;
                launch_ring<2, 2, 1, 1, 8>(epi, g, st);
            } else {
                g.ksplit = std::max(1, std::min({8, nk / 8, 256 / std::max(1, tiles)}));
                launch_ring<2, 2, 1, 2, 8>(epi, g, st);
            }
        }
        return 0;
    }
    g.Wt = nullptr;
    g.tiles_m = (g.M + 127) / 128;
    g.tiles_n = (g.N + 127) / 128;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int per = (ntiles + 7) / 8;
    dim3 grid(8 * per), block(256);
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_STORE>), grid, block, 0, st, g); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_RESID>), grid, block, 0, st, g); break;
        case EPI_QKV: hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_QKV>), grid, block, 0, st, g); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_SWIGLU>), grid, block, 0, st, g); break;
        default: throw Error("bad epilogue");
    }
    MI_HIP(hipGetLastError());
    return 0;
}

}  // namespace

struct mi_encoder {
    mi_encoder_cfg cfg{};
    int device = 0;
    int qk_cols = 0, v_cols = 0, q_cols = 0;
    DevBuf embed, norm_w, dense_w, dense_b, rope_cos, rope_sin, rope_cs;   // rope_cs: [pos][pair] (cos, sin) for the fused QKV epilogue
    std::vector<LayerW> layers;
    std::map<std::string, bool> loaded;
    // activations and staging buffers, one set per stream an encode is issued on: calls on different streams overlap on
    // the GPU and concurrent host threads (one stream each) never share a buffer; threads that share a stream take turns
    // A pinned host staging slot for one call's token ids / positions / work lists: the upload is ONE asynchronous copy
    // out of it, the slot's event says when the GPU has consumed it.  Three slots in rotation: a call never waits for the
    // GPU unless three earlier calls on the stream have not even started -- mi_encoder_encode only enqueues work.
    struct Pinned {
        int32_t *p = nullptr;
        size_t cap = 0;          // int32 elements
        hipEvent_t ev = nullptr;
        bool in_flight = false;
        ~Pinned() {
            if (ev) (void)hipEventDestroy(ev);
            if (p) (void)hipHostFree(p);
        }
    };
    struct WS {
        std::mutex mu;
        DevBuf ws_x, ws_xn, ws_qk, ws_vt, ws_att, ws_h, ws_up, ws_out, ws_stage, ws_part, ws_ssq, ws_arrive, few_ctr, ws_skpart, ws_skctr;
        Pinned pin[3];
        int pin_next = 0;
        size_t vt_zeroed = 0, att_zeroed = 0;
    };
    std::vector<std::pair<void *, std::unique_ptr<WS>>> ws_sets;
    std::mutex mu;           // guards ws_sets, the lazily built fragment-major weights and the profiling events
    DevBuf ws_stage;         // load_tensor staging (exclusive calls)
    bool tiled_ok = false;   // fragment-major weight copies are current
    bool few_ok = false;     // the query-time path's weight pieces are current
    int few_gen = -1;        // ... for this generation of the knobs (g_knob_gen)
    bool few_gu8 = false;    // ... and the gate/up pieces are few_gu8_kernel's (8 gate + 8 up rows per piece)
    bool roped_ok = false;   // the rotary-pair-interleaved QKV copies are current
    // profiling: the GEMM launches of the most recent encode (arguments as launched), replayed back to back between two
    // HIP events by mi_encoder_profile_read -- like the index library's scan replay; per-launch event pairs measured
    // 3-4 % short of rocprofv3's per-kernel durations in the same run
    bool prof = false;
    struct Launch { int epi; GemmArgs g; };
    std::vector<Launch> prof_launches;
    hipStream_t prof_stream = nullptr;
    double prof_flops = 0.0;
};

namespace {

struct EncLease {
    mi_encoder::WS &w;
    std::unique_lock<std::mutex> lk;
};
EncLease lease_ws(mi_encoder *h, void *stream) {
    mi_encoder::WS *w = nullptr;
    {
        std::lock_guard<std::mutex> hl(h->mu);
        for (auto &kv : h->ws_sets)
            if (kv.first == stream) w = kv.second.get();
        if (!w) {
            MI_REQUIRE(h->ws_sets.size() < 64, "too many distinct streams on one encoder handle (max 64)");
            h->ws_sets.emplace_back(stream, std::make_unique<mi_encoder::WS>());
            w = h->ws_sets.back().second.get();
        }
    }
    return EncLease{*w, std::unique_lock<std::mutex>(w->mu)};
}

void register_params(mi_encoder *h) {
    auto &L = h->loaded;
    L["embed_tokens.weight"] = false;
    L["norm.weight"] = false;
    for (int l = 0; l < h->cfg.n_layers; ++l) {
        std::string p = "layers." + std::to_string(l) + ".";
        for (const char *n : {"input_layernorm.weight", "post_attention_layernorm.weight",
                              "self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                              "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
                              "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                              "mlp.down_proj.weight"})
            L[p + n] = false;
    }
    if (h->cfg.dense_out) {
        L["dense.weight"] = false;
        if (h->cfg.dense_bias) L["dense.bias"] = false;
    }
}

// copy a tensor into its internal slot: bf16 (dst) or f32 (dst_f32), rows remapped
void import_tensor(mi_encoder *h, const void *data, int dtype, int64_t rows, int64_t cols, int64_t blk,
                   int64_t stride, int64_t off, bf16_t *dst, float *dst_f32) {
    const size_t esz = dtype == MI_DTYPE_F32 ? 4 : 2;
    const void *src = data;
    if (!is_device_ptr(data)) {
        void *stg = h->ws_stage.reserve((size_t)rows * cols * esz);
        MI_HIP(hipMemcpy(stg, data, (size_t)rows * cols * esz, hipMemcpyHostToDevice));
        src = stg;
    }
    const int64_t n = rows * cols;
    hipLaunchKernelGGL(import_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, src, dtype,
                       rows, cols, blk, stride, off, dst, dst_f32);
    MI_HIP(hipGetLastError());
    MI_HIP(hipDeviceSynchronize());
}

struct Batch {
    int nseq = 0, T_real = 0, T_pad = 0, Lmax = 0, nwork = 0;
    // device pointers into ws_up
    int32_t *ids = nullptr, *pos = nullptr;
    int32_t *seq_start = nullptr, *seq_len = nullptr, *work_seq = nullptr, *work_q0 = nullptr, *tok_map = nullptr, *out_rows = nullptr;
};

// Build the packed layout (sequences back to back, T_pad = the total rounded up to a multiple of 32) straight into a
// pinned staging slot and upload ids / positions / work lists (/ output rows) with one asynchronous copy.  Host `ids` /
// `cu_seqlens` / `out_rows` are consumed before the call returns; nothing here waits for the GPU.
Batch prepare_batch(mi_encoder *h, mi_encoder::WS &ws, int nseq, const int32_t *ids, const int32_t *cu, const int32_t *out_rows,
                    hipStream_t st) {
    MI_REQUIRE(nseq > 0, "encode: nseq must be positive");
    std::vector<int32_t> cu_dev, ids_dev;                 // device-resident inputs (rare): fetched, which does synchronise
    if (is_device_ptr(cu)) {
        cu_dev.resize((size_t)nseq + 1);
        MI_HIP(hipMemcpy(cu_dev.data(), cu, cu_dev.size() * 4, hipMemcpyDeviceToHost));
        cu = cu_dev.data();
    }
    MI_REQUIRE(cu[0] == 0, "encode: cu_seqlens[0] must be 0");
    const int T_real = cu[nseq];
    MI_REQUIRE(T_real >= nseq, "encode: cu_seqlens must be increasing");
    if (is_device_ptr(ids)) {
        ids_dev.resize((size_t)T_real);
        MI_HIP(hipMemcpy(ids_dev.data(), ids, ids_dev.size() * 4, hipMemcpyDeviceToHost));
        ids = ids_dev.data();
    }
    Batch b;
    b.nseq = nseq;
    b.T_real = T_real;
    int cur = 0, nwork = 0;
    for (int i = 0; i < nseq; ++i) {
        const int L = cu[i + 1] - cu[i];
        MI_REQUIRE(L >= 1, "encode: empty sequence");
        MI_REQUIRE(L <= h->cfg.max_seq_len, "encode: sequence longer than max_seq_len");
        b.Lmax = std::max(b.Lmax, L);
        nwork += (L + 63) / 64;
        cur += L;                 // back to back: the attention kernel masks the < 8 foreign keys in front of a sequence
    }
    b.T_pad = (cur + 31) & ~31;   // GEMM rows are clamped / guarded, no tile multiple needed
    b.nwork = nwork;
    const size_t n_up = (size_t)b.T_pad * 2 + (size_t)nseq * 3 + (size_t)nwork * 2 + (size_t)T_real;
    mi_encoder::Pinned &pin = ws.pin[ws.pin_next];
    ws.pin_next = (ws.pin_next + 1) % 3;
    if (pin.in_flight) {                                  // (three calls ago: long consumed unless the stream is that far behind)
        MI_HIP(hipEventSynchronize(pin.ev));
        pin.in_flight = false;
    }
    if (!pin.ev) MI_HIP(hipEventCreateWithFlags(&pin.ev, hipEventDisableTiming));
    if (pin.cap < n_up) {
        if (pin.p) MI_HIP(hipHostFree(pin.p));
        pin.p = nullptr;
        pin.cap = n_up + n_up / 4 + 1024;
        MI_HIP(hipHostMalloc(reinterpret_cast<void **>(&pin.p), pin.cap * 4, hipHostMallocDefault));
    }
    int32_t *ids_pad = pin.p, *pos = ids_pad + b.T_pad, *start = pos + b.T_pad, *len = start + nseq, *wseq = len + nseq,
            *wq0 = wseq + nwork, *tok_map = wq0 + nwork, *rows = tok_map + T_real;
    std::memset(ids_pad, 0, (size_t)b.T_pad * 8);         // ids and positions of the padding tokens
    cur = 0;
    int wk = 0;
    const int vocab = h->cfg.vocab_size;
    for (int i = 0; i < nseq; ++i) {
        const int c0 = cu[i], L = cu[i + 1] - c0;
        start[i] = cur;
        len[i] = L;
        rows[i] = out_rows ? out_rows[i] : i;
        for (int q0 = 0; q0 < L; q0 += 64) {
            wseq[wk] = i;
            wq0[wk++] = q0;
        }
        for (int t = 0; t < L; ++t) {
            const int32_t id = ids[(size_t)c0 + t];
            MI_REQUIRE(id >= 0 && id < vocab, "encode: token id out of range");
            ids_pad[(size_t)cur + t] = id;
            pos[(size_t)cur + t] = t;
            tok_map[(size_t)c0 + t] = cur + t;
        }
        cur += L;
    }
    int32_t *d = ws.ws_up.as<int32_t>(n_up);
    MI_HIP(hipMemcpyAsync(d, pin.p, n_up * 4, hipMemcpyHostToDevice, st));
    MI_HIP(hipEventRecord(pin.ev, st));
    pin.in_flight = true;
    b.ids = d;
    b.pos = d + b.T_pad;
    b.seq_start = b.pos + b.T_pad;
    b.seq_len = b.seq_start + nseq;
    b.work_seq = b.seq_len + nseq;
    b.work_q0 = b.work_seq + nwork;
    b.tok_map = b.work_q0 + nwork;
    b.out_rows = b.tok_map + T_real;
    return b;
}

// profile launch (MI_GEMM_TS=1): in-kernel s_memtime stamps of every workgroup's phases (256x256 tiles), mean / max to stderr
void stamped_launch(int epi, GemmArgs g, hipStream_t stream) {
    const int M = g.M, N = g.N, K = g.K;
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nb = std::max(1024, 8 * ((tiles + 7) / 8) * 2);   // (tail-split / all-tiles-split launches have more workgroups than tiles)
    DevBuf tsb;
    unsigned long long *dts = tsb.as<unsigned long long>((size_t)nb * 8);
    MI_HIP(hipMemsetAsync(dts, 0, (size_t)nb * 64, stream));
    g.ts = dts;
    g.ts_rows = nb;
    launch_gemm(epi, g, stream);
    std::vector<unsigned long long> h((size_t)nb * 8);
    MI_HIP(hipStreamSynchronize(stream));
    MI_HIP(hipMemcpy(h.data(), dts, h.size() * 8, hipMemcpyDeviceToHost));
    const char *names[8] = {"", "first K tile landed", "barrier passed", "K loop issued", "epilogue issued", "stores acknowledged", "", ""};
    std::fprintf(stderr, "[gemm stamps] epilogue %d M %d N %d K %d, %d workgroup slots, s_memtime ticks since the workgroup's start\n", epi, M, N, K, nb);
    const bool slab = h[5] > 4096;           // the slab kernel writes the CU identity into slot 5 and the absolute start into slot 0
    for (int i = 1; i <= (slab ? 4 : 5); ++i) {
        double sum = 0, mx = 0; size_t cnt = 0;
        for (int b = 0; b < nb; ++b) { const double v = (double)h[(size_t)b * 8 + i]; if (v > 0) { sum += v - 1; mx = std::max(mx, v - 1); ++cnt; } }
        if (cnt) std::fprintf(stderr, "  %-22s n=%6zu mean %9.1f max %9.1f\n", slab ? (i == 1 ? "first slabs landed" : i == 2 ? "K loop done" : i == 3 ? "epilogue issued" : "stores acknowledged") : names[i], cnt, sum / cnt, mx);
    }
    if (slab) {
        // per CU: the gap between a workgroup's last stamp and the start of the next workgroup on the same CU
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
        for (int b = 0; b < nb; ++b)
            if (h[(size_t)b * 8 + 4] && h[(size_t)b * 8]) cu[h[(size_t)b * 8 + 5] & 0xf0000ff00ull].push_back({h[(size_t)b * 8], h[(size_t)b * 8] + h[(size_t)b * 8 + 4] - 1});
        double gs = 0, gmax = 0; size_t gn = 0; unsigned long long t_first = ~0ull, t_last = 0;
        for (auto &kv : cu) {
            std::sort(kv.second.begin(), kv.second.end());
            for (size_t i = 0; i < kv.second.size(); ++i) {
                t_first = std::min(t_first, kv.second[i].first); t_last = std::max(t_last, kv.second[i].second);
                if (i) { const double gap = (double)kv.second[i].first - (double)kv.second[i - 1].second; gs += gap; gmax = std::max(gmax, gap); ++gn; }
            }
        }
        std::fprintf(stderr, "  %zu CU identities; gap between consecutive workgroups of a CU: n=%zu mean %.1f max %.1f; first start -> last end %.1f\n",
                     cu.size(), gn, gn ? gs / gn : 0.0, gmax, (double)(t_last - t_first));
    }
}

int timed_gemm(mi_encoder *h, int epi, const GemmArgs &g, hipStream_t st) {
    const int enc_ts = knobs().enc_ts;                                   // stamps of the first layer's four GEMMs, in place
    static std::atomic<int> ts_left{4};
    if (enc_ts > 0 && g.M > (enc_ts == 1 ? 4096 : enc_ts) && ts_left.fetch_sub(1) > 0) stamped_launch(epi, g, st);
    const int normed = launch_gemm(epi, g, st);
    if (!h->prof) return normed;
    std::lock_guard<std::mutex> hl(h->mu);
    h->prof_launches.push_back({epi, g});
    h->prof_stream = st;
    h->prof_flops += 2.0 * (double)g.M * (double)g.N * (double)g.K;
    return normed;
}

void launch_attention(mi_encoder *h, const Batch &b, const bf16_t *qk, const bf16_t *vt, bf16_t *att, int ldvt, hipStream_t st,
                      bf16_t *att_frag = nullptr, int frag_mt = 0) {
    const mi_encoder_cfg &c = h->cfg;
    const int hd = c.head_dim;
    AttnArgs a{};
    a.QK = qk; a.Vt = vt; a.O = att; a.Ofrag = att_frag; a.frag_mt = frag_mt; a.work_seq = b.work_seq; a.work_q0 = b.work_q0;
    a.seq_start = b.seq_start; a.seq_len = b.seq_len; a.ldqk = h->qk_cols; a.ldvt = ldvt;
    a.n_heads = c.n_heads; a.n_kv = c.n_kv_heads; a.causal = c.causal;
    a.scale = 1.0f / std::sqrt((float)hd);
    // every sequence short enough for one pass over its keys (a batch of queries): one wave per (head, 16-query tile), no LDS,
    // no online softmax -- few_attn_kernel (encoder_few.h)
    if (b.Lmax <= FEW_MAX_T && (hd == 64 || hd == 128) && !knobs().no_short_attn && b.nwork > 0 && b.nwork <= 24) {   // (16 queries: 6.3 vs 7.9 us; 64: 13.0 vs 11.0 -- a wave per tile is one latency chain each)
        const dim3 ga((unsigned)c.n_heads, (unsigned)(b.nwork * 3));
        if (hd == 128) hipLaunchKernelGGL((few_attn_kernel<128>), ga, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((few_attn_kernel<64>), ga, dim3(64), 0, st, a);
        MI_HIP(hipGetLastError());
        ++g_short_attn_launches;
        return;
    }
    // two query heads of one K/V head per workgroup when the GQA group allows it
    const bool pair = c.n_heads % 2 == 0 && (c.n_heads / c.n_kv_heads) % 2 == 0;
    a.nwork = b.nwork;
    // persistent workgroups: two per CU (64 KiB of LDS each at head dim 128 x 2 heads), or one per item when there are fewer
    constexpr int wgs_cu = 2;
    const int hpw = pair ? 2 : 1;
    const unsigned nwg = (unsigned)std::min<long>((long)b.nwork * (c.n_heads / hpw), 256L * wgs_cu);
    if (hd == 128 && pair) hipLaunchKernelGGL((attn_kernel<128, 2>), dim3(nwg), dim3(512), 0, st, a);
    else if (hd == 128) hipLaunchKernelGGL((attn_kernel<128, 1>), dim3(nwg), dim3(256), 0, st, a);
    else if (pair) hipLaunchKernelGGL((attn_kernel<64, 2>), dim3(nwg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((attn_kernel<64, 1>), dim3(nwg), dim3(256), 0, st, a);
    MI_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// The query-time path (encoder_few.h): T <= 48 tokens, six launches per layer, no atomics.
// ---------------------------------------------------------------------------------------------------------------
bool few_eligible(const mi_encoder *h, const Batch &b) {
    const bool off = knobs().no_few;
    const mi_encoder_cfg &c = h->cfg;
    const int mt = (b.T_real + 15) / 16;
    return !off && b.T_real <= FEW_MAX_T && c.hidden % 32 == 0 && c.hidden <= 4096 && c.intermediate % 32 == 0 && h->q_cols % 32 == 0 &&
           (h->qk_cols + h->v_cols) % 16 == 0 && c.head_dim % 32 == 0 &&
           (size_t)std::max({c.hidden / 32, h->q_cols / 32, 2 * FEW_NW}) * mt * 1024 + FEW_NW * 64 * 4 <= 160 * 1024;
}

template <int MT>
void few_set_attributes() {
    const int lim = 160 * 1024;
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&few_gemm_kernel<FEW_QKV, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&few_gemm_kernel<FEW_GU, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&few_gu8_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&few_o_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&few_d_kernel<MT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&few_d_kernel<MT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
}

// the weight pieces of every layer, built once per handle (and again after a load_tensor); caller holds h->mu
void few_build_weights(mi_encoder *h, hipStream_t st) {
    const mi_encoder_cfg &c = h->cfg;
    const int H = c.hidden, I = c.intermediate;
    auto tile = [&](const DevBuf &src, int N, int K, int rope_blocks, DevBuf &dstb, bool half) {
        bf16_t *d = dstb.as<bf16_t>((size_t)N * K);
        if (half)
            hipLaunchKernelGGL(few_tile_kernel<8>, dim3((unsigned)((N / 8) * (K / 32))), dim3(64), 0, st, src.get<bf16_t>(), N, K, K, 0, c.head_dim, d);
        else
            hipLaunchKernelGGL(few_tile_kernel<16>, dim3((unsigned)((N / 16) * (K / 32))), dim3(64), 0, st, src.get<bf16_t>(), N, K, K,
                               rope_blocks, c.head_dim, d);
        MI_HIP(hipGetLastError());
    };
    // 8-feature gate/up units: every wave keeps its K range of the fragments in registers (FEW_GKP steps), a workgroup <= FEW_GNU units
    h->few_gu8 = !knobs().no_few_gu8 && I % 32 == 0 && (H / 32 + FEW_GW - 1) / FEW_GW <= FEW_GKP &&
                 (I / 8 + std::min(I / 8, 256) - 1) / std::min(I / 8, 256) <= FEW_GNU;
    for (auto &w : h->layers) {
        tile(w.wqkv, h->qk_cols + h->v_cols, H, h->qk_cols / 16, w.few_qkv, false);
        tile(w.wo, H, h->q_cols, 0, w.few_o, true);
        {
            bf16_t *d = w.few_op.as<bf16_t>((size_t)H * h->q_cols);
            hipLaunchKernelGGL((few_tile_kernel<8, true>), dim3((unsigned)((H / 8) * (h->q_cols / 32))), dim3(64), 0, st, w.wo.get<bf16_t>(), H,
                               h->q_cols, h->q_cols, 0, c.head_dim, d);
            MI_HIP(hipGetLastError());
        }
        if (h->few_gu8) {
            hipLaunchKernelGGL((few_tile_kernel<16, false, true>), dim3((unsigned)((2 * I / 16) * (H / 32))), dim3(64), 0, st, w.wgu.get<bf16_t>(), 2 * I, H,
                               H, 0, c.head_dim, w.few_gu.as<bf16_t>((size_t)2 * I * H));
            MI_HIP(hipGetLastError());
        } else {
            tile(w.wgu, 2 * I, H, 0, w.few_gu, false);
        }
        tile(w.wd, H, I, 0, w.few_d, false);
    }
    few_set_attributes<1>();
    few_set_attributes<2>();
    few_set_attributes<3>();
    MI_HIP(hipStreamSynchronize(st));
    h->few_ok = true;
    h->few_gen = g_knob_gen.load();
}

// leaves the residual stream (before the final norm) in x
template <int MT>
void few_stack(mi_encoder *h, mi_encoder::WS &ws, const Batch &b, float *x, bf16_t *qk, bf16_t *vt, int ldvt, hipStream_t st) {
    const mi_encoder_cfg &c = h->cfg;
    const int H = c.hidden, I = c.intermediate, T = b.T_real, T_pad = b.T_pad;
    // fragments: the normalised stream (H / 32 steps), the attention output (q_cols / 32), h (I / 32); MT KiB per step
    bf16_t *xfrag = ws.ws_xn.as<bf16_t>((size_t)T_pad * H);
    bf16_t *afrag = ws.ws_att.as<bf16_t>((size_t)T_pad * h->q_cols);
    bf16_t *hfrag = ws.ws_h.as<bf16_t>((size_t)T_pad * I);
    float *ssq = ws.ws_stage.as<float>((size_t)(H / 8) * FEW_SSQ_LD);
    auto gemm_grid = [](int nunits) { return std::max(std::min(nunits, 256), (nunits + 2) / 3); };
    auto gemm_smem = [&](int nk, int wn) { return (size_t)std::max(nk, FEW_NW * wn) * MT * 1024 + (size_t)FEW_NW * 64 * 4; };
    // K slices of the down projection: about one workgroup per CU
    FewArgs d0{};
    d0.nk = I / 32; d0.nunits = H / 16;
    const int ugroups = (d0.nunits + 3) / 4;
    {
        constexpr int d_wgs = 248;                           // workgroups the down projection aims at
        const int ns = std::max(1, std::min(d0.nk, (d_wgs + ugroups / 2) / ugroups));
        d0.ks_per_slice = (d0.nk + ns - 1) / ns;
        while ((size_t)d0.ks_per_slice * MT * 1024 > 128 * 1024) d0.ks_per_slice = (d0.ks_per_slice + 1) / 2;
        d0.nslices = (d0.nk + d0.ks_per_slice - 1) / d0.ks_per_slice;
    }
    float *part = static_cast<float *>(ws.ws_part.reserve((size_t)d0.nslices * T_pad * H * 4));
    // MI_FEW_D_FUSE=1: the down projection's reduction inside its own launch (write-through planes, one arrival counter per
    // unit group, the last slice to arrive finishes the group's columns) instead of the few_row_kernel pass.  Measured and
    // OFF by default: 13.2 us against 7.7 + 4.8 for the two launches (one query 1.265 vs 1.211 ms) -- the 8-byte
    // write-through stores and the 80 KB the last arriver reads back past its L1 cost more than a kernel boundary, as the
    // guide's splitk-seam row prices it.  Kept as the parity-tested alternative (tests run both).
    const bool fuse_env = knobs().few_d_fuse;
    const bool fuse_d = fuse_env && ugroups <= FEW_SSQ_LD * 3 && d0.nslices * d0.ks_per_slice >= d0.nk;
    if (ws.few_ctr.cap < (size_t)ugroups * 4) {
        MI_HIP(hipMemsetAsync(ws.few_ctr.reserve((size_t)ugroups * 4 + 1024), 0, (size_t)ugroups * 4 + 1024, st));
    }
    unsigned *ctr = ws.few_ctr.get<unsigned>();
    // MI_FEW_SYNC=1 (debugging): wait after every launch and name the stage on stderr
    const bool dbg_sync = knobs().few_sync;
    // MI_FEW_TS=1 (profiling): s_memtime stamps of every workgroup of the first layer's two fragment GEMMs, mean / max per phase
    const bool dbg_ts = knobs().few_ts;
    DevBuf tsb;
    auto stamps = [&](const char *name, int nwg) {
        std::vector<unsigned long long> hts((size_t)nwg * 8);
        MI_HIP(hipStreamSynchronize(st));
        MI_HIP(hipMemcpy(hts.data(), tsb.p, hts.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int g = 0; g < nwg; ++g) { t0 = std::min(t0, hts[(size_t)g * 8]); t1 = std::max(t1, hts[(size_t)g * 8 + 7]); }
        std::fprintf(stderr, "[few stamps] %s: %d workgroups, first start -> last end %llu ticks; per phase (since the workgroup's start) mean / max; start skew mean / max\n", name, nwg, t1 - t0);
        const char *ph[8] = {"start", "requests issued", "fragments stored", "barrier 1", "stream done", "barrier 2", "partials stored+barrier 3", "end"};
        for (int i = 1; i < 8; ++i) {
            double sum = 0, mx = 0;
            for (int g = 0; g < nwg; ++g) { const double v = (double)(hts[(size_t)g * 8 + i] - hts[(size_t)g * 8]); sum += v; mx = std::max(mx, v); }
            std::fprintf(stderr, "  %-26s %9.0f %9.0f\n", ph[i], sum / nwg, mx);
        }
        double ssum = 0, smx = 0;
        for (int g = 0; g < nwg; ++g) { const double v = (double)(hts[(size_t)g * 8] - t0); ssum += v; smx = std::max(smx, v); }
        std::fprintf(stderr, "  %-26s %9.0f %9.0f\n", "start skew", ssum / nwg, smx);
    };
    auto chk = [&](const char *stage) {
        MI_HIP(hipGetLastError());
        if (dbg_sync) {
            const hipError_t e = hipStreamSynchronize(st);
            std::fprintf(stderr, "[few] %s: %s\n", stage, hipGetErrorString(e));
        }
    };

    // one sequence of <= 32 tokens (the prompted query): the attention runs inside the O projection's workgroups (few_ao_kernel),
    // its operands arrive from the QKV epilogue as pieces (they fit the row buffers: T_pad >= 16 MT)
    const size_t ao_lds = ((size_t)c.n_kv_heads * (2 * (c.head_dim / 32) + c.head_dim / 16) + (size_t)FEW_AW * MT) * 1024;
    const bool fuse_ao = MT <= 2 && b.nseq == 1 && !knobs().no_few_ao && (c.head_dim == 64 || c.head_dim == 128) &&
                         h->q_cols == c.n_heads * c.head_dim && h->v_cols == c.n_kv_heads * c.head_dim &&
                         h->qk_cols == (c.n_heads + c.n_kv_heads) * c.head_dim && c.n_heads % std::max(c.n_kv_heads, 1) == 0 &&
                         T_pad >= 16 * MT && ldvt >= 16 * MT && ao_lds <= 64 * 1024;
    FewArgs e{};
    e.T = T; e.H = H; e.x = x; e.norm_w = h->layers[0].ln1.get<float>(); e.eps = c.rms_eps; e.xfrag = xfrag; e.ids = b.ids;
    e.table = h->embed.get<bf16_t>();
    hipLaunchKernelGGL((few_row_kernel<true, MT>), dim3((unsigned)T), dim3(256), 0, st, e);
    chk("embed");
    for (int l = 0; l < c.n_layers; ++l) {
        LayerW &w = h->layers[l];
        Range layer_range("mi_encoder:layer");
        FewArgs q{};
        q.T = T; q.H = H; q.nk = H / 32; q.nunits = (h->qk_cols + h->v_cols) / 16; q.W = w.few_qkv.get<bf16_t>(); q.afrag = xfrag;
        q.attn_pieces = fuse_ao ? 1 : 0; q.n_heads = c.n_heads; q.n_kv = c.n_kv_heads;
        if (fuse_d && l > 0) { q.ssq = ssq; q.nparts = ugroups; }    // (the previous layer's down projection left bf16(x g) and partial sums of squares)
        q.eps = c.rms_eps; q.bias = w.bqkv.get<float>(); q.qk = qk; q.vt = vt;
        q.ldqk = h->qk_cols; q.ldvt = ldvt; q.qk_cols = h->qk_cols; q.hd = c.head_dim; q.rope_blocks = h->qk_cols / 16;
        q.pos = b.pos; q.cos_t = h->rope_cos.get<float>(); q.sin_t = h->rope_sin.get<float>();
        if (dbg_ts && l == 0) q.ts = tsb.as<unsigned long long>((size_t)gemm_grid(q.nunits) * 8);
        // fragments in registers (one unit per workgroup, a wave's K range <= FEW_GKP steps): few_qkv8_kernel
        const bool qkv8 = !knobs().no_few_qkv8 && (q.nk + FEW_GW - 1) / FEW_GW <= FEW_GKP && q.nunits <= 256 && !(dbg_ts && l == 0);
        if (qkv8) {
            if (l == 0) g_few_qkv8_passes.fetch_add(1);
            hipLaunchKernelGGL((few_qkv8_kernel<MT>), dim3((unsigned)q.nunits), dim3(64 * FEW_GW), 0, st, q);
        } else
        hipLaunchKernelGGL((few_gemm_kernel<FEW_QKV, MT>), dim3((unsigned)gemm_grid(q.nunits)), dim3(64 * FEW_NW), gemm_smem(q.nk, 1), st, q);
        if (q.ts) stamps("qkv", gemm_grid(q.nunits));
        chk("qkv");
        if (l == 0 && fuse_ao) g_few_ao_passes.fetch_add(1);
        if (fuse_ao) {
            if constexpr (MT <= 2) {
                FewArgs o{};
                o.T = T; o.H = H; o.nk = h->q_cols / 32; o.nunits = H / 8; o.W = w.few_op.get<bf16_t>(); o.x = x;
                o.norm_w = w.ln2.get<float>(); o.ssq_out = ssq; o.xfrag = xfrag;
                o.qk = qk; o.vt = vt; o.ldqk = h->qk_cols; o.ldvt = ldvt; o.n_heads = c.n_heads; o.n_kv = c.n_kv_heads; o.causal = c.causal;
                o.scale = 1.0f / std::sqrt((float)c.head_dim);
                if (c.head_dim == 128) hipLaunchKernelGGL((few_ao_kernel<128, MT>), dim3((unsigned)o.nunits), dim3(64 * FEW_AW), ao_lds, st, o);
                else hipLaunchKernelGGL((few_ao_kernel<64, MT>), dim3((unsigned)o.nunits), dim3(64 * FEW_AW), ao_lds, st, o);
                chk("attention + o");
            }
        } else if (b.Lmax > FEW_MAX_T || (c.head_dim != 64 && c.head_dim != 128)) {
            launch_attention(h, b, qk, vt, nullptr, ldvt, st, afrag, MT);
        } else {
            AttnArgs aa{};
            aa.QK = qk; aa.Vt = vt; aa.Ofrag = afrag; aa.frag_mt = MT; aa.work_seq = b.work_seq; aa.work_q0 = b.work_q0;
            aa.seq_start = b.seq_start; aa.seq_len = b.seq_len; aa.ldqk = h->qk_cols; aa.ldvt = ldvt;
            aa.n_heads = c.n_heads; aa.n_kv = c.n_kv_heads; aa.causal = c.causal; aa.scale = 1.0f / std::sqrt((float)c.head_dim);
            aa.nwork = b.nwork;
            const dim3 ga((unsigned)c.n_heads, (unsigned)(b.nwork * 3));
            if (c.head_dim == 128) hipLaunchKernelGGL((few_attn_kernel<128>), ga, dim3(64), 0, st, aa);
            else hipLaunchKernelGGL((few_attn_kernel<64>), ga, dim3(64), 0, st, aa);
        }
        if (!fuse_ao) {
        chk("attention");
        FewArgs o{};
        o.T = T; o.H = H; o.nk = h->q_cols / 32; o.nunits = H / 8; o.W = w.few_o.get<bf16_t>(); o.afrag = afrag; o.x = x;
        o.norm_w = w.ln2.get<float>(); o.ssq_out = ssq; o.xfrag = xfrag;
        if (!knobs().no_few_qkv8 && (o.nk + FEW_OW - 1) / FEW_OW <= 6)   // fragments in registers (MI_NO_FEW_QKV8 turns both projections back)
            hipLaunchKernelGGL((few_o8_kernel<MT>), dim3((unsigned)o.nunits), dim3(64 * FEW_OW), 0, st, o);
        else
        hipLaunchKernelGGL((few_o_kernel<MT>), dim3((unsigned)o.nunits), dim3(64 * FEW_OW), (size_t)std::max(o.nk, FEW_OW) * MT * 1024, st, o);
        chk("o");
        }
        FewArgs u{};
        u.T = T; u.H = H; u.nk = H / 32; u.nunits = I / 16; u.W = w.few_gu.get<bf16_t>(); u.afrag = xfrag; u.eps = c.rms_eps;
        u.ssq = ssq; u.nparts = H / 8; u.hfrag = hfrag;
        if (h->few_gu8) {
            u.nunits = I / 8;
            if (l == 0) g_few_gu8_passes.fetch_add(1);
            if (dbg_ts && l == 0) {
                u.ts = tsb.as<unsigned long long>((size_t)256 * 8);
                MI_HIP(hipMemsetAsync(u.ts, 0, (size_t)256 * 64, st));
            }
            hipLaunchKernelGGL((few_gu8_kernel<MT>), dim3((unsigned)std::min(u.nunits, 256)), dim3(64 * FEW_GW),
                               (size_t)FEW_GW * FEW_GNU * MT * 1024 + (size_t)FEW_GW * 64 * 4, st, u);
            if (u.ts) stamps("gate/up (8-feature units; phases: 1 requests issued, 2 first unit streamed, 4 stream done, 5 barrier, 7 end)", std::min(u.nunits, 256));
        } else {
        if (dbg_ts && l == 0) u.ts = tsb.as<unsigned long long>((size_t)gemm_grid(u.nunits) * 8);
        hipLaunchKernelGGL((few_gemm_kernel<FEW_GU, MT>), dim3((unsigned)gemm_grid(u.nunits)), dim3(64 * FEW_NW), gemm_smem(u.nk, 2), st, u);
        if (u.ts) stamps("gate/up", gemm_grid(u.nunits));
        }
        chk("gu");
        FewArgs d = d0;
        d.T = T; d.H = H; d.W = w.few_d.get<bf16_t>(); d.afrag = hfrag; d.part = part; d.T_pad = T_pad;
        if (fuse_d && l + 1 < c.n_layers) {                  // (the last layer keeps the row kernel: the pooling tail wants the final norm)
            // the last slice of a unit group to arrive finishes the group's columns: no reduction launch
            d.x = x; d.ctr = ctr; d.ssq_out = ssq; d.xfrag = xfrag;
            d.norm_w = l + 1 < c.n_layers ? h->layers[l + 1].ln1.get<float>() : nullptr;
            hipLaunchKernelGGL((few_d_kernel<MT, true>), dim3((unsigned)(ugroups * d.nslices)), dim3(256), (size_t)d.ks_per_slice * MT * 1024 + 16, st, d);
            chk("d+reduce");
            continue;
        }
        hipLaunchKernelGGL((few_d_kernel<MT, false>), dim3((unsigned)(ugroups * d.nslices)), dim3(256), (size_t)d.ks_per_slice * MT * 1024, st, d);
        chk("d");
        FewArgs r{};
        r.T = T; r.H = H; r.x = x; r.part = part; r.T_pad = T_pad; r.nslices = d.nslices; r.eps = c.rms_eps; r.xfrag = xfrag;
        r.norm_w = l + 1 < c.n_layers ? h->layers[l + 1].ln1.get<float>() : h->norm_w.get<float>();   // (last layer: the final norm, for the pooling tail)
        hipLaunchKernelGGL((few_row_kernel<false, MT>), dim3((unsigned)T), dim3(256), 0, st, r);
        chk("reduce");
    }
}

// the decoder stack: leaves the residual stream (before the final norm) in ws_x
// returns the token tiles (1..3) of a pass that took the query-time path -- its final-norm fragments are in ws_xn --, else 0
int run_stack(mi_encoder *h, mi_encoder::WS &ws, const Batch &b, hipStream_t st) {
    const mi_encoder_cfg &c = h->cfg;
    for (auto &kv : h->loaded) MI_REQUIRE(kv.second, std::string("encoder parameter not loaded: ") + kv.first);
    const int H = c.hidden, I = c.intermediate, T = b.T_pad, hd = c.head_dim;
    const int ldvt = T + 64;
    float *x = ws.ws_x.as<float>((size_t)T * H);
    bf16_t *xn = ws.ws_xn.as<bf16_t>((size_t)T * H);
    bf16_t *qk = ws.ws_qk.as<bf16_t>((size_t)T * h->qk_cols);
    const size_t vt_bytes = (size_t)h->v_cols * ldvt * 2;
    bf16_t *vt = static_cast<bf16_t *>(ws.ws_vt.reserve(vt_bytes));
    if (ws.vt_zeroed != ws.ws_vt.cap) {  // fresh allocation: the 64-token slack must hold finite values
        MI_HIP(hipMemsetAsync(ws.ws_vt.p, 0, ws.ws_vt.cap, st));
        ws.vt_zeroed = ws.ws_vt.cap;
    }
    bf16_t *att = ws.ws_att.as<bf16_t>((size_t)T * h->q_cols);
    if (ws.att_zeroed != ws.ws_att.cap) {
        // The attention kernel writes the rows of real tokens only; the rows of padding tokens
        // (between packed sequences, up to T_pad) feed the output projection and from there the
        // padding rows of the residual stream, K and V^T -- which neighbouring real tokens do
        // multiply by their masked (exactly zero) probabilities: 0 x NaN = NaN.  Recycled device
        // memory is not zero, so a fresh buffer is cleared once; afterwards it only ever holds
        // finite values.
        MI_HIP(hipMemsetAsync(ws.ws_att.p, 0, ws.ws_att.cap, st));
        ws.att_zeroed = ws.ws_att.cap;
    }
    bf16_t *hb = ws.ws_h.as<bf16_t>((size_t)T * I);
    // split-K workspace of the residual GEMMs when the tokens are too few for their tiles to fill the chip (launch_slab)
    float *part = nullptr;
    size_t part_bytes = 0;
    const int part_slices = std::min(16, 256 / (((T + 255) / 256) * ((H + 255) / 256)));   // launch_slab's slice count for T x H
    if (T > 64 && part_slices >= 2) {
        part_bytes = (size_t)part_slices * T * H * 4;
        part = static_cast<float *>(ws.ws_part.reserve(part_bytes));
    }

    if (h->prof) {
        std::lock_guard<std::mutex> hl(h->mu);
        h->prof_launches.clear();
        h->prof_flops = 0.0;
    }
    if (few_eligible(h, b)) {
        {
            std::lock_guard<std::mutex> hl(h->mu);         // the pieces are built once; other streams wait for them
            if (!h->few_ok || h->few_gen != g_knob_gen.load()) few_build_weights(h, st);
        }
        Range stack_range("mi_encoder:stack(few)");
        g_few_passes.fetch_add(1);
        const int mt = (b.T_real + 15) / 16;
        if (mt == 1) few_stack<1>(h, ws, b, x, qk, vt, ldvt, st);
        else if (mt == 2) few_stack<2>(h, ws, b, x, qk, vt, ldvt, st);
        else few_stack<3>(h, ws, b, x, qk, vt, ldvt, st);
        return mt;
    }
    // few tokens (a query, or a handful): the GEMMs stream the weights once and are bound by how
    // they read them -- use the fragment-major copies (a second copy of the layer weights, built
    // on first use: 16-row x 64-byte fragments of a row-major matrix are 16 DRAM pages per wave-load)
    const bool few = T <= 64 && H % 32 == 0 && I % 32 == 0 && h->q_cols % 32 == 0 &&
                     (h->qk_cols + h->v_cols) % 16 == 0 && (2 * I) % 16 == 0 && H % 16 == 0;
    std::unique_lock<std::mutex> tiled_lk(h->mu, std::defer_lock);
    if (few) tiled_lk.lock();                            // the copies are built once; other streams wait for them
    if (few && !h->tiled_ok) {
        auto tile = [&](const DevBuf &src, int N, int K, DevBuf &dstb) {
            bf16_t *d = dstb.as<bf16_t>((size_t)N * K);
            hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)((N / 16) * (K / 32))), dim3(64), 0, st,
                               src.get<bf16_t>(), N, K, K, d);
            MI_HIP(hipGetLastError());
        };
        for (auto &w : h->layers) {
            tile(w.wqkv, h->qk_cols + h->v_cols, H, w.wqkv_t);
            tile(w.wo, H, h->q_cols, w.wo_t);
            tile(w.wgu, 2 * I, H, w.wgu_t);
            tile(w.wd, H, I, w.wd_t);
        }
        MI_HIP(hipStreamSynchronize(st));
        h->tiled_ok = true;
    }
    if (few) tiled_lk.unlock();
    // many tokens: the RMSNorms ride in the epilogues of the slab GEMMs either side of them and the rotary embedding in the
    // QKV epilogue (GemmArgs::ssq_out / ssq_in / rope_cs) wherever both GEMMs take the whole-K slab kernel
    GemmArgs probe{};
    probe.M = T; probe.N = h->qk_cols + h->v_cols; probe.K = H;
    const bool qkv_slab = slab_whole_k(EPI_QKV, probe) && T % 8 == 0 && ldvt % 8 == 0;
    // (H < 4096: launch_gemm gives the QKV projection the 8-wave slab kernel, whose waves own 64 columns = 32 rotary pairs)
    probe.qk_cols = h->qk_cols; probe.rope_hd = hd; probe.lda = probe.ldw = H; probe.ldc = h->qk_cols;
    const bool qkv_mid = mid_takes(EPI_QKV, probe, true);   // a few hundred .. few thousand tokens: encoder_mid.h, RoPE in its epilogue
    const bool rope_fused = (qkv_slab && H < 4096 && h->qk_cols % 256 == 0 && h->v_cols % 256 == 0 && !knobs().no_rope_fuse) || qkv_mid;
    const bool norm1_fused = qkv_slab && !knobs().no_norm_fuse;
    probe.N = 2 * I; probe.ldc = I;
    const bool norm2_fused = slab_whole_k(EPI_SWIGLU, probe) && !knobs().no_norm_fuse;
    float *ssq = (norm1_fused || norm2_fused) ? ws.ws_ssq.as<float>((size_t)T * (SSQ_LD + 1)) : nullptr;
    float *inv_rms = ssq ? ssq + (size_t)T * SSQ_LD : nullptr;
    // encoder_mid.h's O projection finishes the RMSNorm itself: one arrival counter per row block (zeroed once: the kernel
    // leaves them at zero)
    unsigned *arrive = nullptr;
    if (norm2_fused) {
        constexpr size_t NCTR = 4096;
        if (ws.ws_arrive.cap < NCTR * 4) MI_HIP(hipMemsetAsync(ws.ws_arrive.reserve(NCTR * 4), 0, NCTR * 4, st));
        arrive = ws.ws_arrive.get<unsigned>();
    }
    // persistent slab GEMMs (>= two rounds of 256 x 256 tiles): the stream-K workspace -- a partial tile per workgroup, arrival
    // counters zeroed once (the kernel leaves them at zero)
    float *sk_part = nullptr;
    unsigned *sk_ctr = nullptr;
    if (knobs().gemm_persist && T >= 1536) {               // (gate/up reaches two rounds of tiles first: ceil(T / 256) x 70 >= 512)
        sk_part = static_cast<float *>(ws.ws_skpart.reserve((size_t)256 * 256 * 256 * 4));
        if (ws.ws_skctr.cap < SK_CTR_WORDS * 4) MI_HIP(hipMemsetAsync(ws.ws_skctr.reserve(SK_CTR_WORDS * 4), 0, SK_CTR_WORDS * 4, st));
        sk_ctr = ws.ws_skctr.get<unsigned>();
    }
    unsigned *sk_err = sk_part ? sk_err_word(h->device) : nullptr;
    auto row_rms = [&](int nslots) -> const float * {       // the slots a residual epilogue left -> 1/rms per row
        hipLaunchKernelGGL(row_rms_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, ssq, nslots, T, H, c.rms_eps, inv_rms);
        MI_HIP(hipGetLastError());
        if (h->prof) {                                       // replayed with the GEMMs (mi_encoder_profile_read)
            GemmArgs r{};
            r.ssq_out = ssq; r.N = nslots; r.M = T; r.K = H; r.norm_eps = c.rms_eps; r.row_scale = inv_rms;
            std::lock_guard<std::mutex> hl(h->mu);
            h->prof_launches.push_back({-1, r});
        }
        return inv_rms;
    };
    if (rope_fused) {
        std::lock_guard<std::mutex> hl(h->mu);             // the interleaved copies are built once; other streams wait for them
        if (!h->roped_ok) {
            const int N = h->qk_cols + h->v_cols;
            for (auto &w : h->layers) {
                const int64_t n = (int64_t)N * (H / 8);
                hipLaunchKernelGGL(interleave_qk_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.wqkv.get<bf16_t>(),
                                   w.bqkv.get<float>(), h->qk_cols, N, H, hd, w.wqkv_r.as<bf16_t>((size_t)N * H), w.bqkv_r.as<float>((size_t)N));
                MI_HIP(hipGetLastError());
            }
            MI_HIP(hipStreamSynchronize(st));
            h->roped_ok = true;
        }
    }
    Range stack_range("mi_encoder:stack");
    int normed = 0;                                  // what the previous residual GEMM left of this layer's first RMSNorm
    if (H <= 2048 && H % 4 == 0) {   // the first layer's RMSNorm rides in the embedding gather
        hipLaunchKernelGGL(embed_norm_kernel, dim3((T + 3) / 4), dim3(256), 0, st, b.ids, h->embed.get<bf16_t>(), H, T, x,
                           h->layers[0].ln1.get<float>(), c.rms_eps, xn);
        normed = GEMM_NORMED;
    } else {
        hipLaunchKernelGGL(embed_kernel, dim3((T + 3) / 4), dim3(256), 0, st, b.ids, h->embed.get<bf16_t>(), H, T, x);
    }
    MI_HIP(hipGetLastError());
    for (int l = 0; l < c.n_layers; ++l) {
        LayerW &w = h->layers[l];
        Range layer_range("mi_encoder:layer");
        if (!(normed & (GEMM_NORMED | GEMM_RAWNORM)))  // (else the previous layer's down projection wrote it)
            hipLaunchKernelGGL(rmsnorm_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, w.ln1.get<float>(), H, T,
                               c.rms_eps, xn);
        GemmArgs g{};
        g.A = xn; g.lda = H; g.W = w.wqkv.get<bf16_t>(); g.ldw = H; g.M = T; g.N = h->qk_cols + h->v_cols; g.K = H;
        g.bias = w.bqkv.get<float>(); g.C = qk; g.ldc = h->qk_cols; g.Vt = vt; g.ldvt = ldvt; g.qk_cols = h->qk_cols;
        if (few) g.Wt = w.wqkv_t.get<bf16_t>();
        g.part = part; g.part_bytes = part_bytes;    // (a few hundred tokens: K split + one pass that also rotates Q and K)
        g.rope_pos = b.pos; g.rope_cos = h->rope_cos.get<float>(); g.rope_sin = h->rope_sin.get<float>(); g.rope_hd = hd;
        g.sk_part = sk_part; g.sk_ctr = sk_ctr; g.sk_err = sk_err;
        if (normed & GEMM_RAWNORM) g.row_scale = row_rms(normed >> 8);
        if (rope_fused) { g.W = w.wqkv_r.get<bf16_t>(); g.bias = w.bqkv_r.get<float>(); g.rope_cs = h->rope_cs.get<float2>(); }
        if (!(timed_gemm(h, EPI_QKV, g, st) & GEMM_ROPED)) {
            const int nh_qk = c.n_heads + c.n_kv_heads;
            const int64_t n = (int64_t)T * nh_qk * (hd / 16);
            hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, qk, h->qk_cols,
                               nh_qk, hd, b.pos, h->rope_cos.get<float>(),
                               h->rope_sin.get<float>(), T);
        }
        launch_attention(h, b, qk, vt, att, ldvt, st);
        GemmArgs o{};
        o.A = att; o.lda = h->q_cols; o.W = w.wo.get<bf16_t>(); o.ldw = h->q_cols; o.M = T; o.N = H; o.K = h->q_cols;
        o.X = x; o.ldc = H; o.part = part; o.part_bytes = part_bytes;
        o.sk_part = sk_part; o.sk_ctr = sk_ctr; o.sk_err = sk_err;
        if (few) o.Wt = w.wo_t.get<bf16_t>();
        o.norm_w = w.ln2.get<float>(); o.norm_y = xn; o.norm_eps = c.rms_eps;   // (rides in a split-K reduction pass when there is one,
        if (norm2_fused) { o.ssq_out = ssq; o.rms_out = inv_rms; o.arrive = arrive; }   //  or in the slab / mid epilogue: unscaled row + sums of squares)
        const int on = timed_gemm(h, EPI_RESID, o, st);
        if (!(on & (GEMM_NORMED | GEMM_RAWNORM)))
            hipLaunchKernelGGL(rmsnorm_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, w.ln2.get<float>(), H, T,
                               c.rms_eps, xn);
        GemmArgs u{};
        u.A = xn; u.lda = H; u.W = w.wgu.get<bf16_t>(); u.ldw = H; u.M = T; u.N = 2 * I; u.K = H; u.C = hb; u.ldc = I;
        if (few) u.Wt = w.wgu_t.get<bf16_t>();
        u.sk_part = sk_part; u.sk_ctr = sk_ctr; u.sk_err = sk_err;
        if (on & GEMM_RAWNORM) u.row_scale = (on & GEMM_RMS_DONE) ? inv_rms : row_rms(on >> 8);
        timed_gemm(h, EPI_SWIGLU, u, st);
        GemmArgs d{};
        d.A = hb; d.lda = I; d.W = w.wd.get<bf16_t>(); d.ldw = I; d.M = T; d.N = H; d.K = I; d.X = x; d.ldc = H;
        d.part = part; d.part_bytes = part_bytes;
        d.sk_part = sk_part; d.sk_ctr = sk_ctr; d.sk_err = sk_err;
        if (few) d.Wt = w.wd_t.get<bf16_t>();
        if (l + 1 < c.n_layers) {                    // the next layer's first RMSNorm can ride in this GEMM the same way
            d.norm_w = h->layers[l + 1].ln1.get<float>(); d.norm_y = xn; d.norm_eps = c.rms_eps;
            if (norm1_fused) d.ssq_out = ssq;
        }
        normed = timed_gemm(h, EPI_RESID, d, st);
    }
    return 0;
}

}  // namespace

extern "C" {

const char *mi_enc_last_error(void) { return last_error().c_str(); }

int mi_encoder_create(const mi_encoder_cfg *cfg, int device, mi_encoder **out) {
    return guard([&] {
        MI_REQUIRE(cfg && out, "null argument");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) {
            (void)hipGetLastError();
            throw Error("no HIP device available: the MI355X encoder has no CPU fallback");
        }
        MI_REQUIRE(device >= 0 && device < ndev, "invalid device ordinal");
        const mi_encoder_cfg &c = *cfg;
        MI_REQUIRE(c.head_dim == 64 || c.head_dim == 128, "head_dim must be 64 or 128");
        MI_REQUIRE(c.hidden > 0 && c.hidden % 64 == 0, "hidden must be a multiple of 64");
        MI_REQUIRE(c.intermediate > 0 && c.intermediate % 64 == 0, "intermediate must be a multiple of 64");
        MI_REQUIRE((c.n_heads * c.head_dim) % 64 == 0, "n_heads*head_dim must be a multiple of 64");
        MI_REQUIRE(c.n_kv_heads > 0 && c.n_heads % c.n_kv_heads == 0, "n_heads must be a multiple of n_kv_heads");
        MI_REQUIRE(c.n_layers > 0 && c.vocab_size > 0 && c.max_seq_len > 0, "bad config");
        MI_REQUIRE(c.dense_out >= 0, "bad dense_out");
        DeviceGuard dg(device);
        auto h = std::make_unique<mi_encoder>();
        h->cfg = c;
        h->device = device;
        h->q_cols = c.n_heads * c.head_dim;
        h->qk_cols = (c.n_heads + c.n_kv_heads) * c.head_dim;
        h->v_cols = c.n_kv_heads * c.head_dim;
        const size_t H = c.hidden, I = c.intermediate;
        h->embed.reserve((size_t)c.vocab_size * H * 2);
        h->norm_w.reserve(H * 4);
        if (c.dense_out) {
            h->dense_w.reserve((size_t)c.dense_out * H * 2);
            h->dense_b.reserve((size_t)c.dense_out * 4);
            MI_HIP(hipMemset(h->dense_b.p, 0, (size_t)c.dense_out * 4));
        }
        h->layers.resize(c.n_layers);
        for (auto &w : h->layers) {
            w.wqkv.reserve((size_t)(h->qk_cols + h->v_cols) * H * 2);
            w.bqkv.reserve((size_t)(h->qk_cols + h->v_cols) * 4);
            w.wo.reserve(H * (size_t)h->q_cols * 2);
            w.wgu.reserve(2 * I * H * 2);
            w.wd.reserve(H * I * 2);
            w.ln1.reserve(H * 4);
            w.ln2.reserve(H * 4);
        }
        // rotary tables in float64 -> f32, exactly like the oracle
        const int half = c.head_dim / 2;
        std::vector<float> cs((size_t)c.max_seq_len * half), sn((size_t)c.max_seq_len * half);
        for (int p = 0; p < c.max_seq_len; ++p)
            for (int i = 0; i < half; ++i) {
                const double inv = 1.0 / std::pow((double)c.rope_theta, (double)(2 * i) / (double)c.head_dim);
                const double ang = (double)p * inv;
                cs[(size_t)p * half + i] = (float)std::cos(ang);
                sn[(size_t)p * half + i] = (float)std::sin(ang);
            }
        MI_HIP(hipMemcpy(h->rope_cos.reserve(cs.size() * 4), cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
        MI_HIP(hipMemcpy(h->rope_sin.reserve(sn.size() * 4), sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> both(cs.size() * 2);
        for (size_t i = 0; i < cs.size(); ++i) { both[2 * i] = cs[i]; both[2 * i + 1] = sn[i]; }
        MI_HIP(hipMemcpy(h->rope_cs.reserve(both.size() * 4), both.data(), both.size() * 4, hipMemcpyHostToDevice));
        register_params(h.get());
        *out = h.release();
    });
}

int mi_encoder_destroy(mi_encoder *h) {
    return guard([&] {
        if (!h) return;
        DeviceGuard dg(h->device);
        delete h;
    });
}

int mi_encoder_load_tensor(mi_encoder *h, const char *name_c, const void *data, int dtype, const int64_t *shape,
                           int ndim) {
    return guard([&] {
        MI_REQUIRE(h && name_c && data && shape, "null argument");
        MI_REQUIRE(dtype >= 0 && dtype <= 2, "bad dtype");
        h->tiled_ok = false;   // any new weight invalidates the fragment-major copies
        h->few_ok = false;
        h->roped_ok = false;
        std::string name(name_c);
        if (name.rfind("model.", 0) == 0) name = name.substr(6);
        auto it = h->loaded.find(name);
        MI_REQUIRE(it != h->loaded.end(), "unknown parameter name: " + name);
        DeviceGuard dg(h->device);
        const mi_encoder_cfg &c = h->cfg;
        const int64_t H = c.hidden, I = c.intermediate, hd = c.head_dim;
        const int64_t BIG = (int64_t)1 << 40;
        auto expect = [&](int64_t r, int64_t cc) {
            const bool ok = (ndim == 2 && shape[0] == r && shape[1] == cc) || (ndim == 1 && cc == 1 && shape[0] == r);
            MI_REQUIRE(ok, "shape mismatch for " + name);
        };
        if (name == "embed_tokens.weight") {
            expect(c.vocab_size, H);
            import_tensor(h, data, dtype, c.vocab_size, H, BIG, 0, 0, h->embed.get<bf16_t>(), nullptr);
        } else if (name == "norm.weight") {
            expect(H, 1);
            import_tensor(h, data, dtype, H, 1, BIG, 0, 0, nullptr, h->norm_w.get<float>());
        } else if (name == "dense.weight") {
            expect(c.dense_out, H);
            import_tensor(h, data, dtype, c.dense_out, H, BIG, 0, 0, h->dense_w.get<bf16_t>(), nullptr);
        } else if (name == "dense.bias") {
            expect(c.dense_out, 1);
            import_tensor(h, data, dtype, c.dense_out, 1, BIG, 0, 0, nullptr, h->dense_b.get<float>());
        } else {
            MI_REQUIRE(name.rfind("layers.", 0) == 0, "unknown parameter name: " + name);
            const size_t dot = name.find('.', 7);
            const int l = std::stoi(name.substr(7, dot - 7));
            MI_REQUIRE(l >= 0 && l < c.n_layers, "layer index out of range");
            const std::string sub = name.substr(dot + 1);
            LayerW &w = h->layers[l];
            const int64_t qc = c.n_heads * hd, kc = c.n_kv_heads * hd;
            if (sub == "input_layernorm.weight") { expect(H, 1); import_tensor(h, data, dtype, H, 1, BIG, 0, 0, nullptr, w.ln1.get<float>()); }
            else if (sub == "post_attention_layernorm.weight") { expect(H, 1); import_tensor(h, data, dtype, H, 1, BIG, 0, 0, nullptr, w.ln2.get<float>()); }
            else if (sub == "self_attn.q_proj.weight") { expect(qc, H); import_tensor(h, data, dtype, qc, H, BIG, 0, 0, w.wqkv.get<bf16_t>(), nullptr); }
            else if (sub == "self_attn.k_proj.weight") { expect(kc, H); import_tensor(h, data, dtype, kc, H, BIG, 0, qc, w.wqkv.get<bf16_t>(), nullptr); }
            else if (sub == "self_attn.v_proj.weight") { expect(kc, H); import_tensor(h, data, dtype, kc, H, BIG, 0, qc + kc, w.wqkv.get<bf16_t>(), nullptr); }
            else if (sub == "self_attn.q_proj.bias") { expect(qc, 1); import_tensor(h, data, dtype, qc, 1, BIG, 0, 0, nullptr, w.bqkv.get<float>()); }
            else if (sub == "self_attn.k_proj.bias") { expect(kc, 1); import_tensor(h, data, dtype, kc, 1, BIG, 0, qc, nullptr, w.bqkv.get<float>()); }
            else if (sub == "self_attn.v_proj.bias") { expect(kc, 1); import_tensor(h, data, dtype, kc, 1, BIG, 0, qc + kc, nullptr, w.bqkv.get<float>()); }
            else if (sub == "self_attn.o_proj.weight") { expect(H, qc); import_tensor(h, data, dtype, H, qc, BIG, 0, 0, w.wo.get<bf16_t>(), nullptr); }
            // gate / up rows interleaved in blocks of 16 so that one MFMA wave tile
            // holds gate_j and up_j of the same j (SwiGLU fused in the GEMM epilogue)
            else if (sub == "mlp.gate_proj.weight") { expect(I, H); import_tensor(h, data, dtype, I, H, 16, 32, 0, w.wgu.get<bf16_t>(), nullptr); }
            else if (sub == "mlp.up_proj.weight") { expect(I, H); import_tensor(h, data, dtype, I, H, 16, 32, 16, w.wgu.get<bf16_t>(), nullptr); }
            else if (sub == "mlp.down_proj.weight") { expect(H, I); import_tensor(h, data, dtype, H, I, BIG, 0, 0, w.wd.get<bf16_t>(), nullptr); }
            else throw Error("unknown parameter name: " + name);
        }
        it->second = true;
    });
}

int mi_encoder_missing(mi_encoder *h, int *count) {
    return guard([&] {
        MI_REQUIRE(h && count, "null argument");
        int n = 0;
        for (auto &kv : h->loaded) n += kv.second ? 0 : 1;
        *count = n;
    });
}

int mi_encoder_out_dim(mi_encoder *h, int *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        *out = h->cfg.dense_out ? h->cfg.dense_out : h->cfg.hidden;
    });
}

}  // extern "C"

namespace {

// out_rows != null: embedding i goes to row out_rows[i] of `out` (device pointer) -- the caller's un-sort, done by the
// last kernel of the call instead of an indexing pass behind it
void encode_impl(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu, int normalize, float *out,
                 const int32_t *out_rows, void *stream) {
    {
        MI_REQUIRE(h && ids && cu && out, "null argument");
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        EncLease lease = lease_ws(h, stream);
        mi_encoder::WS &ws = lease.w;
        const bool od_dev = is_device_ptr(out);
        MI_REQUIRE(!out_rows || od_dev, "encode: out_rows needs a device output");
        Batch b = prepare_batch(h, ws, nseq, ids, cu, out_rows, st);
        const int few_mt = run_stack(h, ws, b, st);
        Range pool_range("mi_encoder:pool+dense+normalise");
        const mi_encoder_cfg &c = h->cfg;
        const int od = c.dense_out ? c.dense_out : c.hidden;
        if (few_mt && c.dense_out && c.hidden <= 2048 && od_dev) {
            // the query-time tail: pooling from the final-norm fragments + Dense, then normalise + place the row (two launches)
            float *raw = ws.ws_out.as<float>((size_t)nseq * od);
            FewPoolArgs p{};
            p.xfrag = ws.ws_xn.get<bf16_t>(); p.MT = few_mt; p.H = c.hidden; p.out_dim = od;
            p.parts = std::max(1, std::min(od / 16, 256 / std::max(1, nseq)));
            p.seq_start = b.seq_start; p.seq_len = b.seq_len; p.dense_w = h->dense_w.get<bf16_t>(); p.dense_b = h->dense_b.get<float>();
            p.raw = raw;
            hipLaunchKernelGGL(few_pool_kernel, dim3((unsigned)nseq, (unsigned)p.parts), dim3(256), 0, st, p);
            MI_HIP(hipGetLastError());
            hipLaunchKernelGGL(few_finish_kernel, dim3((unsigned)nseq), dim3(256), 0, st, raw, od, normalize, out_rows ? b.out_rows : nullptr, out);
            MI_HIP(hipGetLastError());
            return;
        }
        float *o = (od_dev && !out_rows) ? out : ws.ws_out.as<float>((size_t)nseq * od);
        auto finish = [&] {
            if (out_rows) {
                hipLaunchKernelGGL(scatter_rows_kernel, dim3(nseq), dim3(256), 0, st, o, od, b.out_rows, out);
                MI_HIP(hipGetLastError());
            } else if (!od_dev) {
                MI_HIP(hipMemcpyAsync(out, o, (size_t)nseq * od * 4, hipMemcpyDeviceToHost, st));
                MI_HIP(hipStreamSynchronize(st));
            }
        };
        PoolArgs p{};
        p.x = ws.ws_x.get<float>(); p.norm_w = h->norm_w.get<float>();
        p.dense_w = c.dense_out ? h->dense_w.get<bf16_t>() : nullptr;
        p.dense_b = c.dense_out ? h->dense_b.get<float>() : nullptr;
        p.seq_start = b.seq_start; p.seq_len = b.seq_len; p.out = o; p.H = c.hidden; p.out_dim = od;
        p.Lmax = b.Lmax; p.normalize = normalize; p.eps = c.rms_eps;
        const size_t smem = ((size_t)b.Lmax + c.hidden + od + 8) * 4;
        // many sequences: final RMSNorm over all tokens (one wave per token, HBM speed), a mean-pool kernel, then Dense
        // as ONE GEMM over all sequences -- the pooled vectors as its bf16 A operand, residual epilogue on a zeroed
        // output (= A W^T + bias) -- and the row normalisation.  (A workgroup per sequence doing all of that is a
        // chain of dependent row reads plus a private pass over the 3 MB Dense matrix: 0.70 ms per 128-abstract batch.)
        // MI_POOL_GEMM=0: the per-sequence kernel for every batch size.
        const bool pool_gemm_off = knobs().pool_gemm == 0;
        if (c.dense_out && nseq >= 64 && !pool_gemm_off && c.hidden % 64 == 0 && od % 4 == 0) {
            const int H = c.hidden, T = b.T_pad;
            bf16_t *xn = ws.ws_xn.as<bf16_t>((size_t)T * H);
            hipLaunchKernelGGL(rmsnorm_kernel, dim3((T + 3) / 4), dim3(256), 0, st, ws.ws_x.get<float>(), h->norm_w.get<float>(), H, T,
                               c.rms_eps, xn);
            MI_HIP(hipGetLastError());
            bf16_t *pb = ws.ws_stage.as<bf16_t>((size_t)nseq * H);
            hipLaunchKernelGGL(meanpool_kernel, dim3(nseq, (H + 511) / 512), dim3(256), 0, st, xn, b.seq_start, b.seq_len, H, pb);
            MI_HIP(hipGetLastError());
            MI_HIP(hipMemsetAsync(o, 0, (size_t)nseq * od * 4, st));
            GemmArgs g{};
            g.A = pb; g.lda = H; g.W = h->dense_w.get<bf16_t>(); g.ldw = H;
            g.M = nseq; g.N = od; g.K = H; g.bias = h->dense_b.get<float>(); g.X = o; g.ldc = od;
            launch_gemm(EPI_RESID, g, st);
            if (normalize) {
                hipLaunchKernelGGL(l2norm_rows_kernel, dim3(nseq), dim3(256), 0, st, o, od);
                MI_HIP(hipGetLastError());
            }
            finish();
            return;
        }
        // few sequences: split the Dense rows of each over several workgroups (256 CUs to fill)
        constexpr int rows_per_part = 16;
        p.parts = c.dense_out ? std::max(1, std::min(od / rows_per_part, 256 / std::max(1, nseq))) : 1;
        hipLaunchKernelGGL(pool_kernel, dim3(nseq, p.parts), dim3(256), smem, st, p);
        MI_HIP(hipGetLastError());
        if (p.parts > 1 && normalize) {
            hipLaunchKernelGGL(l2norm_rows_kernel, dim3(nseq), dim3(256), 0, st, o, od);
            MI_HIP(hipGetLastError());
        }
        finish();
    }
}

}  // namespace

extern "C" {

int mi_encoder_encode(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu, int normalize, float *out,
                      void *stream) {
    return guard([&] { encode_impl(h, nseq, ids, cu, normalize, out, nullptr, stream); });
}

int mi_encoder_encode_rows(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu, int normalize, float *out,
                           const int32_t *out_rows, void *stream) {
    return guard([&] {
        MI_REQUIRE(out_rows, "null argument");
        MI_REQUIRE(!is_device_ptr(out_rows), "mi_encoder_encode_rows: out_rows is a host array");
        encode_impl(h, nseq, ids, cu, normalize, out, out_rows, stream);
    });
}

int mi_encoder_hidden(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu, float *out, void *stream) {
    return guard([&] {
        MI_REQUIRE(h && ids && cu && out, "null argument");
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        EncLease lease = lease_ws(h, stream);
        mi_encoder::WS &ws = lease.w;
        Batch b = prepare_batch(h, ws, nseq, ids, cu, nullptr, st);
        run_stack(h, ws, b, st);
        const int H = h->cfg.hidden;
        const bool od_dev = is_device_ptr(out);
        float *o = od_dev ? out : ws.ws_out.as<float>((size_t)b.T_real * H);
        hipLaunchKernelGGL(final_norm_kernel, dim3((b.T_real + 3) / 4), dim3(256), 0, st, ws.ws_x.get<float>(),
                           h->norm_w.get<float>(), H, b.T_real, b.tok_map, h->cfg.rms_eps, o);
        MI_HIP(hipGetLastError());
        if (!od_dev) {
            MI_HIP(hipMemcpyAsync(out, o, (size_t)b.T_real * H * 4, hipMemcpyDeviceToHost, st));
            MI_HIP(hipStreamSynchronize(st));
        }
    });
}

int mi_encoder_profile_enable(mi_encoder *h, int on) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        h->prof = on != 0;
    });
}

int mi_encoder_profile_read(mi_encoder *h, double *gemm_ms, double *gemm_flops) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        DeviceGuard dg(h->device);
        std::vector<mi_encoder::Launch> launches;
        hipStream_t st = nullptr;
        {
            std::lock_guard<std::mutex> hl(h->mu);
            MI_REQUIRE(!h->prof_launches.empty(), "mi_encoder_profile_read: no encode has run with profiling on");
            launches = h->prof_launches;
            st = h->prof_stream;
        }
        EncLease lease = lease_ws(h, st);                // the replay writes that stream's workspaces: no encode on it meanwhile
        // the launches of that encode again, back to back on its stream (its workspaces are still in place; the residual
        // GEMMs add into the stream once more, which nobody reads afterwards): one warm pass, then `reps` timed ones
        const int reps = 3;
        hipEvent_t e0, e1;
        MI_HIP(hipEventCreate(&e0));
        MI_HIP(hipEventCreate(&e1));
        // (epi -1: the row_rms_kernel between a fused-RMSNorm residual GEMM and its consumer -- part of that fusion's cost, and
        // what keeps the replayed stream's values finite: without it the consumers would scale by a stale 1/rms, the stream
        // would overflow within a pass, and GEMMs on NaN operands draw less power and clock higher than real ones)
        auto replay = [&](const mi_encoder::Launch &l) {
            if (l.epi >= 0) { launch_gemm(l.epi, l.g, st); return; }
            hipLaunchKernelGGL(row_rms_kernel, dim3((unsigned)((l.g.M + 255) / 256)), dim3(256), 0, st, l.g.ssq_out, l.g.N, l.g.M, l.g.K,
                               l.g.norm_eps, const_cast<float *>(l.g.row_scale));
            MI_HIP(hipGetLastError());
        };
        for (auto &l : launches) replay(l);
        MI_HIP(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r)
            for (auto &l : launches) replay(l);
        MI_HIP(hipEventRecord(e1, st));
        MI_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MI_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (gemm_ms) *gemm_ms = (double)ms / reps;
        std::lock_guard<std::mutex> hl(h->mu);
        if (gemm_flops) *gemm_flops = h->prof_flops;
        h->prof_launches.clear();
        h->prof_flops = 0.0;
    });
}

int mi_enc_debug_counter(const char *name, int64_t *value) {
    return guard([&] {
        MI_REQUIRE(name && value, "null argument");
        if (std::string(name) == "tail_split_launches") *value = g_tail_split_launches.load();
        else if (std::string(name) == "splitk_launches") *value = g_splitk_launches.load();
        else if (std::string(name) == "reduce_norm_launches") *value = g_reduce_norm_launches.load();
        else if (std::string(name) == "n192_launches") *value = g_n192_launches.load();
        else if (std::string(name) == "persist_launches") *value = g_persist_launches.load();
        else if (std::string(name) == "sk_giveups") {          // (synchronises the device: tests only)
            int dev = 0;
            MI_HIP(hipGetDevice(&dev));
            unsigned v = 0;
            MI_HIP(hipDeviceSynchronize());
            MI_HIP(hipMemcpy(&v, sk_err_word(dev), 4, hipMemcpyDeviceToHost));
            *value = v;
        }
        else if (std::string(name) == "m192_launches") *value = g_m192_launches.load();
        else if (std::string(name) == "fused_norm_launches") *value = g_fused_norm_launches.load();
        else if (std::string(name) == "fused_rope_launches") *value = g_fused_rope_launches.load();
        else if (std::string(name) == "few_passes") *value = g_few_passes.load();
        else if (std::string(name) == "few_ao_passes") *value = g_few_ao_passes.load();
        else if (std::string(name) == "few_gu8_passes") *value = g_few_gu8_passes.load();
        else if (std::string(name) == "few_qkv8_passes") *value = g_few_qkv8_passes.load();
        else if (std::string(name) == "mid_launches") *value = g_mid_launches.load();
        else if (std::string(name) == "short_attn_launches") *value = g_short_attn_launches.load();
        else throw Error(std::string("unknown debug counter: ") + name);
    });
}

int mi_encoder_reload_env(void) {
    return guard([&] { knobs_mut().load(); ++g_knob_gen; });
}

int mi_enc_gemm_bf16(int device, int M, int N, int K, const void *A, const void *W, void *C, void *stream) {
    return guard([&] {
        MI_REQUIRE(A && W && C, "null argument");
        MI_REQUIRE(is_device_ptr(A) && is_device_ptr(W) && is_device_ptr(C), "mi_enc_gemm_bf16: device pointers only");
        DeviceGuard dg(device);
        GemmArgs g{};
        g.A = static_cast<const bf16_t *>(A); g.lda = K; g.W = static_cast<const bf16_t *>(W); g.ldw = K;
        g.M = M; g.N = N; g.K = K; g.C = static_cast<bf16_t *>(C); g.ldc = N;
        if (knobs().gemm_persist) {                          // the stream-K workspace of a persistent launch (tools: one call at a time per device)
            static std::mutex mu;
            static float *skp[16] = {};                          // (process-lifetime: 64 MiB + 8 KiB per device that calls this)
            static unsigned *skc[16] = {};
            std::lock_guard<std::mutex> lk(mu);
            const int dv = device & 15;
            if (!skp[dv]) {
                MI_HIP(hipMalloc(reinterpret_cast<void **>(&skp[dv]), (size_t)256 * 256 * 256 * 4));
                MI_HIP(hipMalloc(reinterpret_cast<void **>(&skc[dv]), SK_CTR_WORDS * 4));
                MI_HIP(hipMemset(skc[dv], 0, SK_CTR_WORDS * 4));
            }
            g.sk_part = skp[dv];
            g.sk_ctr = skc[dv];
            g.sk_err = sk_err_word(device);
        }
        if (knobs().gemm_ts) {
            stamped_launch(EPI_STORE, g, as_stream(stream));
            return;
        }
        launch_gemm(EPI_STORE, g, as_stream(stream));
    });
}

}  // extern "C"
