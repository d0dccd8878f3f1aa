// ivfpq.hip -- C ABI (include/mi_ivfpq.h) over the gfx950 kernels of
// ivfpq_kernels.h.  Host orchestration only: no arithmetic of the hot path
// happens on the CPU here, and there is no CPU fallback -- without a HIP
// device every entry point fails with an error.
#include "../../include/mi_ivfpq.h"

#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.h"
#include "ivfpq_kernels.h"
// The ring-pipelined 16-bit MFMA GEMM of the encoder library, f16 instantiation (approximate
// coarse scores).  Wrapped in a namespace of its own: both shared libraries live in one
// process, and identically named kernels / host stubs in two of them preempt each other.
#include <hip/hip_fp16.h>
namespace mi_ring {
#include "encoder_kernels.h"
}
namespace mienc = mi_ring::mienc;

using namespace mi;

namespace {

hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ------------------------------------------------------------------
// Environment knobs.  Read ONCE -- at the first call into the library, or again by mi_ivfpq_reload_env() (tests and tools:
// not to be called beside a running search) -- into this struct: nothing on a search / add call path calls getenv, which is
// not safe against a concurrent setenv.  Every knob here is exercised by a test or a committed tool; the dispatch rules
// they override are the measured defaults.
// ------------------------------------------------------------------
struct Knobs {
    int two_stage;        // MI_TWO_STAGE=0|1: force the two-stage coarse quantiser off / on (-1: by shape)
    bool refine_gmax;     // MI_REFINE_GMAX=0: its second stage reads the score rows instead of the GEMM's group maxima
    bool refine_stats;    // MI_REFINE_STATS=1: print candidates per row of the second stage (synchronises)
    bool refine_ts;       // MI_REFINE_TS=1: print the mean in-kernel phase stamps of the second stage (synchronises; tools/micro/coarse_stage.py)
    int select_big_from;  // MI_SELECT_BIG_FROM: smallest K the sort-based selection takes (65)
    int nslice;           // MI_NSLICE: scan slices per query (0: by shape)
    int scan_nw;          // MI_SCAN_NW=8|16: waves per scan workgroup (0: by shape)
    bool no_allscores;    // MI_NO_ALLSCORES=1: k > 64 by one extraction pass per 64 results
    bool no_fused_merge;  // MI_NO_FUSED_MERGE=1: cross-slice merge as its own launch
    bool no_sliced_select;   // MI_NO_SLICED_SELECT=1: one workgroup per row for a few long rows too
    bool no_topk_rows;    // MI_NO_TOPK_ROWS=1: the re-rank's final top-k through the general merge
    bool scan_ts;         // MI_SCAN_TS=1: mi_index_profile_scan prints in-kernel phase stamps
    bool scan_prune;      // MI_SCAN_PRUNE=0: no exact list pruning (every probed list is scanned, as faiss does)
    int prune_mode;       // MI_SCAN_PRUNE_MODE=1: the two-phase form only; 2: the in-kernel early stop only (0: early stop where it applies, else two phases)
    int prune_p1;         // MI_SCAN_PRUNE_P1: lists per query scanned before the others are tested (0: nprobe / 32, 1..8)
    int64_t prune_min_groups;   // MI_SCAN_PRUNE_MIN_GROUPS: smallest scan (64-code groups of all probed lists of the call) that is pruned
    void load() {
        auto num = [](const char *n, int dflt) { const char *e = std::getenv(n); return e && *e ? std::atoi(e) : dflt; };
        auto set = [](const char *n) { return std::getenv(n) != nullptr; };
        two_stage = num("MI_TWO_STAGE", -1);
        refine_gmax = num("MI_REFINE_GMAX", 1) != 0;
        refine_stats = set("MI_REFINE_STATS");
        refine_ts = set("MI_REFINE_TS");
        select_big_from = num("MI_SELECT_BIG_FROM", 65);
        nslice = num("MI_NSLICE", 0);
        scan_nw = num("MI_SCAN_NW", 0);
        no_allscores = set("MI_NO_ALLSCORES");
        no_fused_merge = set("MI_NO_FUSED_MERGE");
        no_topk_rows = set("MI_NO_TOPK_ROWS");
        no_sliced_select = set("MI_NO_SLICED_SELECT");
        scan_ts = set("MI_SCAN_TS");
        scan_prune = num("MI_SCAN_PRUNE", 1) != 0;
        prune_mode = num("MI_SCAN_PRUNE_MODE", 0);
        prune_p1 = num("MI_SCAN_PRUNE_P1", 0);
        prune_min_groups = num("MI_SCAN_PRUNE_MIN_GROUPS", 200000);
    }
};
Knobs &knobs_mut() {
    static Knobs k = [] { Knobs x{}; x.load(); return x; }();
    return k;
}
inline const Knobs &knobs() { return knobs_mut(); }

// ------------------------------------------------------------------
// kernel launch helpers
// ------------------------------------------------------------------

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, int PF>
void launch_gemm_cfg(const float *A, int na, const float *B, int nb, int d, float *S, int64_t ldS,
                     hipStream_t st, const LutArgs &la) {
    constexpr int BM = 16 * WM * WAVES_M, BN = 16 * WN * WAVES_N;
    int tiles_m = (na + BM - 1) / BM, tiles_n = (nb + BN - 1) / BN;
    int64_t grid = (int64_t)8 * tiles_m * ((tiles_n + 7) / 8);
    MI_REQUIRE(grid + la.nblocks < (int64_t)1 << 31, "ip_gemm: grid too large");
    static_assert(WAVES_M * WAVES_N * 64 == 256, "appended LUT workgroups need 256 threads");
    hipLaunchKernelGGL((ip_gemm_kernel<WM, WN, WAVES_M, WAVES_N, BK, PF>), dim3((unsigned)(grid + la.nblocks)),
                       dim3(256), 0, st, A, na, B, nb, d, S, ldS, tiles_m, tiles_n, (int)grid, la, GatherArgs{nullptr, 0});
    MI_HIP(hipGetLastError());
}

// S[q][c] = <A[q], B[idx[q][c]]> for c < kc: one 16x64 tile row per query (gather mode);
// B rows f32 or IEEE half (widened exactly, same f32 chain)
template <typename TB>
void launch_gemm_gather(const float *A, int nq, const TB *B, int64_t nb, int d, const int64_t *idx, int kc,
                        float *S, int64_t ldS, hipStream_t st) {
    MI_REQUIRE(d % 4 == 0 && nq > 0 && kc > 0 && nb > 0, "bad gather gemm arguments");
    constexpr int BN = 64;
    const int tiles_m = nq, tiles_n = (kc + BN - 1) / BN;
    const int64_t grid = (int64_t)8 * tiles_m * ((tiles_n + 7) / 8);
    MI_REQUIRE(grid < (int64_t)1 << 31, "ip_gemm: grid too large");
    hipLaunchKernelGGL((ip_gemm_kernel<1, 1, 1, 4, 64, 4, TB>), dim3((unsigned)grid), dim3(256), 0, st, A, nq, B,
                       (int)std::min<int64_t>(nb, INT32_MAX), d, S, ldS, tiles_m, tiles_n, (int)grid, LutArgs{},
                       GatherArgs{idx, kc});
    MI_HIP(hipGetLastError());
}

// exact scores of kc candidate rows per query: the streaming kernel when the rows are whole
// 128-byte pieces, else the gather mode of the score GEMM
template <typename TB>
void launch_rerank_scores(const float *q, int nq, const TB *base, int64_t nb, int d, const int64_t *idx, int kc,
                          float *S, int64_t ldS, hipStream_t st) {
    const int tiles = (kc + 63) / 64;
    if (((size_t)d * sizeof(TB)) % 128 == 0 && (int64_t)nq * tiles < ((int64_t)1 << 31)) {
        // a ring of 3 stages: 24.5 KiB of LDS per wave, 6 waves per CU, 2 x 8 KiB each in flight (3 > 4 > 6 > 8 stages measured)
        const unsigned grid = (unsigned)((int64_t)nq * tiles);
        hipLaunchKernelGGL((rerank_rows_kernel<TB, 3>), dim3(grid), dim3(64), 0, st, q, base, nb, d, idx, kc, S, ldS, tiles);
        MI_HIP(hipGetLastError());
        return;
    }
    launch_gemm_gather<TB>(q, nq, base, nb, d, idx, kc, S, ldS, st);
}

#ifndef MI_RERANK_SQ8_NST
#define MI_RERANK_SQ8_NST 3                 // ring stages of rerank_sq8_kernel (2 and 4 measured: profiles/r06_rerank_sq8_stages_ab.txt)
#endif
// the same over a QT_8bit store: the per-query table of the asymmetric score (w | A, in `tab`), then rerank_sq8_kernel (rows
// that are not whole 128-byte pieces: one thread per candidate)
void launch_rerank_sq8(const float *q, int nq, const uint8_t *base, int64_t nb, int d, const float *trained,
                       const int64_t *idx, int kc, float *S, int64_t ldS, DevBuf &tab, hipStream_t st) {
    float *wq = tab.as<float>((size_t)nq * d + (size_t)nq);
    float *Aq = wq + (size_t)nq * d;
    MI_REQUIRE((size_t)d * 8 <= 64 * 1024, "SQ8 re-rank: d too large for the query-table kernel's LDS row");
    hipLaunchKernelGGL(sq8_query_table_kernel, dim3((unsigned)nq), dim3(64), (size_t)d * 8, st, q, (int64_t)nq, d, trained, wq, Aq);
    MI_HIP(hipGetLastError());
    const int tiles = (kc + 63) / 64;
    if (d % 128 == 0 && (int64_t)nq * tiles < ((int64_t)1 << 31)) {
        const unsigned grid = (unsigned)((int64_t)nq * tiles);
        hipLaunchKernelGGL((rerank_sq8_kernel<MI_RERANK_SQ8_NST>), dim3(grid), dim3(64), 0, st, wq, Aq, base, nb, d, idx, kc, S, ldS, tiles);
        MI_HIP(hipGetLastError());
        return;
    }
    const int64_t total = (int64_t)nq * kc;
    hipLaunchKernelGGL(rerank_sq8_simple_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wq, Aq, base, nb, d, idx, kc,
                       total, S, ldS);
    MI_HIP(hipGetLastError());
}

// S[na][nb] = A . B^T (exact f32).  Tile shape by the number of A rows: small
// query batches get many small tiles (one 16x16 MFMA tile per wave, so that a
// 64 x 4096 problem still fills all 1024 SIMDs), big batches get 128x128.
void launch_gemm(const float *A, int64_t na, const float *B, int64_t nb, int d, float *S,
                 int64_t ldS, hipStream_t st, const LutArgs &la = LutArgs{}) {
    MI_REQUIRE(d % 4 == 0, "d must be a multiple of 4");
    MI_REQUIRE(na > 0 && nb > 0, "empty gemm");
    MI_REQUIRE(na < ((int64_t)1 << 31) && nb < ((int64_t)1 << 31), "gemm dims exceed int32");
    if (na <= 16) launch_gemm_cfg<1, 1, 1, 4, 64, 4>(A, (int)na, B, (int)nb, d, S, ldS, st, la);
    else if (na <= 128) launch_gemm_cfg<1, 1, 2, 2, 64, 4>(A, (int)na, B, (int)nb, d, S, ldS, st, la);
    else if (na <= 512) launch_gemm_cfg<2, 2, 2, 2, 32, 2>(A, (int)na, B, (int)nb, d, S, ldS, st, la);
    else launch_gemm_cfg<4, 4, 2, 2, 16, 2>(A, (int)na, B, (int)nb, d, S, ldS, st, la);   // 134 TF at 1024x65536x1024 (BK 32 / PF 1: 123)
}

void launch_select(const float *S, int64_t ldS, int64_t rows, int n, int K, int32_t *oi32,
                   int64_t *oi64, float *os, hipStream_t st, ProbeTables pt = ProbeTables{}, int idx_off = 0, DevBuf *scratch = nullptr) {
    MI_REQUIRE(K >= 1, "select: K < 1");
    // A few long rows (one query against 65 536 centroids): one workgroup per row is a chain of n / 4096 tiles (37 us at 65 536);
    // sliced, every 4096-column slice is a workgroup whose row stays in registers, and a second launch picks the K best of the
    // slices' results (SelSlices; same total order: the result is bit-identical) -- 37 -> ~14 us for one query.
    constexpr int SL = 4096;
    if (scratch && !knobs().no_sliced_select && rows <= 64 && K <= 64 && K < knobs().select_big_from && ldS == n && n % SL == 0 && n / SL >= 4 &&
        n / SL <= 64 && !oi64 && oi32) {
        const int nsl = n / SL;
        const size_t cnt = (size_t)rows * nsl * K;
        float *ss = scratch->as<float>(cnt * 2);
        int32_t *si = reinterpret_cast<int32_t *>(ss + cnt);
        hipLaunchKernelGGL(select_kernel, dim3((unsigned)(rows * nsl)), dim3(256), 0, st, S, (int64_t)SL, SL, K, si, (int64_t *)nullptr, ss,
                           ProbeTables{}, idx_off, SelSlices{nsl, nullptr});
        hipLaunchKernelGGL(select_kernel, dim3((unsigned)rows), dim3(256), 0, st, ss, (int64_t)nsl * K, nsl * K, K, oi32, (int64_t *)nullptr, os, pt,
                           0, SelSlices{0, si});
        MI_HIP(hipGetLastError());
        return;
    }
    // 64 < K <= 4096: threshold + bitonic sort (K = 256 of 4096: 16 us; the rank-counting path of
    // select_kernel 55 us, its insertion path 764 us at K = 1024).  MI_SELECT_BIG_FROM moves the border.
    const int big_from = knobs().select_big_from;
    if (K >= big_from && K <= SELB_CAP) {
        hipLaunchKernelGGL(select_big_kernel, dim3((unsigned)rows), dim3(256), 0, st, S, ldS, n, K, oi32,
                           oi64, os, pt, idx_off);
        MI_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)rows), dim3(256), 0, st, S, ldS, n, K, oi32,
                       oi64, os, pt, idx_off, SelSlices{0, nullptr});
    MI_HIP(hipGetLastError());
}

void launch_to_f16_rows(const float *x, int64_t rows, int d, f16_t *y, float *scale_out, float scale_all,
                        hipStream_t st) {
    hipLaunchKernelGGL(to_f16_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, (int)rows, d, y,
                       scale_out, scale_all);
    MI_HIP(hipGetLastError());
}

template <int WMT, int WNT, int WAVES_M, int WAVES_N, int ST>
void launch_ring_f16(const f16_t *A, int64_t na, const f16_t *B, int64_t nb, int K, float *S, int64_t ldS, hipStream_t st) {
    constexpr int BM = 16 * WMT * WAVES_M, BN = 16 * WNT * WAVES_N;
    mienc::GemmArgs g{};
    g.A = reinterpret_cast<const mienc::bf16_t *>(A);
    g.W = reinterpret_cast<const mienc::bf16_t *>(B);
    g.lda = K; g.ldw = K; g.M = (int)na; g.N = (int)nb; g.K = K;
    g.X = S; g.ldc = (int)ldS;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    g.ksplit = 1; g.tail_first = 0; g.tail_split = 1;
    const int per = (g.tiles_m * g.tiles_n + 7) / 8;
    hipLaunchKernelGGL((mienc::gemm_bf16_ring_kernel<mienc::EPI_F32H, WMT, WNT, WAVES_M, WAVES_N, ST>), dim3(8u * per),
                       dim3(64 * WAVES_M * WAVES_N), 0, st, g);
    MI_HIP(hipGetLastError());
}

// the encoder's slab kernel (256 x 256 tiles, hand-ordered K loop, staged whole-line f32 stores), f16 instantiation
void launch_slab_f16(const f16_t *A, int64_t na, const f16_t *B, int64_t nb, int K, float *S, int64_t ldS, hipStream_t st,
                     float *gmax = nullptr, int ld_gmax = 0) {
    mienc::GemmArgs g{};
    // MI_REFINE_TS=1 (tools/micro/coarse_stage.py): the tiles' in-kernel phase stamps, mean over the workgroups
    DevBuf tsb;
    const unsigned nwg = 8u * (unsigned)((((na + 255) / 256) * ((nb + 255) / 256) + 7) / 8);
    if (knobs().refine_ts) {
        g.ts = tsb.as<unsigned long long>((size_t)nwg * 8);
        g.ts_rows = nwg;
        MI_HIP(hipMemsetAsync(g.ts, 0, (size_t)nwg * 64, st));
    }
    g.A = reinterpret_cast<const mienc::bf16_t *>(A);
    g.W = reinterpret_cast<const mienc::bf16_t *>(B);
    g.lda = K; g.ldw = K; g.M = (int)na; g.N = (int)nb; g.K = K;
    g.X = S; g.ldc = (int)ldS;
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 255) / 256;
    g.ksplit = 1; g.tail_first = 0; g.tail_split = 1;
    g.gmax = gmax; g.ld_gmax = ld_gmax;
    const int per = (g.tiles_m * g.tiles_n + 7) / 8;
    hipLaunchKernelGGL((mienc::gemm_bf16_slab_kernel<mienc::EPI_F32H, 4>), dim3(8u * per), dim3(512), 0, st, g);
    MI_HIP(hipGetLastError());
    if (g.ts) {
        std::vector<unsigned long long> h((size_t)nwg * 8);
        MI_HIP(hipStreamSynchronize(st));
        MI_HIP(hipMemcpy(h.data(), g.ts, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[5] = {0}; size_t cnt = 0;
        for (unsigned b = 0; b < nwg; ++b)
            if (h[(size_t)b * 8 + 4]) { for (int i = 1; i <= 4; ++i) sum[i] += (double)h[(size_t)b * 8 + i] - 1; ++cnt; }
        if (cnt) std::fprintf(stderr, "f16 slab GEMM stamps (s_memtime ticks since the workgroup's start, mean of %zu tiles; %lld x %lld x %d): first slabs landed %.0f | "
                              "K loop done %.0f | epilogue issued %.0f | stores acknowledged %.0f\n", cnt, (long long)na, (long long)nb, K,
                              sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt);
    }
}

// gmax (optional, [na][nb / 64]): filled with the 64-column group maxima when the slab kernel runs; returns whether it was
bool launch_gemm_f16(const f16_t *A, int64_t na, const f16_t *B, int64_t nb, int K, float *S, int64_t ldS,
                     hipStream_t st, float *gmax = nullptr) {
    MI_REQUIRE(K % 64 == 0 && ldS % 4 == 0 && ldS < ((int64_t)1 << 31), "f16 gemm: K % 64, ldS % 4, ldS < 2^31");
    // the encoder's GEMM kernels, f16 instantiation: 256 x 256 slab tiles from 512 tiles up, 128 x 128 ring tiles below
    const int64_t tiles_big = ((na + 255) / 256) * ((nb + 255) / 256);
    if (tiles_big >= 512) {
        const bool with_gmax = gmax && nb % 64 == 0;
        launch_slab_f16(A, na, B, nb, K, S, ldS, st, with_gmax ? gmax : nullptr, (int)(nb / 64));
        return with_gmax;
    }
    launch_ring_f16<4, 4, 2, 2, 4>(A, na, B, nb, K, S, ldS, st);
    return false;
}

// ---- two-stage coarse quantiser (select_refine_kernel): shared by search, add and k-means ----
// f16 image of a centroid matrix + the two numbers the error bound needs
void prepare_cent16(const float *c, int64_t nc, int d, DevBuf &c16, DevBuf &stat, float &cmax, float &cscale,
                    hipStream_t st) {
    unsigned *cm = static_cast<unsigned *>(stat.reserve(8));
    MI_HIP(hipMemsetAsync(cm, 0, 8, st));
    hipLaunchKernelGGL(max_row_norm_kernel, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, st, c, (int)nc, d, cm);
    MI_HIP(hipGetLastError());
    unsigned bits[2] = {0, 0};
    MI_HIP(hipMemcpyAsync(bits, cm, 8, hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    float maxabs = 0.f;
    std::memcpy(&cmax, &bits[0], 4);
    std::memcpy(&maxabs, &bits[1], 4);
    cscale = 1.f;
    if (maxabs > 0.f && maxabs <= 3.0e38f) {
        int e = 0;
        (void)std::frexp(maxabs, &e);
        cscale = std::ldexp(1.f, std::min(std::max(14 - e, -100), 100));
    }
    launch_to_f16_rows(c, nc, d, static_cast<f16_t *>(c16.reserve((size_t)nc * d * 2)), nullptr, cscale, st);
}

// Large batches only: below, the exact GEMM is latency-bound and the second stage costs more
// than it saves.  MI_TWO_STAGE=0 / 1 forces it off / on (tests).
bool two_stage_wanted(int64_t nq, int64_t nc, int d, int K) {
    bool on = nq >= 256 && nc >= 8192 && nq * nc >= ((int64_t)1 << 24) && (K <= 128 || nq * nc >= ((int64_t)1 << 26));   // (1024 x 65536, K 256: 2.68 -> 2.16 ms)
    // the 8 192-centroid slice a rank of the 8-GPU job quantises against (sharded coarse quantiser, batch 1024): 0.181 -> 0.139 ms,
    // the rank's rehearsed step 0.536 -> 0.478 ms (bench.py --emulate-rank-of 8, MI_TWO_STAGE unset / 1 alternating)
    on = on || (nq >= 1024 && nc >= 8192 && K <= 128);
    if (knobs().two_stage >= 0) on = knobs().two_stage != 0;
    return on && d % 128 == 0 && d <= 4096 && nc % 4 == 0 && K <= 1024 && nc < ((int64_t)1 << 31);
}

// best K of q [nq][d] against c32 [nc][d]: approximate scores into `scores` [nq][nc], then the
// exact second stage; outputs as launch_select's
void launch_two_stage(const float *q, int64_t nq, const float *c32, const f16_t *c16, int64_t nc, int d, int K,
                      float cmax, float cscale, float *scores, DevBuf &q16buf, DevBuf &qscalebuf, DevBuf &statbuf,
                      int32_t *out_i32, float *out_s, ProbeTables pt, hipStream_t st, int idx_off = 0) {
    f16_t *q16 = static_cast<f16_t *>(q16buf.reserve((size_t)nq * d * 2));
    // behind the per-row scales: the 64-column group maxima of the approximate scores (MI_REFINE_GMAX=0: without)
    const bool want_gmax = nc % 64 == 0 && nc / 64 <= 1024 && knobs().refine_gmax;
    const size_t qs_bytes = (((size_t)nq * 4 + 255) / 256) * 256;
    float *qscale = static_cast<float *>(qscalebuf.reserve(qs_bytes + (want_gmax ? (size_t)nq * (nc / 64) * 4 : 0)));
    float *gmax = want_gmax ? reinterpret_cast<float *>(reinterpret_cast<char *>(qscale) + qs_bytes) : nullptr;
    launch_to_f16_rows(q, nq, d, q16, qscale, 0.f, st);
    const bool have_gmax = launch_gemm_f16(q16, nq, c16, nc, d, scores, nc, st, gmax);
    RefineArgs ra{};
    ra.gmax = have_gmax ? gmax : nullptr; ra.ngroups = (int)(nc / 64);
    ra.q = q; ra.cent = c32; ra.Sa = scores; ra.ldS = nc;
    ra.n = (int)nc; ra.d = d; ra.K = K;
    // |approx - exact| <= eps_rel |q| |c|  (derivation: ivfpq_kernels.h); the 1 % covers every
    // second-order term (products of the relative errors, d u / (1 - d u))
    ra.eps_rel = (0x1p-10f + (float)d * (0x1p-22f + 0x1p-24f) + 0x1p-26f * std::sqrt((float)d)) * 1.01f;
    ra.qscale = qscale; ra.cscale = cscale; ra.cmax = cmax;
    ra.out_i32 = out_i32; ra.out_s = out_s; ra.pt = pt; ra.idx_off = idx_off;
    const bool want_stats = knobs().refine_stats;
    if (want_stats) {
        ra.stats = static_cast<unsigned *>(statbuf.reserve(8));
        MI_HIP(hipMemsetAsync(ra.stats, 0, 8, st));
    }
    DevBuf tsbuf;
    if (knobs().refine_ts) {
        ra.ts = static_cast<unsigned long long *>(tsbuf.reserve((size_t)nq * 64));
        MI_HIP(hipMemsetAsync(ra.ts, 0, (size_t)nq * 64, st));
    }
    if (d <= 1024) hipLaunchKernelGGL(select_refine_kernel<1024>, dim3((unsigned)nq), dim3(256), 0, st, ra);
    else hipLaunchKernelGGL(select_refine_kernel<4096>, dim3((unsigned)nq), dim3(256), 0, st, ra);
    MI_HIP(hipGetLastError());
    if (ra.ts) {
        std::vector<unsigned long long> ts((size_t)nq * 8);
        MI_HIP(hipMemcpyAsync(ts.data(), ra.ts, ts.size() * 8, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        double sum[8] = {0};
        for (int64_t r = 0; r < nq; ++r)
            for (int i = 1; i < 8; ++i) sum[i] += (double)(ts[r * 8 + i] - ts[r * 8 + i - 1]);
        std::fprintf(stderr, "select_refine stamps (s_memtime ticks, mean of %lld workgroups; K %d): load+norm+maxima %.0f | descent %.0f | "
                     "group list %.0f | candidates %.0f | chains %.0f | sort %.0f | output+tables %.0f\n",
                     (long long)nq, K, sum[1] / nq, sum[2] / nq, sum[3] / nq, sum[4] / nq, sum[5] / nq, sum[6] / nq, sum[7] / nq);
    }
    if (want_stats) {
        unsigned hs[2] = {0, 0};
        MI_HIP(hipMemcpyAsync(hs, ra.stats, 8, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        std::fprintf(stderr, "two-stage coarse: %lld rows, K %d: %.2f candidates per row, %u exact-fallback rows\n",
                     (long long)nq, K, (double)hs[0] / (double)nq, hs[1]);
    }
}

LutArgs make_lut_args(const float *q, int nq, int d, int M, const float *cb, float *lut) {
    LutArgs a{};
    a.q = q; a.codebook = cb; a.lut = lut; a.nq = nq; a.d = d; a.M = M; a.dsub = d / M; a.qtile = 16;   // 16 queries per LUT workgroup: half the workgroups of 8 and +4 % QPS under stream overlap (32: single-stream latency suffers)
    a.nblocks = M * ((nq + a.qtile - 1) / a.qtile);
    return a;
}

void launch_lut(const float *q, int nq, int d, int M, const float *cb, float *lut, hipStream_t st) {
    const LutArgs a = make_lut_args(q, nq, d, M, cb, lut);
    dim3 grid(a.nblocks), block(256);
    switch (a.dsub) {
#define MI_LUT_CASE(DS)                                                      \
    case DS:                                                                 \
        hipLaunchKernelGGL((lut_kernel<DS>), grid, block, 0, st, a);         \
        break;
        MI_LUT_CASE(1)
        MI_LUT_CASE(2)
        MI_LUT_CASE(4)
        MI_LUT_CASE(8)
        MI_LUT_CASE(16)
        MI_LUT_CASE(32)
        MI_LUT_CASE(64)
#undef MI_LUT_CASE
        default:
            throw Error("unsupported d/M (sub-vector length must be 1,2,4,8,16,32 or 64)");
    }
    MI_HIP(hipGetLastError());
}

void launch_pq_encode(const float *x, int64_t n, int d, int M, const float *cb, const float *cent,
                      const int32_t *assign, uint8_t *codes, hipStream_t st) {
    const int dsub = d / M;
    dim3 grid((unsigned)((n + 255) / 256), M), block(256);
    switch (dsub) {
#define MI_ENC_CASE(DS)                                                                        \
    case DS:                                                                                   \
        hipLaunchKernelGGL((pq_encode_kernel<DS>), grid, block, 0, st, x, n, d, M, cb, cent, assign, \
                           codes);                                                             \
        break;
        MI_ENC_CASE(1)
        MI_ENC_CASE(2)
        MI_ENC_CASE(4)
        MI_ENC_CASE(8)
        MI_ENC_CASE(16)
        MI_ENC_CASE(32)
        MI_ENC_CASE(64)
#undef MI_ENC_CASE
        default:
            throw Error("unsupported d/M (sub-vector length must be 1,2,4,8,16,32 or 64)");
    }
    MI_HIP(hipGetLastError());
}

template <int M, int NW, bool ALL>
void launch_scan_mw(const ScanArgs &a, hipStream_t st) {
    const size_t smem = scan_smem_bytes(M, a.nprobe, NW);
    MI_REQUIRE(smem <= 160 * 1024, "nprobe too large for the LDS probe tables");
    set_max_dynamic_lds(reinterpret_cast<const void *>(&scan_kernel<M, NW, ALL>), 160 * 1024);
    hipLaunchKernelGGL((scan_kernel<M, NW, ALL>), dim3(scan_grid(a.nq, a.nslice)), dim3(NW * 64), smem,
                       st, a);
    MI_HIP(hipGetLastError());
}
template <int M, bool ALL>
void launch_scan_l2(const ScanArgs &a, hipStream_t st) {
    const size_t smem = scan_smem_bytes(M, a.nprobe, 8);
    MI_REQUIRE(smem <= 160 * 1024, "nprobe too large for the LDS probe tables");
    set_max_dynamic_lds(reinterpret_cast<const void *>(&scan_kernel<M, 8, ALL, true>), 160 * 1024);
    hipLaunchKernelGGL((scan_kernel<M, 8, ALL, true>), dim3(scan_grid(a.nq, a.nslice)), dim3(8 * 64), smem, st, a);
    MI_HIP(hipGetLastError());
}
template <int M>
void launch_scan_m(const ScanArgs &a, hipStream_t st) {
    if (a.tnorm) {   // METRIC_L2: 8 waves per workgroup only
        if (a.all_s) launch_scan_l2<M, true>(a, st);
        else launch_scan_l2<M, false>(a, st);
        return;
    }
    if (a.all_s) {
        if (a.nw == 16) launch_scan_mw<M, 16, true>(a, st);
        else launch_scan_mw<M, 8, true>(a, st);
    } else if (a.nw == 16) launch_scan_mw<M, 16, false>(a, st);
    else launch_scan_mw<M, 8, false>(a, st);
}

bool scan_supports_M(int M) {
    switch (M) {
        case 4: case 8: case 16: case 32: case 48: case 64: case 96: case 128: return true;
        default: return false;
    }
}

void launch_scan(int M, const ScanArgs &a, hipStream_t st) {
    switch (M) {
        case 4: launch_scan_m<4>(a, st); break;
        case 8: launch_scan_m<8>(a, st); break;
        case 16: launch_scan_m<16>(a, st); break;
        case 32: launch_scan_m<32>(a, st); break;
        case 48: launch_scan_m<48>(a, st); break;
        case 64: launch_scan_m<64>(a, st); break;
        case 96: launch_scan_m<96>(a, st); break;
        case 128: launch_scan_m<128>(a, st); break;
        default: throw Error("unsupported M (PQ sub-quantisers: 4,8,16,32,48,64,96,128)");
    }
}

// the K best pairs of every row (K <= 4096: 48 KiB of LDS per workgroup; K <= 8192: 96 KiB)
// as_set: ids only (D may be null), in no particular order -- what the first stage of a refine search needs
// IDS null: the ids of the survivors come from the lists through the probe tables (p_goff [rows][nprobe], list_ids = the scan
// image's id array) -- the all-scores scan then stores scores only
void launch_select_pairs(const float *S, const int64_t *IDS, int64_t ld, const int32_t *p_prefix, int nprobe, int K, int64_t rows,
                         float *D, int64_t *I, int64_t ldo, hipStream_t st, bool as_set = false, const int32_t *p_goff = nullptr,
                         const int64_t *list_ids = nullptr) {
    MI_REQUIRE(K <= SELP_CAP, "select_pairs: K too large (max 8192)");
    MI_REQUIRE(as_set || D, "select_pairs: scores requested without a buffer");
    MI_REQUIRE(IDS || (p_goff && list_ids), "select_pairs: neither an id row nor the probe tables");
    const dim3 grid((unsigned)rows), block(256);
    if (as_set) {
        // (rows longer than the 4096-slot kernel keeps resident -- 4 tiles of 4096 -- go to the 8192-slot one whatever K: it holds
        //  32 k scores in registers; streamed, a 25.6 k row took 591 us where the resident pass takes ~210)
        if (K <= SELB_CAP && ld <= 16384) hipLaunchKernelGGL((select_pairs_kernel<16, true>), grid, block, 0, st, S, IDS, ld, p_prefix, nprobe, K, D, I, ldo, p_goff, list_ids);
        else hipLaunchKernelGGL((select_pairs_kernel<16, true, 512>), grid, dim3(512), 0, st, S, IDS, ld, p_prefix, nprobe, K, D, I, ldo, p_goff, list_ids);
    } else {
        if (K <= SELB_CAP) hipLaunchKernelGGL((select_pairs_kernel<16, false>), grid, block, 0, st, S, IDS, ld, p_prefix, nprobe, K, D, I, ldo, p_goff, list_ids);
        else hipLaunchKernelGGL((select_pairs_kernel<32, false>), grid, block, 0, st, S, IDS, ld, p_prefix, nprobe, K, D, I, ldo, p_goff, list_ids);
    }
    MI_HIP(hipGetLastError());
}

// k-way merge of nparts candidate lists per query.  Three tiers by the LDS one query needs
// (12 B per candidate + 16 B per result): several queries per workgroup in the default 64 KiB;
// one query per workgroup in up to 160 KiB (gfx950's LDS per CU); beyond that -- and for long candidate
// lists whatever their LDS need -- the candidates are laid out as one row of pairs per query in `scratch`
// and reduced by select_pairs_kernel.  8 shards x k = 4096, or re-ranking k * k_factor = 50 000 candidates, all work.
void launch_merge(const float *ps, const int64_t *pid, int nparts, int64_t stride_p,
                  int64_t stride_q, int64_t nq, int k, float *D, int64_t *I, int64_t ldo,
                  int out_off, float *bs, int64_t *bid, hipStream_t st, int64_t stride_p_id = -1,
                  IdMap im = IdMap{}, DevBuf *scratch = nullptr) {
    if (stride_p_id < 0) stride_p_id = stride_p;
    const size_t per_wave = merge_wave_bytes(nparts, k);
    // the counting tiers rank every candidate against every other one -- (nparts k)^2 / 64 steps per lane: a re-rank of 5 120
    // candidates took 0.77 ms of a 4.6 ms step there -- so long candidate lists go to the selection tier (linear in nparts k)
    constexpr int64_t big_from = 2048;
    // (only for callers that bring a kept scratch buffer: without one the tier would hipMalloc + synchronise + hipFree on every
    // call -- a single query's cross-slice merge must stay enqueue-only)
    const bool by_selection = scratch && !bs && !bid && (int64_t)nparts * k >= big_from && k <= SELB_CAP;
    if (per_wave <= 64 * 1024 && !by_selection) {
        int qpb = (int)std::min<size_t>(4, (64 * 1024) / per_wave);   // queries (waves) per workgroup
        if (qpb == 3) qpb = 2;
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)((nq + qpb - 1) / qpb)), dim3(64 * qpb),
                           per_wave * qpb, st, ps, pid, nparts, stride_p, stride_p_id, stride_q, nq, k, D, I, ldo,
                           out_off, bs, bid, im);
        MI_HIP(hipGetLastError());
        return;
    }
    if (per_wave <= 160 * 1024 && !by_selection) {
        set_max_dynamic_lds(reinterpret_cast<const void *>(&merge_kernel), 160 * 1024);
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(64), per_wave, st, ps, pid, nparts, stride_p,
                           stride_p_id, stride_q, nq, k, D, I, ldo, out_off, bs, bid, im);
        MI_HIP(hipGetLastError());
        return;
    }
    MI_REQUIRE(k <= SELP_CAP, "merge: k too large (max 8192)");
    MI_REQUIRE(!bs && !bid, "internal: bounded merge on the large path");
    const int64_t ld = (((int64_t)nparts * k + 63) / 64) * 64;
    DevBuf local;
    DevBuf &sb = scratch ? *scratch : local;
    // queries in sub-batches so that the rows stay under 1 GiB
    const int64_t qc = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)1 << 30) / (ld * 12)));
    char *base = static_cast<char *>(sb.reserve((size_t)qc * ld * 12 + (size_t)qc * 8 + 64));
    int64_t *rid = reinterpret_cast<int64_t *>(base);
    float *rs = reinterpret_cast<float *>(base + (size_t)qc * ld * 8);
    int32_t *prefix = reinterpret_cast<int32_t *>(base + (size_t)qc * ld * 12);
    for (int64_t c0 = 0; c0 < nq; c0 += qc) {
        const int64_t m = std::min(qc, nq - c0);
        hipLaunchKernelGGL(gather_parts_kernel, dim3((unsigned)m), dim3(256), 0, st, ps + (size_t)c0 * stride_q,
                           pid + (size_t)c0 * stride_q, nparts, stride_p, stride_p_id, stride_q, k, ld, rs, rid, prefix, im);
        launch_select_pairs(rs, rid, ld, prefix, 1, k, m, D + (size_t)c0 * ldo + out_off, I + (size_t)c0 * ldo + out_off, ldo, st);
        MI_HIP(hipGetLastError());
    }
    if (!scratch) MI_HIP(hipStreamSynchronize(st));   // `local` dies with this scope
}

// copy `bytes` to the device if `src` is a host pointer; returns a device pointer.
const void *to_device(const void *src, size_t bytes, DevBuf &stage, hipStream_t st) {
    if (is_device_ptr(src)) return src;
    void *dst = stage.reserve(bytes);
    MI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return dst;
}

}  // namespace

// ------------------------------------------------------------------
// handles
// ------------------------------------------------------------------

// per-stream search workspaces (see mi_index::ws_sets)
struct SearchWS {
    std::mutex mu;   // held by the one call that is enqueuing work through this set (two host threads on one stream take turns)
    DevBuf selsc;   // the sliced selection's per-slice results (launch_select)
    DevBuf q, scores, cidx, cdis, lut, ps, pid, bs, bid, D, I, pgoff, plen, pprefix, counters, all_s, all_id, q16, qscale, rstats, qaug, qn, cscan;
    DevBuf plen1, ppre1, plen2, ppre2, comb_s, comb_id, pruneA;   // exact list pruning: the two phases' probe tables, their results side by side; the early stop's A[q]
    size_t counters_zeroed = 0;  // bytes of `counters` known to be zero
    // most recent scan launch on this stream (mi_index_profile_scan replays it)
    ScanArgs last_scan{};
    bool have_last_scan = false;
    int64_t last_nq = 0;
    int last_nprobe = 0;
};

struct mi_index {
    int d = 0, nlist = 0, M = 0, dsub = 0, metric = 0, by_residual = 1, device = 0;
    int nprobe_attr = 1;   // faiss's index.nprobe as stored in / restored from index files (search takes nprobe per call)
    bool has_coarse = false, has_codebook = false;
    DevBuf centroids, codebook;
    // two-stage coarse quantiser: f16 copy of the centroids, their largest norm
    DevBuf cent16, cmax_dev;
    float cmax = 0.f, cscale = 1.f;
    bool cent16_ok = false;
    // master copy of the inverted lists: an append log in HBM, insertion order (codes row-major
    // [n][M], list number, slot inside the list, id) -- see "Inverted-list maintenance" in
    // ivfpq_kernels.h.  d_cnt[l] = current length of list l; h_len mirrors it on the host when
    // len_ok.  Nothing proportional to ntotal lives on the host.
    DevBuf log_codes, log_list, log_pos, log_ids, d_cnt;
    // METRIC_L2 (oracle: "METRIC_L2" section): centroids augmented to `da` columns
    // [c, -|c|^2/2, 0..] for the coarse quantiser; per-vector term t in the log and in the image
    DevBuf cent_aug, log_t, d_tnorm, ws_xaug;
    int da = 0;
    int64_t log_cap = 0;
    bool sealed = false;   // mi_index_seal: the log is freed (80 B per vector); rebuilt from the image by the next call that needs it
    std::vector<int32_t> h_len;
    bool len_ok = true;
    int64_t ntotal = 0;
    bool dirty = true;
    // device image (group-interleaved, see ivfpq_kernels.h)
    DevBuf d_codes, d_ids, d_goff, d_len;
    int64_t ngroups = 0;
    // code groups of the cap_nprobe longest lists (row capacity of the all-scores path)
    int cap_nprobe = -1;
    int64_t cap_groups = 0;
    // search workspaces: one set per stream the index is searched on, so that
    // batches issued on different streams overlap on the GPU (a serving loop
    // round-robins 2-4 streams; each kernel of one batch leaves most CUs idle)
    std::vector<std::pair<void *, std::shared_ptr<SearchWS>>> ws_sets;   // shared: a leased set outlives release_workspaces
    // Concurrent readers (mi_ivfpq.h "Threading"): `mu` guards the list of workspace sets and every piece of state a
    // search builds lazily (the scan image after an add, the f16 centroid image, the all-scores row capacity); the
    // arithmetic of a search only reads the index.
    std::mutex mu;
    // add()/encode() workspaces
    DevBuf ws_scores, ws_x, ws_assign, ws_codes, ws_ids, ws_count, ws_x16, ws_xscale, ws_rstats;
    // exact list pruning (search_chunk): {groups scanned, groups of all probed lists, queries} summed over the pruned calls
    // The device counters only grow; mi_index_prune_stats reports them against `prune_base` (its reset moves the base).  A copy
    // lands in pinned host memory every few pruned calls, asynchronously: the NEXT calls read from it whether this index's data
    // prunes at all -- when most of the probed groups are scanned anyway the early stop keeps the balanced slices of the
    // exhaustive launch instead of one workgroup per query (no synchronisation on a search path; a stale or torn read only
    // picks the other, equally exact, launch shape).
    DevBuf prune_stats;
    bool prune_stats_ok = false;
    unsigned long long prune_base[3] = {0, 0, 0};
    unsigned long long *prune_seen = nullptr;   // pinned, [PRUNE_SLOTS][3]
    std::atomic<unsigned> prune_calls{0};

    int nch() const { return (M + 15) / 16; }
};

struct mi_flat {
    int d = 0, device = 0;
    int elem = 4;          // bytes per stored component: 4 = f32 (IndexFlat), 2 = IEEE half (IndexScalarQuantizer QT_fp16),
                           // 1 = QT_8bit codes with per-dimension ranges in `sq_trained`
    DevBuf sq_trained, sq_lohi;   // QT_8bit: vmin | vdiff (what the kernels read); min | max as trained so far
    bool sq_ok = false;
    int metric = MI_METRIC_INNER_PRODUCT;
    int da = 0;            // stored row width: d, or d + 4 for METRIC_L2 ([x, -|x|^2/2, 0, 0, 0]: oracle flat_l2)
    int64_t ntotal = 0;
    DevBuf base;
    DevBuf ws_q, ws_scores;   // add() / reconstruct_n() staging (exclusive calls)
    // search / re-rank workspaces, one set per stream a call is issued on: batches on different streams overlap on the
    // GPU, and concurrent host threads (one stream each) never share a buffer
    struct WS {
        std::mutex mu;
        DevBuf q, cand, scores, D, I, qaug, qn, bigmerge, rerank;
    };
    std::vector<std::pair<void *, std::shared_ptr<WS>>> ws_sets;
    std::mutex mu;            // guards ws_sets
};

namespace {

constexpr size_t MAX_STREAMS = 64;   // distinct streams one handle keeps workspaces for

// caller holds h->mu
std::shared_ptr<SearchWS> ws_for(mi_index *h, void *stream) {
    for (auto &kv : h->ws_sets)
        if (kv.first == stream) return kv.second;
    MI_REQUIRE(h->ws_sets.size() < MAX_STREAMS, "too many distinct streams on one index handle (max 64)");
    h->ws_sets.emplace_back(stream, std::make_shared<SearchWS>());
    return h->ws_sets.back().second;
}

void sync_lists(mi_index *h);

// The workspace set of `stream`, leased to the calling thread until the lease dies.  Threads on distinct streams run
// concurrently; threads that share a stream take turns enqueuing (the GPU serialises their work anyway).  `sync`: also
// bring the scan image up to date (the first search after an add builds it, once, under the handle lock).
// The lease owns a reference: mi_index_release_workspaces between the look-up and the lock drops the handle's reference only,
// the set lives until the call that leased it returns.
struct WsLease {
    std::shared_ptr<SearchWS> keep;
    SearchWS &w;
    std::unique_lock<std::mutex> lk;
};
WsLease lease_ws(mi_index *h, void *stream, bool sync) {
    std::shared_ptr<SearchWS> w;
    {
        std::lock_guard<std::mutex> hl(h->mu);
        w = ws_for(h, stream);
        if (sync) sync_lists(h);
    }
    SearchWS &r = *w;
    return WsLease{std::move(w), r, std::unique_lock<std::mutex>(r.mu)};
}

struct FlatLease {
    std::shared_ptr<mi_flat::WS> keep;
    mi_flat::WS &w;
    std::unique_lock<std::mutex> lk;
};
FlatLease lease_ws(mi_flat *h, void *stream) {
    std::shared_ptr<mi_flat::WS> w;
    {
        std::lock_guard<std::mutex> hl(h->mu);
        for (auto &kv : h->ws_sets)
            if (kv.first == stream) w = kv.second;
        if (!w) {
            MI_REQUIRE(h->ws_sets.size() < MAX_STREAMS, "too many distinct streams on one flat index handle (max 64)");
            h->ws_sets.emplace_back(stream, std::make_shared<mi_flat::WS>());
            w = h->ws_sets.back().second;
        }
    }
    mi_flat::WS &r = *w;
    return FlatLease{std::move(w), r, std::unique_lock<std::mutex>(r.mu)};
}

void require_trained(mi_index *h) {
    MI_REQUIRE(h->has_coarse && h->has_codebook, "index is not trained (set_coarse/set_codebook)");
}

// host mirror of the list lengths
void refresh_len(mi_index *h) {
    if (h->len_ok) return;
    h->h_len.resize((size_t)h->nlist);
    MI_HIP(hipMemcpy(h->h_len.data(), h->d_cnt.p, (size_t)h->nlist * 4, hipMemcpyDeviceToHost));
    h->len_ok = true;
}

void refresh_len(mi_index *h);

// A sealed index (mi_index_seal) holds its lists in the scan image only; whoever needs the log again -- add, export, save --
// gets it back from the image, in list order (the slots inside a list, i.e. every list's insertion order, are the image's).
void unseal(mi_index *h) {
    if (!h->sealed) return;
    h->sealed = false;
    if (h->ntotal == 0) return;
    refresh_len(h);
    const int64_t n = h->ntotal;
    h->log_cap = n;
    uint8_t *lc = static_cast<uint8_t *>(h->log_codes.reserve((size_t)n * h->M));
    int32_t *ll = static_cast<int32_t *>(h->log_list.reserve((size_t)n * 4));
    int32_t *lp = static_cast<int32_t *>(h->log_pos.reserve((size_t)n * 4));
    int64_t *li = static_cast<int64_t *>(h->log_ids.reserve((size_t)n * 8));
    float *lt = h->metric == MI_METRIC_L2 ? static_cast<float *>(h->log_t.reserve((size_t)n * 4)) : nullptr;
    std::vector<int64_t> start((size_t)h->nlist + 1, 0);
    for (int l = 0; l < h->nlist; ++l) start[(size_t)l + 1] = start[(size_t)l] + h->h_len[(size_t)l];
    MI_REQUIRE(start.back() == n, "unseal: list lengths do not add up to ntotal");
    DevBuf dstart;
    MI_HIP(hipMemcpyAsync(dstart.reserve(start.size() * 8), start.data(), start.size() * 8, hipMemcpyHostToDevice, nullptr));
    const int64_t nslots = h->ngroups * 64;
    if (nslots > 0) {
        hipLaunchKernelGGL(image_to_log_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, nullptr, h->d_codes.get<uint8_t>(),
                           h->d_ids.get<int64_t>(), lt ? h->d_tnorm.get<float>() : nullptr, nslots, h->d_goff.get<int32_t>(),
                           h->d_len.get<int32_t>(), dstart.get<int64_t>(), h->nlist, h->M, h->nch(), lc, ll, lp, li, lt);
        MI_HIP(hipGetLastError());
    }
    MI_HIP(hipStreamSynchronize(nullptr));
}

// room for `need` entries in the append log (grows by half, contents preserved)
void ensure_log_cap(mi_index *h, int64_t need) {
    unseal(h);
    if (need <= h->log_cap) return;
    const int64_t cap = std::max<int64_t>(need, std::max<int64_t>(h->log_cap + h->log_cap / 2, 4096));
    auto grow = [&](DevBuf &b, size_t elem) {
        DevBuf nb;
        nb.reserve((size_t)cap * elem);
        if (h->ntotal) MI_HIP(hipMemcpyAsync(nb.p, b.p, (size_t)h->ntotal * elem, hipMemcpyDeviceToDevice, nullptr));
        MI_HIP(hipStreamSynchronize(nullptr));
        std::swap(nb.p, b.p);
        std::swap(nb.cap, b.cap);
    };
    grow(h->log_codes, (size_t)h->M);
    grow(h->log_list, 4);
    grow(h->log_pos, 4);
    grow(h->log_ids, 8);
    if (h->metric == MI_METRIC_L2) grow(h->log_t, 4);
    h->log_cap = cap;
}

// pos[i] = slot of entry i inside its list (entries of one list in ascending i), cnt[l] += members;
// stream-ordered chunks of <= 65536 (list_rank_kernel).
void rank_and_count(const int32_t *list_no, int64_t n, int32_t *cnt, int32_t *pos, hipStream_t st) {
    for (int64_t c0 = 0; c0 < n; c0 += 65536) {
        const int m = (int)std::min<int64_t>(65536, n - c0);
        hipLaunchKernelGGL(list_rank_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, list_no + c0, m, cnt, pos + c0);
        hipLaunchKernelGGL(list_count_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, list_no + c0, m, cnt);
        MI_HIP(hipGetLastError());
    }
}

// Entries [ntotal, ntotal + n) of the log have codes / list numbers / ids in place: give them
// their slots (insertion order) and count them.
void commit_log_entries(mi_index *h, int64_t n, hipStream_t st) {
    if (h->metric == MI_METRIC_L2) {   // the per-vector term of the expansion, from the codes just written
        MI_REQUIRE(h->has_coarse && h->has_codebook, "METRIC_L2: the index must be trained before entries are added");
        hipLaunchKernelGGL(l2_term_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           h->log_codes.get<uint8_t>() + (size_t)h->ntotal * h->M, h->log_list.get<int32_t>() + h->ntotal, n,
                           h->d, h->M, h->codebook.get<float>(), h->centroids.get<float>(), h->by_residual,
                           h->log_t.get<float>() + h->ntotal);
        MI_HIP(hipGetLastError());
    }
    rank_and_count(h->log_list.get<int32_t>() + h->ntotal, n, h->d_cnt.get<int32_t>(), h->log_pos.get<int32_t>() + h->ntotal, st);
    h->ntotal += n;
    h->dirty = true;
    h->len_ok = false;
}

// rebuild the group-interleaved device image of the inverted lists from the log
void sync_lists(mi_index *h) {
    if (!h->dirty) return;
    refresh_len(h);
    const int nlist = h->nlist, M = h->M, NCH = h->nch();
    std::vector<int32_t> goff(nlist + 1, 0);
    int64_t g = 0;
    for (int l = 0; l < nlist; ++l) {
        goff[l] = (int32_t)g;
        g += ((int64_t)h->h_len[l] + 63) / 64;
        MI_REQUIRE(g * 64 < ((int64_t)1 << 32), "more than 2^32 padded codes on one device");
    }
    goff[nlist] = (int32_t)g;
    h->ngroups = g;
    h->cap_nprobe = -1;
    const size_t gbytes = (size_t)NCH * 1024;
    const size_t ng = (size_t)std::max<int64_t>(g, 1);
    uint8_t *img = static_cast<uint8_t *>(h->d_codes.reserve(ng * gbytes));
    int64_t *img_ids = static_cast<int64_t *>(h->d_ids.reserve(ng * 64 * 8));
    MI_HIP(hipMemsetAsync(img, 0, ng * gbytes, nullptr));
    MI_HIP(hipMemsetAsync(img_ids, 0xFF, ng * 64 * 8, nullptr));   // every id -1
    float *img_t = nullptr;
    if (h->metric == MI_METRIC_L2) {
        img_t = static_cast<float *>(h->d_tnorm.reserve(ng * 64 * 4));
        MI_HIP(hipMemsetAsync(img_t, 0, ng * 64 * 4, nullptr));
    }
    MI_HIP(hipMemcpyAsync(h->d_goff.reserve(goff.size() * 4), goff.data(), goff.size() * 4, hipMemcpyHostToDevice, nullptr));
    MI_HIP(hipMemcpyAsync(h->d_len.reserve((size_t)nlist * 4), h->d_cnt.p, (size_t)nlist * 4, hipMemcpyDeviceToDevice, nullptr));
    if (h->ntotal) {
        const int64_t threads = h->ntotal * NCH;
        hipLaunchKernelGGL(build_image_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr,
                           h->log_codes.get<uint8_t>(), h->log_list.get<int32_t>(), h->log_pos.get<int32_t>(),
                           h->log_ids.get<int64_t>(), h->ntotal, h->d_goff.get<int32_t>(), M, NCH, img, img_ids,
                           img_t ? h->log_t.get<float>() : nullptr, img_t);
        MI_HIP(hipGetLastError());
    }
    MI_HIP(hipStreamSynchronize(nullptr));   // searches run on other (non-blocking) streams
    h->dirty = false;
}

// coarse assign + PQ encode of n vectors (device pointer xdev) into assign [n] / codes [n][M]
// (device pointers: the add() path passes the tail of the append log).
void launch_augment(const float *x, int64_t n, int d, int da, int mode, float *out, hipStream_t st) {
    hipLaunchKernelGGL(augment_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, x, n, d, da, mode, out);
    MI_HIP(hipGetLastError());
}

void encode_chunk(mi_index *h, const float *xdev, int64_t n, int32_t *assign, uint8_t *codes, hipStream_t st) {
    float *scores = h->ws_scores.as<float>((size_t)n * h->nlist);
    if (h->metric == MI_METRIC_L2) {
        // arg min |x - c|^2 = arg max of the augmented inner product: the same kernels on `da` columns
        float *xa = h->ws_xaug.as<float>((size_t)n * h->da);
        launch_augment(xdev, n, h->d, h->da, 0, xa, st);
        if (two_stage_wanted(n, h->nlist, h->da, 1)) {
            launch_two_stage(xa, n, h->cent_aug.get<float>(), static_cast<const f16_t *>(h->cent16.p), h->nlist, h->da, 1,
                             h->cmax, h->cscale, scores, h->ws_x16, h->ws_xscale, h->ws_rstats, assign, nullptr,
                             ProbeTables{}, st);
        } else {
            launch_gemm(xa, n, h->cent_aug.get<float>(), h->nlist, h->da, scores, h->nlist, st);
            launch_select(scores, h->nlist, n, h->nlist, 1, assign, nullptr, nullptr, st);
        }
        launch_pq_encode(xdev, n, h->d, h->M, h->codebook.get<float>(),
                         h->by_residual ? h->centroids.get<float>() : nullptr, assign, codes, st);
        return;
    }
    if (two_stage_wanted(n, h->nlist, h->d, 1)) {
        if (!h->cent16_ok) {
            prepare_cent16(h->centroids.get<float>(), h->nlist, h->d, h->cent16, h->cmax_dev, h->cmax, h->cscale, st);
            h->cent16_ok = true;
        }
        launch_two_stage(xdev, n, h->centroids.get<float>(), static_cast<const f16_t *>(h->cent16.p), h->nlist, h->d, 1,
                         h->cmax, h->cscale, scores, h->ws_x16, h->ws_xscale, h->ws_rstats, assign, nullptr,
                         ProbeTables{}, st);
    } else {
        launch_gemm(xdev, n, h->centroids.get<float>(), h->nlist, h->d, scores, h->nlist, st);
        launch_select(scores, h->nlist, n, h->nlist, 1, assign, nullptr, nullptr, st);
    }
    launch_pq_encode(xdev, n, h->d, h->M, h->codebook.get<float>(),
                     h->by_residual ? h->centroids.get<float>() : nullptr, assign, codes, st);
}

int64_t add_chunk_size(const mi_index *h) {
    int64_t c = ((int64_t)1 << 28) / std::max(1, h->nlist);  // <= 1 GiB of scores
    return std::min<int64_t>(std::max<int64_t>(c, 256), 65536);
}

}  // namespace

// ------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------

extern "C" {

const char *mi_last_error(void) { return last_error().c_str(); }

int mi_device_count(int *count) {
    return guard([&] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            n = 0;
        }
        *count = n;
    });
}

int mi_index_create(int d, int nlist, int M, int nbits, int metric, int by_residual, int device,
                    mi_index **out) {
    return guard([&] {
        MI_REQUIRE(out != nullptr, "out is null");
        MI_REQUIRE(metric == MI_METRIC_INNER_PRODUCT || metric == MI_METRIC_L2, "metric must be MI_METRIC_INNER_PRODUCT or MI_METRIC_L2");
        MI_REQUIRE(nbits == 8, "only nbits == 8 is implemented");
        MI_REQUIRE(d > 0 && M > 0 && d % M == 0, "d must be a positive multiple of M");
        MI_REQUIRE(d % 4 == 0, "d must be a multiple of 4");
        MI_REQUIRE(nlist > 0, "nlist must be positive");
        MI_REQUIRE(scan_supports_M(M), "unsupported M (PQ sub-quantisers: 4,8,16,32,48,64,96,128)");
        int dsub = d / M;
        MI_REQUIRE(dsub == 1 || dsub == 2 || dsub == 4 || dsub == 8 || dsub == 16 || dsub == 32 ||
                       dsub == 64,
                   "unsupported d/M (sub-vector length must be 1,2,4,8,16,32 or 64)");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) {
            (void)hipGetLastError();
            throw Error("no HIP device available: the MI355X index has no CPU fallback");
        }
        MI_REQUIRE(device >= 0 && device < ndev, "invalid device ordinal");
        auto h = std::make_unique<mi_index>();
        h->d = d; h->nlist = nlist; h->M = M; h->dsub = dsub; h->metric = metric;
        h->by_residual = by_residual ? 1 : 0; h->device = device;
        // augmented width for METRIC_L2: one more column, padded so that the two-stage coarse
        // quantiser (d % 128 == 0) still applies to big indexes
        h->da = (d % 128 == 0 && nlist >= 8192) ? d + 128 : d + 4;
        {
            DeviceGuard dg(device);
            MI_HIP(hipMemset(h->d_cnt.reserve((size_t)nlist * 4), 0, (size_t)nlist * 4));
        }
        h->h_len.assign((size_t)nlist, 0);
        *out = h.release();
    });
}

int mi_index_destroy(mi_index *h) {
    return guard([&] {
        if (!h) return;
        DeviceGuard dg(h->device);
        if (h->prune_seen) {
            (void)hipDeviceSynchronize();      // an asynchronous copy of the counters may still be on its way
            (void)hipHostFree(h->prune_seen);
        }
        delete h;
    });
}

int mi_index_set_coarse(mi_index *h, const float *centroids) {
    return guard([&] {
        MI_REQUIRE(h && centroids, "null argument");
        DeviceGuard dg(h->device);
        size_t bytes = (size_t)h->nlist * h->d * sizeof(float);
        MI_HIP(hipMemcpy(h->centroids.reserve(bytes), centroids, bytes, hipMemcpyDefault));
        h->has_coarse = true;
        h->cent16_ok = false;
        const float *cq = h->centroids.get<float>();   // what the coarse quantiser multiplies
        int dq = h->d;
        if (h->metric == MI_METRIC_L2) {
            launch_augment(cq, h->nlist, h->d, h->da, 1, h->cent_aug.as<float>((size_t)h->nlist * h->da), nullptr);
            cq = h->cent_aug.get<float>();
            dq = h->da;
        }
        // the f16 image the two-stage coarse quantiser uses is built here, not lazily inside a
        // search (concurrent searches on several streams only read the index)
        if (h->nlist >= 8192 && dq % 128 == 0 && dq <= 4096 && h->nlist % 4 == 0) {
            prepare_cent16(cq, h->nlist, dq, h->cent16, h->cmax_dev, h->cmax, h->cscale, nullptr);
            h->cent16_ok = true;
        }
        MI_HIP(hipStreamSynchronize(nullptr));
    });
}

int mi_index_set_codebook(mi_index *h, const float *codebook) {
    return guard([&] {
        MI_REQUIRE(h && codebook, "null argument");
        DeviceGuard dg(h->device);
        size_t bytes = (size_t)h->M * 256 * h->dsub * sizeof(float);
        MI_HIP(hipMemcpy(h->codebook.reserve(bytes), codebook, bytes, hipMemcpyDefault));
        h->has_codebook = true;
    });
}

int mi_index_get_coarse(mi_index *h, float *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        MI_REQUIRE(h->has_coarse, "no coarse centroids set");
        DeviceGuard dg(h->device);
        MI_HIP(hipMemcpy(out, h->centroids.p, (size_t)h->nlist * h->d * 4, hipMemcpyDeviceToHost));
    });
}

int mi_index_get_codebook(mi_index *h, float *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        MI_REQUIRE(h->has_codebook, "no codebook set");
        DeviceGuard dg(h->device);
        MI_HIP(hipMemcpy(out, h->codebook.p, (size_t)h->M * 256 * h->dsub * 4, hipMemcpyDeviceToHost));
    });
}

int mi_index_is_trained(mi_index *h, int *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        *out = (h->has_coarse && h->has_codebook) ? 1 : 0;
    });
}

int mi_index_ntotal(mi_index *h, int64_t *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        *out = h->ntotal;
    });
}

int mi_index_reset(mi_index *h) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        DeviceGuard dg(h->device);
        MI_HIP(hipMemset(h->d_cnt.p, 0, (size_t)h->nlist * 4));
        h->h_len.assign((size_t)h->nlist, 0);
        h->len_ok = true;
        h->ntotal = 0;
        h->dirty = true;
        h->sealed = false;
    });
}

int mi_index_encode(mi_index *h, int64_t n, const float *x, int32_t *list_no, uint8_t *codes) {
    return guard([&] {
        MI_REQUIRE(h && (n == 0 || x), "null argument");
        require_trained(h);
        DeviceGuard dg(h->device);
        const int64_t chunk = add_chunk_size(h);
        const bool xdev = is_device_ptr(x);
        for (int64_t c0 = 0; c0 < n; c0 += chunk) {
            int64_t m = std::min(chunk, n - c0);
            const float *xs = x + (size_t)c0 * h->d;
            if (!xdev) {
                float *stage = h->ws_x.as<float>((size_t)m * h->d);
                MI_HIP(hipMemcpy(stage, xs, (size_t)m * h->d * 4, hipMemcpyHostToDevice));
                xs = stage;
            }
            encode_chunk(h, xs, m, h->ws_assign.as<int32_t>((size_t)m), h->ws_codes.as<uint8_t>((size_t)m * h->M), nullptr);
            if (list_no) MI_HIP(hipMemcpy(list_no + c0, h->ws_assign.p, (size_t)m * 4, hipMemcpyDeviceToHost));
            if (codes) MI_HIP(hipMemcpy(codes + (size_t)c0 * h->M, h->ws_codes.p, (size_t)m * h->M, hipMemcpyDeviceToHost));
        }
    });
}

int mi_index_add_codes(mi_index *h, int64_t n, const int32_t *list_no, const uint8_t *codes,
                       const int64_t *ids) {
    return guard([&] {
        MI_REQUIRE(h && (n == 0 || (list_no && codes)), "null argument");
        if (n == 0) return;
        DeviceGuard dg(h->device);
        const bool dev = is_device_ptr(list_no);
        MI_REQUIRE(is_device_ptr(codes) == dev && (!ids || is_device_ptr(ids) == dev),
                   "add_codes: list_no, codes and ids must be all host or all device pointers");
        if (!dev)
            for (int64_t i = 0; i < n; ++i)
                MI_REQUIRE(list_no[i] >= 0 && list_no[i] < h->nlist, "list number out of range");
        ensure_log_cap(h, h->ntotal + n);
        const int64_t at = h->ntotal;
        MI_HIP(hipMemcpyAsync(h->log_list.get<int32_t>() + at, list_no, (size_t)n * 4, hipMemcpyDefault, nullptr));
        MI_HIP(hipMemcpyAsync(h->log_codes.get<uint8_t>() + (size_t)at * h->M, codes, (size_t)n * h->M, hipMemcpyDefault, nullptr));
        if (ids) MI_HIP(hipMemcpyAsync(h->log_ids.get<int64_t>() + at, ids, (size_t)n * 8, hipMemcpyDefault, nullptr));
        else {
            hipLaunchKernelGGL(iota_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                               h->log_ids.get<int64_t>() + at, n, at);
            MI_HIP(hipGetLastError());
        }
        commit_log_entries(h, n, nullptr);
        if (!dev) MI_HIP(hipStreamSynchronize(nullptr));   // the caller may reuse its host buffers
    });
}

int mi_index_reserve(mi_index *h, int64_t n) {
    return guard([&] {
        MI_REQUIRE(h && n >= 0, "bad argument");
        DeviceGuard dg(h->device);
        ensure_log_cap(h, n);
    });
}

int mi_index_add(mi_index *h, int64_t n, const float *x, const int64_t *ids) {
    return guard([&] {
        MI_REQUIRE(h && (n == 0 || x), "null argument");
        require_trained(h);
        if (n == 0) return;
        DeviceGuard dg(h->device);
        ensure_log_cap(h, h->ntotal + n);
        const int64_t at = h->ntotal;
        if (ids) MI_HIP(hipMemcpyAsync(h->log_ids.get<int64_t>() + at, ids, (size_t)n * 8, hipMemcpyDefault, nullptr));
        else {
            hipLaunchKernelGGL(iota_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                               h->log_ids.get<int64_t>() + at, n, at);
            MI_HIP(hipGetLastError());
        }
        const int64_t chunk = add_chunk_size(h);
        const bool xdev = is_device_ptr(x);
        for (int64_t c0 = 0; c0 < n; c0 += chunk) {
            int64_t m = std::min(chunk, n - c0);
            const float *xs = x + (size_t)c0 * h->d;
            if (!xdev) {
                float *stage = h->ws_x.as<float>((size_t)m * h->d);
                MI_HIP(hipMemcpy(stage, xs, (size_t)m * h->d * 4, hipMemcpyHostToDevice));
                xs = stage;
            }
            // list numbers and codes are written straight into the tail of the log
            encode_chunk(h, xs, m, h->log_list.get<int32_t>() + at + c0,
                         h->log_codes.get<uint8_t>() + (size_t)(at + c0) * h->M, nullptr);
        }
        commit_log_entries(h, n, nullptr);
        if (!xdev || (ids && !is_device_ptr(ids))) MI_HIP(hipStreamSynchronize(nullptr));
    });
}

int mi_index_list_size(mi_index *h, int list_no, int64_t *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        MI_REQUIRE(list_no >= 0 && list_no < h->nlist, "list number out of range");
        DeviceGuard dg(h->device);
        refresh_len(h);
        *out = (int64_t)h->h_len[(size_t)list_no];
    });
}

int mi_index_list_sizes(mi_index *h, int64_t *sizes) {
    return guard([&] {
        MI_REQUIRE(h && sizes, "null argument");
        DeviceGuard dg(h->device);
        refresh_len(h);
        for (int l = 0; l < h->nlist; ++l) sizes[l] = h->h_len[(size_t)l];
    });
}

int mi_index_export_lists(mi_index *h, int list_lo, int list_hi, uint8_t *codes, int64_t *ids) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        MI_REQUIRE(0 <= list_lo && list_lo <= list_hi && list_hi <= h->nlist, "list range out of bounds");
        DeviceGuard dg(h->device);
        unseal(h);
        refresh_len(h);
        std::vector<int64_t> start((size_t)(list_hi - list_lo) + 1, 0);
        for (int l = list_lo; l < list_hi; ++l) start[(size_t)(l - list_lo) + 1] = start[(size_t)(l - list_lo)] + h->h_len[(size_t)l];
        const int64_t rows = start.back();
        if (rows == 0 || (!codes && !ids)) return;
        DevBuf dstart, dc, di;
        MI_HIP(hipMemcpyAsync(dstart.reserve(start.size() * 8), start.data(), start.size() * 8, hipMemcpyHostToDevice, nullptr));
        const bool cdev = codes && is_device_ptr(codes), idev = ids && is_device_ptr(ids);
        uint8_t *oc = cdev ? codes : dc.as<uint8_t>((size_t)rows * h->M);
        int64_t *oi = idev ? ids : di.as<int64_t>((size_t)rows);
        const int64_t threads = h->ntotal * h->nch();
        hipLaunchKernelGGL(export_lists_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr,
                           h->log_codes.get<uint8_t>(), h->log_list.get<int32_t>(), h->log_pos.get<int32_t>(),
                           h->log_ids.get<int64_t>(), h->ntotal, list_lo, list_hi, dstart.get<int64_t>(), h->M, h->nch(),
                           oc, oi);
        MI_HIP(hipGetLastError());
        if (codes && !cdev) MI_HIP(hipMemcpyAsync(codes, oc, (size_t)rows * h->M, hipMemcpyDeviceToHost, nullptr));
        if (ids && !idev) MI_HIP(hipMemcpyAsync(ids, oi, (size_t)rows * 8, hipMemcpyDeviceToHost, nullptr));
        MI_HIP(hipStreamSynchronize(nullptr));
    });
}

// The index is built and will be searched: free the append log (80 B per vector: 16.6 GB of the 207 M-vector index), keep the
// scan image.  Nothing is lost -- a later add / add_codes / export / save rebuilds the log from the image first (one pass, list
// order) -- and searches are unaffected.  An exclusive call, like add().
int mi_index_seal(mi_index *h) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        DeviceGuard dg(h->device);
        {
            std::lock_guard<std::mutex> lk(h->mu);
            sync_lists(h);
        }
        if (h->sealed) return;
        h->log_codes.release(); h->log_list.release(); h->log_pos.release(); h->log_ids.release(); h->log_t.release();
        h->log_cap = 0;
        h->sealed = true;
    });
}

int mi_index_get_list(mi_index *h, int list_no, uint8_t *codes, int64_t *ids) {
    if (!h || list_no < 0 || list_no >= h->nlist) {
        last_error() = !h ? "null argument" : "list number out of range";
        return 1;
    }
    return mi_index_export_lists(h, list_no, list_no + 1, codes, ids);
}

int mi_index_profile_scan(mi_index *h, int reps, void *stream, double *scan_ms_avg, int64_t *scan_bytes) {
    return guard([&] {
        MI_REQUIRE(h && reps >= 1, "bad argument");
        WsLease lease = lease_ws(h, stream, false);
        SearchWS &w = lease.w;
        MI_REQUIRE(w.have_last_scan, "mi_index_profile_scan: no search has run on this stream yet");
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        // algorithmic bytes of the launch: sum of the probed list lengths x (M + 8)
        unsigned long long *cnt = h->ws_count.as<unsigned long long>(1);
        MI_HIP(hipMemsetAsync(cnt, 0, 8, st));
        const int64_t n = w.last_nq * w.last_nprobe;
        // (the launch's own table of list lengths: a pruned search replays its first phase, whose other probes have length 0)
        hipLaunchKernelGGL(sum_len_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.last_scan.p_len, n, cnt);
        MI_HIP(hipGetLastError());
        hipEvent_t e0, e1;
        MI_HIP(hipEventCreate(&e0));
        MI_HIP(hipEventCreate(&e1));
        ScanArgs replay = w.last_scan;
        replay.prune_stats = nullptr;        // (the replays are not searches)
        launch_scan(h->M, replay, st);  // warm
        MI_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_scan(h->M, replay, st);
        MI_HIP(hipEventRecord(e1, st));
        MI_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MI_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (knobs().scan_ts) {
            // one more replay with in-kernel s_memtime stamps: mean/max time of each phase
            // boundary relative to the earliest workgroup start, printed to stderr
            const size_t nb = scan_grid(w.last_scan.nq, w.last_scan.nslice);
            DevBuf tsb;
            unsigned long long *dts = tsb.as<unsigned long long>(nb * SCAN_TS);
            MI_HIP(hipMemsetAsync(dts, 0, nb * SCAN_TS * 8, st));
            ScanArgs sa = replay;
            sa.ts = dts;
            launch_scan(h->M, sa, st);
            std::vector<unsigned long long> hts(nb * SCAN_TS);
            MI_HIP(hipStreamSynchronize(st));
            MI_HIP(hipMemcpy(hts.data(), dts, nb * SCAN_TS * 8, hipMemcpyDeviceToHost));
            // every XCD has its own counter: only differences inside one workgroup mean anything
            std::fprintf(stderr, "[scan stamps] %zu workgroups, s_memtime ticks since the workgroup's own start\n", nb);
            {   // slot 14 = survivors of the block threshold + 1
                double sum = 0; unsigned long long mx = 0; size_t big = 0;
                for (size_t b = 0; b < nb; ++b) {
                    const unsigned long long c = hts[b * SCAN_TS + 14];
                    if (!c) continue;
                    sum += (double)(c - 1); mx = std::max(mx, c - 1); big += (c - 1) > 64;
                }
                std::fprintf(stderr, "  survivors per workgroup: mean %.1f max %llu, %zu workgroups above 64\n", sum / nb, mx, big);
            }
            for (int i = 1; i < SCAN_TS; ++i) {
                if (i == 14) continue;
                double sum = 0, mx = 0, mn = 1e30;
                size_t cntb = 0;
                for (size_t b = 0; b < nb; ++b) {
                    const unsigned long long v = hts[b * SCAN_TS + i], v0 = hts[b * SCAN_TS];
                    if (!v || !v0) continue;
                    const double dv = (double)(v - v0);
                    sum += dv; mx = std::max(mx, dv); mn = std::min(mn, dv);
                    ++cntb;
                }
                if (cntb) std::fprintf(stderr, "  stamp %2d: n=%5zu  min %8.0f  mean %8.0f  max %8.0f\n", i, cntb, mn, sum / cntb, mx);
            }
        }
        unsigned long long c = 0;
        MI_HIP(hipMemcpy(&c, cnt, 8, hipMemcpyDeviceToHost));
        if (replay.prune_A) {
            // an early-stop launch reads only the groups its waves reach: counted by one more replay (whole 64-code groups)
            DevBuf tmp;
            unsigned long long *t3 = tmp.as<unsigned long long>(PRUNE_SLOTS * 3);
            MI_HIP(hipMemsetAsync(t3, 0, PRUNE_SLOTS * 24, st));
            ScanArgs sa = replay;
            sa.prune_stats = t3;
            launch_scan(h->M, sa, st);
            MI_HIP(hipStreamSynchronize(st));
            unsigned long long g3[PRUNE_SLOTS * 3], groups = 0;
            MI_HIP(hipMemcpy(g3, t3, sizeof(g3), hipMemcpyDeviceToHost));
            for (int sl = 0; sl < PRUNE_SLOTS; ++sl) groups += g3[sl * 3];
            c = std::min<unsigned long long>(c, groups * 64ull);
        }
        if (scan_ms_avg) *scan_ms_avg = (double)ms / reps;
        if (scan_bytes) *scan_bytes = (int64_t)c * (h->M + 8);
    });
}

int mi_index_prune_stats(mi_index *h, unsigned long long *out3, int reset) {
    return guard([&] {
        MI_REQUIRE(h && out3, "bad argument");
        DeviceGuard dg(h->device);
        std::lock_guard<std::mutex> hl(h->mu);
        out3[0] = out3[1] = out3[2] = 0;
        if (!h->prune_stats_ok) return;
        MI_HIP(hipDeviceSynchronize());
        unsigned long long slots[PRUNE_SLOTS * 3], life[3] = {0, 0, 0};
        MI_HIP(hipMemcpy(slots, h->prune_stats.p, sizeof(slots), hipMemcpyDeviceToHost));
        for (int sl = 0; sl < PRUNE_SLOTS; ++sl)
            for (int i = 0; i < 3; ++i) life[i] += slots[sl * 3 + i];
        for (int i = 0; i < 3; ++i) {
            out3[i] = life[i] - h->prune_base[i];
            if (reset) h->prune_base[i] = life[i];
        }
    });
}

// Number of (query, slice) workgroups: enough to give every CU one, but never
// fewer than ~16 code groups (two per wave) of expected work per slice: every
// slice re-stages the query's 64 KiB LUT, so more slices = more LDS fill traffic
// (PMC: at 8 slices the LUT reads equal the code bytes).
static int choose_nslice(const mi_index *h, int64_t nq, int nprobe, int k = 10) {
    double avg_groups = h->nlist > 0 ? (double)h->ngroups / h->nlist : 0.0;
    double per_query = avg_groups * nprobe;
    int64_t by_fill = (256 + nq - 1) / nq;              // one workgroup per CU
    int64_t by_work = (int64_t)(per_query / 16.0);       // >= 2 groups per wave
    int64_t s = std::min(by_fill, by_work);
    // Big scans (cfg4: 1024 queries x 3160 groups): one workgroup per query is two rounds of
    // 512 co-resident workgroups whose lengths differ with the probed lists -- the last CUs idle
    // while the longest finish.  Slices of >= 128 groups even that out (207 M index, 1024 x
    // nprobe 64: scan 2.77 -> 2.49 ms, 5.27 -> 5.85 TB/s; profiles/r02_cfg4_scan_sweep.txt);
    // at most 64 / k slices so that the cross-slice merge stays on its one-wave path.
    const int64_t by_tail = std::min<int64_t>((int64_t)(per_query / 128.0), std::max(1, std::min(8, 64 / std::max(1, std::min(k, 64)))));
    s = std::max(s, by_tail);
    if (knobs().nslice > 0) s = knobs().nslice;          // tuning knob
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, 32));
}

static void search_chunk(mi_index *h, SearchWS &w, int64_t nq, const float *qdev, int k, int nprobe, float *Ddev,
                         int64_t *Idev, hipStream_t st, int32_t *cI_out, float *cD_out,
                         float *lut_out, bool stop_after_lut, const int32_t *pre_I = nullptr,
                         const float *pre_D = nullptr, bool cand_set = false) {
    const int M = h->M;
    const bool l2 = h->metric == MI_METRIC_L2;
    std::unique_ptr<Range> stage = std::make_unique<Range>("mi_ivfpq:coarse+lut");
    int32_t *cidx = w.cidx.as<int32_t>((size_t)nq * nprobe);
    float *cdis = w.cdis.as<float>((size_t)nq * nprobe);
    float *lut = w.lut.as<float>((size_t)nq * M * 256);
    ProbeTables pt{};
    if (!stop_after_lut) {
        pt.list_goff = h->d_goff.get<int32_t>();
        pt.list_len = h->d_len.get<int32_t>();
        pt.p_goff = w.pgoff.as<int32_t>((size_t)nq * nprobe);
        pt.p_len = w.plen.as<int32_t>((size_t)nq * nprobe);
        pt.p_prefix = w.pprefix.as<int32_t>((size_t)nq * (nprobe + 1));
    }
    if (pre_I) {
        // search_preassigned: coarse result given (device pointers); tables + LUT only
        MI_HIP(hipMemcpyAsync(cidx, pre_I, (size_t)nq * nprobe * 4, hipMemcpyDeviceToDevice, st));
        MI_HIP(hipMemcpyAsync(cdis, pre_D, (size_t)nq * nprobe * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(probe_tables_kernel, dim3((unsigned)nq), dim3(256), 0, st, pt, nprobe, cidx);
        MI_HIP(hipGetLastError());
        launch_lut(qdev, (int)nq, h->d, M, h->codebook.get<float>(), lut, st);
    } else {
    float *scores = w.scores.as<float>((size_t)nq * h->nlist);
    // what the coarse quantiser multiplies: the vectors themselves, or (METRIC_L2) their augmented images
    const float *qc = qdev, *cc = h->centroids.get<float>();
    int dc = h->d;
    if (l2) {
        float *qa = w.qaug.as<float>((size_t)nq * h->da);
        launch_augment(qdev, nq, h->d, h->da, 0, qa, st);
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, qdev, nq, h->d, w.qn.as<float>((size_t)nq));
        MI_HIP(hipGetLastError());
        qc = qa; cc = h->cent_aug.get<float>(); dc = h->da;
    }
    // Large batches: f16 MFMA scores + exact re-scoring of the few centroids within a proven
    // error margin of the cut (bit-identical result, see select_refine_kernel) instead of the
    // exact f32 GEMM over all of them.
    if (two_stage_wanted(nq, h->nlist, dc, nprobe)) {
        {
            std::lock_guard<std::mutex> hl(h->mu);       // (set_coarse builds it for the shapes that qualify; this is the fallback)
            if (!h->cent16_ok) {
                prepare_cent16(cc, h->nlist, dc, h->cent16, h->cmax_dev, h->cmax, h->cscale, st);
                MI_HIP(hipStreamSynchronize(st));        // other streams may read it as soon as the flag is up
                h->cent16_ok = true;
            }
        }
        launch_two_stage(qc, nq, cc, static_cast<const f16_t *>(h->cent16.p), h->nlist, dc,
                         nprobe, h->cmax, h->cscale, scores, w.q16, w.qscale, w.rstats, cidx, cdis, pt, st);
        launch_lut(qdev, (int)nq, h->d, M, h->codebook.get<float>(), lut, st);
    } else {
    const bool lut_in_gemm = h->dsub == 4 || h->dsub == 8 || h->dsub == 16;
    launch_gemm(qc, nq, cc, h->nlist, dc, scores, h->nlist, st,
                lut_in_gemm ? make_lut_args(qdev, (int)nq, h->d, M, h->codebook.get<float>(), lut) : LutArgs{});
    launch_select(scores, h->nlist, nq, h->nlist, nprobe, cidx, nullptr, cdis, st, pt, 0, &w.selsc);
    if (!lut_in_gemm) launch_lut(qdev, (int)nq, h->d, M, h->codebook.get<float>(), lut, st);
    }
    }
    const float *cscan = cdis;   // the per-(query, probe) term the scan adds
    if (l2) {
        MI_REQUIRE(!pre_I, "METRIC_L2: search_preassigned is not implemented");
        float *cs = w.cscan.as<float>((size_t)nq * nprobe);
        hipLaunchKernelGGL(l2_coarse_term_kernel, dim3((unsigned)(((size_t)nq * nprobe + 255) / 256)), dim3(256), 0, st, cdis, cs,
                           cidx, w.qn.get<float>(), nq, nprobe, h->by_residual);
        MI_HIP(hipGetLastError());
        cscan = cs;
    }
    auto l2_finish = [&] {       // scores -> ascending squared distances (faiss METRIC_L2 results)
        if (!l2) return;
        hipLaunchKernelGGL(l2_finish_kernel, dim3((unsigned)(((size_t)nq * k + 255) / 256)), dim3(256), 0, st, Ddev, Idev, nq * (int64_t)k);
        MI_HIP(hipGetLastError());
    };
    if (cI_out) MI_HIP(hipMemcpyAsync(cI_out, cidx, (size_t)nq * nprobe * 4, hipMemcpyDeviceToHost, st));
    if (cD_out) MI_HIP(hipMemcpyAsync(cD_out, cdis, (size_t)nq * nprobe * 4, hipMemcpyDeviceToHost, st));
    if (lut_out) MI_HIP(hipMemcpyAsync(lut_out, lut, (size_t)nq * M * 256 * 4, hipMemcpyDeviceToHost, st));
    if (stop_after_lut) return;
    stage.reset();
    stage = std::make_unique<Range>("mi_ivfpq:scan+topk");

    int nslice = choose_nslice(h, nq, nprobe, std::min(k, 64));
    const int npass = (k + 63) / 64;
    float *ps = w.ps.as<float>((size_t)nq * nslice * 64);
    int64_t *pid = w.pid.as<int64_t>((size_t)nq * nslice * 64);
    float *bs = nullptr;
    int64_t *bid = nullptr;
    if (npass > 1) {
        bs = w.bs.as<float>((size_t)nq);
        bid = w.bid.as<int64_t>((size_t)nq);
    }
    // Waves per scan workgroup.  16 (one round less for the first wave of a cfg2-sized
    // slice) measured the same kernel time -- the slice is bound by the CU's LDS gather
    // rate, not by the number of rounds -- and costs the second co-resident workgroup
    // (3 streams: 2.7M vs 3.2M QPS), so 8 it is; MI_SCAN_NW=16 keeps the experiment.
    // Large nprobe at a small batch is the other regime: at most one workgroup per CU and
    // >= 128 code groups per slice.  There 16 waves (4 per SIMD) hide the gather/load
    // latency that a second co-resident workgroup would: 254 -> 232 us at 64 x nprobe 1024.
    int scan_nw = 8;
    {
        const double groups_per_slice = (h->nlist > 0 ? (double)h->ngroups / h->nlist : 0.0) * nprobe / nslice;
        if (scan_grid((int)nq, nslice) <= 256 && groups_per_slice >= 128.0) scan_nw = 16;
    }
    if (knobs().scan_nw) scan_nw = knobs().scan_nw == 16 ? 16 : 8;
    if (npass > 1 && !knobs().no_allscores) {
        // k > 64: one pass over the codes that stores every (score, id) of the probed lists,
        // then the k best of each row (select_pairs_kernel) -- instead of one scan per 64
        // results.  Row capacity = the groups of the nprobe longest lists; queries go in
        // sub-batches so that the rows stay under 2 GiB.
        int64_t R;
        {
            std::lock_guard<std::mutex> hl(h->mu);
            if (h->cap_nprobe != nprobe) {
                std::vector<int64_t> g((size_t)h->nlist);
                for (int l = 0; l < h->nlist; ++l) g[(size_t)l] = ((int64_t)h->h_len[(size_t)l] + 63) / 64;   // valid: sync_lists ran
                std::nth_element(g.begin(), g.begin() + (nprobe - 1), g.end(), std::greater<int64_t>());
                int64_t tot = 0;
                for (int i = 0; i < nprobe; ++i) tot += g[(size_t)i];
                h->cap_groups = std::max<int64_t>(tot, 1);
                h->cap_nprobe = nprobe;
            }
            R = h->cap_groups * 64;
        }
        if (pre_I) {   // caller-assigned lists may repeat: nprobe times the longest list
            int64_t gmax = 1;
            for (int l = 0; l < h->nlist; ++l) gmax = std::max(gmax, ((int64_t)h->h_len[(size_t)l] + 63) / 64);
            R = gmax * nprobe * 64;
        }
        const int64_t qc = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)2 << 30) / (R * 12)));
        float *all_s = w.all_s.as<float>((size_t)(qc * R));
        // scores only (inner product): the selection fetches the survivors' ids from the lists; METRIC_L2 (whose scan variant
        // was left as it is) keeps the row of ids beside the scores
        const bool ids_row = l2;
        int64_t *all_id = ids_row ? w.all_id.as<int64_t>((size_t)(qc * R)) : nullptr;
        for (int64_t c0 = 0; c0 < nq; c0 += qc) {
            const int64_t m = std::min(qc, nq - c0);
            ScanArgs a{};
            a.lut = lut + (size_t)c0 * M * 256; a.coarse_dis = cscan + (size_t)c0 * nprobe;
            a.tnorm = l2 ? h->d_tnorm.get<float>() : nullptr;
            a.p_goff = pt.p_goff + (size_t)c0 * nprobe; a.p_len = pt.p_len + (size_t)c0 * nprobe;
            a.p_prefix = pt.p_prefix + (size_t)c0 * (nprobe + 1);
            a.codes = h->d_codes.get<uint8_t>(); a.ids = h->d_ids.get<int64_t>();
            a.nq = (int)m; a.nprobe = nprobe; a.nslice = choose_nslice(h, m, nprobe); a.k = 64;
            if (knobs().nslice <= 0) {
                // the all-scores scan has no cross-slice merge to keep short: slices of ~64 groups even out the rounds of
                // workgroups better than the k-limited count above (1024 queries x 395 groups: 1 / 2 / 3 / 6 / 8 / 12 slices
                // 449 / 401 / 389 / 366 / 384 / 393 us -- tools/micro/nslice_sweep.sh)
                const double per_query = (h->nlist > 0 ? (double)h->ngroups / h->nlist : 0.0) * nprobe;
                a.nslice = std::max(a.nslice, (int)std::min(6.0, per_query / 64.0));
            }
            a.by_residual = l2 ? 1 : h->by_residual;
            a.nw = 8;
            {
                const double groups_per_slice = (h->nlist > 0 ? (double)h->ngroups / h->nlist : 0.0) * nprobe / a.nslice;
                if (scan_grid((int)m, a.nslice) <= 256 && groups_per_slice >= 128.0) a.nw = 16;
            }
            if (knobs().scan_nw) a.nw = knobs().scan_nw == 16 ? 16 : 8;
            a.all_s = all_s; a.all_id = all_id; a.all_ld = R;
            launch_scan(M, a, st);
            launch_select_pairs(all_s, all_id, R, a.p_prefix, nprobe, k, m, Ddev ? Ddev + (size_t)c0 * k : nullptr, Idev + (size_t)c0 * k,
                                (int64_t)k, st, cand_set && !l2, a.p_goff, a.ids);
            MI_HIP(hipGetLastError());
        }
        l2_finish();
        return;
    }
    // ---- exact list pruning, two forms (ivfpq_kernels.h, above prune_tables_kernel).  By-residual inner product, k <= 64:
    //  - early stop inside the scan kernel: the lists of a query come in descending coarse order, so do their bounds; a wave
    //    stops at the first list whose bound is below a threshold it holds.  One launch more (lut_maxsum_kernel, ~5 us);
    //    nprobe <= 64 (the probe tables in registers), lists from the library's own coarse quantiser (sorted);
    //  - two phases (prune_tables_kernel): the P1 best lists of every query first, then only the lists whose bound reaches the
    //    k-th score found there -- any nprobe, any list order, a threshold shared by all slices of a query; two more small
    //    launches, a second scan launch and a merge: for scans large enough to pay for them (a query batch, not a query).
    const double avg_groups_pr = h->nlist > 0 ? (double)h->ngroups / h->nlist : 0.0;
    const bool prunable = knobs().scan_prune && !l2 && h->by_residual && npass == 1 && M <= 128;
    // (a query's slices each hold their own thresholds, and only the first has the best lists: the early stop wants ONE workgroup
    // per query -- batches that fill the chip that way; smaller ones take the two phases, whose threshold all slices share.  One
    // workgroup per query gives up the balanced rounds of the sliced launch: only while the index's earlier calls did prune)
    // (caller-assigned lists -- mi_index_search_preassigned: ShardedIndex hands over the merged coarse result, in order -- are
    // checked row by row on the device: a query whose lists are not in descending order is scanned in full)
    const bool early = prunable && knobs().prune_mode != 1 && nprobe <= 64 &&
                       (knobs().prune_mode == 2 || (nq >= 512 && avg_groups_pr * nprobe * (double)nq >= 4096.0));
    unsigned long long *pstats = nullptr;
    if (prunable) {
        std::lock_guard<std::mutex> hl(h->mu);
        pstats = h->prune_stats.as<unsigned long long>(PRUNE_SLOTS * 3);
        if (!h->prune_stats_ok) {
            MI_HIP(hipMemsetAsync(pstats, 0, PRUNE_SLOTS * 24, st));
            MI_HIP(hipStreamSynchronize(st));
            MI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->prune_seen), PRUNE_SLOTS * 24, hipHostMallocDefault));
            std::memset(h->prune_seen, 0, PRUNE_SLOTS * 24);
            h->prune_stats_ok = true;
        }
    }
    // does this index's data prune?  (counters of earlier calls, as last copied to the host; nothing seen yet: assume it does)
    bool prunes_well = true;
    if (prunable && h->prune_seen) {
        unsigned long long g_scanned = 0, g_all = 0;
        for (int sl = 0; sl < PRUNE_SLOTS; ++sl) {
            g_scanned += reinterpret_cast<volatile unsigned long long *>(h->prune_seen)[sl * 3];
            g_all += reinterpret_cast<volatile unsigned long long *>(h->prune_seen)[sl * 3 + 1];
        }
        if (g_all >= 100000 && g_scanned <= g_all) prunes_well = (double)g_scanned < 0.5 * (double)g_all;
    }
    if (early && knobs().nslice <= 0 && nslice > 1 && prunes_well) {
        nslice = 1;
        ps = w.ps.as<float>((size_t)nq * nslice * 64);
        pid = w.pid.as<int64_t>((size_t)nq * nslice * 64);
    }
    float *pruneA = nullptr;
    int32_t *prune_sorted = nullptr;
    if (early) {
        pruneA = w.pruneA.as<float>((size_t)nq * 2);
        hipLaunchKernelGGL(lut_maxsum_kernel, dim3((unsigned)nq), dim3(256), 0, st, lut, M, pruneA);
        MI_HIP(hipGetLastError());
        if (pre_I) {
            prune_sorted = reinterpret_cast<int32_t *>(pruneA + nq);
            hipLaunchKernelGGL(rows_descending_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, cscan, nq, nprobe, prune_sorted);
            MI_HIP(hipGetLastError());
        }
    }
    {
        const bool prune = prunable && !early && knobs().prune_mode != 2 && nprobe >= 8 && Ddev &&
                           avg_groups_pr * nprobe * (double)nq >= (double)knobs().prune_min_groups;
        if (prune) {
            const int kp = k;
            const int P1 = std::max(1, std::min(knobs().prune_p1 > 0 ? knobs().prune_p1 : std::min(8, nprobe / 32), nprobe - 1));
            unsigned long long *stats = pstats;
            float *comb_s = w.comb_s.as<float>((size_t)nq * 2 * kp);
            int64_t *comb_id = w.comb_id.as<int64_t>((size_t)nq * 2 * kp);
            int32_t *len_ph[2] = {w.plen1.as<int32_t>((size_t)nq * nprobe), w.plen2.as<int32_t>((size_t)nq * nprobe)};
            int32_t *pre_ph[2] = {w.ppre1.as<int32_t>((size_t)nq * (nprobe + 1)), w.ppre2.as<int32_t>((size_t)nq * (nprobe + 1))};
            const int ns_ph[2] = {choose_nslice(h, nq, P1, kp), choose_nslice(h, nq, std::max(1, (nprobe - P1) / 8), kp)};
            const int ns_max = std::max(ns_ph[0], ns_ph[1]);
            float *pps = w.ps.as<float>((size_t)nq * ns_max * 64);
            int64_t *ppid = w.pid.as<int64_t>((size_t)nq * ns_max * 64);
            for (int ph = 0; ph < 2; ++ph) {
                PruneArgs pa{};
                pa.lut = lut; pa.coarse_dis = cscan; pa.p_len = pt.p_len; pa.p_prefix = pt.p_prefix;
                pa.t_s = comb_s; pa.t_id = comb_id; pa.ld_t = 2 * kp; pa.k = kp; pa.nprobe = nprobe; pa.P1 = P1; pa.M = M; pa.mode = ph;
                pa.len_out = len_ph[ph]; pa.prefix_out = pre_ph[ph]; pa.stats = ph ? stats : nullptr;
                hipLaunchKernelGGL(prune_tables_kernel, dim3((unsigned)nq), dim3(256), 0, st, pa);
                MI_HIP(hipGetLastError());
                ScanArgs a{};
                a.lut = lut; a.coarse_dis = cscan;
                a.p_goff = pt.p_goff; a.p_len = len_ph[ph]; a.p_prefix = pre_ph[ph];
                a.codes = h->d_codes.get<uint8_t>(); a.ids = h->d_ids.get<int64_t>();
                a.part_s = pps; a.part_id = ppid;
                a.nq = (int)nq; a.nprobe = nprobe; a.nslice = ns_ph[ph]; a.k = kp; a.by_residual = 1;
                a.nw = 8;
                if (knobs().scan_nw) a.nw = knobs().scan_nw == 16 ? 16 : 8;
                const bool fuse = scan_fused_merge_bytes(a.nslice, kp) <= scan_lut_bytes(M, a.nw) && !knobs().no_fused_merge;
                a.D = comb_s; a.I = comb_id; a.ldo = 2 * kp; a.out_off = ph * kp;
                if (fuse) {
                    unsigned *cnt = w.counters.as<unsigned>((size_t)nq);
                    if (w.counters_zeroed < w.counters.cap) {
                        MI_HIP(hipMemsetAsync(w.counters.p, 0, w.counters.cap, st));
                        w.counters_zeroed = w.counters.cap;
                    }
                    a.counters = cnt;
                }
                launch_scan(M, a, st);
                if (ph == 0) {
                    w.last_scan = a;
                    w.have_last_scan = true;
                    w.last_nq = nq;
                    w.last_nprobe = nprobe;
                }
                if (!fuse)
                    launch_merge(pps, ppid, a.nslice, kp, (int64_t)a.nslice * kp, nq, kp, comb_s, comb_id, 2 * kp, ph * kp, nullptr, nullptr, st);
            }
            launch_merge(comb_s, comb_id, 2, kp, (int64_t)2 * kp, nq, kp, Ddev, Idev, k, 0, nullptr, nullptr, st);
            if ((h->prune_calls.fetch_add(1, std::memory_order_relaxed) & 3u) == 0)
                MI_HIP(hipMemcpyAsync(h->prune_seen, pstats, PRUNE_SLOTS * 24, hipMemcpyDeviceToHost, st));
            return;
        }
    }
    for (int pass = 0; pass < npass; ++pass) {
        const int kp = std::min(64, k - pass * 64);
        ScanArgs a{};
        a.lut = lut; a.coarse_dis = cscan;
        a.tnorm = l2 ? h->d_tnorm.get<float>() : nullptr;
        a.p_goff = pt.p_goff; a.p_len = pt.p_len; a.p_prefix = pt.p_prefix;
        a.codes = h->d_codes.get<uint8_t>(); a.ids = h->d_ids.get<int64_t>();
        a.part_s = ps; a.part_id = pid;
        a.bound_s = pass ? bs : nullptr; a.bound_id = pass ? bid : nullptr;
        a.nq = (int)nq; a.nprobe = nprobe; a.nslice = nslice; a.k = kp; a.by_residual = l2 ? 1 : h->by_residual;
        a.nw = scan_nw;
        a.debug = 0;
        a.ts = nullptr;
        // the last slice of each query merges the partial lists in-kernel when
        // they fit in the LUT's LDS region; otherwise a separate merge kernel
        const bool fuse = scan_fused_merge_bytes(nslice, kp) <= scan_lut_bytes(M, scan_nw) &&
                          !knobs().no_fused_merge;
        a.counters = nullptr; a.D = Ddev; a.I = Idev; a.ldo = k; a.out_off = pass * 64;
        a.next_bound_s = npass > 1 ? bs : nullptr; a.next_bound_id = npass > 1 ? bid : nullptr;
        a.prune_A = pruneA; a.prune_sorted = prune_sorted; a.prune_stats = pruneA ? pstats : nullptr;
        if (fuse) {
            const size_t cb = (size_t)nq * sizeof(unsigned);
            unsigned *cnt = w.counters.as<unsigned>((size_t)nq);
            if (w.counters_zeroed < w.counters.cap) {  // fresh allocation: zero it once
                MI_HIP(hipMemsetAsync(w.counters.p, 0, w.counters.cap, st));
                w.counters_zeroed = w.counters.cap;
            }
            (void)cb;
            a.counters = cnt;
        }
        launch_scan(M, a, st);
        if (a.prune_stats && (h->prune_calls.fetch_add(1, std::memory_order_relaxed) & 3u) == 0)
            MI_HIP(hipMemcpyAsync(h->prune_seen, pstats, PRUNE_SLOTS * 24, hipMemcpyDeviceToHost, st));
        if (pass == 0) {
            w.last_scan = a;
            w.have_last_scan = true;
            w.last_nq = nq;
            w.last_nprobe = nprobe;
        }
        if (!fuse)
            launch_merge(ps, pid, nslice, kp, (int64_t)nslice * kp, nq, kp, Ddev, Idev, k, pass * 64,
                         npass > 1 ? bs : nullptr, npass > 1 ? bid : nullptr, st);
    }
    l2_finish();
}

static int64_t query_chunk_size(const mi_index *h) {
    int64_t c = ((int64_t)1 << 28) / std::max(1, h->nlist);  // scores <= 1 GiB
    c = std::min<int64_t>(c, 4096);                          // LUT <= 4096 * M KiB
    return std::max<int64_t>(c, 64);
}

int mi_index_search(mi_index *h, int64_t nq, const float *q, int k, int nprobe, float *D, int64_t *I,
                    void *stream) {
    return guard([&] {
        MI_REQUIRE(h && (nq == 0 || (q && D && I)), "null argument");
        MI_REQUIRE(k >= 1 && k <= SELP_CAP, "k must be in [1, 8192]");
        MI_REQUIRE(nprobe >= 1, "nprobe must be >= 1");
        require_trained(h);
        if (nq == 0) return;
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        WsLease lease = lease_ws(h, stream, true);
        SearchWS &w = lease.w;
        nprobe = std::min(nprobe, h->nlist);
        const bool qd = is_device_ptr(q), Dd = is_device_ptr(D), Id = is_device_ptr(I);
        const int64_t chunk = query_chunk_size(h);
        for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
            const int64_t m = std::min(chunk, nq - c0);
            const float *qs = q + (size_t)c0 * h->d;
            if (!qd) qs = static_cast<const float *>(to_device(qs, (size_t)m * h->d * 4, w.q, st));
            float *Dc = Dd ? D + (size_t)c0 * k : w.D.as<float>((size_t)m * k);
            int64_t *Ic = Id ? I + (size_t)c0 * k : w.I.as<int64_t>((size_t)m * k);
            search_chunk(h, w, m, qs, k, nprobe, Dc, Ic, st, nullptr, nullptr, nullptr, false);
            if (!Dd) MI_HIP(hipMemcpyAsync(D + (size_t)c0 * k, Dc, (size_t)m * k * 4, hipMemcpyDeviceToHost, st));
            if (!Id) MI_HIP(hipMemcpyAsync(I + (size_t)c0 * k, Ic, (size_t)m * k * 8, hipMemcpyDeviceToHost, st));
            if (!qd || !Dd || !Id) MI_HIP(hipStreamSynchronize(st));
        }
    });
}

int mi_index_search_candidates(mi_index *h, int64_t nq, const float *q, int kc, int nprobe, int64_t *I, void *stream) {
    return guard([&] {
        MI_REQUIRE(h && (nq == 0 || (q && I)), "null argument");
        MI_REQUIRE(kc >= 1 && kc <= SELP_CAP, "kc must be in [1, 8192]");
        MI_REQUIRE(nprobe >= 1, "nprobe must be >= 1");
        require_trained(h);
        if (nq == 0) return;
        MI_REQUIRE(is_device_ptr(q) && is_device_ptr(I), "mi_index_search_candidates: device pointers only");
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        WsLease lease = lease_ws(h, stream, true);
        SearchWS &w = lease.w;
        nprobe = std::min(nprobe, h->nlist);
        const int64_t chunk = query_chunk_size(h);
        for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
            const int64_t m = std::min(chunk, nq - c0);
            // kc <= 64 (selected inside the scan) and METRIC_L2 (scores are finished in place) keep the sorted path
            const bool set = kc > 64 && h->metric == MI_METRIC_INNER_PRODUCT && !knobs().no_allscores;
            float *Dc = set ? nullptr : w.D.as<float>((size_t)m * kc);
            search_chunk(h, w, m, q + (size_t)c0 * h->d, kc, nprobe, Dc, I + (size_t)c0 * kc, st, nullptr, nullptr, nullptr, false,
                         nullptr, nullptr, set);
        }
    });
}

int mi_index_coarse_lut(mi_index *h, int64_t nq, const float *q, int nprobe, int32_t *cI, float *cD,
                        float *lut) {
    return guard([&] {
        MI_REQUIRE(h && (nq == 0 || q), "null argument");
        MI_REQUIRE(nprobe >= 1, "nprobe must be >= 1");
        require_trained(h);
        if (nq == 0) return;
        DeviceGuard dg(h->device);
        WsLease lease = lease_ws(h, nullptr, false);
        SearchWS &w = lease.w;
        nprobe = std::min(nprobe, h->nlist);
        const int64_t chunk = query_chunk_size(h);
        const bool qd = is_device_ptr(q);
        for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
            const int64_t m = std::min(chunk, nq - c0);
            const float *qs = q + (size_t)c0 * h->d;
            if (!qd) qs = static_cast<const float *>(to_device(qs, (size_t)m * h->d * 4, w.q, nullptr));
            search_chunk(h, w, m, qs, 1, nprobe, nullptr, nullptr, nullptr,
                         cI ? cI + (size_t)c0 * nprobe : nullptr, cD ? cD + (size_t)c0 * nprobe : nullptr,
                         lut ? lut + (size_t)c0 * h->M * 256 : nullptr, true);
            MI_HIP(hipStreamSynchronize(nullptr));
        }
    });
}

int mi_index_coarse_slice(mi_index *h, int64_t nq, const float *q, int nprobe, int list_lo, int list_hi,
                          int32_t *coarse_I, float *coarse_D, void *stream) {
    return guard([&] {
        MI_REQUIRE(h && q && coarse_I && coarse_D, "null argument");
        MI_REQUIRE(0 <= list_lo && list_lo < list_hi && list_hi <= h->nlist, "bad centroid slice");
        MI_REQUIRE(nprobe >= 1, "nprobe must be >= 1");
        MI_REQUIRE(h->has_coarse, "no coarse centroids set");
        MI_REQUIRE(h->metric == MI_METRIC_INNER_PRODUCT, "mi_index_coarse_slice: METRIC_INNER_PRODUCT only");
        MI_REQUIRE(is_device_ptr(q) && is_device_ptr(coarse_I) && is_device_ptr(coarse_D),
                   "mi_index_coarse_slice: device pointers only");
        if (nq == 0) return;
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        WsLease lease = lease_ws(h, stream, false);
        SearchWS &w = lease.w;
        const int n = list_hi - list_lo;
        const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(4096, ((int64_t)1 << 28) / n));
        for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
            const int64_t m = std::min(chunk, nq - c0);
            float *scores = w.scores.as<float>((size_t)m * n);
            // big slices: the two-stage quantiser on the slice's rows of the f16 image (bit-identical
            // result; max |c| over the whole table is a valid bound for the slice)
            if (h->cent16_ok && two_stage_wanted(m, n, h->d, nprobe) && nprobe <= n) {
                launch_two_stage(q + (size_t)c0 * h->d, m, h->centroids.get<float>() + (size_t)list_lo * h->d,
                                 static_cast<const f16_t *>(h->cent16.p) + (size_t)list_lo * h->d, n, h->d, nprobe, h->cmax,
                                 h->cscale, scores, w.q16, w.qscale, w.rstats, coarse_I + (size_t)c0 * nprobe,
                                 coarse_D + (size_t)c0 * nprobe, ProbeTables{}, st, list_lo);
                continue;
            }
            launch_gemm(q + (size_t)c0 * h->d, m, h->centroids.get<float>() + (size_t)list_lo * h->d, n, h->d,
                        scores, n, st);
            // K = nprobe even if the slice is smaller: the tail is -1 / -FLT_MAX padded
            launch_select(scores, n, m, n, nprobe, coarse_I + (size_t)c0 * nprobe, nullptr,
                          coarse_D + (size_t)c0 * nprobe, st, ProbeTables{}, list_lo);
        }
    });
}

int mi_index_search_preassigned(mi_index *h, int64_t nq, const float *q, int k, int nprobe,
                                const int32_t *coarse_I, const float *coarse_D, float *D, int64_t *I,
                                void *stream) {
    return guard([&] {
        MI_REQUIRE(h && (nq == 0 || (q && D && I && coarse_I && coarse_D)), "null argument");
        MI_REQUIRE(k >= 1 && k <= SELP_CAP, "k must be in [1, 8192]");
        MI_REQUIRE(nprobe >= 1, "nprobe must be >= 1");
        require_trained(h);
        if (nq == 0) return;
        MI_REQUIRE(is_device_ptr(q) && is_device_ptr(D) && is_device_ptr(I) && is_device_ptr(coarse_I) &&
                       is_device_ptr(coarse_D),
                   "mi_index_search_preassigned: device pointers only");
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        WsLease lease = lease_ws(h, stream, true);
        SearchWS &w = lease.w;
        const int64_t chunk = query_chunk_size(h);
        for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
            const int64_t m = std::min(chunk, nq - c0);
            search_chunk(h, w, m, q + (size_t)c0 * h->d, k, nprobe, D + (size_t)c0 * k, I + (size_t)c0 * k, st,
                         nullptr, nullptr, nullptr, false, coarse_I + (size_t)c0 * nprobe,
                         coarse_D + (size_t)c0 * nprobe);
        }
    });
}

namespace {
// Scratch rows of the merge's selection tier for the two handle-less entry points, one buffer per (device, stream) of the
// calling thread: a host thread that round-robins several streams (the documented serving pattern) has merges in flight
// on all of them, and a buffer shared between streams -- or left on the first device the thread used -- would be written
// by two launches at once.  Never shrinks; at most 64 pairs per thread (the oldest entry is recycled after a device sync).
DevBuf *merge_scratch(int device, void *stream) {
    struct Slot { int device; void *stream; std::unique_ptr<DevBuf> buf; };
    static thread_local std::vector<Slot> slots;
    for (auto &sl : slots)
        if (sl.device == device && sl.stream == stream) return sl.buf.get();
    if (slots.size() >= 64) {                            // recycle the oldest: nothing on ITS device may still read it
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(slots.front().device);
        (void)hipDeviceSynchronize();
        slots.erase(slots.begin());                      // (frees on the slot's device)
        (void)hipSetDevice(cur);
    }
    slots.push_back(Slot{device, stream, std::make_unique<DevBuf>()});
    return slots.back().buf.get();
}
}  // namespace

int mi_merge_topk(int device, int nparts, int64_t nq, int k, const float *D_parts,
                  const int64_t *I_parts, float *D, int64_t *I, void *stream) {
    return guard([&] {
        MI_REQUIRE(nparts >= 1 && k >= 1 && nq >= 0, "bad sizes");
        MI_REQUIRE(D_parts && I_parts && D && I, "null argument");
        if (nq == 0) return;
        DeviceGuard dg(device);
        hipStream_t st = as_stream(stream);
        const bool all_dev = is_device_ptr(D_parts) && is_device_ptr(I_parts) && is_device_ptr(D) && is_device_ptr(I);
        const size_t n_in = (size_t)nparts * nq * k, n_out = (size_t)nq * k;
        if (all_dev) {
            // the selection tier's rows (long lists): kept per (device, stream), so the call only enqueues
            launch_merge(D_parts, I_parts, nparts, nq * k, k, nq, k, D, I, k, 0, nullptr, nullptr, st, -1, IdMap{}, merge_scratch(device, stream));
            return;
        }
        MI_REQUIRE(!is_device_ptr(D_parts) && !is_device_ptr(I_parts) && !is_device_ptr(D) && !is_device_ptr(I),
                   "mi_merge_topk: pointers must be all host or all device");
        DevBuf dD, dI, oD, oI;
        MI_HIP(hipMemcpyAsync(dD.reserve(n_in * 4), D_parts, n_in * 4, hipMemcpyHostToDevice, st));
        MI_HIP(hipMemcpyAsync(dI.reserve(n_in * 8), I_parts, n_in * 8, hipMemcpyHostToDevice, st));
        launch_merge(dD.get<float>(), dI.get<int64_t>(), nparts, nq * k, k, nq, k,
                     oD.as<float>(n_out), oI.as<int64_t>(n_out), k, 0, nullptr, nullptr, st);
        MI_HIP(hipMemcpyAsync(D, oD.p, n_out * 4, hipMemcpyDeviceToHost, st));
        MI_HIP(hipMemcpyAsync(I, oI.p, n_out * 8, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
    });
}

int mi_merge_topk_gathered(int device, int nparts, int64_t nq, int k, const void *gathered, int64_t blk_bytes,
                           int64_t id_mul, int64_t id_add, int64_t id_step, int64_t q_lo, int64_t nq_out, float *D,
                           int64_t *I, void *stream) {
    return guard([&] {
        MI_REQUIRE(nparts >= 1 && k >= 1 && nq >= 0, "bad sizes");
        MI_REQUIRE(q_lo >= 0 && nq_out >= 0 && q_lo + nq_out <= nq, "merge: query slice out of range");
        MI_REQUIRE(gathered && D && I, "null argument");
        const int64_t d_bytes = ((nq * k * 4 + 7) / 8) * 8;
        MI_REQUIRE(blk_bytes >= d_bytes + nq * k * 8 && blk_bytes % 8 == 0, "merge: block size does not hold nq*k (f32, i64) pairs");
        MI_REQUIRE(is_device_ptr(gathered) && is_device_ptr(D) && is_device_ptr(I), "mi_merge_topk_gathered: device pointers only");
        if (nq_out == 0) return;
        DeviceGuard dg(device);
        const char *base = static_cast<const char *>(gathered);
        IdMap im;
        im.mul = id_mul; im.add = id_add; im.step = id_step;
        launch_merge(reinterpret_cast<const float *>(base) + (size_t)q_lo * k,
                     reinterpret_cast<const int64_t *>(base + d_bytes) + (size_t)q_lo * k, nparts, blk_bytes / 4, k, nq_out, k,
                     D, I, k, 0, nullptr, nullptr, as_stream(stream), blk_bytes / 8, im, merge_scratch(device, stream));   // (the selection tier and the > 160 KiB tier use it)
    });
}

// ---- vector-sharded search over RCCL (one process per GPU): the exchange step in the C ABI ----
// RCCL is bound at run time (dlopen): a host that never shards does not need it, and a Python
// host passes the path of the librccl torch already loaded so that one copy serves both.

}  // extern "C"

#include <dlfcn.h>

namespace {

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ struct Id128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct Id128 { char b[128]; };

RcclApi &rccl(const char *path) {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (api.lib) return api;
    const char *cands[] = {path, std::getenv("MI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *c : cands) {
        if (!c || !*c) continue;
        api.lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) {
        const char *why = dlerror();                     // one call: dlerror() clears the message it returns
        throw Error(std::string("cannot load RCCL (librccl.so): ") + (why ? why : "not found"));
    }
    auto sym = [&](const char *n) {
        void *p = dlsym(api.lib, n);
        if (!p) {
            dlclose(api.lib);
            api = RcclApi{};                             // a half-bound table must not look loaded to the next call
            throw Error(std::string("RCCL symbol missing: ") + n);
        }
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    return api;
}

void nccl_check(RcclApi &r, int rc, const char *what) {
    if (rc != 0) throw Error(std::string(what) + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
}

}  // namespace

struct mi_shards {
    mi_index *local = nullptr;
    mi_flat *refine = nullptr;     // optional: the shard's raw vectors (IndexRefine over the local index)
    int k_factor = 1;
    int rank = 0, world = 1, device = 0;
    IdMap im;
    void *comm = nullptr;
    // exchange buffers, one set per stream a search is issued on (two batches in flight on two streams must not share
    // a send buffer).  Collectives on one communicator must be issued in the same order on every rank, so ONE host
    // thread drives a mi_shards handle; the set is still leased under a mutex.
    struct Bufs {
        std::mutex mu;
        DevBuf send, recv, cand_D, cand_I, big;
    };
    std::vector<std::pair<void *, std::unique_ptr<Bufs>>> bufs;
    std::mutex mu;
};

extern "C" {

int mi_shards_unique_id(const char *rccl_lib, void *id128) {
    return guard([&] {
        MI_REQUIRE(id128, "null argument");
        RcclApi &r = rccl(rccl_lib);
        nccl_check(r, r.GetUniqueId(id128), "ncclGetUniqueId");
    });
}

int mi_shards_create(mi_index *local, mi_flat *refine, int k_factor, int rank, int world, const void *id128,
                     const char *rccl_lib, int64_t id_mul, int64_t id_add, int64_t id_step, mi_shards **out) {
    return guard([&] {
        MI_REQUIRE(local && id128 && out, "null argument");
        MI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
        MI_REQUIRE(!refine || k_factor >= 1, "k_factor must be >= 1");
        RcclApi &r = rccl(rccl_lib);
        DeviceGuard dg(local->device);
        auto s = std::make_unique<mi_shards>();
        s->local = local; s->refine = refine; s->k_factor = refine ? k_factor : 1;
        s->rank = rank; s->world = world; s->device = local->device;
        s->im.mul = id_mul; s->im.add = id_add; s->im.step = id_step;
        Id128 id;
        std::memcpy(id.b, id128, 128);
        nccl_check(r, r.CommInitRank(&s->comm, world, id, rank), "ncclCommInitRank");
        *out = s.release();
    });
}

int mi_shards_destroy(mi_shards *s) {
    return guard([&] {
        if (!s) return;
        DeviceGuard dg(s->device);
        if (s->comm) (void)rccl(nullptr).CommDestroy(s->comm);
        delete s;
    });
}

int mi_shards_search(mi_shards *s, int64_t nq, const float *q, int k, int nprobe, float *D, int64_t *I, void *stream) {
    return guard([&] {
        MI_REQUIRE(s && (nq == 0 || (q && D && I)), "null argument");
        MI_REQUIRE(k >= 1 && k <= 4096, "k must be in [1, 4096]");
        if (nq == 0) return;
        MI_REQUIRE(is_device_ptr(q) && is_device_ptr(D) && is_device_ptr(I), "mi_shards_search: device pointers only");
        DeviceGuard dg(s->device);
        hipStream_t st = as_stream(stream);
        const size_t d_bytes = (((size_t)nq * k * 4 + 7) / 8) * 8, blk = d_bytes + (size_t)nq * k * 8;
        mi_shards::Bufs *b = nullptr;
        {
            std::lock_guard<std::mutex> hl(s->mu);
            for (auto &kv : s->bufs)
                if (kv.first == stream) b = kv.second.get();
            if (!b) {
                MI_REQUIRE(s->bufs.size() < MAX_STREAMS, "too many distinct streams on one shards handle (max 64)");
                s->bufs.emplace_back(stream, std::make_unique<mi_shards::Bufs>());
                b = s->bufs.back().second.get();
            }
        }
        std::lock_guard<std::mutex> bl(b->mu);
        char *send = static_cast<char *>(b->send.reserve(blk));
        char *recv = static_cast<char *>(b->recv.reserve(blk * s->world));
        float *Dl = reinterpret_cast<float *>(send);
        int64_t *Il = reinterpret_cast<int64_t *>(send + d_bytes);
        // this shard, all queries: straight into the two halves of the send buffer
        if (s->refine) {
            const int kb = k * s->k_factor;
            MI_REQUIRE(kb <= SELP_CAP, "k * k_factor must be <= 8192");
            float *cD = b->cand_D.as<float>((size_t)nq * kb);
            int64_t *cI = b->cand_I.as<int64_t>((size_t)nq * kb);
            (void)cD;
            if (mi_index_search_candidates(s->local, nq, q, kb, nprobe, cI, stream)) throw Error(last_error());
            if (mi_flat_rerank(s->refine, nq, q, kb, cI, k, Dl, Il, stream)) throw Error(last_error());
        } else {
            if (mi_index_search(s->local, nq, q, k, nprobe, Dl, Il, stream)) throw Error(last_error());
        }
        // the path's one exchange step
        Range stage("mi_ivfpq:exchange+merge");
        RcclApi &r = rccl(nullptr);
        nccl_check(r, r.AllGather(send, recv, blk, /* ncclInt8 */ 0, s->comm, st), "ncclAllGather");
        launch_merge(reinterpret_cast<const float *>(recv), reinterpret_cast<const int64_t *>(recv + d_bytes), s->world,
                     (int64_t)(blk / 4), k, nq, k, D, I, k, 0, nullptr, nullptr, st, (int64_t)(blk / 8), s->im, &b->big);
    });
}

// ---- write_index / read_index: faiss's binary IndexIVFPQ format ---------------------
// (IwPQ + IndexFlat quantiser + ArrayInvertedLists `ilar` or OnDiskInvertedLists `ilod`: the
// reference's index.faiss + ondisk.ivfdata pair, Makefile:11-12.)  Layout restated from
// faiss's documented serialisation -- abstracts-search_amd/faiss_io.py holds the field-by-
// field description and the same validation caveat.  Lists stream between the device log and
// the file in bounded slabs: the host never holds more than one slab.

}  // extern "C"

namespace {

struct FileW {
    FILE *f = nullptr;
    std::string name;
    explicit FileW(const char *path) : f(std::fopen(path, "wb")), name(path) {
        if (!f) throw Error("cannot open " + name + " for writing");
    }
    ~FileW() { if (f) std::fclose(f); }
    void raw(const void *p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) throw Error("short write to " + name);
    }
    template <class T> void one(T v) { raw(&v, sizeof(T)); }
    void cc(const char *four) { raw(four, 4); }
    void close() {
        if (f && std::fclose(f) != 0) { f = nullptr; throw Error("error closing " + name); }
        f = nullptr;
    }
};

struct FileR {
    FILE *f = nullptr;
    std::string name;
    explicit FileR(const char *path) : f(std::fopen(path, "rb")), name(path) {
        if (!f) throw Error("cannot open " + name);
    }
    ~FileR() { if (f) std::fclose(f); }
    void raw(void *p, size_t n) {
        if (n && std::fread(p, 1, n, f) != n) throw Error(name + ": truncated file");
    }
    template <class T> T one() { T v; raw(&v, sizeof(T)); return v; }
    std::string cc() { char c[4]; raw(c, 4); return std::string(c, 4); }
    void seek(uint64_t off) {
        if (fseeko(f, (off_t)off, SEEK_SET) != 0) throw Error(name + ": seek failed");
    }
};

void write_index_header(FileW &w, int d, int64_t ntotal, bool trained, int metric) {
    w.one<int32_t>(d); w.one<int64_t>(ntotal); w.one<int64_t>((int64_t)1 << 20); w.one<int64_t>((int64_t)1 << 20);
    w.one<uint8_t>(trained ? 1 : 0); w.one<int32_t>(metric);
}

struct IndexHeader { int d; int64_t ntotal; bool trained; int metric; };
IndexHeader read_index_header(FileR &r) {
    IndexHeader h{};
    h.d = r.one<int32_t>(); h.ntotal = r.one<int64_t>(); (void)r.one<int64_t>(); (void)r.one<int64_t>();
    h.trained = r.one<uint8_t>() != 0; h.metric = r.one<int32_t>();
    if (h.metric > 1) (void)r.one<float>();
    return h;
}

// lists [lo, hi) such that a slab stays under ~256 MiB of codes
int slab_end(const std::vector<int32_t> &len, int lo, int M) {
    int64_t rows = 0;
    int hi = lo;
    while (hi < (int)len.size() && (hi == lo || (rows + len[(size_t)hi]) * (int64_t)(M + 8) <= ((int64_t)256 << 20))) rows += len[(size_t)hi++];
    return hi;
}

}  // namespace

extern "C" {

int mi_index_save(mi_index *h, const char *fname, const char *ondisk_data) {
    return guard([&] {
        MI_REQUIRE(h && fname, "null argument");
        DeviceGuard dg(h->device);
        refresh_len(h);
        const int d = h->d, nlist = h->nlist, M = h->M;
        const bool trained = h->has_coarse && h->has_codebook;
        FileW w(fname);
        w.cc("IwPQ");
        write_index_header(w, d, h->ntotal, trained, h->metric);
        w.one<uint64_t>((uint64_t)nlist); w.one<uint64_t>((uint64_t)std::max(1, h->nprobe_attr));
        w.cc(h->metric == MI_METRIC_INNER_PRODUCT ? "IxFI" : "IxF2");
        write_index_header(w, d, h->has_coarse ? nlist : 0, true, h->metric);
        {
            std::vector<float> c(h->has_coarse ? (size_t)nlist * d : 0);
            if (h->has_coarse) MI_HIP(hipMemcpy(c.data(), h->centroids.p, c.size() * 4, hipMemcpyDeviceToHost));
            w.one<uint64_t>(c.size()); w.raw(c.data(), c.size() * 4);
        }
        w.one<int8_t>(0); w.one<uint64_t>(0);                         // DirectMap::NoMap, empty array
        w.one<uint8_t>(h->by_residual ? 1 : 0); w.one<uint64_t>((uint64_t)M);
        w.one<uint64_t>((uint64_t)d); w.one<uint64_t>((uint64_t)M); w.one<uint64_t>(8);
        {
            std::vector<float> c(h->has_codebook ? (size_t)M * 256 * h->dsub : 0);
            if (h->has_codebook) MI_HIP(hipMemcpy(c.data(), h->codebook.p, c.size() * 4, hipMemcpyDeviceToHost));
            w.one<uint64_t>(c.size()); w.raw(c.data(), c.size() * 4);
        }
        std::vector<uint8_t> cbuf;
        std::vector<int64_t> ibuf;
        auto stream_lists = [&](FileW &out) {   // per list: codes then ids
            for (int lo = 0; lo < nlist;) {
                const int hi = slab_end(h->h_len, lo, M);
                int64_t rows = 0;
                for (int l = lo; l < hi; ++l) rows += h->h_len[(size_t)l];
                if (rows) {
                    cbuf.resize((size_t)rows * M); ibuf.resize((size_t)rows);
                    if (mi_index_export_lists(h, lo, hi, cbuf.data(), ibuf.data())) throw Error(last_error());
                    int64_t o = 0;
                    for (int l = lo; l < hi; ++l) {
                        const int64_t k = h->h_len[(size_t)l];
                        out.raw(cbuf.data() + (size_t)o * M, (size_t)k * M);
                        out.raw(ibuf.data() + o, (size_t)k * 8);
                        o += k;
                    }
                }
                lo = hi;
            }
        };
        if (!ondisk_data) {
            w.cc("ilar");
            w.one<uint64_t>((uint64_t)nlist); w.one<uint64_t>((uint64_t)M);
            int64_t nz = 0;
            for (int l = 0; l < nlist; ++l) nz += h->h_len[(size_t)l] != 0;
            if (nz > nlist / 2) {
                w.cc("full"); w.one<uint64_t>((uint64_t)nlist);
                for (int l = 0; l < nlist; ++l) w.one<uint64_t>((uint64_t)h->h_len[(size_t)l]);
            } else {
                w.cc("sprs"); w.one<uint64_t>((uint64_t)2 * (uint64_t)nz);   // faiss WRITEVECTORs the flattened {list, size} pairs
                for (int l = 0; l < nlist; ++l)
                    if (h->h_len[(size_t)l]) { w.one<uint64_t>((uint64_t)l); w.one<uint64_t>((uint64_t)h->h_len[(size_t)l]); }
            }
            stream_lists(w);
        } else {
            FileW dw(ondisk_data);
            stream_lists(dw);
            dw.close();
            w.cc("ilod");
            w.one<uint64_t>((uint64_t)nlist); w.one<uint64_t>((uint64_t)M);
            w.one<uint64_t>((uint64_t)nlist);
            uint64_t pos = 0;
            for (int l = 0; l < nlist; ++l) {                         // {size, capacity, offset}
                const uint64_t k = (uint64_t)h->h_len[(size_t)l];
                w.one<uint64_t>(k); w.one<uint64_t>(k); w.one<uint64_t>(pos);
                pos += k * (uint64_t)(M + 8);
            }
            w.one<uint64_t>(0);                                       // free slots
            std::string base(ondisk_data);
            const size_t sl = base.find_last_of('/');
            if (sl != std::string::npos) base = base.substr(sl + 1);
            w.one<uint64_t>(base.size()); w.raw(base.data(), base.size());
            w.one<uint64_t>(pos);
        }
        w.close();
    });
}

int mi_index_load(const char *fname, int device, mi_index **out) { return mi_index_load_at(fname, 0, device, out); }

int mi_index_load_at(const char *fname, int64_t offset, int device, mi_index **out) {
    mi_index *h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(fname && out && offset >= 0, "null argument");
        FileR r(fname);
        if (offset) r.seek((uint64_t)offset);
        const std::string name(fname);
        const std::string cc = r.cc();
        if (cc == "IvPQ" || cc == "IvQR" || cc == "IwQR") throw Error(name + ": " + cc + " (legacy / refined IVFPQ) is not supported, only IwPQ");
        if (cc != "IwPQ") throw Error(name + ": fourcc '" + cc + "' is not an IndexIVFPQ (IwPQ)");
        const IndexHeader ih = read_index_header(r);
        const uint64_t nlist = r.one<uint64_t>(), nprobe = r.one<uint64_t>();
        MI_REQUIRE(nlist > 0 && nlist < ((uint64_t)1 << 31), "implausible nlist");
        std::string qcc = r.cc();
        if (qcc == "IHNf") {
            // [PRIOR layout: faiss write_index(IndexHNSW) = fourcc, index header, write_HNSW, storage index]  "IVF65536_HNSW32,..."
            // puts an IndexHNSWFlat in front of the lists.  Its flat storage IS the centroid table: it is taken over and searched
            // exactly (the probes of an exact search are what the HNSW graph approximates -- a superset in recall terms, not the
            // same list set faiss would visit); the graph itself is skipped.
            (void)read_index_header(r);
            const size_t elem[5] = {8, 4, 4, 8, 4};   // assign_probas f64, cum_nneighbor_per_level i32, levels i32, offsets u64, neighbors i32
            for (int v = 0; v < 5; ++v) {
                const uint64_t n = r.one<uint64_t>();
                MI_REQUIRE(n < ((uint64_t)1 << 40), "implausible HNSW vector length");
                r.seek((uint64_t)ftello(r.f) + n * elem[v]);
            }
            for (int v = 0; v < 5; ++v) (void)r.one<int32_t>();   // entry_point, max_level, efConstruction, efSearch, upper_beam (deprecated)
            qcc = r.cc();
            if (qcc != "IxFI" && qcc != "IxF2" && qcc != "IxFl") throw Error(name + ": HNSW coarse quantiser over '" + qcc + "' storage: only flat storage (IndexHNSWFlat) is read");
        } else if (qcc == "IHNp" || qcc == "IHNs" || qcc == "IHN2" || qcc == "IHNc") {
            throw Error(name + ": coarse quantiser '" + qcc + "' (HNSW over compressed storage) holds no exact centroid table and is not supported; IHNf (IndexHNSWFlat) is");
        }
        if (qcc != "IxFI" && qcc != "IxF2" && qcc != "IxFl") throw Error(name + ": coarse quantiser '" + qcc + "' is not an IndexFlat");
        const IndexHeader qh = read_index_header(r);
        const uint64_t ncf = r.one<uint64_t>();
        if (qh.d != ih.d || ncf != (uint64_t)qh.ntotal * (uint64_t)ih.d) throw Error(name + ": quantiser shape mismatch");
        if (ih.trained && (uint64_t)qh.ntotal != nlist) throw Error(name + ": quantiser holds " + std::to_string(qh.ntotal) + " centroids, nlist is " + std::to_string(nlist));
        std::vector<float> cent((size_t)ncf);
        r.raw(cent.data(), cent.size() * 4);
        (void)r.one<int8_t>();
        { const uint64_t n = r.one<uint64_t>(); std::vector<int64_t> dm((size_t)n); r.raw(dm.data(), dm.size() * 8); }
        const bool by_res = r.one<uint8_t>() != 0;
        const uint64_t code_size = r.one<uint64_t>();
        const uint64_t pd = r.one<uint64_t>(), M = r.one<uint64_t>(), nbits = r.one<uint64_t>();
        const uint64_t ncb = r.one<uint64_t>();
        if (pd != (uint64_t)ih.d || M == 0 || ih.d % (int)M || nbits != 8 || code_size != M)
            throw Error(name + ": unsupported PQ (d=" + std::to_string(pd) + ", M=" + std::to_string(M) + ", nbits=" + std::to_string(nbits) + ", code_size=" + std::to_string(code_size) + ")");
        if (ncb != 0 && ncb != M * 256 * ((uint64_t)ih.d / M)) throw Error(name + ": PQ codebook has " + std::to_string(ncb) + " floats");
        std::vector<float> cb((size_t)ncb);
        r.raw(cb.data(), cb.size() * 4);
        if (mi_index_create(ih.d, (int)nlist, (int)M, 8, ih.metric, by_res ? 1 : 0, device, &h)) throw Error(last_error());
        h->nprobe_attr = (int)std::max<uint64_t>(1, std::min<uint64_t>(nprobe, nlist));
        if (ih.trained && !cent.empty()) {
            if (mi_index_set_coarse(h, cent.data())) throw Error(last_error());
            if (!cb.empty() && mi_index_set_codebook(h, cb.data())) throw Error(last_error());
        }
        // inverted lists
        const std::string lcc = r.cc();
        std::vector<uint64_t> sizes((size_t)nlist, 0), caps, offs;
        std::unique_ptr<FileR> data;
        FileR *src = &r;
        bool ondisk = false;
        if (lcc == "il00") {
            // none
        } else if (lcc == "ilar" || lcc == "ilod") {
            const uint64_t nl = r.one<uint64_t>(), cs = r.one<uint64_t>();
            if (nl != nlist || cs != code_size) throw Error(name + ": inverted lists are " + std::to_string(nl) + " x " + std::to_string(cs) + " B");
            if (lcc == "ilar") {
                const std::string kind = r.cc();
                const uint64_t n = r.one<uint64_t>();
                if (kind == "full") {
                    if (n != nlist) throw Error(name + ": " + std::to_string(n) + " list sizes for " + std::to_string(nlist) + " lists");
                    r.raw(sizes.data(), (size_t)n * 8);
                } else if (kind == "sprs") {
                    // faiss: the count is that of the flattened words (2 per non-empty list).  Files written by this library
                    // before round 3's fix carry the number of PAIRS there: told apart by which reading adds up to ntotal
                    const off_t at = ftello(r.f);
                    auto read_pairs = [&](uint64_t npairs) -> bool {
                        std::fill(sizes.begin(), sizes.end(), 0);
                        uint64_t sum = 0;
                        for (uint64_t i = 0; i < npairs; ++i) {
                            const uint64_t l = r.one<uint64_t>(), k = r.one<uint64_t>();
                            if (l >= nlist) return false;
                            sizes[(size_t)l] = k;
                            sum += k;
                        }
                        return sum == (uint64_t)ih.ntotal;
                    };
                    bool ok = n % 2 == 0 && n / 2 <= nlist && read_pairs(n / 2);
                    if (!ok && n <= nlist) {                             // the legacy count
                        r.seek((uint64_t)at);
                        ok = read_pairs(n);
                    }
                    if (!ok) throw Error(name + ": sparse list sizes (" + std::to_string(n) + " words) do not add up to ntotal " +
                                         std::to_string(ih.ntotal) + " under faiss's layout nor under this library's pre-round-3 one: re-save the index");
                } else throw Error(name + ": list size encoding '" + kind + "'");
            } else {
                ondisk = true;
                const uint64_t n = r.one<uint64_t>();
                if (n != nlist) throw Error(name + ": " + std::to_string(n) + " on-disk list records for " + std::to_string(nlist) + " lists");
                caps.resize((size_t)nlist); offs.resize((size_t)nlist);
                for (uint64_t l = 0; l < nlist; ++l) { sizes[(size_t)l] = r.one<uint64_t>(); caps[(size_t)l] = r.one<uint64_t>(); offs[(size_t)l] = r.one<uint64_t>(); }
                { const uint64_t nf = r.one<uint64_t>(); std::vector<uint64_t> fs((size_t)nf * 2); r.raw(fs.data(), fs.size() * 8); }
                const uint64_t nn = r.one<uint64_t>();
                std::string dname((size_t)nn, '\0');
                r.raw(&dname[0], (size_t)nn);
                (void)r.one<uint64_t>();
                // faiss stores the path it was written with; the file travels next to the index
                std::string dir = name;
                const size_t sl = dir.find_last_of('/');
                dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
                std::string bn = dname;
                const size_t s2 = bn.find_last_of('/');
                if (s2 != std::string::npos) bn = bn.substr(s2 + 1);
                FILE *t = std::fopen(dname.c_str(), "rb");
                std::string path = dname;
                if (!t) { path = dir + "/" + bn; t = std::fopen(path.c_str(), "rb"); }
                if (!t) throw Error(name + ": on-disk list data '" + dname + "' not found (also tried " + path + ")");
                std::fclose(t);
                data = std::make_unique<FileR>(path.c_str());
                src = data.get();
            }
        } else throw Error(name + ": inverted lists '" + lcc + "' are not supported (ilar, ilod, il00)");
        uint64_t tot = 0;
        for (uint64_t l = 0; l < nlist; ++l) {
            if (sizes[(size_t)l] >= ((uint64_t)1 << 31)) throw Error(name + ": list too long");
            tot += sizes[(size_t)l];
        }
        if (tot != (uint64_t)ih.ntotal) throw Error(name + ": lists hold " + std::to_string(tot) + " vectors, header says " + std::to_string(ih.ntotal));
        if (tot) {
            DeviceGuard dg(device);
            ensure_log_cap(h, (int64_t)tot);
            std::vector<int32_t> len32((size_t)nlist);
            for (uint64_t l = 0; l < nlist; ++l) len32[(size_t)l] = (int32_t)sizes[(size_t)l];
            std::vector<uint8_t> cbuf;
            std::vector<int64_t> ibuf;
            std::vector<int32_t> lbuf;
            for (int lo = 0; lo < (int)nlist;) {
                const int hi = slab_end(len32, lo, (int)M);
                int64_t rows = 0;
                for (int l = lo; l < hi; ++l) rows += len32[(size_t)l];
                if (rows) {
                    cbuf.resize((size_t)rows * M); ibuf.resize((size_t)rows); lbuf.resize((size_t)rows);
                    int64_t o = 0;
                    for (int l = lo; l < hi; ++l) {
                        const int64_t k = len32[(size_t)l];
                        if (!k) continue;
                        if (ondisk) {
                            if (sizes[(size_t)l] > caps[(size_t)l]) throw Error(name + ": list " + std::to_string(l) + " exceeds its capacity");
                            src->seek(offs[(size_t)l]);
                            src->raw(cbuf.data() + (size_t)o * M, (size_t)k * M);
                            src->seek(offs[(size_t)l] + caps[(size_t)l] * M);
                            src->raw(ibuf.data() + o, (size_t)k * 8);
                        } else {
                            src->raw(cbuf.data() + (size_t)o * M, (size_t)k * M);
                            src->raw(ibuf.data() + o, (size_t)k * 8);
                        }
                        std::fill(lbuf.begin() + o, lbuf.begin() + o + k, (int32_t)l);
                        o += k;
                    }
                    if (mi_index_add_codes(h, rows, lbuf.data(), cbuf.data(), ibuf.data())) throw Error(last_error());
                }
                lo = hi;
            }
        }
        *out = h;
    });
    if (rc && h) {
        const std::string keep = last_error();
        (void)mi_index_destroy(h);
        last_error() = keep;
    }
    return rc;
}

int mi_index_get_params(mi_index *h, int *d, int *nlist, int *M, int *nbits, int *metric, int *by_residual, int *nprobe) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        if (d) *d = h->d;
        if (nlist) *nlist = h->nlist;
        if (M) *M = h->M;
        if (nbits) *nbits = 8;
        if (metric) *metric = h->metric;
        if (by_residual) *by_residual = h->by_residual;
        if (nprobe) *nprobe = h->nprobe_attr;
    });
}

int mi_index_set_nprobe(mi_index *h, int nprobe) {
    return guard([&] {
        MI_REQUIRE(h && nprobe >= 1, "bad argument");
        h->nprobe_attr = nprobe;
    });
}

// ---- IndexFlatIP -----------------------------------------------------

int mi_flat_create(int d, int device, mi_flat **out) {
    return guard([&] {
        MI_REQUIRE(out != nullptr, "out is null");
        MI_REQUIRE(d > 0 && d % 4 == 0, "d must be a positive multiple of 4");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) {
            (void)hipGetLastError();
            throw Error("no HIP device available: the MI355X index has no CPU fallback");
        }
        MI_REQUIRE(device >= 0 && device < ndev, "invalid device ordinal");
        auto h = std::make_unique<mi_flat>();
        h->d = d;
        h->da = d;
        h->device = device;
        *out = h.release();
    });
}

int mi_flat_create_ex(int d, int device, int storage, mi_flat **out) {
    if (storage != MI_STORE_F32 && storage != MI_STORE_F16 && storage != MI_STORE_SQ8) {
        last_error() = "mi_flat_create_ex: storage must be MI_STORE_F32, MI_STORE_F16 or MI_STORE_SQ8";
        return 1;
    }
    int rc = mi_flat_create(d, device, out);
    if (rc == 0) (*out)->elem = storage == MI_STORE_F16 ? 2 : storage == MI_STORE_SQ8 ? 1 : 4;   // (inner product only)
    return rc;
}

int mi_flat_sq_train(mi_flat *h, int64_t n, const float *x, int merge) {
    return guard([&] {
        MI_REQUIRE(h && x && n > 0, "bad argument");
        MI_REQUIRE(h->elem == 1, "mi_flat_sq_train: not a QT_8bit store");
        MI_REQUIRE(h->ntotal == 0, "mi_flat_sq_train: the store already holds vectors encoded with the current ranges");
        DeviceGuard dg(h->device);
        const int d = h->d;
        float *lohi = h->sq_lohi.as<float>((size_t)2 * d);
        const bool have = merge && h->sq_ok;
        const int64_t chunk_rows = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)d * 4));
        const bool xdev = is_device_ptr(x);
        DevBuf part;
        bool first = !have;
        for (int64_t c0 = 0; c0 < n; c0 += chunk_rows) {
            const int64_t m = std::min(chunk_rows, n - c0);
            const float *xs = x + (size_t)c0 * d;
            if (!xdev) {
                float *stage = h->ws_q.as<float>((size_t)m * d);
                MI_HIP(hipMemcpy(stage, xs, (size_t)m * d * 4, hipMemcpyHostToDevice));
                xs = stage;
            }
            const int chunks = (int)std::min<int64_t>(1024, (m + 255) / 256);
            const int64_t rpc = (m + chunks - 1) / chunks;
            float *pp = part.as<float>((size_t)chunks * 2 * d);
            hipLaunchKernelGGL(sq8_minmax_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)chunks), dim3(256), 0, nullptr, xs, m, d, rpc, pp);
            hipLaunchKernelGGL(sq8_minmax_fold_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, nullptr, pp, chunks, d, lohi, first ? 0 : 1);
            MI_HIP(hipGetLastError());
            MI_HIP(hipStreamSynchronize(nullptr));   // the staging buffer and `part` are reused
            first = false;
        }
        hipLaunchKernelGGL(sq8_ranges_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, nullptr, lohi, d,
                           h->sq_trained.as<float>((size_t)2 * d));
        MI_HIP(hipGetLastError());
        MI_HIP(hipStreamSynchronize(nullptr));
        h->sq_ok = true;
    });
}

int mi_flat_sq_get_trained(mi_flat *h, float *trained) {
    return guard([&] {
        MI_REQUIRE(h && trained, "null argument");
        MI_REQUIRE(h->elem == 1 && h->sq_ok, "mi_flat_sq_get_trained: not a trained QT_8bit store");
        DeviceGuard dg(h->device);
        MI_HIP(hipMemcpy(trained, h->sq_trained.p, (size_t)2 * h->d * 4, hipMemcpyDeviceToHost));
    });
}

int mi_flat_sq_set_trained(mi_flat *h, const float *trained) {
    return guard([&] {
        MI_REQUIRE(h && trained, "null argument");
        MI_REQUIRE(h->elem == 1, "mi_flat_sq_set_trained: not a QT_8bit store");
        MI_REQUIRE(h->ntotal == 0, "mi_flat_sq_set_trained: the store already holds vectors encoded with the current ranges");
        DeviceGuard dg(h->device);
        const int d = h->d;
        std::vector<float> t(trained, trained + 2 * (size_t)d), lohi(2 * (size_t)d);
        for (int i = 0; i < d; ++i) { lohi[(size_t)i] = t[(size_t)i]; lohi[(size_t)d + i] = t[(size_t)i] + t[(size_t)d + i]; }
        MI_HIP(hipMemcpy(h->sq_trained.reserve(t.size() * 4), t.data(), t.size() * 4, hipMemcpyHostToDevice));
        MI_HIP(hipMemcpy(h->sq_lohi.reserve(t.size() * 4), lohi.data(), t.size() * 4, hipMemcpyHostToDevice));
        h->sq_ok = true;
    });
}

int mi_flat_sq_is_trained(mi_flat *h, int *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        *out = (h->elem != 1 || h->sq_ok) ? 1 : 0;
    });
}

int mi_flat_create_metric(int d, int metric, int device, mi_flat **out) {
    if (metric != MI_METRIC_INNER_PRODUCT && metric != MI_METRIC_L2) {
        last_error() = "mi_flat_create_metric: metric must be MI_METRIC_INNER_PRODUCT or MI_METRIC_L2";
        return 1;
    }
    int rc = mi_flat_create(d, device, out);
    if (rc == 0 && metric == MI_METRIC_L2) {
        (*out)->metric = MI_METRIC_L2;
        (*out)->da = d + 4;
    }
    return rc;
}

int mi_flat_destroy(mi_flat *h) {
    return guard([&] {
        if (!h) return;
        DeviceGuard dg(h->device);
        delete h;
    });
}

int mi_flat_add(mi_flat *h, int64_t n, const float *x) {
    return guard([&] {
        MI_REQUIRE(h && (n == 0 || x), "null argument");
        if (n == 0) return;
        DeviceGuard dg(h->device);
        const size_t row = (size_t)h->da * h->elem;
        size_t old_bytes = (size_t)h->ntotal * row, add_bytes = (size_t)n * row;
        if (old_bytes + add_bytes > h->base.cap) {
            DevBuf nb;
            nb.reserve((old_bytes + add_bytes) * 3 / 2);
            if (old_bytes) MI_HIP(hipMemcpy(nb.p, h->base.p, old_bytes, hipMemcpyDeviceToDevice));
            std::swap(nb.p, h->base.p);
            std::swap(nb.cap, h->base.cap);
        }
        char *dst = static_cast<char *>(h->base.p) + old_bytes;
        if (h->elem == 4 && h->metric == MI_METRIC_L2) {
            const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)h->d * 4));
            const bool xdev = is_device_ptr(x);
            for (int64_t c0 = 0; c0 < n; c0 += chunk) {
                const int64_t m = std::min(chunk, n - c0);
                const float *xs = x + (size_t)c0 * h->d;
                if (!xdev) {
                    float *stage = h->ws_q.as<float>((size_t)m * h->d);
                    MI_HIP(hipMemcpy(stage, xs, (size_t)m * h->d * 4, hipMemcpyHostToDevice));
                    xs = stage;
                }
                launch_augment(xs, m, h->d, h->da, 1, reinterpret_cast<float *>(dst + (size_t)c0 * row), nullptr);
            }
            MI_HIP(hipStreamSynchronize(nullptr));
        } else if (h->elem == 4) {
            MI_HIP(hipMemcpy(dst, x, add_bytes, hipMemcpyDefault));
        } else if (h->elem == 1) {
            MI_REQUIRE(h->sq_ok, "add: the QT_8bit store is not trained (mi_flat_sq_train / mi_flat_sq_set_trained)");
            const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)h->d * 4));
            const bool xdev = is_device_ptr(x);
            for (int64_t c0 = 0; c0 < n; c0 += chunk) {
                const int64_t m = std::min(chunk, n - c0);
                const float *xs = x + (size_t)c0 * h->d;
                if (!xdev) {
                    float *stage = h->ws_q.as<float>((size_t)m * h->d);
                    MI_HIP(hipMemcpy(stage, xs, (size_t)m * h->d * 4, hipMemcpyHostToDevice));
                    xs = stage;
                }
                const int64_t n4 = m * h->d / 4;
                hipLaunchKernelGGL(sq8_encode_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, nullptr, xs, n4, h->d,
                                   h->sq_trained.get<float>(), reinterpret_cast<uint8_t *>(dst + (size_t)c0 * row));
                MI_HIP(hipGetLastError());
                if (!xdev) MI_HIP(hipStreamSynchronize(nullptr));   // the staging buffer is reused
            }
            MI_HIP(hipStreamSynchronize(nullptr));
        } else {
            // QT_fp16: every component rounded to nearest-even half, no scaling (faiss ScalarQuantizer)
            const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)h->d * 4));
            const bool xdev = is_device_ptr(x);
            for (int64_t c0 = 0; c0 < n; c0 += chunk) {
                const int64_t m = std::min(chunk, n - c0);
                const float *xs = x + (size_t)c0 * h->d;
                if (!xdev) {
                    float *stage = h->ws_q.as<float>((size_t)m * h->d);
                    MI_HIP(hipMemcpy(stage, xs, (size_t)m * h->d * 4, hipMemcpyHostToDevice));
                    xs = stage;
                }
                launch_to_f16_rows(xs, m, h->d, reinterpret_cast<f16_t *>(dst + (size_t)c0 * row), nullptr, 1.f, nullptr);
            }
            MI_HIP(hipStreamSynchronize(nullptr));
        }
        h->ntotal += n;
    });
}

int mi_flat_reserve(mi_flat *h, int64_t n) {
    return guard([&] {
        MI_REQUIRE(h && n >= 0, "bad argument");
        DeviceGuard dg(h->device);
        const size_t want = (size_t)n * h->da * h->elem, old_bytes = (size_t)h->ntotal * h->da * h->elem;
        if (want <= h->base.cap) return;
        DevBuf nb;
        nb.reserve(want);
        if (old_bytes) MI_HIP(hipMemcpy(nb.p, h->base.p, old_bytes, hipMemcpyDeviceToDevice));
        std::swap(nb.p, h->base.p);
        std::swap(nb.cap, h->base.cap);
    });
}

int mi_flat_rerank(mi_flat *h, int64_t nq, const float *q, int kc, const int64_t *cand_I, int k, float *D,
                   int64_t *I, void *stream) {
    return guard([&] {
        MI_REQUIRE(h && (nq == 0 || (q && cand_I && D && I)), "null argument");
        MI_REQUIRE(k >= 1 && k <= 1024 && kc >= k && kc % k == 0, "kc must be a positive multiple of k (k <= 1024)");
        MI_REQUIRE(h->ntotal > 0, "rerank: the refine index is empty");
        MI_REQUIRE(nq < ((int64_t)1 << 24), "rerank: too many queries in one call");
        if (nq == 0) return;
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        const bool dev = is_device_ptr(q);
        MI_REQUIRE(is_device_ptr(cand_I) == dev && is_device_ptr(D) == dev && is_device_ptr(I) == dev,
                   "rerank: q, cand_I, D and I must be all host or all device pointers");
        FlatLease lease = lease_ws(h, stream);
        mi_flat::WS &w = lease.w;
        Range stage("mi_ivfpq:rerank");
        const float *qs = q;
        const int64_t *ci = cand_I;
        if (!dev) {
            // ids are validated on the host copy; device callers are trusted (ids come from
            // the base index over the same vectors)
            for (int64_t i = 0; i < nq * kc; ++i)
                MI_REQUIRE(cand_I[i] < h->ntotal, "rerank: candidate id out of range");
            qs = static_cast<const float *>(to_device(q, (size_t)nq * h->d * 4, w.q, st));
            ci = static_cast<const int64_t *>(to_device(cand_I, (size_t)nq * kc * 8, w.cand, st));
        }
        float *scores = w.rerank.as<float>((size_t)nq * kc);
        float *Dc = dev ? D : w.D.as<float>((size_t)nq * k);
        int64_t *Ic = dev ? I : w.I.as<int64_t>((size_t)nq * k);
        const bool l2 = h->metric == MI_METRIC_L2;
        if (l2) {   // augmented queries; the candidates rank by S = <q, x> - |x|^2/2, reported as squared distances
            float *qa = w.qaug.as<float>((size_t)nq * h->da);
            launch_augment(qs, nq, h->d, h->da, 0, qa, st);
            hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, qs, nq, h->d, w.qn.as<float>((size_t)nq));
            MI_HIP(hipGetLastError());
            qs = qa;
        }
        if (h->elem == 4) launch_rerank_scores<float>(qs, (int)nq, h->base.get<float>(), h->ntotal, h->da, ci, kc, scores, kc, st);
        else if (h->elem == 1) launch_rerank_sq8(qs, (int)nq, h->base.get<uint8_t>(), h->ntotal, h->d, h->sq_trained.get<float>(), ci, kc, scores, kc, w.qaug, st);
        else launch_rerank_scores<f16_t>(qs, (int)nq, h->base.get<f16_t>(), h->ntotal, h->d, ci, kc, scores, kc, st);
        // the k best under (score desc, id asc), negative ids skipped: a few results of a long list in one pass over the rows
        // (topk_rows_kernel); otherwise the candidate list as kc/k "parts" of k entries through the k-way merge
        const bool no_rows = knobs().no_topk_rows;
        if (k <= 32 && kc >= 256 && kc <= 8192 && !no_rows) {
            const int vpt = (kc + 255) / 256;
            if (vpt <= 8) hipLaunchKernelGGL((topk_rows_kernel<8>), dim3((unsigned)nq), dim3(256), 0, st, scores, ci, kc, k, Dc, Ic, (int64_t)k);
            else if (vpt <= 20) hipLaunchKernelGGL((topk_rows_kernel<20>), dim3((unsigned)nq), dim3(256), 0, st, scores, ci, kc, k, Dc, Ic, (int64_t)k);
            else hipLaunchKernelGGL((topk_rows_kernel<32>), dim3((unsigned)nq), dim3(256), 0, st, scores, ci, kc, k, Dc, Ic, (int64_t)k);
            MI_HIP(hipGetLastError());
        } else {
            launch_merge(scores, ci, kc / k, k, kc, nq, k, Dc, Ic, k, 0, nullptr, nullptr, st, -1, IdMap{}, &w.bigmerge);
        }
        if (l2) {
            hipLaunchKernelGGL(l2_flat_finish_kernel, dim3((unsigned)(((size_t)nq * k + 255) / 256)), dim3(256), 0, st, Dc, Ic,
                               w.qn.get<float>(), nq, k);
            MI_HIP(hipGetLastError());
        }
        if (!dev) {
            MI_HIP(hipMemcpyAsync(D, Dc, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
            MI_HIP(hipMemcpyAsync(I, Ic, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
            MI_HIP(hipStreamSynchronize(st));
        }
    });
}

int mi_flat_reconstruct_n(mi_flat *h, int64_t i0, int64_t n, float *out) {
    return guard([&] {
        MI_REQUIRE(h && (n == 0 || out), "null argument");
        MI_REQUIRE(i0 >= 0 && n >= 0 && i0 + n <= h->ntotal, "reconstruct_n: range out of bounds");
        if (n == 0) return;
        DeviceGuard dg(h->device);
        if (h->elem == 4 && h->da == h->d) {
            MI_HIP(hipMemcpy(out, h->base.get<float>() + (size_t)i0 * h->d, (size_t)n * h->d * 4, hipMemcpyDefault));
            return;
        }
        if (h->elem == 4) {   // augmented rows (METRIC_L2): drop the extra columns
            const bool od = is_device_ptr(out);
            float *dst = od ? out : h->ws_scores.as<float>((size_t)n * h->d);
            hipLaunchKernelGGL(unaugment_rows_kernel, dim3((unsigned)(((size_t)n * h->d + 255) / 256)), dim3(256), 0, nullptr,
                               h->base.get<float>() + (size_t)i0 * h->da, n, h->d, h->da, dst);
            MI_HIP(hipGetLastError());
            if (!od) MI_HIP(hipMemcpy(out, dst, (size_t)n * h->d * 4, hipMemcpyDeviceToHost));
            else MI_HIP(hipStreamSynchronize(nullptr));
            return;
        }
        const bool odev = is_device_ptr(out);
        float *dst = odev ? out : h->ws_scores.as<float>((size_t)n * h->d);
        if (h->elem == 1) {
            hipLaunchKernelGGL(sq8_decode_kernel, dim3((unsigned)(((size_t)n * h->d + 255) / 256)), dim3(256), 0, nullptr,
                               h->base.get<uint8_t>() + (size_t)i0 * h->d, n, h->d, h->sq_trained.get<float>(), dst);
            MI_HIP(hipGetLastError());
            if (!odev) MI_HIP(hipMemcpy(out, dst, (size_t)n * h->d * 4, hipMemcpyDeviceToHost));
            else MI_HIP(hipStreamSynchronize(nullptr));
            return;
        }
        hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)(((size_t)n * h->d + 255) / 256)), dim3(256), 0, nullptr,
                           h->base.get<f16_t>() + (size_t)i0 * h->d, (int64_t)n * h->d, dst);
        MI_HIP(hipGetLastError());
        if (!odev) MI_HIP(hipMemcpy(out, dst, (size_t)n * h->d * 4, hipMemcpyDeviceToHost));
        else MI_HIP(hipStreamSynchronize(nullptr));
    });
}

#ifdef MI_SELP_TS
int mi_debug_selp_stamps(unsigned long long *out) { return guard([&] { MI_HIP(hipDeviceSynchronize()); MI_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(selp_ts), sizeof(unsigned long long) * 8 * 4096)); }); }
#endif

int mi_flat_release_workspaces(mi_flat *h) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        DeviceGuard dg(h->device);
        std::lock_guard<std::mutex> hl(h->mu);
        MI_HIP(hipDeviceSynchronize());                  // nothing in flight reads the buffers any more
        for (auto &kv : h->ws_sets) {
            std::unique_lock<std::mutex> lk(kv.second->mu, std::try_to_lock);
            MI_REQUIRE(lk.owns_lock(), "release_workspaces: a call is running on this handle");
        }
        h->ws_sets.clear();
    });
}

int mi_index_release_workspaces(mi_index *h) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        DeviceGuard dg(h->device);
        std::lock_guard<std::mutex> hl(h->mu);
        MI_HIP(hipDeviceSynchronize());
        for (auto &kv : h->ws_sets) {
            std::unique_lock<std::mutex> lk(kv.second->mu, std::try_to_lock);
            MI_REQUIRE(lk.owns_lock(), "release_workspaces: a call is running on this handle");
        }
        h->ws_sets.clear();
    });
}

int mi_flat_get_rows(mi_flat *h, int64_t n, const int64_t *ids, void *out) {
    return guard([&] {
        MI_REQUIRE(h && (n == 0 || (ids && out)), "null argument");
        if (n == 0) return;
        DeviceGuard dg(h->device);
        // not a search-type call: it reads the store on the null stream, so it takes the handle lock and waits for whatever an
        // add() on another stream still has in flight
        std::lock_guard<std::mutex> hl(h->mu);
        MI_HIP(hipDeviceSynchronize());
        const size_t row = (size_t)h->da * h->elem;                     // bytes per stored row
        std::vector<int64_t> hid;
        const int64_t *hi = ids;
        if (is_device_ptr(ids)) {
            hid.resize((size_t)n);
            MI_HIP(hipMemcpy(hid.data(), ids, (size_t)n * 8, hipMemcpyDeviceToHost));
            hi = hid.data();
        }
        for (int64_t i = 0; i < n; ++i) MI_REQUIRE(hi[i] >= 0 && hi[i] < h->ntotal, "mi_flat_get_rows: id out of range");
        // a gather kernel into a device buffer, then one copy (a sample of a 212 GB store is a few hundred thousand rows)
        DevBuf dids, dout;
        const int64_t *d_ids = ids;
        if (!is_device_ptr(ids)) {
            MI_HIP(hipMemcpy(dids.reserve((size_t)n * 8), hi, (size_t)n * 8, hipMemcpyHostToDevice));
            d_ids = dids.get<int64_t>();
        }
        const bool od = is_device_ptr(out);
        unsigned char *dst = od ? static_cast<unsigned char *>(out) : static_cast<unsigned char *>(dout.reserve((size_t)n * row));
        const bool wide = row % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0;   // 16-byte copies only for an aligned destination
        hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)n), dim3(256), 0, nullptr, h->base.get<unsigned char>(), row, d_ids, dst, wide ? 1 : 0);
        MI_HIP(hipGetLastError());
        if (!od) MI_HIP(hipMemcpy(out, dst, (size_t)n * row, hipMemcpyDeviceToHost));
        else MI_HIP(hipStreamSynchronize(nullptr));
    });
}

int mi_flat_ntotal(mi_flat *h, int64_t *out) {
    return guard([&] {
        MI_REQUIRE(h && out, "null argument");
        *out = h->ntotal;
    });
}

int mi_flat_reset(mi_flat *h) {
    return guard([&] {
        MI_REQUIRE(h, "null argument");
        h->ntotal = 0;
    });
}

int mi_flat_search(mi_flat *h, int64_t nq, const float *q, int k, float *D, int64_t *I, void *stream) {
    return guard([&] {
        MI_REQUIRE(h && (nq == 0 || (q && D && I)), "null argument");
        MI_REQUIRE(k >= 1 && k <= 4096, "k must be in [1, 4096]");
        MI_REQUIRE(h->elem == 4, "mi_flat_search: the half-precision store serves re-ranking (mi_flat_rerank) only");
        if (nq == 0) return;
        DeviceGuard dg(h->device);
        hipStream_t st = as_stream(stream);
        const bool qd = is_device_ptr(q), Dd = is_device_ptr(D), Id = is_device_ptr(I);
        const bool l2 = h->metric == MI_METRIC_L2;
        if (h->ntotal == 0) {  // faiss: all -1
            MI_REQUIRE(!Dd && !Id, "empty flat index: host outputs only");
            for (int64_t i = 0; i < nq * k; ++i) {
                D[i] = l2 ? FLT_MAX : -FLT_MAX;
                I[i] = -1;
            }
            return;
        }
        int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(4096, ((int64_t)1 << 28) / h->ntotal));
        FlatLease lease = lease_ws(h, stream);
        mi_flat::WS &w = lease.w;
        Range stage("mi_ivfpq:flat_search");
        for (int64_t c0 = 0; c0 < nq; c0 += chunk) {
            const int64_t m = std::min(chunk, nq - c0);
            const float *qs = q + (size_t)c0 * h->d;
            if (!qd) qs = static_cast<const float *>(to_device(qs, (size_t)m * h->d * 4, w.q, st));
            float *scores = w.scores.as<float>((size_t)m * h->ntotal);
            float *Dc = Dd ? D + (size_t)c0 * k : w.D.as<float>((size_t)m * k);
            int64_t *Ic = Id ? I + (size_t)c0 * k : w.I.as<int64_t>((size_t)m * k);
            if (l2) {
                float *qa = w.qaug.as<float>((size_t)m * h->da);
                launch_augment(qs, m, h->d, h->da, 0, qa, st);
                hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, qs, m, h->d, w.qn.as<float>((size_t)m));
                MI_HIP(hipGetLastError());
                qs = qa;
            }
            launch_gemm(qs, m, h->base.get<float>(), h->ntotal, h->da, scores, h->ntotal, st);
            launch_select(scores, h->ntotal, m, (int)h->ntotal, k, nullptr, Ic, Dc, st);
            if (l2) {
                hipLaunchKernelGGL(l2_flat_finish_kernel, dim3((unsigned)(((size_t)m * k + 255) / 256)), dim3(256), 0, st, Dc, Ic,
                                   w.qn.get<float>(), m, k);
                MI_HIP(hipGetLastError());
            }
            if (!Dd) MI_HIP(hipMemcpyAsync(D + (size_t)c0 * k, Dc, (size_t)m * k * 4, hipMemcpyDeviceToHost, st));
            if (!Id) MI_HIP(hipMemcpyAsync(I + (size_t)c0 * k, Ic, (size_t)m * k * 8, hipMemcpyDeviceToHost, st));
            if (!qd || !Dd || !Id) MI_HIP(hipStreamSynchronize(st));
        }
    });
}

// ---- building blocks for train() --------------------------------------

// re-read the MI_* knobs (tests and tools; not beside a running search)
int mi_ivfpq_reload_env(void) {
    return guard([&] { knobs_mut().load(); });
}

// out[n][nc] = x . c^T (+ bias[nc]): the exact f32 GEMM of the coarse quantiser (ascending-k fmaf chain per element) as a
// plain operator -- the VectorTransform in front of an IndexPreTransform (OPQ / random rotation: x -> A x + b)
int mi_ip_gemm(int device, int64_t n, const float *x, int64_t nc, const float *c, int d, const float *bias, float *out,
               void *stream) {
    return guard([&] {
        MI_REQUIRE(x && c && out, "null argument");
        MI_REQUIRE(n > 0 && nc > 0 && d > 0 && d % 4 == 0, "mi_ip_gemm: empty input or d not a multiple of 4");
        DeviceGuard dg(device);
        hipStream_t st = as_stream(stream);
        MI_REQUIRE(is_device_ptr(x) && is_device_ptr(c) && is_device_ptr(out) && (!bias || is_device_ptr(bias)),
                   "mi_ip_gemm: x, c, bias and out must be device pointers");
        launch_gemm(x, n, c, nc, d, out, nc, st);
        if (bias) {
            const int64_t total = n * nc;
            hipLaunchKernelGGL(add_row_bias_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, st, out, total, (int)nc, bias);
            MI_HIP(hipGetLastError());
        }
    });
}

int mi_ip_assign(int device, int64_t n, const float *x, int64_t nc, const float *c, int d,
                 int32_t *assign, float *score, void *stream) {
    return guard([&] {
        MI_REQUIRE(x && c && assign, "null argument");
        MI_REQUIRE(n > 0 && nc > 0, "empty input");
        DeviceGuard dg(device);
        hipStream_t st = as_stream(stream);
        MI_REQUIRE(is_device_ptr(x) && is_device_ptr(c), "mi_ip_assign: x and c must be device pointers");
        const bool ad = is_device_ptr(assign), sd = score ? is_device_ptr(score) : true;
        DevBuf scores, da, ds, c16, cstat, x16, xscale, rstat;
        int64_t chunk = std::max<int64_t>(256, std::min<int64_t>(65536, ((int64_t)1 << 28) / nc));
        const bool two = two_stage_wanted(std::min(chunk, n), nc, d, 1);
        float cmax = 0.f, cscale = 1.f;
        if (two) prepare_cent16(c, nc, d, c16, cstat, cmax, cscale, st);
        scores.reserve((size_t)std::min(chunk, n) * nc * 4);
        if (!ad) da.reserve((size_t)n * 4);
        if (score && !sd) ds.reserve((size_t)n * 4);
        int32_t *ap = ad ? assign : da.get<int32_t>();
        float *sp = score ? (sd ? score : ds.get<float>()) : nullptr;
        for (int64_t c0 = 0; c0 < n; c0 += chunk) {
            int64_t m = std::min(chunk, n - c0);
            if (two) {
                launch_two_stage(x + (size_t)c0 * d, m, c, static_cast<const f16_t *>(c16.p), nc, d, 1, cmax, cscale,
                                 scores.get<float>(), x16, xscale, rstat, ap + c0, sp ? sp + c0 : nullptr, ProbeTables{}, st);
                continue;
            }
            launch_gemm(x + (size_t)c0 * d, m, c, nc, d, scores.get<float>(), nc, st);
            launch_select(scores.get<float>(), nc, m, (int)nc, 1, ap + c0, nullptr, sp ? sp + c0 : nullptr, st);
        }
        if (!ad) MI_HIP(hipMemcpyAsync(assign, ap, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        if (score && !sd) MI_HIP(hipMemcpyAsync(score, sp, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));  // scratch buffers die with this scope
    });
}

int mi_cluster_means(int device, int64_t n, const float *x, int d, const int32_t *assign, int k, float *centroids,
                     int32_t *counts, void *stream) {
    return guard([&] {
        MI_REQUIRE(x && assign && centroids, "null argument");
        MI_REQUIRE(n > 0 && n < ((int64_t)1 << 31) && d > 0 && k > 0, "bad sizes");
        DeviceGuard dg(device);
        hipStream_t st = as_stream(stream);
        MI_REQUIRE(is_device_ptr(x) && is_device_ptr(assign) && is_device_ptr(centroids),
                   "mi_cluster_means: x, assign and centroids must be device pointers");
        DevBuf cnt, pos, perm, start;
        int32_t *dc = cnt.as<int32_t>((size_t)k);
        MI_HIP(hipMemsetAsync(dc, 0, (size_t)k * 4, st));
        rank_and_count(assign, n, dc, pos.as<int32_t>((size_t)n), st);
        std::vector<int32_t> hc((size_t)k);
        MI_HIP(hipMemcpyAsync(hc.data(), dc, (size_t)k * 4, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        std::vector<int64_t> hs((size_t)k);
        int64_t o = 0;
        for (int c = 0; c < k; ++c) { hs[(size_t)c] = o; o += hc[(size_t)c]; }
        MI_REQUIRE(o == n, "mi_cluster_means: an assignment is out of range");
        MI_HIP(hipMemcpyAsync(start.as<int64_t>((size_t)k), hs.data(), (size_t)k * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(cluster_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, assign,
                           pos.get<int32_t>(), start.get<int64_t>(), n, perm.as<int32_t>((size_t)n));
        hipLaunchKernelGGL(cluster_mean_kernel, dim3((unsigned)k), dim3(256), 0, st, x, d, perm.get<int32_t>(),
                           start.get<int64_t>(), dc, centroids);
        MI_HIP(hipGetLastError());
        if (counts) MI_HIP(hipMemcpyAsync(counts, dc, (size_t)k * 4, hipMemcpyDefault, st));
        MI_HIP(hipStreamSynchronize(st));  // scratch buffers die with this scope
    });
}

int mi_neg_half_sqnorm(int device, int64_t n, const float *x, int d, float *out, void *stream) {
    return guard([&] {
        MI_REQUIRE(x && out && n > 0 && d > 0, "bad argument");
        DeviceGuard dg(device);
        MI_REQUIRE(is_device_ptr(x) && is_device_ptr(out), "mi_neg_half_sqnorm: device pointers only");
        hipLaunchKernelGGL(neg_half_sqnorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, n, d, out);
        MI_HIP(hipGetLastError());
    });
}

int mi_pq_encode(int device, int64_t n, const float *x, int d, int M, const float *codebook,
                 uint8_t *codes, void *stream) {
    return guard([&] {
        MI_REQUIRE(x && codebook && codes, "null argument");
        MI_REQUIRE(n > 0 && d > 0 && M > 0 && d % M == 0, "bad sizes");
        DeviceGuard dg(device);
        hipStream_t st = as_stream(stream);
        MI_REQUIRE(is_device_ptr(x) && is_device_ptr(codebook), "mi_pq_encode: x and codebook must be device pointers");
        if (is_device_ptr(codes)) {
            launch_pq_encode(x, n, d, M, codebook, nullptr, nullptr, codes, st);
            return;
        }
        DevBuf dc;
        launch_pq_encode(x, n, d, M, codebook, nullptr, nullptr, dc.as<uint8_t>((size_t)n * M), st);
        MI_HIP(hipMemcpyAsync(codes, dc.p, (size_t)n * M, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
    });
}

}  // extern "C"
