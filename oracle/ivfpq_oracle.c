/*
 * oracle/ivfpq_oracle.c -- CPU restatement of the IVF-PQ half of the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (abstracts-search_amd/)
 * may import, link or call this file.  It is the checker used by tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED.  The reference (/root/reference) holds none of this
 * arithmetic: it only *invokes* it through `sidecar-search index train|fill|tune`
 * (reference Makefile:39, Makefile:25, Makefile:32) and the query-time app
 * (reference README.md:28).  The arithmetic lives in faiss, an un-vendored,
 * un-pinned transitive dependency of sidecar-search@0.3.0
 * (reference requirements.txt:1).  faiss is not importable in the build
 * container, the reference ships no tests and no golden vectors, so this file
 * restates the *published* algorithm (Jegou, Douze, Schmid: "Product
 * quantization for nearest neighbor search", PAMI 2011, the IVFADC scheme)
 * with faiss's documented IndexIVFPQ search semantics:
 *
 *   - coarse quantiser = flat inner-product search over the nlist centroids,
 *     best `nprobe` kept                                  (IndexIVF::search)
 *   - by_residual: the vector stored in list c is PQ-encoded as (x - c)
 *                                                        (IndexIVFPQ::encode)
 *   - PQ encode: sub-vector m -> index of the L2-nearest codeword of
 *     codebook[m]                                 (ProductQuantizer::compute_code)
 *   - ADC, inner product: LUT[m][j] = <q_m, codebook[m][j]>;
 *     score(code) = <q, c> (if by_residual) + sum_{m ascending} LUT[m][code[m]]
 *                                         (IVFPQScanner, METRIC_INNER_PRODUCT)
 *   - keep the k best, best first; unfilled slots are I = -1 and
 *     D = -FLT_MAX (the neutral element of faiss's CMin heap).
 *
 * Where faiss leaves the floating-point evaluation order to its BLAS / SIMD
 * build, this restatement fixes ONE order so that "bit-exact" is a meaningful
 * statement between this file and the HIP kernels:
 *
 *   dot(a, b, n)  = fmaf chain, k ascending, accumulator starts at +0.0f
 *                   (this is bit-for-bit what gfx950's v_mfma_f32_16x16x4_f32
 *                   computes, see MI355X guide "FP32-input MFMA")
 *   l2sqr(a,b,n)  = t = a[k]-b[k]; acc = fmaf(t, t, acc), k ascending
 *   ADC sum       = acc = 0; acc += LUT[m][code[m]] for m ascending;
 *                   score = coarse_term + acc
 *   ties          = total order (score descending, then id ascending); for the
 *                   coarse quantiser the "id" is the list number.  faiss breaks
 *                   exact ties by heap insertion order, which depends on its
 *                   threading/scan order; a total order makes the result
 *                   independent of scan order and of how the index is sharded.
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit
 * fmaf(); nothing else may be contracted.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* elementary arithmetic, fixed evaluation order                      */
/* ------------------------------------------------------------------ */

static inline float dot_chain(const float *a, const float *b, int n) {
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

static inline float l2sqr_chain(const float *a, const float *b, int n) {
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) {
        float t = a[k] - b[k];
        acc = fmaf(t, t, acc);
    }
    return acc;
}

/* 8 independent ascending-k chains at once: same bits as dot_chain, but the
 * chains pipeline instead of serialising on fma latency. */
static inline void dot_chain_x8(const float *q, const float *rows, int ld, int n,
                                float out[8]) {
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const float *r0 = rows, *r1 = rows + ld, *r2 = rows + 2 * (size_t)ld,
                *r3 = rows + 3 * (size_t)ld, *r4 = rows + 4 * (size_t)ld,
                *r5 = rows + 5 * (size_t)ld, *r6 = rows + 6 * (size_t)ld,
                *r7 = rows + 7 * (size_t)ld;
    for (int k = 0; k < n; ++k) {
        float x = q[k];
        a0 = fmaf(x, r0[k], a0);
        a1 = fmaf(x, r1[k], a1);
        a2 = fmaf(x, r2[k], a2);
        a3 = fmaf(x, r3[k], a3);
        a4 = fmaf(x, r4[k], a4);
        a5 = fmaf(x, r5[k], a5);
        a6 = fmaf(x, r6[k], a6);
        a7 = fmaf(x, r7[k], a7);
    }
    out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
    out[4] = a4; out[5] = a5; out[6] = a6; out[7] = a7;
}

/* all inner products of one query against n rows */
static void ip_row(const float *q, const float *rows, int64_t n, int d, float *out) {
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) dot_chain_x8(q, rows + i * d, d, d, out + i);
    for (; i < n; ++i) out[i] = dot_chain(q, rows + i * d, d);
}

/* ------------------------------------------------------------------ */
/* top-k under the total order (score desc, id asc)                   */
/* ------------------------------------------------------------------ */

typedef struct { float s; int64_t id; } cand_t;

static inline int better(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

/* sorted insertion list, best first; n = current fill, k = capacity */
static inline void topk_push(cand_t *L, int *n, int k, float s, int64_t id) {
    if (s != s) return; /* NaN never enters (faiss: comparison false) */
    if (*n == k && !better(s, id, L[k - 1].s, L[k - 1].id)) return;
    int pos = (*n < k) ? *n : k - 1;
    while (pos > 0 && better(s, id, L[pos - 1].s, L[pos - 1].id)) {
        L[pos] = L[pos - 1];
        --pos;
    }
    L[pos].s = s;
    L[pos].id = id;
    if (*n < k) ++*n;
}

/* ------------------------------------------------------------------ */
/* exported entry points                                              */
/* ------------------------------------------------------------------ */

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Flat inner-product top-k: the coarse quantiser (IndexFlatIP::search), also
 * config #1's IndexFlatIP.  idx is int64 (faiss idx_t); unfilled = -1/-FLT_MAX. */
ORACLE_API void oracle_flat_ip(int64_t nq, int d, const float *q, int64_t nb,
                               const float *base, int k, float *D, int64_t *I) {
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            ip_row(q + qi * d, base, nb, d, row);
            int n = 0;
            for (int64_t i = 0; i < nb; ++i) topk_push(L, &n, k, row[i], i);
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? L[j].s : -FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
        }
        free(row);
        free(L);
    }
}

/* IndexRefineFlat::search, re-ranking step (faiss: the base index returns
 * k * k_factor labels, the refine index -- an IndexFlat over the same vectors in
 * add() order -- recomputes their exact distances, the k best are kept).  cand_I
 * [nq][kc], negative = empty slot.  Same dot product and the same total order
 * (score desc, id asc) as everywhere else in this file. */
ORACLE_API void oracle_rerank(int64_t nq, int d, const float *q, const float *base, int kc,
                              const int64_t *cand_I, int k, float *D, int64_t *I) {
#pragma omp parallel
    {
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            int n = 0;
            for (int c = 0; c < kc; ++c) {
                const int64_t id = cand_I[qi * kc + c];
                if (id < 0) continue;
                topk_push(L, &n, k, dot_chain(q + qi * d, base + id * (int64_t)d, d), id);
            }
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? L[j].s : -FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
        }
        free(L);
    }
}

/* ---- ScalarQuantizer QT_8bit (faiss index_factory "...,Refine(SQ8)": the refine store that lets
 * the recall >= 0.95 operating point keep ALL 207 M vectors of BASELINE.json configs[3] beside the index in one
 * GPU's HBM: 1 byte per component, 212 GB).  Reference call site: `sidecar-search index ... tune`
 * (reference Makefile:32) explores the index's search parameters, README.md:28 the query-time app; the
 * arithmetic is faiss's ScalarQuantizer, restated from its published behaviour:
 *
 *   train  (ScalarQuantizer::train, QT_8bit = per-dimension ranges, RangeStat RS_minmax, rangestat_arg 0):
 *          vmin[i] = min_r x[r][i], vdiff[i] = max_r x[r][i] - vmin[i]; trained = [vmin | vdiff] (2 d floats)
 *   encode (QuantizerTemplate<Codec8bit, non-uniform>::encode_vector):
 *          xi = (x[i] - vmin[i]) / vdiff[i] clipped to [0, 1] (0 when vdiff[i] == 0); code[i] = (int)(255 * xi)
 *   decode (reconstruct_component; faiss's SIMD build evaluates it with two fused multiply-adds):
 *          t = fmaf((float)code[i], 1/255, 0.5/255);  x^[i] = fmaf(t, vdiff[i], vmin[i])
 *   score  (DCTemplate<..., SimilarityIP>::query_to_code): <q, x^>, with x^[i] = a[i] + b[i] code[i] the same affine
 *          map written out (a[i] = vmin[i] + vdiff[i] 0.5/255, b[i] = vdiff[i] / 255).  Like the PQ scan, which adds
 *          look-up-table entries precomputed per query instead of decoding every code, the score is evaluated in its
 *          asymmetric form: a per-query table first, then one multiply-add per stored byte --
 *              A(q)   = chain_i fmaf(q[i], a[i], .)   from +0, i ascending      (once per query)
 *              w(q)[i] = q[i] * b[i]                                             (once per query)
 *              score  = chain_i fmaf(w(q)[i], (float)code[i], .)  from A(q), i ascending
 *          = sum_i q[i] (a[i] + b[i] code[i]) exactly in real arithmetic; the rounding sequence is this file's ONE fixed
 *          order for it (faiss's own order is its SIMD build's: 8 lane accumulators over decoded components).  Two VALU
 *          operations per byte instead of five: the re-rank kernel streams 1 KiB rows at HBM speed.
 */
ORACLE_API void oracle_sq8_train(int64_t n, int d, const float *x, float *trained /* [2 d]: vmin | vdiff */) {
    for (int i = 0; i < d; ++i) {
        float lo = HUGE_VALF, hi = -HUGE_VALF;
        for (int64_t r = 0; r < n; ++r) {
            const float v = x[r * (int64_t)d + i];
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        trained[i] = lo;
        trained[d + i] = hi - lo;
    }
}

ORACLE_API void oracle_sq8_encode(int64_t n, int d, const float *x, const float *trained, uint8_t *codes) {
    const float *vmin = trained, *vdiff = trained + d;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int i = 0; i < d; ++i) {
            float xi = 0.f;
            if (vdiff[i] != 0.f) {
                xi = (x[r * (int64_t)d + i] - vmin[i]) / vdiff[i];
                if (xi < 0.f) xi = 0.f;
                if (xi > 1.f) xi = 1.f;
            }
            codes[r * (int64_t)d + i] = (uint8_t)(int)(255.f * xi);
        }
}

static inline float sq8_component(uint8_t c, float vmin, float vdiff) {
    const float t = fmaf((float)c, 1.0f / 255.0f, 0.5f / 255.0f);
    return fmaf(t, vdiff, vmin);
}

/* per-query table of the asymmetric score: w [d] and the constant A (see the section header) */
ORACLE_API void oracle_sq8_query_table(int d, const float *q, const float *trained, float *w, float *A) {
    const float *vmin = trained, *vdiff = trained + d;
    float acc = 0.f;
    for (int i = 0; i < d; ++i) {
        const float a = fmaf(vdiff[i], 0.5f / 255.0f, vmin[i]), b = vdiff[i] / 255.0f;
        acc = fmaf(q[i], a, acc);
        w[i] = q[i] * b;
    }
    *A = acc;
}

ORACLE_API void oracle_sq8_decode(int64_t n, int d, const uint8_t *codes, const float *trained, float *x) {
    const float *vmin = trained, *vdiff = trained + d;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int i = 0; i < d; ++i) x[r * (int64_t)d + i] = sq8_component(codes[r * (int64_t)d + i], vmin[i], vdiff[i]);
}

/* IndexRefine(base, IndexScalarQuantizer(QT_8bit)) re-ranking step: like oracle_rerank over the decoded rows */
ORACLE_API void oracle_rerank_sq8(int64_t nq, int d, const float *q, const uint8_t *codes, const float *trained, int kc,
                                  const int64_t *cand_I, int k, float *D, int64_t *I) {
#pragma omp parallel
    {
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
        float *w = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            int n = 0;
            float A;
            oracle_sq8_query_table(d, q + qi * d, trained, w, &A);
            for (int c = 0; c < kc; ++c) {
                const int64_t id = cand_I[qi * kc + c];
                if (id < 0) continue;
                const uint8_t *row = codes + id * (int64_t)d;
                float acc = A;
                for (int i = 0; i < d; ++i) acc = fmaf(w[i], (float)row[i], acc);
                topk_push(L, &n, k, acc, id);
            }
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? L[j].s : -FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
        }
        free(w);
        free(L);
    }
}

/* ADC look-up table for one query: lut[m*ksub + j] = <q_m, codebook[m][j]> */
ORACLE_API void oracle_lut(int d, int M, int ksub, const float *q,
                           const float *codebook, float *lut) {
    int dsub = d / M;
    for (int m = 0; m < M; ++m)
        for (int j = 0; j < ksub; ++j)
            lut[m * ksub + j] = dot_chain(
                q + m * dsub, codebook + ((size_t)m * ksub + j) * dsub, dsub);
}

/* Index.add(): coarse assign (arg max inner product, ties -> smallest list
 * number) + PQ encode of the residual (or of x itself if !by_residual).
 * ksub <= 256, one byte per sub-quantiser. */
ORACLE_API void oracle_encode(int64_t n, int d, const float *x, int nlist,
                              const float *centroids, int M, int ksub,
                              const float *codebook, int by_residual,
                              int32_t *list_no, uint8_t *codes) {
    int dsub = d / M;
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)nlist);
        float *r = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < n; ++i) {
            const float *xi = x + i * d;
            ip_row(xi, centroids, nlist, d, row);
            int best = 0;
            for (int c = 1; c < nlist; ++c)
                if (row[c] > row[best]) best = c;
            list_no[i] = best;
            const float *cen = centroids + (size_t)best * d;
            for (int t = 0; t < d; ++t) r[t] = by_residual ? xi[t] - cen[t] : xi[t];
            for (int m = 0; m < M; ++m) {
                int bj = 0;
                float bd = l2sqr_chain(r + m * dsub, codebook + (size_t)m * ksub * dsub, dsub);
                for (int j = 1; j < ksub; ++j) {
                    float dj = l2sqr_chain(
                        r + m * dsub, codebook + ((size_t)m * ksub + j) * dsub, dsub);
                    if (dj < bd) { bd = dj; bj = j; }
                }
                codes[i * M + m] = (uint8_t)bj;
            }
        }
        free(row);
        free(r);
    }
}

/* Index.search() over inverted lists in CSR form:
 *   list l holds codes[list_off[l] .. list_off[l+1]) (M bytes each, row-major)
 *   and ids[...] in the same order.
 * Returns the coarse result too (coarse_I/coarse_D may be NULL). */
ORACLE_API void oracle_search(int64_t nq, int d, const float *q, int nlist,
                              const float *centroids, int M, int ksub,
                              const float *codebook, int by_residual,
                              const int64_t *list_off, const uint8_t *codes,
                              const int64_t *ids, int nprobe, int k, float *D,
                              int64_t *I, int32_t *coarse_I, float *coarse_D) {
    if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)nlist);
        float *lut = (float *)malloc(sizeof(float) * (size_t)M * ksub);
        cand_t *P = (cand_t *)malloc(sizeof(cand_t) * (size_t)nprobe);
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            const float *qv = q + qi * d;
            /* step 1: coarse quantise */
            ip_row(qv, centroids, nlist, d, row);
            int np = 0;
            for (int c = 0; c < nlist; ++c) topk_push(P, &np, nprobe, row[c], c);
            /* step 2: distance LUT */
            oracle_lut(d, M, ksub, qv, codebook, lut);
            /* step 3+4: scan the probed lists, keep the k best */
            int n = 0;
            for (int p = 0; p < np; ++p) {
                int64_t l = P[p].id;
                float dis0 = by_residual ? P[p].s : 0.0f;
                for (int64_t e = list_off[l]; e < list_off[l + 1]; ++e) {
                    const uint8_t *c = codes + e * M;
                    float acc = 0.0f;
                    for (int m = 0; m < M; ++m) acc += lut[m * ksub + c[m]];
                    topk_push(L, &n, k, dis0 + acc, ids[e]);
                }
            }
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? L[j].s : -FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
            if (coarse_I)
                for (int p = 0; p < nprobe; ++p) coarse_I[qi * nprobe + p] = p < np ? (int32_t)P[p].id : -1;
            if (coarse_D)
                for (int p = 0; p < nprobe; ++p) coarse_D[qi * nprobe + p] = p < np ? P[p].s : -FLT_MAX;
        }
        free(row);
        free(lut);
        free(P);
        free(L);
    }
}

/* IndexIVF.search_preassigned: steps 2-4 with a given coarse assignment
 * (coarse_I [nq][nprobe], -1 = none; coarse_D = <q, centroid>). */
ORACLE_API void oracle_search_preassigned(int64_t nq, int d, const float *q, int M, int ksub,
                                          const float *codebook, int by_residual,
                                          const int64_t *list_off, const uint8_t *codes,
                                          const int64_t *ids, int nprobe, const int32_t *coarse_I,
                                          const float *coarse_D, int k, float *D, int64_t *I) {
#pragma omp parallel
    {
        float *lut = (float *)malloc(sizeof(float) * (size_t)M * ksub);
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            oracle_lut(d, M, ksub, q + qi * d, codebook, lut);
            int n = 0;
            for (int p = 0; p < nprobe; ++p) {
                int64_t l = coarse_I[qi * nprobe + p];
                if (l < 0) continue;
                float dis0 = by_residual ? coarse_D[qi * nprobe + p] : 0.0f;
                for (int64_t e = list_off[l]; e < list_off[l + 1]; ++e) {
                    const uint8_t *c = codes + e * M;
                    float acc = 0.0f;
                    for (int m = 0; m < M; ++m) acc += lut[m * ksub + c[m]];
                    topk_push(L, &n, k, dis0 + acc, ids[e]);
                }
            }
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? L[j].s : -FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
        }
        free(lut);
        free(L);
    }
}

/* Merge `nparts` per-shard top-k lists ([part][q][k], best first, -1 padded)
 * into one, under the same total order.  This is the exchange step's
 * arithmetic (faiss: IndexShards / merge_knn_results). */
ORACLE_API void oracle_merge(int nparts, int64_t nq, int k, const float *Dp,
                             const int64_t *Ip, float *D, int64_t *I) {
    cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
    for (int64_t qi = 0; qi < nq; ++qi) {
        int n = 0;
        for (int p = 0; p < nparts; ++p)
            for (int j = 0; j < k; ++j) {
                size_t o = ((size_t)p * nq + qi) * k + j;
                if (Ip[o] >= 0) topk_push(L, &n, k, Dp[o], Ip[o]);
            }
        for (int j = 0; j < k; ++j) {
            D[qi * k + j] = j < n ? L[j].s : -FLT_MAX;
            I[qi * k + j] = j < n ? L[j].id : -1;
        }
    }
    free(L);
}

/* ------------------------------------------------------------------
 * k-means building blocks (IndexIVFPQ.train; reference Makefile:39 `index train`).
 * The product's update step (mi_cluster_means) fixes the summation order -- members of a
 * cluster in ascending row order, sequential f32 adds, one division by the count -- so
 * that training is bit-reproducible; this is that order as a plain loop.
 * ------------------------------------------------------------------ */
ORACLE_API void oracle_cluster_means(int64_t n, int d, const float *x, const int32_t *assign, int k,
                                     float *centroids, int32_t *counts) {
    float *acc = (float *)calloc((size_t)k * d, sizeof(float));
    int32_t *cnt = (int32_t *)calloc((size_t)k, sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) {
        const int c = assign[i];
        float *a = acc + (size_t)c * d;
        const float *xi = x + (size_t)i * d;
        for (int t = 0; t < d; ++t) a[t] = a[t] + xi[t];
        ++cnt[c];
    }
    for (int c = 0; c < k; ++c) {
        if (counts) counts[c] = cnt[c];
        if (!cnt[c]) continue;                      /* an empty cluster keeps its row */
        for (int t = 0; t < d; ++t) centroids[(size_t)c * d + t] = acc[(size_t)c * d + t] / (float)cnt[c];
    }
    free(acc);
    free(cnt);
}

/* out[r] = -0.5 * dot(x_r, x_r) */
ORACLE_API void oracle_neg_half_sqnorm(int64_t n, int d, const float *x, float *out) {
    for (int64_t r = 0; r < n; ++r) out[r] = -0.5f * dot_chain(x + (size_t)r * d, x + (size_t)r * d, d);
}

/* ------------------------------------------------------------------
 * METRIC_L2 (faiss's default metric; IndexIVFPQ / IndexFlatL2 with squared L2 distances).
 * faiss evaluates |q - x|^2 either directly or, in its BLAS path, through the expansion
 * |q|^2 + |x|^2 - 2<q,x>; which one runs -- and in which order it sums -- depends on its
 * build and on the batch size.  This restatement fixes ONE evaluation, the expansion, so
 * that the inner-product machinery (and its kernels) carry over unchanged:
 *
 *   S(x, c)   = dot(x, c) + (-0.5 * dot(c, c))        one rounding at the "+"; bitwise the
 *               ascending-k fmaf chain over the augmented vectors [x, 1, 0..] . [c, -|c|^2/2, 0..]
 *   coarse    = the nprobe lists with the largest S(q, c)  (ties: smaller list number)
 *   qn        = dot(q, q)
 *   dis0      = 2 * S(q, c) - qn                       (= -|q - c|^2 up to rounding)
 *   add       : list = arg max_c S(x, c); r = x - c; codes = nearest codeword per sub-vector
 *               t(v) = dot(r^, r^) + 2 * dot(c, r^)     r^ = decoded residual (by_residual)
 *                    = dot(x^, x^), dis0 = -qn           (!by_residual)
 *   ADC       : acc = sum_m LUT_ip[m][code_m] (m ascending);  s = dis0 + (2 * acc - t)
 *   result    : the k largest s under (s desc, id asc), reported as D = -s (ascending squared
 *               distances, best first); unfilled slots I = -1, D = +FLT_MAX (faiss's CMax neutral).
 * ------------------------------------------------------------------ */
static inline float aug_score(const float *x, const float *c, float nh_c, int d) {
    return dot_chain(x, c, d) + nh_c;
}

ORACLE_API void oracle_flat_l2(int64_t nq, int d, const float *q, int64_t nb, const float *base, int k,
                               float *D, int64_t *I) {
    float *nh = (float *)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
    for (int64_t i = 0; i < nb; ++i) nh[i] = -0.5f * dot_chain(base + i * d, base + i * d, d);
#pragma omp parallel
    {
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            const float *qv = q + qi * d;
            const float qn = dot_chain(qv, qv, d);
            int n = 0;
            for (int64_t i = 0; i < nb; ++i) topk_push(L, &n, k, 2.0f * aug_score(qv, base + i * d, nh[i], d) - qn, i);
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? -L[j].s : FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
        }
        free(L);
    }
    free(nh);
}

/* Index.add() for METRIC_L2: list numbers, codes and the per-vector term t */
ORACLE_API void oracle_encode_l2(int64_t n, int d, const float *x, int nlist, const float *centroids, int M, int ksub,
                                 const float *codebook, int by_residual, int32_t *list_no, uint8_t *codes, float *tnorm) {
    const int dsub = d / M;
    float *nh = (float *)malloc(sizeof(float) * (size_t)nlist);
    for (int c = 0; c < nlist; ++c) nh[c] = -0.5f * dot_chain(centroids + (size_t)c * d, centroids + (size_t)c * d, d);
#pragma omp parallel
    {
        float *r = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < n; ++i) {
            const float *xi = x + i * d;
            int best = 0;
            float bs = aug_score(xi, centroids, nh[0], d);
            for (int c = 1; c < nlist; ++c) {
                const float s = aug_score(xi, centroids + (size_t)c * d, nh[c], d);
                if (s > bs) { bs = s; best = c; }
            }
            list_no[i] = best;
            const float *cen = centroids + (size_t)best * d;
            for (int t = 0; t < d; ++t) r[t] = by_residual ? xi[t] - cen[t] : xi[t];
            float a1 = 0.0f, a2 = 0.0f;
            for (int m = 0; m < M; ++m) {
                int bj = 0;
                float bd = l2sqr_chain(r + m * dsub, codebook + (size_t)m * ksub * dsub, dsub);
                for (int j = 1; j < ksub; ++j) {
                    const float dj = l2sqr_chain(r + m * dsub, codebook + ((size_t)m * ksub + j) * dsub, dsub);
                    if (dj < bd) { bd = dj; bj = j; }
                }
                codes[i * M + m] = (uint8_t)bj;
                const float *cw = codebook + ((size_t)m * ksub + bj) * dsub;
                for (int t = 0; t < dsub; ++t) {
                    a1 = fmaf(cw[t], cw[t], a1);
                    if (by_residual) a2 = fmaf(cen[m * dsub + t], cw[t], a2);
                }
            }
            tnorm[i] = a1 + 2.0f * a2;
        }
        free(r);
    }
    free(nh);
}

ORACLE_API void oracle_search_l2(int64_t nq, int d, const float *q, int nlist, const float *centroids, int M, int ksub,
                                 const float *codebook, int by_residual, const int64_t *list_off, const uint8_t *codes,
                                 const int64_t *ids, const float *tnorm, int nprobe, int k, float *D, int64_t *I,
                                 int32_t *coarse_I, float *coarse_D) {
    if (nprobe > nlist) nprobe = nlist;
    float *nh = (float *)malloc(sizeof(float) * (size_t)nlist);
    for (int c = 0; c < nlist; ++c) nh[c] = -0.5f * dot_chain(centroids + (size_t)c * d, centroids + (size_t)c * d, d);
#pragma omp parallel
    {
        float *lut = (float *)malloc(sizeof(float) * (size_t)M * ksub);
        cand_t *P = (cand_t *)malloc(sizeof(cand_t) * (size_t)nprobe);
        cand_t *L = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            const float *qv = q + qi * d;
            const float qn = dot_chain(qv, qv, d);
            int np = 0;
            for (int c = 0; c < nlist; ++c) topk_push(P, &np, nprobe, aug_score(qv, centroids + (size_t)c * d, nh[c], d), c);
            oracle_lut(d, M, ksub, qv, codebook, lut);
            int n = 0;
            for (int p = 0; p < np; ++p) {
                const int64_t l = P[p].id;
                const float dis0 = by_residual ? 2.0f * P[p].s - qn : -qn;
                for (int64_t e = list_off[l]; e < list_off[l + 1]; ++e) {
                    const uint8_t *c = codes + e * M;
                    float acc = 0.0f;
                    for (int m = 0; m < M; ++m) acc += lut[m * ksub + c[m]];
                    topk_push(L, &n, k, dis0 + (2.0f * acc - tnorm[e]), ids[e]);
                }
            }
            for (int j = 0; j < k; ++j) {
                D[qi * k + j] = j < n ? -L[j].s : FLT_MAX;
                I[qi * k + j] = j < n ? L[j].id : -1;
            }
            if (coarse_I)
                for (int p = 0; p < nprobe; ++p) coarse_I[qi * nprobe + p] = p < np ? (int32_t)P[p].id : -1;
            if (coarse_D)   /* reported as squared distances, like faiss's quantizer.search */
                for (int p = 0; p < nprobe; ++p) coarse_D[qi * nprobe + p] = p < np ? -(2.0f * P[p].s - qn) : FLT_MAX;
        }
        free(lut);
        free(P);
        free(L);
    }
    free(nh);
}
