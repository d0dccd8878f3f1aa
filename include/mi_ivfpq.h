/*
 * mi_ivfpq.h -- C ABI of the MI355X-native IVF-PQ index (search hot path).
 *
 * This is the drop-in boundary for the index half of the hot path.  The
 * reference has no in-repo FFI for it: it reaches the arithmetic through the
 * `sidecar-search index` CLI (reference Makefile:39 train, Makefile:25 fill,
 * Makefile:32 tune, README.md:28 query-time app), which calls the Python faiss
 * API.  Each entry point below names the faiss call it replaces; the Python
 * mirror of that API is abstracts-search_amd/faiss.py, which binds exactly
 * these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is
 *     in mi_last_error() (thread-local).  No C++ exception crosses the ABI.
 *   - plain pointers and sizes only.  A data pointer may be a host pointer or
 *     a HIP device pointer on the index's device (e.g. torch.Tensor.data_ptr());
 *     the library detects which.  With device pointers for all inputs and
 *     outputs, mi_index_search() only enqueues work on `stream` and returns
 *     without synchronising; with host pointers it stages and synchronises.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Metrics: MI_METRIC_INNER_PRODUCT (the metric the reference's normalised stella
 *     embeddings use; results best first = largest first, unfilled D = -FLT_MAX) and
 *     MI_METRIC_L2 (faiss's default: squared L2 distances, smallest first, unfilled
 *     D = +FLT_MAX), evaluated through the expansion -|q - x|^2 = 2 S - |q|^2 with
 *     S = <q, x> - |x|^2/2 one ascending-k fmaf chain over vectors augmented by one column
 *     (oracle/ivfpq_oracle.c, section METRIC_L2) -- the inner-product kernels run unchanged;
 *     an IVF-PQ vector stores one more f32 (|r^|^2 + 2<c, r^>).  mi_index_coarse_slice /
 *     mi_index_search_preassigned and the half-precision flat store are inner-product only.
 *   - nbits == 8 only (one byte per sub-quantiser, ksub = 256).
 *   - handles are opaque, freed only by *_destroy; one handle per device.
 *   - Threading (faiss: "search is thread-safe for concurrent readers, add/train are
 *     exclusive"): the search-type calls -- mi_index_search, mi_index_search_preassigned,
 *     mi_index_coarse_slice, mi_flat_search, mi_flat_rerank -- may be issued on one
 *     handle from several host threads at once.  Every call leases the workspace set of the
 *     stream it is given (one set per stream, up to 64 per handle) for as long as it enqueues
 *     work: threads on distinct streams run concurrently and overlap on the GPU, threads that
 *     share a stream (e.g. NULL) take turns.  State a search builds lazily (the scan image after
 *     an add) is built once under the handle's lock.  add / train / set_* / reset / save /
 *     destroy are exclusive: no other call may run on the handle meanwhile.  mi_shards_search
 *     issues a collective: one host thread per handle (every rank must issue in the same order).
 *   - result semantics (faiss): best first; unfilled slots I = -1,
 *     D = -FLT_MAX.  Exact score ties are ordered by ascending id.
 */
#ifndef MI_IVFPQ_H
#define MI_IVFPQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_METRIC_INNER_PRODUCT 0 /* faiss.METRIC_INNER_PRODUCT */
#define MI_METRIC_L2 1            /* faiss.METRIC_L2 */

typedef struct mi_index mi_index; /* faiss.IndexIVFPQ */
typedef struct mi_flat mi_flat;   /* faiss.IndexFlatIP / IndexScalarQuantizer(QT_fp16) */

/* Last error message of the calling thread ("" if none). */
const char *mi_last_error(void);

/* Number of visible HIP devices (faiss.get_num_gpus). */
int mi_device_count(int *count);

/* ---- IndexIVFPQ ---------------------------------------------------- */

/* faiss.index_factory(d, "IVF{nlist},PQ{M}", metric) / IndexIVFPQ ctor. */
int mi_index_create(int d, int nlist, int M, int nbits, int metric, int by_residual,
                    int device, mi_index **out);
int mi_index_destroy(mi_index *h);

/* Result of IndexIVFPQ.train: coarse centroids [nlist][d] and PQ codebook
 * [M][256][d/M], float32.  (faiss: quantizer.add(centroids); pq.centroids.) */
int mi_index_set_coarse(mi_index *h, const float *centroids);
int mi_index_set_codebook(mi_index *h, const float *codebook);
int mi_index_get_coarse(mi_index *h, float *centroids_host);
int mi_index_get_codebook(mi_index *h, float *codebook_host);
int mi_index_is_trained(mi_index *h, int *out);

/* IndexIVFPQ.ntotal / .reset() */
int mi_index_ntotal(mi_index *h, int64_t *out);
int mi_index_reset(mi_index *h);

/* No faiss counterpart (faiss keeps one copy of the lists): the index is built and will be searched -- free the append log
 * (80 B per vector; the scan image, 72 B per padded vector, stays).  A later add / add_codes / export / save rebuilds the log
 * from the image first; results of every call are unchanged.  Exclusive, like mi_index_add.  (reference Makefile:25
 * `index fill` is followed by searches only.) */
int mi_index_seal(mi_index *h);

/* IndexIVFPQ.add (ids == NULL: sequential from ntotal) / add_with_ids.
 * x: float32 [n][d]. */
int mi_index_add(mi_index *h, int64_t n, const float *x, const int64_t *ids);

/* The arithmetic of add without the append (IndexIVFPQ.encode_multiple +
 * quantizer.assign): list_no[n] int32, codes[n][M] uint8, host outputs. */
int mi_index_encode(mi_index *h, int64_t n, const float *x, int32_t *list_no_host,
                    uint8_t *codes_host);

/* Append pre-encoded entries (read_index path; InvertedLists.add_entries). */
int mi_index_add_codes(mi_index *h, int64_t n, const int32_t *list_no,
                       const uint8_t *codes, const int64_t *ids); /* all host or all device pointers */

/* InvertedLists.list_size / get_codes+get_ids (host outputs, insertion order). */
int mi_index_list_size(mi_index *h, int list_no, int64_t *out);
int mi_index_get_list(mi_index *h, int list_no, uint8_t *codes_host, int64_t *ids_host);
/* All list sizes at once: sizes_host int64 [nlist]. */
int mi_index_list_sizes(mi_index *h, int64_t *sizes_host);
/* The lists [list_lo, list_hi) concatenated in list order, each in insertion order:
 * codes uint8 [rows][M], ids int64 [rows] with rows = the sum of their sizes (host or
 * device outputs; either may be NULL).  What write_index streams to disk slab by slab
 * (the lists live in HBM; the host never holds the whole index). */
int mi_index_export_lists(mi_index *h, int list_lo, int list_hi, uint8_t *codes, int64_t *ids);
/* Capacity hint for add(): room for n vectors in total (std::vector::reserve). */
int mi_index_reserve(mi_index *h, int64_t n);

/* faiss.write_index / faiss.read_index for this index family, in faiss's binary format
 * (IwPQ + IndexFlat quantiser + in-file `ilar` lists, or -- with ondisk_data != NULL -- the
 * OnDiskInvertedLists pair: the reference's index.faiss + ondisk.ivfdata, Makefile:11-12).
 * The lists stream between HBM and the file in bounded slabs. */
int mi_index_save(mi_index *h, const char *fname, const char *ondisk_data);
int mi_index_load(const char *fname, int device, mi_index **out);
/* The same with the IwPQ record starting `offset` bytes into the file: the sub-index of a faiss IndexPreTransform file
 * ("IxPT": header, the VectorTransform chain, then the index -- sidecar-search's `index train` may wrap the IVF-PQ index
 * in an OPQ rotation; faiss.py parses the chain and applies it to the vectors before add() / search()). */
int mi_index_load_at(const char *fname, int64_t offset, int device, mi_index **out);
/* Parameters of a handle (e.g. one returned by mi_index_load); any output may be NULL.
 * nprobe is faiss's index.nprobe attribute as stored in index files -- mi_index_search()
 * takes nprobe per call. */
int mi_index_get_params(mi_index *h, int *d, int *nlist, int *M, int *nbits, int *metric,
                        int *by_residual, int *nprobe);
int mi_index_set_nprobe(mi_index *h, int nprobe);

/* IndexIVFPQ.search with index.nprobe = nprobe.
 * q float32 [nq][d]; D float32 [nq][k]; I int64 [nq][k].  1 <= k <= 8192 (k > 64: every (score, id) of the
 * probed lists is stored in one pass and the k best kept: the candidate lists of the refine stage). */
int mi_index_search(mi_index *h, int64_t nq, const float *q, int k, int nprobe,
                    float *D, int64_t *I, void *stream);

/* The first stage of IndexRefine::search (base_index->search(n, x, k * k_factor, ...) whose labels feed the re-rank):
 * the kc best entries of every query under the same total order as mi_index_search, as a SET -- ids only, in no
 * particular order, -1 in unfilled slots.  The re-rank's result does not depend on the order of its candidates, and
 * skipping the sort of a several-thousand-entry list is 1 ms of a 4.6 ms step at the recall >= 0.95 point of the 207 M
 * index.  Device pointers only; 1 <= kc <= 8192. */
int mi_index_search_candidates(mi_index *h, int64_t nq, const float *q, int kc, int nprobe,
                               int64_t *I, void *stream);

/* Steps 1-2 of search exposed for parity tests: quantizer.search(q, nprobe)
 * -> coarse_I int32 [nq][nprobe], coarse_D float32 [nq][nprobe] (host), and the
 * ADC table pq.compute_inner_prod_tables -> lut float32 [nq][M][256] (host).
 * Any output may be NULL. */
int mi_index_coarse_lut(mi_index *h, int64_t nq, const float *q, int nprobe,
                        int32_t *coarse_I_host, float *coarse_D_host, float *lut_host);

/* Coarse quantisation over a slice [list_lo, list_hi) of the centroids
 * (IndexFlatIP.search on a sub-range; list numbers are global): used to split
 * the coarse GEMM across the GPUs of a node -- at IVF65536 it costs more than
 * the scan of a shard.  coarse_I int32 / coarse_D float32 [nq][nprobe], padded
 * with -1 / -FLT_MAX when the slice holds fewer than nprobe lists.  Device
 * pointers only; enqueues on `stream`. */
int mi_index_coarse_slice(mi_index *h, int64_t nq, const float *q, int nprobe, int list_lo,
                          int list_hi, int32_t *coarse_I, float *coarse_D, void *stream);

/* IndexIVF.search_preassigned: search with a given coarse assignment
 * (coarse_I int32 [nq][nprobe], -1 = no list; coarse_D float32 [nq][nprobe] =
 * <q, centroid>), e.g. the merged result of mi_index_coarse_slice() calls.
 * Device pointers only; enqueues on `stream`. */
int mi_index_search_preassigned(mi_index *h, int64_t nq, const float *q, int k, int nprobe,
                                const int32_t *coarse_I, const float *coarse_D, float *D,
                                int64_t *I, void *stream);

/* Timing of the dominant kernel for the roofline: re-launches the PQ-code scan
 * kernel of the most recent mi_index_search() call (same arguments, idempotent)
 * `reps` times back to back on `stream`, bracketed by two HIP events recorded on
 * that stream, and returns the average duration per launch in ms together with
 * the algorithmic bytes of one launch (codes scanned x (M + 8)). */
int mi_index_profile_scan(mi_index *h, int reps, void *stream, double *scan_ms_avg,
                          int64_t *scan_bytes);
/* Exact list pruning (replaces nothing in the reference: faiss's IndexIVFPQ::search scans every probed list; same (D, I)).
 * A by-residual inner-product code scores <q, centroid> + the f32 chain of M table entries, so <q, centroid> + the same chain
 * over the tables' row maxima bounds every code of a list (rounded addition is monotone: no slack term); a list whose bound is
 * below a score that k found codes already reach provably holds no result.  mi_index_search uses it in two forms, k <= 64:
 *  - batches of >= 512 queries, nprobe <= 64: inside the scan kernel -- one workgroup per query walks the lists in descending
 *    coarse order and every wave stops at the first list whose bound is below a threshold it holds;
 *  - other large scans (mi_index_search_preassigned's caller-ordered lists, nprobe > 64, 16..511 queries): two scan launches
 *    -- the best nprobe/32 lists of every query, then only the lists whose bound reaches the k-th score found there.
 * out3 = {64-code groups scanned (early stop) or scanned by the second launch (two phases), groups of all probed lists,
 * queries}, summed over the pruned calls since the last reset.  MI_SCAN_PRUNE=0 turns the pruning off, MI_SCAN_PRUNE_MODE=1 / 2
 * keeps one form (mi_ivfpq_reload_env).  mi_index_profile_scan replays the (first) scan launch of the last search. */
int mi_index_prune_stats(mi_index *h, unsigned long long *out3, int reset);

/* ---- exchange step ------------------------------------------------- */

/* Merge nparts per-shard results (faiss IndexShards merge): D_parts/I_parts are
 * [nparts][nq][k] (what an all-gather of per-shard (D, I) produces). */
int mi_merge_topk(int device, int nparts, int64_t nq, int k, const float *D_parts,
                  const int64_t *I_parts, float *D, int64_t *I, void *stream);

/* The same merge straight on the receive buffer of ONE all-gather (the path's one exchange
 * step, SURVEY 8(e): ncclAllGather of nq*k*12 bytes per rank).  `gathered` = nparts blocks
 * of blk_bytes; block p = { float D[nq*k]; pad to 8 bytes; int64 I[nq*k] } exactly as rank
 * p's mi_index_search() wrote its (D, I) into the two halves of its send buffer.  Ids are
 * translated while they are loaded: global = local * id_mul + id_add + p * id_step
 * (round-robin shards numbered by position: {nparts, 0, 1}; identity: {1, 0, 0}); negative
 * ids stay empty slots.  Only the queries [q_lo, q_lo + nq_out) of the nq in every block are
 * merged (a rank that brought its own batch merges its own slice): D, I are [nq_out][k].
 * Device pointers only; enqueues on `stream`. */
int mi_merge_topk_gathered(int device, int nparts, int64_t nq, int k, const void *gathered,
                           int64_t blk_bytes, int64_t id_mul, int64_t id_add, int64_t id_step,
                           int64_t q_lo, int64_t nq_out, float *D, int64_t *I, void *stream);

/* ---- vector-sharded search, one process per GPU ----------
 * The whole sharded search step behind one call, for a host without torch.distributed: the
 * local shard is searched for all nq (replicated) queries straight into a send buffer, ONE
 * ncclAllGather (RCCL over xGMI, nq*k*12 bytes per rank) moves the per-shard (D, I), and the
 * k-way merge -- ids translated local -> global as global = local * id_mul + id_add +
 * rank * id_step -- writes the result; everything is enqueued on `stream`.
 * RCCL is bound at run time: rccl_lib = path of the librccl.so to use (NULL: $MI_RCCL_LIB,
 * then the system's); a Python host passes the copy torch already loaded.
 * mi_shards_unique_id: rank 0 creates the 128-byte id and ships it to the other ranks out of
 * band (file, socket, MPI, torch.distributed...); every rank then calls mi_shards_create with
 * the same id (ncclCommInitRank: a collective call).  refine != NULL makes every shard an
 * IndexRefine: k * k_factor candidates from `local`, exact re-ranking against `refine`
 * (numbered by position like `local`), the exact per-shard lists exchanged. */
typedef struct mi_shards mi_shards;
int mi_shards_unique_id(const char *rccl_lib, void *id128);
int mi_shards_create(mi_index *local, mi_flat *refine, int k_factor, int rank, int world,
                     const void *id128, const char *rccl_lib, int64_t id_mul, int64_t id_add,
                     int64_t id_step, mi_shards **out);
int mi_shards_search(mi_shards *s, int64_t nq, const float *q, int k, int nprobe, float *D,
                     int64_t *I, void *stream);
int mi_shards_destroy(mi_shards *s);

/* ---- IndexFlatIP (config #1; also the coarse quantiser's arithmetic) ---- */

int mi_flat_create(int d, int device, mi_flat **out);
/* storage = MI_STORE_F16: components kept as IEEE half, rounded to nearest-even at add() with no
 * scaling -- faiss's IndexScalarQuantizer(d, QT_fp16), the refine index of the factory string
 * "...,Refine(SQfp16)".  Scores are <q, (float)x16> in the same f32 chain as the f32 store: half
 * the bytes per re-ranked candidate.  Such a store serves mi_flat_rerank / reconstruct_n only. */
#define MI_STORE_F32 0
#define MI_STORE_F16 1
/* storage = MI_STORE_SQ8: faiss's IndexScalarQuantizer(d, QT_8bit), the refine index of "...,Refine(SQ8)": one
 * byte per component with per-dimension ranges.  The ranges must be trained (mi_flat_sq_train = faiss
 * ScalarQuantizer::train with RS_minmax, rangestat_arg 0: vmin[i] = min, vdiff[i] = max - min over the training
 * rows) or set (mi_flat_sq_set_trained) before add(); add() stores code = (int)(255 * clip((x - vmin) / vdiff));
 * reconstruct_n decodes x^[i] = fma(fma(code, 1/255, 0.5/255), vdiff[i], vmin[i]); re-ranking scores <q, x^> in the
 * asymmetric form A(q) + sum_i (q[i] vdiff[i] / 255) code[i] -- a per-query table, then one ascending f32 fma chain over
 * the stored bytes (oracle/ivfpq_oracle.c, section "ScalarQuantizer QT_8bit").  207 M x 1024
 * components are 212 GB: the whole refine store of BASELINE.json configs[3] beside the index in one GPU's HBM.
 * Inner product only; serves mi_flat_rerank / reconstruct_n only. */
#define MI_STORE_SQ8 2
int mi_flat_create_ex(int d, int device, int storage, mi_flat **out);
/* ScalarQuantizer.train(x): x float32 [n][d], host or device.  `merge` != 0 widens the ranges already trained
 * instead of replacing them (training over a corpus that arrives in chunks). */
int mi_flat_sq_train(mi_flat *h, int64_t n, const float *x, int merge);
/* ScalarQuantizer.trained: float32 [2 d] = vmin[d] | vdiff[d] (host pointers). */
int mi_flat_sq_get_trained(mi_flat *h, float *trained);
int mi_flat_sq_set_trained(mi_flat *h, const float *trained);
int mi_flat_sq_is_trained(mi_flat *h, int *out);
/* faiss.IndexFlat(d, metric): MI_METRIC_L2 = IndexFlatL2 -- squared L2 distances, ascending,
 * unfilled slots +FLT_MAX -- evaluated through the expansion |q|^2 + |x|^2 - 2<q, x> on
 * augmented rows (see "Metrics" above). */
int mi_flat_create_metric(int d, int metric, int device, mi_flat **out);
int mi_flat_destroy(mi_flat *h);
int mi_flat_add(mi_flat *h, int64_t n, const float *x);
/* Capacity hint (std::vector::reserve on faiss's IndexFlat::codes): room for n vectors in
 * total, so that a 100 GB store filled in chunks never holds two copies while growing. */
int mi_flat_reserve(mi_flat *h, int64_t n);
/* IndexFlat.reconstruct_n: vectors [i0, i0 + n) as stored, float32 [n][d] (host or device
 * output) -- e.g. the centroids of an IndexFlat handed to IndexIVFPQ as its quantizer. */
int mi_flat_reconstruct_n(mi_flat *h, int64_t i0, int64_t n, float *out);
/* The STORED bytes of the rows `ids` (int64 [n], host or device, each in [0, ntotal)) -> out (host or device):
 * d floats (f32 store; d + 4 for METRIC_L2's augmented rows), d IEEE halves (QT_fp16) or d code bytes (QT_8bit) per
 * row, in the order of `ids`.  faiss reads the same bytes through IndexFlatCodes::codes / sa_encode; here it is
 * the parity hook that hands a sample of a store too large to export (212 GB at cfg4) to the oracle
 * (bench.py at_recall_095.parity_vs_oracle; reference call site Makefile:32 `tune` -> the operating point). */
int mi_flat_get_rows(mi_flat *h, int64_t n, const int64_t *ids, void *out);
/* Frees the per-stream scratch buffers (a set per stream that has ever searched: ~2-3 GB each at batch 1024 over
 * 65 536 lists) after the device has drained; the next call on a stream allocates its set again.  Not to be called
 * while another thread is inside a call on the same handle.  (faiss keeps no such state: its search allocates per
 * call; a caller that searched on many streams and now needs the HBM -- bench.py before it exports the lists --
 * gives it back here.) */
int mi_flat_release_workspaces(mi_flat *h);
int mi_index_release_workspaces(mi_index *h);
int mi_flat_ntotal(mi_flat *h, int64_t *out);
int mi_flat_reset(mi_flat *h);
int mi_flat_search(mi_flat *h, int64_t nq, const float *q, int k, float *D, int64_t *I,
                   void *stream);
/* Re-ranking step of faiss's IndexRefineFlat::search (the refine index is an IndexFlat;
 * reference call site: the same `index tune` / query-time search, Makefile:32, README.md:28,
 * when the factory string carries ",RFlat").  cand_I int64 [nq][kc] are candidate ids from a
 * base index (positions in this flat index; negative = empty slot), kc a multiple of k.
 * Exact inner products of every query with its candidates -- the same arithmetic as
 * mi_flat_search -- then the k best under (score desc, id asc); unfilled: -1 / -FLT_MAX.
 * q, cand_I, D, I: all host or all device pointers. */
int mi_flat_rerank(mi_flat *h, int64_t nq, const float *q, int kc, const int64_t *cand_I, int k,
                   float *D, int64_t *I, void *stream);

/* ---- tuning knobs ------------------------------------------------------ */

/* The library reads its MI_* environment knobs (README.md lists them; each overrides a measured dispatch rule) ONCE, at the
 * first call that needs one, into a process-wide struct: no search / add call path calls getenv.  This re-reads them --
 * for tests and tools that switch a knob inside one process; not to be called while another thread is inside the library.
 * No faiss counterpart. */
int mi_ivfpq_reload_env(void);

/* ---- building blocks used by train() in the Python mirror ---------- */

/* out[n][nc] = x . c^T (+ bias[nc] when not NULL), exact f32 (every element an ascending-k fmaf chain: the coarse
 * quantiser's GEMM as a plain operator).  Replaces faiss VectorTransform::apply / LinearTransform::apply_noalloc (sgemm)
 * in front of an IndexPreTransform ("OPQ64,IVF...": the factory string reference Makefile:39 leaves to sidecar-search's
 * default).  Device pointers; d % 4 == 0. */
int mi_ip_gemm(int device, int64_t n, const float *x, int64_t nc, const float *c, int d,
               const float *bias, float *out, void *stream);

/* arg max_j <x_i, c_j> (ties: smallest j) -> assign int32 [n], score float32 [n]
 * (host or device outputs; score may be NULL).  Clustering assignment step. */
int mi_ip_assign(int device, int64_t n, const float *x, int64_t nc, const float *c, int d,
                 int32_t *assign, float *score, void *stream);

/* The update step of Lloyd's k-means, deterministic: centroids[c] = mean of the rows x_i with
 * assign[i] == c, the members summed in ascending i by sequential f32 adds, divided by their
 * count (faiss Clustering's centroid update, with a fixed summation order so that training is
 * bit-reproducible and restated by the oracle).  Clusters without members keep their row of
 * `centroids`; counts int32 [k] (host or device, may be NULL).  x, assign, centroids: device. */
int mi_cluster_means(int device, int64_t n, const float *x, int d, const int32_t *assign, int k,
                     float *centroids, int32_t *counts, void *stream);

/* out[r] = -0.5 * <x_r, x_r> (the dot an ascending-k fmaf chain from +0): the augmenting column
 * that makes arg max of an inner product the arg min of the L2 distance in mi_ip_assign. */
int mi_neg_half_sqnorm(int device, int64_t n, const float *x, int d, float *out, void *stream);

/* ProductQuantizer.compute_codes: codes uint8 [n][M] (host or device output),
 * codebook float32 [M][256][d/M]. */
int mi_pq_encode(int device, int64_t n, const float *x, int d, int M, const float *codebook,
                 uint8_t *codes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_IVFPQ_H */
