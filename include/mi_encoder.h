/*
 * mi_encoder.h -- C ABI of the MI355X-native text encoder (encode hot path).
 *
 * Replaces what runs underneath `SentenceTransformer(model).encode(batch)` in
 * the reference: `sidecar-search build -b 32` (reference Makefile:65,
 * README.md:60) and the query-time app (reference README.md:28) -- the Qwen2
 * decoder stack of stella_en_1.5B_v5, mean pooling, the Dense 1536->1024 module
 * and L2 normalisation.  Tokenisation stays on the host (the `tokenizers`
 * library); this ABI starts at token ids.  The Python mirror of the
 * sentence-transformers surface is abstracts-search_amd/sentence_transformers.py.
 *
 * Conventions: as in mi_ivfpq.h (int status, mi_enc_last_error(), opaque
 * handle, plain pointers, hipStream_t as void*).  Threading: mi_encoder_encode /
 * mi_encoder_hidden are safe to call from several host threads on one handle --
 * activations live in one workspace set per stream, leased for the duration of a
 * call (threads on distinct streams overlap on the GPU, threads sharing a stream
 * take turns); create / load_tensor / destroy are exclusive.  Token ids arrive PACKED:
 * ids[T] with cu_seqlens[nseq+1] (sequence i is ids[cu[i] .. cu[i+1])), host or
 * device pointers.  Weights are stored and multiplied in bf16 with f32
 * accumulation (MFMA); norms, softmax, pooling, Dense and the final
 * normalisation are f32.  Output is float32 [nseq][out_dim].
 */
#ifndef MI_ENCODER_H
#define MI_ENCODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_encoder mi_encoder;

typedef struct mi_encoder_cfg {
    int32_t vocab_size;
    int32_t hidden;        /* 1536 */
    int32_t n_layers;      /* 28 */
    int32_t n_heads;       /* 12 */
    int32_t n_kv_heads;    /* 2 */
    int32_t head_dim;      /* 128 (64 also supported) */
    int32_t intermediate;  /* 8960 */
    float rms_eps;         /* 1e-6 */
    float rope_theta;      /* 1e6 */
    int32_t causal;        /* 0 = bidirectional (stella), 1 = causal (plain Qwen2) */
    int32_t dense_out;     /* 1024; 0 = no Dense module */
    int32_t dense_bias;    /* 1 */
    int32_t max_seq_len;   /* 512 */
} mi_encoder_cfg;

#define MI_DTYPE_F32 0
#define MI_DTYPE_BF16 1
#define MI_DTYPE_F16 2

const char *mi_enc_last_error(void);

/* SentenceTransformer(model_name_or_path, device=...) : allocate the model. */
int mi_encoder_create(const mi_encoder_cfg *cfg, int device, mi_encoder **out);
int mi_encoder_destroy(mi_encoder *h);

/* Load one parameter by its HF name without the "model." prefix, e.g.
 * "embed_tokens.weight", "layers.3.self_attn.q_proj.weight", "norm.weight",
 * "dense.weight", "dense.bias".  `data` is a host or device pointer to a dense
 * row-major tensor of `dtype`; it is converted to the internal layout. */
int mi_encoder_load_tensor(mi_encoder *h, const char *name, const void *data, int dtype,
                           const int64_t *shape, int ndim);
/* Number of parameters still missing (0 = ready to encode). */
int mi_encoder_missing(mi_encoder *h, int *count);
int mi_encoder_out_dim(mi_encoder *h, int *out);

/* SentenceTransformer.encode on token ids: Transformer -> mean Pooling ->
 * Dense -> optional L2 normalisation.  out: float32 [nseq][out_dim], host or
 * device.  total tokens = cu_seqlens[nseq]; each sequence 1..max_seq_len tokens. */
int mi_encoder_encode(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu_seqlens,
                      int normalize, float *out, void *stream);

/* The same with the caller's un-sort folded in: embedding i is written to row out_rows[i] of `out` (a device pointer;
 * out_rows a host int32 [nseq]).  sentence-transformers sorts its inputs by length and restores the input order
 * afterwards; here the last kernel of the call does that.  With host `ids` / `cu_seqlens` both encode calls only ENQUEUE
 * work on `stream` and return (token ids travel through pinned staging slots): the host prepares the next pass while the
 * GPU runs this one. */
int mi_encoder_encode_rows(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu_seqlens,
                           int normalize, float *out, const int32_t *out_rows, void *stream);

/* Parity hook: the stack's last hidden state after the final norm,
 * float32 [T][hidden] (host or device), packed like ids. */
int mi_encoder_hidden(mi_encoder *h, int nseq, const int32_t *ids, const int32_t *cu_seqlens,
                      float *out, void *stream);

/* Timing of the dominant kernel family for the roofline: with profiling on, mi_encoder_encode() records the arguments of
 * its bf16 MFMA GEMM launches; mi_encoder_profile_read() re-issues exactly those launches back to back on the same
 * stream between two HIP events (one warm pass, three timed) and returns the device time of one pass (ms) and the FLOPs
 * the launches perform.  (The residual GEMMs add into the activations again: call it after the embeddings were read.) */
int mi_encoder_profile_enable(mi_encoder *h, int on);
int mi_encoder_profile_read(mi_encoder *h, double *gemm_ms, double *gemm_flops);

/* Test hook: process-wide counters of code paths the dispatcher takes on its own.  "tail_split_launches" = GEMM
 * launches whose last (partial) round of 256x256 tiles was split along K with f32 atomics into the residual stream;
 * "splitk_launches" = residual GEMMs whose every tile was split along K through the workspace (a few hundred to ~5 000
 * tokens); "reduce_norm_launches" = those whose reduction pass also wrote the next RMSNorm; "n192_launches" = residual
 * GEMMs run on 256 x 192 tiles. */
int mi_enc_debug_counter(const char *name, int64_t *value);

/* The library reads its MI_* environment knobs (README.md lists them; each overrides a measured dispatch rule) ONCE, at the
 * first call that needs one, into a process-wide struct: no encode call path calls getenv.  This re-reads them -- for tests
 * and tools that switch a knob inside one process; not to be called while another thread is inside the library. */
int mi_encoder_reload_env(void);

/* Building block exposed for numerics tests: C[M][N] = A[M][K] . W[N][K]^T in
 * bf16 with f32 accumulation (device pointers; C bf16 row-major). */
int mi_enc_gemm_bf16(int device, int M, int N, int K, const void *A, const void *W, void *C,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_ENCODER_H */
