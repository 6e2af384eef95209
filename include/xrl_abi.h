/*
 * xrl_abi.h -- C ABI of libxrl_amd.so, the MI355X-native XR-Linear inference library.
 *
 * Part 1 declares, with IDENTICAL names, argument order and meaning, the entry points the
 * reference (amzn/pecos @ 2024-10-20) exports from pecos.core.libpecos_float32 for its
 * XR-Linear inference path, so that the reference's ctypes binding
 * (pecos/core/base.py:799-976 link_xlinear_methods, :1499-1534 link_sparse_operations) can bind
 * this library for that path without change.  Every declaration cites the reference line it
 * replaces.  Part 2 is ADDITIVE (prefix xrl_): error reporting, device selection and a
 * device-resident variant of predict used for benchmarking and multi-GPU sharding.
 *
 * All pointers are plain host pointers unless a parameter is documented as a device pointer.
 * No torch / C++ types appear in any signature.
 */
#ifndef XRL_ABI_H
#define XRL_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------------------------------
 * Matrix views handed over by the caller (read-only, valid for the duration of the call).
 * Layout-identical to pecos/core/utils/matrix.hpp:49-71 and the ctypes mirrors in
 * pecos/core/base.py:172-354.
 * ---------------------------------------------------------------------------------------- */
typedef struct {            /* matrix.hpp:49-55  ScipyCsrF32 */
    uint32_t rows, cols;
    uint64_t* row_ptr;      /* [rows+1] */
    uint32_t* col_idx;      /* [nnz], ascending inside a row (base.py:1073-1076) */
    float* val;             /* [nnz] */
} ScipyCsrF32;

typedef struct {            /* matrix.hpp:56-62  ScipyCscF32 */
    uint32_t rows, cols;
    uint64_t* col_ptr;      /* [cols+1] */
    uint32_t* row_idx;      /* [nnz], ascending inside a column */
    float* val;             /* [nnz] */
} ScipyCscF32;

typedef struct {            /* matrix.hpp:63-67  ScipyDrmF32: dense row-major */
    uint32_t rows, cols;
    float* val;             /* [rows*cols] */
} ScipyDrmF32;

typedef struct {            /* matrix.hpp:68-72  ScipyDcmF32: dense column-major */
    uint32_t rows, cols;
    float* val;             /* [rows*cols] */
} ScipyDcmF32;

/* matrix.hpp:47.  The callee passes the ADDRESSES of its three result pointers; the caller
 * allocates indices u32[nnz], indptr u64[rows+1 | cols+1], data f32[nnz] and writes their
 * addresses back (pecos/core/base.py:431-464).  Invoked exactly once per predict call,
 * synchronously on the calling thread, after all device work has completed. */
typedef void (*py_sparse_allocator_t)(bool is_col_major, uint64_t rows, uint64_t cols,
                                      uint64_t nnz, void* indices_pp, void* indptr_pp,
                                      void* data_pp);

/* ------------------------------------------------------------------------------------------
 * Part 1: reference-compatible entry points
 * ---------------------------------------------------------------------------------------- */

/* pecos/core/libpecos.cpp:116-119.  model_path = <model>/ranker (param.json + {d}.model/). */
void* c_xlinear_load_model_from_disk(const char* model_path);

/* libpecos.cpp:121-126.  weight_matrix_type in {0 CSC, 1 HASH_CHUNKED, 2 BINARY_SEARCH_CHUNKED}
 * (pecos/core/base.py:49).  This library has ONE device layout; the value is remembered only so
 * that c_xlinear_get_layer_type answers like the reference. */
void* c_xlinear_load_model_from_disk_ext(const char* model_path, int weight_matrix_type);

/* libpecos.cpp:128-131: load a memory-mapped model folder (*.mmap_store, param.json with is_mmap).
 * lazy_load is accepted and ignored: the model is copied to HBM either way. */
void* c_xlinear_load_mmap_model_from_disk(const char* model_path, const bool lazy_load);

/* libpecos.cpp:133-138: npz model folder -> mmap model folder in the reference's byte layout
 * (readable by the reference's own loader).  Host-only, needs no GPU. */
void c_xlinear_compile_mmap_model(const char* model_path, const char* mmap_model_path);

/* libpecos.cpp:140-143 */
void c_xlinear_destruct_model(void* ptr);

/* libpecos.cpp:147-150; attr in {depth, nr_features, nr_labels, nr_codes} (inference.hpp:2367-2379).
 * Additive attrs: nr_pred_cols (columns of predict()'s CSR), nr_bucket_layers / nr_bitmap64_layers (layers on the bucket / 64-feature-word row lookup). */
uint32_t c_xlinear_get_int_attr(void* ptr, const char* attr);

/* libpecos.cpp:152-156 */
int c_xlinear_get_layer_type(void* ptr, int layer_depth);

/* libpecos.cpp:158-175 (C_XLINEAR_PREDICT).  0 / NULL overrides mean "use the model's per-layer
 * param.json value" (inference.hpp:2055-2058).  `threads` is accepted and ignored (the work runs
 * on the GPU).  Result: CSR rows x nr_labels, rows score-sorted, at most only_topk per row. */
void c_xlinear_predict_csr_f32(void* ptr, const ScipyCsrF32* input_x,
                               const uint32_t overridden_beam_size,
                               const char* overridden_post_processor_str,
                               const uint32_t overridden_only_topk, const int threads,
                               py_sparse_allocator_t pred_alloc);
void c_xlinear_predict_drm_f32(void* ptr, const ScipyDrmF32* input_x,
                               const uint32_t overridden_beam_size,
                               const char* overridden_post_processor_str,
                               const uint32_t overridden_only_topk, const int threads,
                               py_sparse_allocator_t pred_alloc);

/* libpecos.cpp:179-198 (C_XLINEAR_PREDICT_ON_SELECTED_OUTPUTS): scores for a given (query, label)
 * pattern.  The reference only offers this for weight_matrix_type == CSC (inference.hpp:2143-2147) and
 * uses the CSC arithmetic (vector_ops::inner_product, :1018-1078); this library accepts any handle and
 * reproduces that arithmetic and the reference's output order (the walk of
 * prolongate_sparse_predictions, :1302-1358).  Result CSR has selected_outputs_csr's row_ptr. */
void c_xlinear_predict_on_selected_outputs_csr_f32(void* ptr, const ScipyCsrF32* input_x,
                                                   const ScipyCsrF32* selected_outputs_csr,
                                                   const char* overridden_post_processor_str,
                                                   const int threads, py_sparse_allocator_t pred_alloc);
void c_xlinear_predict_on_selected_outputs_drm_f32(void* ptr, const ScipyDrmF32* input_x,
                                                   const ScipyCsrF32* selected_outputs_csr,
                                                   const char* overridden_post_processor_str,
                                                   const int threads, py_sparse_allocator_t pred_alloc);

/* libpecos.cpp:201-235 (C_XLINEAR_SINGLE_LAYER_PREDICT): one layer from caller-owned W / C,
 * optional previous-layer predictions csr_codes (NULL => all-ones N x C.cols, no combine). */
void c_xlinear_single_layer_predict_csr_f32(const ScipyCsrF32* input_x, const ScipyCsrF32* csr_codes,
                                            ScipyCscF32* W, ScipyCscF32* C,
                                            const char* post_processor_str, const uint32_t only_topk,
                                            const int num_threads, const float bias,
                                            py_sparse_allocator_t pred_alloc);
void c_xlinear_single_layer_predict_drm_f32(const ScipyDrmF32* input_x, const ScipyCsrF32* csr_codes,
                                            ScipyCscF32* W, ScipyCscF32* C,
                                            const char* post_processor_str, const uint32_t only_topk,
                                            const int num_threads, const float bias,
                                            py_sparse_allocator_t pred_alloc);

/* libpecos.cpp:237-274 (C_XLINEAR_SINGLE_LAYER_PREDICT_ON_SELECTED_OUTPUTS) */
void c_xlinear_single_layer_predict_on_selected_outputs_csr_f32(const ScipyCsrF32* input_x, const ScipyCsrF32* selected_outputs_csr,
                                                                const ScipyCsrF32* csr_codes, ScipyCscF32* W, ScipyCscF32* C,
                                                                const char* post_processor_str, const int num_threads,
                                                                const float bias, py_sparse_allocator_t pred_alloc);
void c_xlinear_single_layer_predict_on_selected_outputs_drm_f32(const ScipyDrmF32* input_x, const ScipyCsrF32* selected_outputs_csr,
                                                                const ScipyCsrF32* csr_codes, ScipyCscF32* W, ScipyCscF32* C,
                                                                const char* post_processor_str, const int num_threads,
                                                                const float bias, py_sparse_allocator_t pred_alloc);

/* libpecos.cpp:337-355 (C_SPARSE_INNER_PRODUCTS): val[i] = <X[X_row_idx[i],:], W[:,W_col_idx[i]]>,
 * `val` is caller-allocated f32[len] (pecos/core/utils/matrix.hpp:1049-1060). */
void c_sparse_inner_products_csr2csc_f32(const ScipyCsrF32* pX, const ScipyCscF32* pW, uint64_t len,
                                         uint32_t* X_row_idx, uint32_t* W_col_idx, float* val, int threads);
void c_sparse_inner_products_drm2csc_f32(const ScipyDrmF32* pX, const ScipyCscF32* pW, uint64_t len,
                                         uint32_t* X_row_idx, uint32_t* W_col_idx, float* val, int threads);
void c_sparse_inner_products_csr2dcm_f32(const ScipyCsrF32* pX, const ScipyDcmF32* pW, uint64_t len,
                                         uint32_t* X_row_idx, uint32_t* W_col_idx, float* val, int threads);
void c_sparse_inner_products_drm2dcm_f32(const ScipyDrmF32* pX, const ScipyDcmF32* pW, uint64_t len,
                                         uint32_t* X_row_idx, uint32_t* W_col_idx, float* val, int threads);

/* ------------------------------------------------------------------------------------------
 * Part 2: additive entry points (no reference counterpart)
 * ---------------------------------------------------------------------------------------- */

/* The reference throws C++ exceptions through extern "C" (process abort).  This library catches
 * everything; a failed call returns (NULL / 0 / without invoking the allocator) and leaves a
 * message here.  Thread-local; NULL when the last call on this thread succeeded. */
const char* xrl_last_error(void);
void xrl_clear_error(void);

const char* xrl_version(void);
int xrl_device_count(void);            /* hipGetDeviceCount; 0 when no GPU is visible */
int xrl_set_device(int device);        /* device used by subsequent loads on this thread; 0 = ok */

/* Host-only: parse a model folder exactly like the loader does (param.json + uncompressed npz),
 * WITHOUT touching a GPU.  Writes per layer {W.rows, W.cols, nnz(W), C.rows, C.cols, nnz(C)} into
 * out[6*layer ...] (up to cap values) and returns the depth, or -1 on error (see xrl_last_error). */
int xrl_inspect_model(const char* model_path, uint64_t* out, uint32_t cap);

/* Build a model from in-memory CSC layers (same semantics as loading the folder). */
void* xrl_model_create(uint32_t depth, const ScipyCscF32* const* W, const ScipyCscF32* const* C,
                       const float* bias, const uint32_t* only_topk,
                       const char* const* post_processor);

/* Device-resident queries: upload once, predict many times (bench / multi-GPU shards). */
void* xrl_queries_upload_csr(void* model, const ScipyCsrF32* X);
void* xrl_queries_upload_drm(void* model, const ScipyDrmF32* X);
/* Queries that are ALREADY in HBM (a GPU featurizer, a torch tensor): wrap device pointers without copying -- CSR with u64
 * row_ptr[rows+1], u32 col_idx (sorted inside every row, as the reference requires, pecos/core/base.py:1073-1076), f32 val; or a
 * dense row-major f32 matrix.  The handle does not own the arrays; they must stay valid until xrl_queries_free.  This is the
 * hand-off for the callers of SURVEY.md N4 (c_tfidf_predict's output, Text2Text, pecos/apps/text2text/model.py:416-417): no host
 * round trip of X. */
void* xrl_queries_from_device_csr(void* model, uint32_t rows, uint32_t cols, const uint64_t* d_row_ptr,
                                  const uint32_t* d_col_idx, const float* d_val, uint64_t nnz);
void* xrl_queries_from_device_drm(void* model, uint32_t rows, uint32_t cols, const float* d_val);
/* The weighting half of the reference's TF-IDF vectorizer on the device (BaseVectorizer::get_sorted_feature,
 * pecos/core/utils/tfidf.hpp:798-822; c_tfidf_predict, libpecos.cpp:427-445): a device CSR of term COUNTS (column ids ascending
 * inside every row, what the reference's tokenizer + n-gram lookup produce) -> binary / sublinear tf -> x idf (d_idf[cols], NULL =
 * use_idf false) -> l1 / l2 normalisation (norm_p 1 | 2), float32 operation by operation like the reference (bit-identical; with
 * sublinear_tf <= 1 ulp, the device's logf).  The weighted values go to d_out (nnz floats, caller-owned; may alias d_count) or, with
 * d_out == NULL, into a buffer the returned query handle owns; the handle REFERENCES d_row_ptr / d_col_idx (and d_out): keep them
 * alive.  It feeds xrl_predict_device directly: the queries never visit the host. */
void* xrl_queries_tfidf_device(void* model, uint32_t rows, uint32_t cols, const uint64_t* d_row_ptr, const uint32_t* d_col_idx,
                               const float* d_count, uint64_t nnz, const float* d_idf, int binary, int sublinear_tf, int norm_p,
                               float* d_out, void* hip_stream);
/* ---- TF-IDF query producer with the reference's own entry points (libpecos.cpp:398-445 -> pecos/core/utils/tfidf.hpp) --------------
 * c_tfidf_load reads a folder written by the reference's Tfidf.save (one BaseVectorizer folder: tokenizer/{config.json, vocab.txt},
 * vectorizer/{config.json, tfidf-model.txt}; or an ensemble: meta.json + <i>.base/) -- host only.  c_tfidf_predict has the
 * reference's signature and result (host CSR through the allocator, rows = documents, sorted feature ids): the tokenizer and the
 * n-gram lookup run on host threads (`threads`, <= 0: all), their term counts go to the device once, weighting and normalisation run
 * there (K5) and the result is copied back -- bit-identical to the reference (sublinear_tf: <= 1 ulp).  Training, saving and the
 * *_from_file variants stay the reference's.
 * xrl_tfidf_predict_device is the same pipeline WITHOUT the copy back: it returns a query handle (xrl_queries_free) whose X lives in
 * the HBM of `model`'s device and feeds xrl_predict_device / xrl_predict_device_rows directly -- the call sites SURVEY.md 8f N4 names
 * (pecos/apps/text2text/model.py:416-417: preprocessor.predict -> xlinear predict) without any host round trip of X.
 * xrl_tfidf_counts (host only, tests): the hstacked CSR of term COUNTS the device half starts from. */
void* c_tfidf_load(const char* model_dir);
void c_tfidf_destruct(void* ptr);
void c_tfidf_predict(void* ptr, void* corpus_ptr /* const char** */, const size_t* doc_lens, size_t nr_doc, int threads, py_sparse_allocator_t pred_alloc);
/* libpecos.cpp:413-425: one document per line of a text file (buffer_size only sizes the reference's read chunks: ignored) */
void c_tfidf_predict_from_file(void* ptr, void* corpus_fname_ptr /* const char* */, size_t fname_len, size_t buffer_size, int threads, py_sparse_allocator_t pred_alloc);
uint32_t xrl_tfidf_nr_features(void* ptr);
void* xrl_tfidf_predict_device(void* vectorizer, void* model, void* corpus_ptr /* const char** */, const size_t* doc_lens, size_t nr_doc, int threads);
void xrl_tfidf_counts(void* ptr, void* corpus_ptr /* const char** */, const size_t* doc_lens, size_t nr_doc, int threads, py_sparse_allocator_t alloc);
/* xrl_queries_concat_device_ex for a CSR query handle (e.g. xrl_tfidf_predict_device's): [X of the handle | X_emb] as a NEW handle on the
 * same device (XR-Transformer's concat_model call site, pecos/xmc/xtransformer/model.py:589-603, with both halves device-resident). */
void* xrl_queries_concat_handle(void* model, void* queries, uint32_t dense_cols, const float* d_emb, int normalize_emb, void* hip_stream);
/* The query form of XR-Transformer's concat_model (TransformerMatcher.concat_features + smat_util.hstack_csr,
 * pecos/xmc/xtransformer/matcher.py:864-890, model.py:589-603): [X_feat (device CSR, sparse_cols columns) | X_emb (device dense
 * rows x dense_cols)] assembled into one device CSR owned by the returned handle; every cell of the dense block becomes a stored
 * entry (zeros included), as smat_util.dense_to_csr does.  _ex with normalize_emb != 0 also applies sklearn's row-wise l2
 * normalize to X_emb on the device (the reference's default, matcher.py:879-880; values agree to ~1e-7 relative). */
void* xrl_queries_concat_device(void* model, uint32_t rows, uint32_t sparse_cols, const uint64_t* d_row_ptr,
                                const uint32_t* d_col_idx, const float* d_val, uint64_t nnz, uint32_t dense_cols,
                                const float* d_emb, void* hip_stream);
void* xrl_queries_concat_device_ex(void* model, uint32_t rows, uint32_t sparse_cols, const uint64_t* d_row_ptr,
                                   const uint32_t* d_col_idx, const float* d_val, uint64_t nnz, uint32_t dense_cols,
                                   const float* d_emb, int normalize_emb, void* hip_stream);
/* Shape of a query handle: out4 = {rows, cols, stored values (nnz, or rows x cols for a dense one), dense (0/1)}; and a copy of its
 * arrays back to the host (tests, debugging): CSR handles fill row_ptr[rows+1] / col_idx[nnz] / val[nnz], dense ones val[rows x cols]. */
int xrl_queries_info(void* queries, uint64_t* out4);
int xrl_queries_download(void* queries, uint64_t* row_ptr, uint32_t* col_idx, float* val);
void xrl_queries_free(void* queries);

/* Beam search with inputs already resident in HBM.  Writes fixed-stride results
 *   d_out_idx u32[rows*out_stride], d_out_val f32[rows*out_stride], d_out_cnt u32[rows]
 * into caller-provided DEVICE buffers (e.g. torch tensors) on `hip_stream` (a hipStream_t, NULL =
 * the library's own stream -- so a caller working on the legacy default stream, whose handle is NULL, must either pass
 * `sync` != 0 or use an explicit stream) and returns without synchronising when `sync` == 0.
 * out_stride must be >= the effective only_topk.  Returns 0 on success. */
int xrl_predict_device(void* model, void* queries, uint32_t beam_size, const char* post_processor,
                       uint32_t only_topk, uint32_t* d_out_idx, float* d_out_val,
                       uint32_t* d_out_cnt, uint32_t out_stride, void* hip_stream, int sync);

/* The same for rows [row_begin, row_begin + row_count) of the queries only; results land at the SAME rows of the output buffers.
 * Lets a caller pipeline a shard in pieces (bench.py: the all-gather of the first half runs under the second half's kernels). */
int xrl_predict_device_rows(void* model, void* queries, uint32_t beam_size, const char* post_processor,
                            uint32_t only_topk, uint32_t* d_out_idx, float* d_out_val,
                            uint32_t* d_out_cnt, uint32_t out_stride, void* hip_stream, int sync,
                            uint32_t row_begin, uint32_t row_count);

/* Effective only_topk of the last layer for the given override (0 = model default). */
uint32_t xrl_effective_topk(void* model, uint32_t only_topk);

/* Profiling: when enabled, every kernel launch of predict is bracketed by a hipEvent pair recorded
 * on the stream the kernel is launched on (no synchronisation is added to the predict call).
 * xrl_profile_get synchronises, folds the pending pairs and fills up to `cap` records; it returns
 * the number of records available. */
typedef struct {
    char name[32];          /* kernel family: "k0_prolongate", "k1_sparse", "k1_dense", "k1q_dense", "k1q_dense_x", "k1c_csc", "k2_topk" */
    uint32_t layer;
    uint32_t launches;
    double ms;              /* accumulated GPU time of those launches */
    double reserved;
} xrl_profile_rec_t;
void xrl_profile_enable(void* model, int enable);
void xrl_profile_reset(void* model);
uint32_t xrl_profile_get(void* model, xrl_profile_rec_t* out, uint32_t cap);

/* One untimed predict (tile-format kernels) that also counts, per layer l, 8 doubles at stats_out[8l ...]:
 *   [0] algorithmic bytes of the reference-layout chunks streamed (8*E_p + 4*R_p + 4*(R_p+1) per (query, beam parent),
 *       every active chunk whole, no inter-query reuse)      [1] candidates evaluated
 *   [2] (query, tile) items   [3] row lookups (query features x items)   [4] matched tile rows   [5] entries of the matched rows
 *   [6] tile columns over the items (scores written)   [7] query features x tile columns over the items (dense-format weights)
 * [2..7] are the MATCHED work bench.py's roofline model is built from.  stats_cap >= 8*depth.  Returns 0 on success. */
int xrl_predict_stats(void* model, void* queries, uint32_t beam_size, const char* post_processor,
                      uint32_t only_topk, double* stats_out, uint32_t stats_cap);

/* Host-only pieces of the model compiler, exported so that they can be tested without a GPU.
 * xrl_debug_split_chunk: number of even column tiles for a chunk of n columns whose entry-count prefix sums are
 *   cum[0..n], such that every tile has <= 128 columns and fewer than `limit` entries (0: impossible).
 * xrl_debug_layout_rows: places a tile's rows (rptr[0..nrows] = packed CSR starts); with align != 0 a row that would touch
 *   more 128-byte lines than its length requires starts on the next 16-entry boundary.  Writes one packed extent
 *   `offset | (len-1) << 25` per row to ext_out (may be NULL) and returns the padded entry count of the tile. */
uint32_t xrl_debug_split_chunk(const uint64_t* cum, uint32_t n, uint64_t limit);
uint64_t xrl_debug_layout_rows(const uint32_t* rptr, uint32_t nrows, int align, uint32_t* ext_out);

/* Tuning knobs (benchmark / tests only).  Results never depend on them.
 *   "k1_group"            lanes per (query, tile) item in K1: 0 = auto, else a power of two <= 64
 *   "max_batch_rows"      rows of X per internal batch (0 = auto: candidate buffer <= 6 GiB)
 *   "sort_min_tiles"      tile-sort the items of layers with at least this many tiles (0 = never)
 *   "adaptive"            1 (default): PRUNING FEEDBACK -- the handle remembers, per layer, what the first stage of the bound pruning achieved on its
 *                         previous predicts (sampled device counters / the later stage's item count, read back without any synchronisation);
 *                         a layer whose first stage settled fewer than ~35 % of the queries runs UNSTAGED on the following predicts (every
 *                         candidate in one pass; tile format: on tile-sorted items) and is probed again every 32nd predict; 0: always staged
 *   "presence"            1 (default): K1Q on sparse X asks the layer's PRESENCE words (one bit per (feature, 16..32-column parent tile): "holds a weight")
 *                         before requesting a weight segment, on the layers that run unstaged -- prune = 0, or switched by the pruning feedback:
 *                         there every beam parent's segments are addressed and a third to a half of them are empty (Amazon-670K-hard: levels 0-3
 *                         11.9 -> 7.5 ms); 2: on every layer that has the words; 0: never.  XRL_PRESENCE=0 at load builds none.
 *   "prune_mid"           1 (default): bound-pruned tile-format layers entered with >= 16 beam parents score slots 1..4 in a MIDDLE stage before
 *                         "every remaining slot" (three stages instead of two; Wiki10-31K's beam of 20); 0: two stages
 *   "sort_rest"           1 (default): the second phase of a bound-pruned tile-format layer runs on tile-sorted items (counting sort of the
 *                         compacted list by tile: the items of a tile run back to back on one XCD and share its lookup words and entries in
 *                         that XCD's L2); 0: in query order
 *   "sort_rest_min"       32768 (default): ... and only while the later stages of the handle's previous predicts held at least max(this many, 32 per tile
 *                         of the layer) items (the pruning feedback's count): below that the sort's four launches cost more than the locality buys; 0: always sort
 *   "qsort"               1 (default): K1Q, sparse X -- the LAST layer of a run of dense-format layers is launched on its own, on queries counting-sorted
 *                         by the best parent of their beam, every XCD taking a contiguous eighth of the sorted order (queries of one region of the
 *                         tree share (feature, parent) segments in that XCD's L2), when the layer has >= "qsort_min_parents" (64) parents and the
 *                         row batch >= "qsort_min_rows" (131072) rows; 0: never.  Results do not depend on it (round 5: L2 misses -27 %, time unchanged).
 *   "host_register"       host ABI: 1 = page-lock the caller's X arrays in place for the duration of the call (hipHostRegister) and let
 *                         the copy engine read them directly, instead of staging them through two pinned buffers with host threads
 *   "devices"             MULTI-GPU BEHIND THE DROP-IN ENTRY POINTS: the handle serves c_xlinear_predict_{csr,drm}_f32 from this many
 *                         devices -- its own plus value-1 replicas of the compiled model on the following devices (wrapping around when
 *                         the box has fewer).  A predict then cuts X into nnz-balanced row shards, one host thread + stream + pinned
 *                         staging per device, no inter-GPU traffic, results of all shards into the arrays of the ONE allocator call.
 *                         Only for models loaded from a folder.  c_xlinear_get_int_attr "nr_devices" reads it back.
 *   "prune"               1 (default): EXACT bound pruning -- a layer scores the children of the best beam parent(s) first (tile format: one
 *                         parent; dense row format and the dense-X SGEMM: as many as fill 64 candidates) and skips the other parents
 *                         for every query whose k-th best candidate already reaches the next parent's bound (every post-processor with
 *                         a combiner keeps a child's score <= max(its parent's, 0) resp. <= its parent's, and later candidates lose
 *                         ties by position), so the result is unchanged bit for bit; NaN parent scores and "noop" disable it per
 *                         query / layer; 0: every candidate of every beam parent is scored (what the reference evaluates)
 *   "overlap_min_rows"    split predicts of at least this many rows into two batches on two streams (0 = never)
 *   "reserve_rows"        sizes the host ABI's result buffers (pinned host + device, rows x the model's default top-k) for calls of up to this many rows NOW
 *                         instead of inside the first large call.  (Everything that does not depend on X -- copy threads, streams, the pinned upload
 *                         ring, kernel code objects, both scratch lanes for 65 536-row batches -- is created when a model is loaded from a folder;
 *                         the environment variable XRL_WARM=0 turns that off.)
 *   "host_batch_mb"       12 (default): CSR input of the pipelined host ABI is computed in batches that grow x1.6 from a third of this
 *                         many megabytes of (column id, value) pairs up to three times it (measured on Amazon-670K: 12 -> 10.4 ms per
 *                         call, 24 -> 12.1, 36 -> 13.0)
 *   "host_pipeline"       1 (default): c_xlinear_predict_* cut a large X into nnz-balanced row batches; batch b+1 is staged into
 *                         pinned memory and uploaded on a copy stream while batch b computes; 0: one synchronous upload
 *   "dense_layers"        1 (default): layers that carry the dense row format run the fused query-stationary kernel K1Q
 *                         whenever the beam's candidates fit its registers and, for sparse X, the format moves fewer lines than the
 *                         tile format's row lookup (parents of <= 32 padded columns, or >= 1 weight per (feature, parent) segment);
 *                         2: whenever they fit (tests); 0: tile-format kernels K0 -> K1 -> K2 everywhere
 *   "k1q_fuse"            3 (default): consecutive dense-format layers of <= 3 candidate registers per lane (<= 192 candidates per
 *                         query) run in ONE K1Q launch, the wavefront that owns a query carries its beam through them in LDS;
 *                         1 / 2: only layers of <= 1 / 2 registers share a launch; 0: one launch per layer
 *   "k1g_min_items"       dense X: a dense-format layer runs the tiled, k-ordered SGEMM K1G (tile-sorted items, weight and query
 *                         panels staged in LDS) once a parent serves this many queries on average (default 16; 0 = never: K1Q)
 *   "k1g_variant"         1: the alternative register-tile / panel shapes of K1G (A/B, tests; results identical)
 *   "k1_wpb", "k1_lds_pad", "k1_ablate"   debug: wavefronts per K1 workgroup, extra LDS per wavefront, phase ablation
 * Environment read at model load: XRL_LOOKUP=bitmap|bitmap64|bucket (force the row
 * lookup structure; default per layer: bucket table if rank-bitmaps would take more than a quarter of the free HBM, else
 * 64-feature words carrying the first row's extent on sparse tiles, else 32-feature words),
 * XRL_ROW_ALIGN=0 (keep tile rows packed instead of line-aligned), XRL_MAX_TILE_ENTRIES (lower the tile splitter's
 * limit; tests), XRL_DENSE=0 (never build the dense row format), XRL_DENSE_MAX_MB (cap of one layer's dense matrix;
 * default 65536, and never more than a quarter of the free HBM). */
int xrl_set_option(void* model, const char* key, int64_t value);

/* Debug: with option k1_ablate bit 6 set, K1 accumulates per-phase shader cycles
 * [prologue, fill, D1, D3, epilogue, #waves, -, -]; this reads (and optionally resets) them. */
void xrl_debug_k1_phases(unsigned long long* out8, int reset);

/* Single-layer API (c_xlinear_single_layer_predict*): compiled one-layer handles are cached by the identity of the caller's
 * W / C value arrays (pointers, shapes, nnz, bias) plus a 64-bit hash of EVERY byte of all their arrays (an in-place edit is seen on
 * the next call), at most 8 entries, least recently used evicted.  stats: cumulative hits / misses and live entries (tests). */
void xrl_single_layer_cache_clear(void);
void xrl_single_layer_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* entries);

/* Device layout of one layer: out[0..12) = {row lookup of the tile format (0 = 32-feature rank-bitmap, 1 = bucket table,
 * 2 = 64-feature words with the first row's extent), bucket search levels, carries the dense row format (0/1), padded dense
 * tile width, padded columns per dense row, tiles, weights (entries), bytes of the dense matrix, W.rows, children kept,
 * widest tile, bytes of HBM}.  Returns the number of values available. */
uint32_t xrl_layer_info(void* model, uint32_t layer, uint64_t* out, uint32_t cap);

/* Bytes of HBM held by the compiled model. */
uint64_t xrl_model_device_bytes(void* model);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* XRL_ABI_H */
