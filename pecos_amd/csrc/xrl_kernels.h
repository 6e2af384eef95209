// Launch interface of the HIP kernels (xrl_kernels.hip).  All pointers are device pointers.
#pragma once
#include "xrl_model.h"

namespace xrl {

// Device-resident query matrix (CSR with 32-bit offsets relative to the matrix, or dense row-major).
struct QueriesDev {
    const uint64_t* row_ptr;   // [rows+1] (CSR) or nullptr
    const uint32_t* col_idx;
    const float* val;          // CSR values, or the dense matrix
    uint32_t rows, cols;
    int dense;
    uint64_t nnz;              // CSR only
};

// Beam of the previous layer, fixed stride per query; idx == nullptr means the implicit root
// (one parent, id 0, score 1: HierarchicalMLModel::predict, inference.hpp:2462-2463).
struct BeamDev {
    uint32_t* idx;
    float* val;
    uint32_t* cnt;
    uint32_t stride;
};

struct K1Tune { int wpb = 1, lds_pad = 0, ablate = 0, k1g_variant = 0, pres_mode = 1, tile_rows = 1, k2_big_min_k = 0; };   // per-model tuning / debug knobs (xrl_set_option k1_wpb, k1_lds_pad, k1_ablate, k1g_variant)

struct LayerPlan {
    uint32_t row0, nrows;       // query rows [row0, row0+nrows) of the query matrix
    uint32_t beam_in;           // max #parents per query entering the layer
    uint32_t k;                 // #survivors kept by this layer
    uint32_t cand_stride;       // floats reserved per query in `cand`
    PostProc pp;
    int first_layer;            // no combine (no_prev_pred)
    int implicit_root;          // previous beam is the implicit all-ones root
    int prune;                  // exact bound pruning allowed (Model::prune)
    int bias_first;             // sparse X under weight_matrix_type HASH_CHUNKED: the bias row is applied BEFORE the query's features
                                // (chunk_ops<csr, hash>, inference.hpp:705-735); dense X is bias-first in every layout
    int layer;                  // index in the chain (profiling, feedback slot)
    K1Tune tune;
    uint32_t* fb_host = nullptr;  // pruning feedback (Model::fb_host), or nullptr
    uint32_t* fb_dev = nullptr;   // K1Q's sampled counters (Model::fb_dev)
};

// K0  prolongate: per query, offsets of every beam parent's child block + candidate count, and one
//     16-byte item descriptor per (query, beam slot, tile-in-parent) for K1.
void launch_k0_prolongate(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev, uint32_t* cand_off,
                          uint32_t* ncand, void* items, hipStream_t s, uint32_t item_ranks = 0xFFFFFFFFu /* beam slots that get item descriptors */);
// bound-pruned layers, second phase: items of the beam slots >= first_rank of the queries with done[q] == 0, compact; *n_items = their number
void launch_k0b_remaining(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev, const uint32_t* cand_off, const uint32_t* done,
                          uint32_t first_rank, void* items, uint32_t* n_items, hipStream_t s, uint32_t end_rank = 0xFFFFFFFFu);   // beam slots [first_rank, end_rank)
bool k2_wave_path(const LayerPlan& P);   // the register top-k kernel serves this layer (what bound pruning needs)
size_t k0_item_bytes();
// K1  (query, tile) inner products + bias + post-processor + combine, one item per G lanes.
void launch_k1(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items, const uint32_t* n_items,
               float* cand, int group, hipStream_t s);
// counting sort of the item descriptors by tile (LDS histograms, no global atomics); start[n_tiles] = #items
void launch_sort_items(const LayerDev& L, uint64_t n_slots, const void* items, void* sorted, uint32_t* H,
                       uint32_t* start, hipStream_t s,
                       const uint32_t* n_dev = nullptr);   // n_dev: device count of a compacted list (<= n_slots)
// counting sort of the QUERIES of a row batch by the best parent of their beam (slot 0), as a permutation: perm[slot] = query (K1Q's sorted launch)
void launch_sort_queries(BeamDev prev, uint32_t nrows, uint32_t n_keys, uint32_t* H, uint32_t* start, uint32_t* perm, hipStream_t s);
uint32_t qsort_max_keys();
size_t qsort_hist_bytes(uint32_t nrows, uint32_t n_keys);
void launch_step_marker(hipStream_t s);   // XRL_STEP_MARKER=1: an empty kernel at the start of every predict (step boundaries in kernel traces)
uint32_t sort_max_tiles();
size_t sort_hist_bytes(uint64_t n_slots, uint32_t n_tiles);
// K2  per-query top-k with (value desc, position asc) order; maps positions to original child ids.
void launch_k2_topk(const LayerDev& L, const LayerPlan& P, BeamDev prev, const uint32_t* cand_off,
                    const uint32_t* ncand, const float* cand, uint32_t* out_idx, float* out_val,
                    uint32_t* out_cnt, uint32_t out_stride, hipStream_t s,
                    uint32_t rank_limit = 0 /* > 0: only the candidates of the first rank_limit beam slots */, uint32_t limited_cands = 0 /* their maximum number */,
                    uint32_t* done = nullptr /* out: that selection is final (exact bound, see K2Args) */, const uint32_t* skip_done = nullptr /* queries to skip */,
                    const uint32_t* xok = nullptr /* with done: the per-query pruning guard (launch_xguard / K1Q's out_xok) */);
// stats: sum over (query, parent) of the reference chunk's algorithmic bytes, and of candidates
constexpr int kStatsPerLayer = 8;   // [0] reference-chunk bytes, [1] candidates, [2] items, [3] probes, [4] matched rows, [5] their entries,
                                    // [6] tile columns over the items, [7] query features x tile columns over the items
void launch_stats(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev, const uint32_t* ncand, const void* items,
                  double* out8, hipStream_t s, uint64_t item_slots = 0 /* slots of `items` (0: rows x beam x tiles per parent) */);
// K3  sparse_inner_products (pecos/core/utils/matrix.hpp:1049-1060), 4 layout combos
void launch_k3_inner_products(const uint64_t* x_ptr, const uint32_t* x_idx, const float* x_val, int x_dense,
                              const uint64_t* w_ptr, const uint32_t* w_idx, const float* w_val, int w_dense,
                              uint32_t dim, uint64_t len, const uint32_t* rows, const uint32_t* cols,
                              float* out, hipStream_t s);

unsigned long long* k1_phase_buffer();
void k1_phase_read(unsigned long long out[8], bool reset);   // debug: per-phase cycle totals of K1
// K4  predict_on_selected_outputs: one layer of (query, node) pairs against CSC W
void launch_k4_selected(const uint64_t* col_ptr, const uint32_t* row_idx, const float* val, uint32_t w_rows, float bias,
                        const QueriesDev& X, const uint32_t* pair_q, const uint32_t* node, const uint32_t* ppos,
                        const uint64_t* prev_off, const float* prev_val, float* out_val, uint64_t n_pairs,
                        const PostProc& pp, int first_layer, hipStream_t s);
// K1G (xrl_k1g.hip): dense queries against a dense-format layer as a tiled, k-ordered SGEMM over tile-sorted items
uint32_t k1g_cols(const LayerDev& L);                 // 0: the layer cannot be served by K1G
void launch_k1g(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items_sorted, const uint32_t* start,
                uint32_t* blk_start, const uint32_t* x_ok, float* cand, hipStream_t s);
void launch_xguard(const QueriesDev& X, uint32_t row0, uint32_t nrows, float wmax, uint32_t* ok, hipStream_t s);   // per query row (CSR or dense): finite and too small to overflow any accumulator (prune_guard_ok)
// K1C (xrl_pairs.hip): the CSC route of a layer (w_ops<csc_t>, inference.hpp:1081-1149) over the candidates K0 laid out
void launch_k1c_csc(const LayerDev& L, const uint64_t* col_ptr, const uint32_t* row_idx, const float* val, const LayerPlan& P,
                    const QueriesDev& X, BeamDev prev, const uint32_t* cand_off, const uint32_t* ncand, float* cand, hipStream_t s);
// [X_feat | X_emb] -> one CSR on the device (concat_model's query form, matcher.py:864-890)
void launch_concat_csr(const uint64_t* in_ptr, const uint32_t* in_idx, const float* in_val, const float* emb, uint32_t rows,
                       uint32_t sparse_cols, uint32_t dense_cols, int normalize_emb, uint64_t* out_ptr, uint32_t* out_idx, float* out_val, hipStream_t s);
// xrl_features.hip: the weighting half of the reference's TF-IDF vectorizer (tfidf.hpp:798-822) on a device CSR of term counts
void launch_tfidf_weight(const uint64_t* row_ptr, const uint32_t* col_idx, const float* count, const float* idf, uint32_t rows, uint32_t cols,
                         int binary, int sublinear_tf, int norm_p, float* out, hipStream_t s,
                         uint32_t seg_stride = 1, uint32_t seg_off = 0, uint32_t* err = nullptr);   // rows = segments of row_ptr; *err = 1 on a column id >= cols
int k1_auto_group(const LayerDev& L, const Layer& host, int dense);
// K1T (xrl_k1t.hip): K1 on the densely held tile rows (LayerDev::wt), accumulators in registers; launch_k1 routes to it when k1t_serves
bool k1t_serves(const LayerDev& L, const QueriesDev& X);
void launch_k1t(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items, const uint32_t* n_items, float* cand, hipStream_t s);
// K1Q (xrl_k1q.hip): a whole layer -- prolongate, chunk products against the DENSE row format, post-processor,
// combine, top-k, child re-ordering -- in one query-stationary kernel: previous beam in, next beam out.
uint32_t k1q_regs(const LayerDev& L, uint32_t beam_in, uint32_t k, bool dense_x);   // 0: the layer / beam / k cannot (or should not) be served by K1Q
// n consecutive dense-format layers in ONE launch (the beam stays in LDS between them); n <= 8
void launch_k1q(const LayerDev* const* Ls, const LayerPlan* Ps, int n, const QueriesDev& X, BeamDev prev, uint32_t* out_idx, float* out_val,
                uint32_t* out_cnt, uint32_t out_stride, hipStream_t s, float prune_wmax, uint32_t* out_xok = nullptr,
                const uint32_t* qperm = nullptr /* launch slot -> query (launch_sort_queries); every XCD then takes a contiguous range of slots */);
                // prune_wmax / out_xok: the bound-pruning guard (prune_guard_ok, xrl_device.h); out_xok[q] receives every query's flag
size_t k2_max_k();
// xrl_topk_big.hip: top-k sizes beyond k2_max_k() -- one segmented radix sort over the batch's candidate rows (no cap, like the reference's sorted_csr)
void launch_k2_topk_big(const LayerDev& L, const LayerPlan& P, BeamDev prev, const uint32_t* cand_off, const uint32_t* ncand, const float* cand,
                        uint32_t* out_idx, float* out_val, uint32_t* out_cnt, uint32_t out_stride, hipStream_t s);

}  // namespace xrl
