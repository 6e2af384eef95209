// Device-resident model layout + host-side model compiler interface.
//
// The reference keeps, per layer, a "chunked" W: one chunk per parent cluster holding the
// weights of that parent's children, row-major inside the chunk
// (pecos/core/xmc/inference.hpp:292-329 bin_search_chunk_view_t, :244-254 chunk_entry_t,
// built by make_chunked_from_csc :557-650).  The MI355X layout keeps the same unit of work
// (one (query, parent) pair = one chunk product, arithmetic order untouched) but is laid out for
// HBM + 64-wide wavefronts instead of a CPU cache:
//
//   * a chunk wider than kMaxTileCols children is split into column TILES (each output column's
//     accumulation order is unaffected by the split);
//   * the row lookup "is feature f present in this tile, and where" is a rank-bitmap:
//     one 8-byte {bits, rank} word per 32 features, i.e. ONE load per probe instead of the
//     reference's ~log2(R) binary-search steps (:786-803).  It costs rows/4 bytes per tile,
//     which is what 288 GB of HBM3E is for; sparse tiles use 64-feature words that also return the first
//     row's extent, and layers whose bitmaps would not fit fall back to a bucket table + binary search;
//   * rows are {start, length} packed in 32 bits, tile-relative, laid out so that no row touches more 128-byte
//     lines than its length requires; entries stay {u32 col_offset, f32 val} (8 B).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "xrl_common.h"
#include "xrl_io.h"

namespace xrl {

constexpr uint32_t kMaxTileCols = 128;       // accumulators per (query, tile) item held in LDS
constexpr uint32_t kNoBias = 0xFFFFFFFFu;
constexpr uint32_t kMissing = 0x80000000u;   // dense row format: "W has no entry here" = -0.0.  For finite x the product x * (-0.0) is a zero, which leaves every
                                             // reachable accumulator unchanged (accumulators are never -0.0: they start at +0.0 or +0.0 + bias*w, and a + (-a) rounds
                                             // to +0.0), so the kernels' fast loops multiply and add it like any weight; only a non-finite x needs the
                                             // "is there an entry" test.  An explicit -0.0 stored in W is written as +0.0 (densify_kernel): same products.

enum PPKind : int { PP_NOOP = 0, PP_SIGMOID = 1, PP_LOG_SIGMOID = 2, PP_LP_HINGE = 3, PP_LOG_LP_HINGE = 4 };
struct PostProc { int kind = PP_NOOP; int p = 0; };
PostProc parse_post_processor(const char* name);  // inference.hpp:192-240

struct TileDesc {            // 32 bytes, device
    uint32_t col_begin;      // first child column (rearranged space) covered by the tile
    uint32_t ncols;
    uint32_t nrows;          // distinct non-zero rows R_t
    uint32_t bias_slot;      // row slot holding W's bias row (== rows-1), or kNoBias
    uint64_t rowptr_base;    // index of this tile's first row in row_ext[] and row_idx[]
    uint64_t ent_base;       // index of this tile's first entry
};

struct Entry { uint32_t col; float val; };          // 8 bytes, == chunk_entry_t
struct BmWord { uint32_t bits; uint32_t rank; };    // 8 bytes per 32 features
struct alignas(16) BmWord64 { uint32_t lo, hi, rank, ext0; };   // 16 bytes per 64 features: bits, rank, packed extent of the first set row
// one tile row in 4 bytes: tile-relative entry offset (25 bits, < max_tile_entries) | (length - 1) << 25 (rows hold 1..128 entries)
inline uint32_t pack_row_extent(uint32_t start, uint32_t len) { return start | ((len - 1u) << 25); }

// Plain-pointer view handed to kernels (all device pointers).
struct LayerDev {
    const TileDesc* tiles;
    const uint32_t* ptile;       // [n_parents+1] tiles of parent p = [ptile[p], ptile[p+1])
    const uint32_t* chunk_col;   // [n_parents+1] child-column range of parent p (rearranged space)
    const BmWord* bitmap;        // [n_tiles * nwords], or nullptr when the layer uses the bucket lookup
    const BmWord64* bitmap64;    // [n_tiles * nwords64] sparse tiles (few rows per word): 64-feature words that carry the first row's extent, else nullptr
    uint32_t nwords64;
    const uint32_t* bucket;      // [n_tiles * (bk_n+1)] first row slot of every feature-id bucket (bitmap too large for HBM), else nullptr
    uint32_t bk_shift, bk_n, bk_levels;   // bucket = feature >> bk_shift; binary-search steps that cover the longest bucket
    const uint32_t* row_ext;     // [sum(nrows)] packed {entry offset, length} of every tile row (pack_row_extent)
    const uint32_t* row_idx;     // [sum(nrows)] feature id of every tile row (dense-query path)
    const Entry* entries;        // [nnz]
    const uint32_t* perm_inv;    // [n_children] rearranged -> original child id, or nullptr
    const float* chunk_alg_bytes;  // [n_parents] algorithmic bytes of the reference chunk (stats)
    const float* bias_prod;      // [n_children] fl32(bias * W[bias_row, child]) or +0.0 (no explicit entry / no bias)
    uint32_t n_parents, n_children, n_tiles, nwords, w_rows;
    uint32_t max_tiles_per_parent, max_tile_cols;
    float bias;
    int has_bias;
    // DENSE row format (K1Q, xrl_k1q.hip), nullptr when the layer is held in the tile format only:
    //   wd[feature * d_ld + (dense tile << d_gp_log2) + column] = weight bits, kMissing where W has no entry; w_rows + 1 rows,
    //   the last one all kMissing (where K1Q sends features the layer does not know).
    // Dense tiles partition every parent's children into runs of <= 2^d_gp_log2 columns (their own tiling: <= 64 wide).
    const uint32_t* wd;
    uint64_t d_ld;               // padded columns per feature row (a multiple of 32: rows start on 128-byte lines)
    uint32_t d_gp_log2;          // log2 of the padded dense tile width
    uint32_t d_max_tiles;        // max dense tiles per parent
    const uint32_t* d_ptile;     // [n_parents+1] dense tiles of parent p
    const uint32_t* d_tcol;      // [n_dtiles+1] first child column of every dense tile (children are contiguous)
    int d_sparse_ok;             // sparse X may use the dense format (K1Q): parents of <= 32 padded columns (half a line / one line per
                                 // (feature, parent)), or wider ones whose (feature, parent) segments hold >= 1 weight on average; otherwise the
                                 // tile format's row lookup moves fewer lines (Wiki10-31K's leaf: 0.18 weights per 64-column segment)
    const uint32_t* pres;        // PRESENCE words of the dense row format, or nullptr: pres[f * pres_words + (dt >> 5)] bit (dt & 31) = "dense tile dt holds
    uint32_t pres_words;         //   at least one weight at feature row f" ((w_rows + 1) rows like wd).  K1Q (sparse X) requests a (feature, tile) segment
                                 //   only when its bit is set: on deep trees a third to a half of the segments a query addresses are empty, and the
                                 //   kernel runs on the number of fabric requests (profiles/r04_hard_config.md)
    int d_full;                  // every (feature, kept child) cell of the dense matrix holds a weight (no kMissing): K1G's 2-op inner loop
    const uint32_t* tile_parent; // [n_tiles] parent of every tile-format tile (K1G walks tile-sorted items)
    // ROOT layer only (finalize_model): levels 0 and 1 in ONE dense matrix of 64 columns per feature row -- columns [0, wd01_c1) are level 1's dense row
    // (same offsets as in its own matrix), columns [wd01_c1, wd01_c1 + K0) level 0's -- so that K1Q's fused walk of the two levels issues one
    // load per feature; nullptr when the two levels do not fit 64 columns
    const uint32_t* wd01; uint32_t wd01_c1;
    // TILE ROWS held densely (K1T, xrl_k1t.hip), nullptr when not built: tile t's rows (every feature with a weight in the tile, ascending: the
    // slots the rank-bitmaps return) are wt_stride floats each, column c of the tile at position c, kMissing where W has no entry, followed by
    // one all-missing pad row; they start at float index wt_base[t].  wt_stride = G * NR of k1t_shape(max_tile_cols).
    const float* wt; const uint64_t* wt_base; uint32_t wt_stride; uint64_t wt_bytes;
    int d_regular;               // every parent owns exactly ONE dense tile and every tile is full (2^d_gp_log2 children): dense tile = parent, first child = parent << d_gp_log2
                                 // -- K1Q's prolongation then needs neither d_ptile nor d_tcol (two dependent loads per candidate register); balanced trees are like this
};

struct Layer {
    // host metadata
    uint32_t w_rows = 0, w_cols = 0, c_rows = 0, c_cols = 0;
    float bias = 0.f;
    float w_absmax = 0.f;                  // max |weight| of the layer times max(1, |bias|); +inf when a weight (or the bias) is not finite -- what the
                                           // pruning guard prices a query's largest possible partial sum with (xrl_predict.cpp: prune_wmax)
    uint32_t only_topk = 0;
    PostProc pp;
    std::string pp_name;
    bool reordered = false;
    uint32_t n_children = 0;               // nnz(C)
    uint32_t n_tiles = 0, nwords = 0, max_tiles_per_parent = 0, max_tile_cols = 0, max_chunk_cols = 0;
    uint64_t nnz = 0, total_rows = 0;
    std::vector<uint32_t> chunk_sizes_desc;  // chunk sizes sorted descending (cand stride bound)
    // predict_on_selected_outputs (inference.hpp:2507-2571): host copy of C's pattern, child -> parent,
    // and W in CSC form on the device (uploaded lazily: from w_path, or from w_host for in-memory models)
    std::vector<uint64_t> h_c_ptr; std::vector<uint32_t> h_c_idx, h_parent;
    std::string w_path; std::shared_ptr<HostCsc> w_host;
    DevBuf d_csc_ptr, d_csc_idx, d_csc_val; bool csc_ready = false;
    // device storage
    DevBuf d_tiles, d_ptile, d_chunk_col, d_bitmap, d_row_ptr, d_row_idx, d_entries, d_perm_inv, d_chunk_alg, d_bias_prod;
    DevBuf d_bucket, d_bitmap64;
    DevBuf d_wd, d_dptile, d_dtcol, d_tile_parent, d_pres;   // dense row format (see LayerDev::wd)
    DevBuf d_wt, d_wt_base;                                   // tile rows held densely (see LayerDev::wt)
    uint64_t dense_bytes = 0;
    uint32_t bk_shift = 0, bk_n = 0, bk_levels = 0;
    LayerDev dev{};
    uint64_t device_bytes = 0;
    // sum of the `beam` largest chunks: upper bound on candidates per query entering this layer
    uint64_t cand_bound(uint32_t beam) const;
};

struct ProfileSlot { std::string name; uint32_t layer; uint32_t launches = 0; double ms = 0; };
struct PendingEvent { hipEvent_t a, b; size_t slot; };

// predict scratch, grow-only, owned by the model handle
struct LaneWs {      // scratch of one row batch in flight
    DevBuf beam_idx[2], beam_val[2], beam_cnt[2];
    DevBuf cand_off, ncand, cand;
    DevBuf items, items_sorted, sort_hist, sort_start;   // item descriptors (K0) and their tile-sorted copy
    DevBuf blk_start, x_ok;                              // K1G: first workgroup of every tile; per-row "all x finite" flags
    DevBuf qperm, qsort_hist, qsort_start;               // K1Q's sorted launch: launch slot -> query, and the counting sort's scratch
    DevBuf prune_done, prune_cnt;                        // bound-pruned layers: per-query "first phase was final" flags; item count of the second phase
};
struct Workspace {
    LaneWs lane[2];      // two row batches are in flight on two streams (xrl_predict.cpp)
    DevBuf stats;
    // host-ABI predict: uploaded X + result staging
    DevBuf x_ptr, x_idx, x_val;
    DevBuf out_idx, out_val, out_cnt;
    PinnedBuf h_idx, h_val, h_cnt;
    PinnedBuf stage[3];   // pinned staging ring of the pipelined host-ABI upload (kStageSlots, xrl_abi.cpp)
    PinnedBuf stage_ptr;  // ... and the row pointer's own pinned staging buffer (it travels first, on the copy stream)
    // initial beam for the single-layer API
    DevBuf init_idx, init_val, init_cnt;
};

struct Model {
    int device = 0;
    int weight_matrix_type = 2;
    std::vector<std::unique_ptr<Layer>> layers;
    uint32_t nr_features = 0, nr_labels = 0, nr_codes = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;       // second lane of the batch pipeline
    hipEvent_t ws_done = nullptr;           // end of the most recent predict that used the workspace, recorded on ws_stream:
    hipStream_t ws_stream = nullptr;        //   a predict arriving on ANOTHER stream waits for it before touching the scratch buffers
    hipStream_t copy_stream = nullptr;      // H2D of the pipelined host-ABI path
    hipStream_t d2h_stream = nullptr;       // D2H of its results: a batch's rows travel back under the next batch's kernels
    std::vector<hipEvent_t> d2h_events;     // "batch b's kernels are queued" (grow-only, reused)
    int host_register = 0;                  // host ABI: 1 = hipHostRegister the caller's arrays for the call and DMA from them directly (no staging copy)
    int host_batch_mb = 12;                 // host ABI, CSR input: megabytes of (column id, value) pairs per compute batch
    int host_pipeline = 1;                  // host ABI: cut large X into row batches whose upload overlaps the previous batch's kernels
    std::vector<hipEvent_t> events;         // cross-stream ordering (timing disabled), reused across predicts
    // host ABI, two compute lanes (xrl_abi.cpp host_compute): lane 0 uses ws_done / ws_stream above; lane 1 keeps its own "scratch in use
    // until" event between calls, and one join event orders the lanes.  Owned by the handle: ~Model destroys them on the handle's device.
    struct HostLanes { hipEvent_t done[2] = {nullptr, nullptr}; hipStream_t strm[2] = {nullptr, nullptr}; hipEvent_t join = nullptr; } host_lanes;
    std::mutex mu;                          // one predict at a time per handle
    std::unique_ptr<Workspace> ws;
    // options
    int k1_group = 0;                       // 0 = auto
    int k1_wpb = 1, k1_lds_pad = 0, k1_ablate = 0;   // K1 tuning / debug knobs (xrl_set_option), per handle
    int k1g_variant = 0;                            // 1: K1G's alternative register-tile / panel shapes (A/B, tests)
    int64_t max_batch_rows = 0;             // 0 = auto
    int overlap_min_rows = 0;               // split a predict of at least this many rows into two half batches on two streams so that one half's
                                            // K0/K2 run under the other half's K1; 0 = never (measured on Amazon-670K: 25.9 vs 25.5 ms, no gain)
    int prune = 1;                          // exact bound pruning (xrl_predict.cpp): 1 = a layer first scores the children of the best beam parent(s) only and
                                            // skips the rest for every query whose k-th best already reaches the next parent's score; 0 = score every candidate
    int k2_big_min_k = 0;                   // > 0: top-k sizes from this value on take the segmented-sort K2 (xrl_topk_big.hip) that otherwise serves k > 20 480 (tests)
    int tile_rows = 1;                      // tile-format layers that carry densely held tile rows (LayerDev::wt), sparse X: 1 = launches on items in query order run K1T (xrl_k1t.hip), 2 = every launch, 0 = always the entry-list kernel K1
    int dense_layers = 1;                   // 1 = layers that carry the dense row format run the fused query-stationary kernel K1Q (0: K0 -> K1 -> K2 everywhere)
    bool csc_route = false;                 // weight_matrix_type == CSC: every layer runs the reference's CSC arithmetic (K0 -> K1C -> K2)
    int k1q_fuse = 3;                       // consecutive dense-format layers of <= this many candidate registers (1..3) share one K1Q launch (the beam stays in LDS); 0: one launch per layer
    int k1g_first = 0;                      // K1G layers under bound pruning: beam parents scored in the first stage (0 = about one candidate register, 64 / children per parent)
    int k1g_min_items = 16;                 // dense X: run a dense-format layer as the tiled SGEMM K1G once a parent serves this many queries on average (0 = never)
    // ---- pruning feedback (xrl_predict.cpp): what the bound pruning of the PREVIOUS predicts of this handle achieved, per layer, so that a
    // model on which the first stage settles almost nothing (scores that do not saturate, routing spread over the tree) stops paying for
    // the staging -- the layer then scores every candidate in one pass (tile format: on tile-sorted items).  Results never depend on it.
    static constexpr int kFbLayers = 16;
    uint32_t* fb_host = nullptr;            // pinned, device-visible: [0, 2*kFbLayers) K1Q's sampled counters {queries seen, queries that needed the second pass}
                                            //   per layer (copied from fb_dev by the first wavefront of the next K1Q launch), [2*kFbLayers, 3*kFbLayers) the
                                            //   second stage's item count of tile-format layers (written by its K1 launch)
    DevBuf fb_dev;                          // K1Q's counters (device atomics)
    DevBuf d_wd01;                          // levels 0 + 1 merged dense rows (LayerDev::wd01 of the root layer)
    uint32_t fb_seen[kFbLayers] = {0}, fb_second[kFbLayers] = {0};   // K1Q counters at the last decision
    uint64_t fb_tile_slots[kFbLayers] = {0};                          // second-stage slots the item count of a tile-format layer refers to
    uint32_t fb_unstaged_calls[kFbLayers] = {0};                      // predicts in a row a layer has run unstaged (re-probed every kFbReprobe)
    uint8_t fb_unstaged[kFbLayers] = {0};
    uint8_t fb_probing[kFbLayers] = {0};                              // an unstaged layer was staged ONCE (the probe) and its outcome has not arrived yet: it keeps running unstaged meanwhile
    int adaptive = 1;                       // 0: always stage (xrl_set_option "adaptive")
    int presence = 1;                       // K1Q, sparse X: 1 = layers that run UNSTAGED (prune off, or switched by the pruning feedback) request a (feature, parent) weight
                                            // segment only when the layer's presence word says it holds a weight; 2 = every layer that has presence words; 0 = never
    int prune_mid = 1;                      // bound-pruned tile-format layers with >= 16 beam parents: a middle stage (slots 1..4) between the first parent and "everything else"
    int sort_rest = 1;                      // bound-pruned tile-format layers: the second phase's compacted items are tile-sorted before K1 runs on them (0: query order)
    int sort_rest_min = 32768;              // ... only when the previous predicts' later stages held at least this many items (pruning feedback's count; 0 = always)
    int qsort = 1;                          // K1Q, sparse X: the last layer of a run of dense-format layers runs on queries SORTED by the best parent of their beam,
                                            // every XCD on a contiguous range of them (xrl_predict.cpp), when it has >= qsort_min_parents parents and the batch >= qsort_min_rows rows
    int qsort_min_parents = 64, qsort_min_rows = 131072;
    int sort_min_tiles = 0;                 // tile-sort a layer's items once it has this many tiles (0 = never; measured: cuts HBM fetch 15x at the leaf but K1 is issue-bound, not HBM-bound, so it does not pay yet)
    // multi-GPU behind the drop-in entry points (xrl_set_option "devices"): further copies of the compiled model on other devices; the
    // host-ABI predict shards the rows over this handle's device and the replicas' (xrl_abi.cpp predict_host)
    std::string src_path; int src_kind = -1;   // where the model came from: 0 = npz folder, 1 = mmap folder, -1 = arrays (no replicas)
    std::vector<std::unique_ptr<Model>> replicas;
    bool profiling = false;
    std::vector<ProfileSlot> profile;
    std::vector<PendingEvent> pending;     // recorded, not yet resolved (no sync on the timed path)
    Model();
    ~Model();
    uint64_t device_bytes() const;
};

// host-only pieces of the model compiler (also exported for tests: xrl_debug_split_chunk / xrl_debug_layout_rows)
uint32_t split_chunk(const uint64_t* cum, uint32_t n, uint64_t limit);
uint64_t layout_tile_rows(const uint32_t* rptr, uint32_t nrows, bool align, uint32_t* ext);

// Build one layer from host CSC W / C (LayerData<chunked>::init, inference.hpp:1849-1883).
// perm_inv_override / orig_rows: W and C are already in the rearranged (contiguous) child order and the
// given map takes rearranged -> original ids (mmap model folders, LayerData::init_mmap :1885-1908).
// structure_only: keep the tree bookkeeping (chunks, child order, perm) but no weights in the tile / dense formats --
// layers that run the CSC route (K1C, xrl_pairs.hip) read W in CSC form instead.
std::unique_ptr<Layer> compile_layer(const HostCsc& W, const HostCsc& C, float bias, uint32_t only_topk,
                                     const std::string& post_processor,
                                     const std::vector<uint32_t>* perm_inv_override = nullptr, uint32_t orig_rows = 0,
                                     bool structure_only = false);
// Load <path>/param.json + {d}.model/ (HierarchicalMLModel::load, inference.hpp:2616-2655).
std::unique_ptr<Model> load_model_from_disk(const std::string& path, int weight_matrix_type);
void finalize_model(Model& m);
std::unique_ptr<Model> load_mmap_model_from_disk(const std::string& path);        // xrl_mmap.cpp
void compile_mmap_model(const std::string& npz_path, const std::string& mmap_path);   // xrl_mmap.cpp
void ensure_device_csc(Layer& L);
// xrl_k1q.hip: memset wd to kMissing and scatter the CSC columns src_col[c] to padded column dst_off[c]
// xrl_k1q.hip: presence words of a dense-format layer (LayerDev::pres) from its matrix
void k1t_shape(uint32_t max_tile_cols, int& g, int& nr);   // lanes per item / columns per lane of K1T for a layer's widest tile
void launch_tile_rows(const LayerDev& L, uint64_t total_floats, uint32_t* wt, hipStream_t s);   // fills LayerDev::wt from the tile format on the device (xrl_k1t.hip)
void launch_merge01(const uint32_t* wd0, uint64_t ld0, uint32_t k0, const uint32_t* wd1, uint64_t ld1, uint32_t c1, uint32_t rows, uint32_t* out, hipStream_t s);
void launch_presence(const uint32_t* wd, uint64_t ld, uint32_t rows, uint32_t gp_log2, uint32_t n_tiles, uint32_t pres_words, uint32_t* pres, hipStream_t s);
void launch_densify(const uint64_t* col_ptr, const uint32_t* row_idx, const float* val, const uint32_t* src_col,
                    const uint32_t* dst_off, uint32_t n_children, uint32_t w_rows, uint64_t ld, uint32_t* wd, hipStream_t s);   // upload W as CSC (original column ids) if not there yet

}  // namespace xrl
