// Hand-written HIP kernels for gfx950 (CDNA4, 64-wide wavefronts) -- the XR-Linear beam-search
// hot path of pecos/core/xmc/inference.hpp, restated for the MI355X execution model.
//
//   K0 k0_prolongate   prolongate_predictions               inference.hpp:1155-1219
//   K1 k1_kernel       compute_sparse_predictions + chunk_ops + transform + combine
//                                                           inference.hpp:925-1007, 769-839, 506-518,
//                                                           1360-1384, PostProcessor :192-240
//   K2 k2_topk_*       sorted_csr + reorder_prediction      inference.hpp:1223-1298, 1919-1923
//   K3 k3_kernel       sparse_inner_products                matrix.hpp:1049-1060, 836-877
//   K4 k4_selected     predict_on_selected_outputs (CSC)    inference.hpp:1018-1078, 1302-1358
//
// Arithmetic contract (verified bit-for-bit against the compiled reference, see oracle/):
// every output column accumulates fl32(acc + fl32(x_f * w)) over matched features in ASCENDING
// feature id, bias last (sparse X) / first (dense X); no FMA anywhere (built with
// -ffp-contract=off and explicit __fmul_rn/__fadd_rn); transforms in fp64 then rounded to fp32.
//
// Work decomposition: one ITEM = (query, beam parent, column tile).  A wavefront carries 64/G
// items, G lanes each ("wavefront-segmented"): the G lanes probe G consecutive query features
// per step against the tile's rank-bitmap (one 8-byte load per probe, coalesced x reads), hits are
// compacted IN ORDER into a small per-item LDS FIFO with a segmented ballot/popcount, and the
// FIFO is drained row by row with the G lanes striding over the row's entries (distinct output
// columns -> no conflicts), accumulators living in LDS.  Rows are drained in feature order, so
// the per-column summation order is exactly the reference's.
//
// What bounds K1 on MI355X (profiles/): not HBM bytes and, since the drain was rewritten, not
// VALU issue (50-65 % busy) but the stream of 128-byte lines its 8-byte gathers request from the L2 -- hence one
// 4-byte packed extent per row, rows placed so that none straddles an extra line, bitmap words that return the
// first row's extent with the probe, lanes past a unit's end re-reading its first entry.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "xrl_device.h"
#include "xrl_kernels.h"
#include "xrl_items.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// K0: one thread per query.  Besides the child-block offsets (prolongate) it writes one 32-byte
// ITEM DESCRIPTOR per (query, beam slot, tile-in-parent) so that K1 starts from a single coalesced
// load instead of a chain of dependent lookups (beam -> parent -> tile range -> offsets).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k0_prolongate(const uint32_t* __restrict__ chunk_col, const uint32_t* __restrict__ ptile,
              const TileDesc* __restrict__ tiles, uint32_t nrows, uint32_t beam_in, uint32_t item_ranks, uint32_t TT, uint32_t cand_stride,
              int implicit_root, const uint32_t* __restrict__ p_idx, const float* __restrict__ p_val,
              const uint32_t* __restrict__ p_cnt, uint32_t p_stride, uint32_t* __restrict__ cand_off,
              uint32_t* __restrict__ ncand, ItemDesc* __restrict__ items, const uint64_t* __restrict__ x_row_ptr) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= nrows) return;
    uint64_t xb = 0; uint32_t xl = 0;
    if (x_row_ptr) { xb = x_row_ptr[q]; xl = (uint32_t)(x_row_ptr[q + 1] - xb); }
    const uint32_t cnt = implicit_root ? 1u : min(p_cnt[q], beam_in);
    uint32_t off = 0;
    // item descriptors are written for the first `item_ranks` beam slots only (all of them unless the layer runs the bound-pruned
    // two-phase scheme, where the later slots' items are laid out by k0b_remaining for the queries that still need them)
    for (uint32_t j = 0; j < beam_in; ++j) {
        ItemDesc* it = items + ((size_t)q * item_ranks + j) * TT;
        uint32_t nt = 0;
        if (j < cnt) {
            const uint32_t parent = implicit_root ? 0u : p_idx[(size_t)q * p_stride + j];
            const float ps = implicit_root ? 1.0f : p_val[(size_t)q * p_stride + j];
            const uint32_t cb = chunk_col[parent], t0 = ptile[parent];
            nt = ptile[parent + 1] - t0;
            cand_off[(size_t)q * beam_in + j] = off;
            if (j < item_ranks)
                for (uint32_t tt = 0; tt < nt; ++tt)
                    it[tt] = make_item(q, t0 + tt, q * cand_stride + off + (tiles[t0 + tt].col_begin - cb), ps, xb, xl);
            off += chunk_col[parent + 1] - cb;
        }
        if (j < item_ranks) for (uint32_t tt = nt; tt < TT; ++tt) it[tt] = make_item(q, kNoTile, 0u, 0.f, 0, 0u);
    }
    ncand[q] = off;
}

void launch_k0_prolongate(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev, uint32_t* cand_off,
                          uint32_t* ncand, void* items, hipStream_t s, uint32_t item_ranks) {
    if (P.nrows == 0) return;
    if ((uint64_t)P.nrows * P.cand_stride > 0xFFFFFFFFull) fail("k0: candidate buffer exceeds 2^32 floats; lower max_batch_rows");
    hipLaunchKernelGGL(k0_prolongate, dim3((P.nrows + 255) / 256), dim3(256), 0, s, L.chunk_col, L.ptile, L.tiles,
                       P.nrows, P.beam_in, std::min(item_ranks, P.beam_in), L.max_tiles_per_parent, P.cand_stride, P.implicit_root, prev.idx, prev.val,
                       prev.cnt, prev.stride, cand_off, ncand, static_cast<ItemDesc*>(items),
                       X.dense ? nullptr : X.row_ptr + P.row0);
    XRL_LAUNCH_CHECK();
}
size_t k0_item_bytes() { return sizeof(ItemDesc); }

// Bound-pruned layers, second phase: the items of beam slots >= first_rank, for the queries whose first phase did NOT already
// prove its top-k final (done[q] == 0), appended to a compact list (one atomicAdd per wavefront).
__global__ void __launch_bounds__(256)
k0b_remaining(const uint32_t* __restrict__ chunk_col, const uint32_t* __restrict__ ptile, const TileDesc* __restrict__ tiles,
              uint32_t nrows, uint32_t beam_in, uint32_t first_rank, uint32_t end_rank, uint32_t cand_stride, const uint32_t* __restrict__ p_idx,
              const float* __restrict__ p_val, const uint32_t* __restrict__ p_cnt, uint32_t p_stride, const uint32_t* __restrict__ cand_off,
              const uint32_t* __restrict__ done, ItemDesc* __restrict__ items, uint32_t* __restrict__ n_items,
              const uint64_t* __restrict__ x_row_ptr) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = q < nrows && !done[q];
    const uint32_t cnt = live ? min(min(p_cnt[q], beam_in), end_rank) : 0u;      // slots [first_rank, end_rank) of the unfinished queries
    uint32_t n = 0;
    for (uint32_t j = first_rank; j < cnt; ++j) { const uint32_t parent = p_idx[(size_t)q * p_stride + j]; n += ptile[parent + 1] - ptile[parent]; }
    uint32_t incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += y; }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(n_items, total);
    base = (uint32_t)__shfl((int)base, 63, 64) + incl - n;
    if (n == 0) return;
    uint64_t xb = 0; uint32_t xl = 0;
    if (x_row_ptr) { xb = x_row_ptr[q]; xl = (uint32_t)(x_row_ptr[q + 1] - xb); }
    for (uint32_t j = first_rank; j < cnt; ++j) {
        const uint32_t parent = p_idx[(size_t)q * p_stride + j];
        const float ps = p_val[(size_t)q * p_stride + j];
        const uint32_t cb = chunk_col[parent], t0 = ptile[parent], nt = ptile[parent + 1] - t0;
        const uint32_t off = cand_off[(size_t)q * beam_in + j];
        for (uint32_t tt = 0; tt < nt; ++tt)
            items[base++] = make_item(q, t0 + tt, q * cand_stride + off + (tiles[t0 + tt].col_begin - cb), ps, xb, xl);
    }
}

void launch_k0b_remaining(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev, const uint32_t* cand_off, const uint32_t* done,
                          uint32_t first_rank, void* items, uint32_t* n_items, hipStream_t s, uint32_t end_rank) {
    if (P.nrows == 0) return;
    XRL_HIP(hipMemsetAsync(n_items, 0, 4, s));
    hipLaunchKernelGGL(k0b_remaining, dim3((P.nrows + 255) / 256), dim3(256), 0, s, L.chunk_col, L.ptile, L.tiles, P.nrows, P.beam_in, first_rank, end_rank,
                       P.cand_stride, prev.idx, prev.val, prev.cnt, prev.stride, cand_off, done, static_cast<ItemDesc*>(items), n_items,
                       X.dense ? nullptr : X.row_ptr + P.row0);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Item ordering: counting sort of the layer's item descriptors by tile id, so that the wavefronts
// working on one tile run back to back on ONE XCD and find the tile's bitmap / rows / entries in
// that XCD's L2 instead of HBM (the reference sorts its (query, chunk) pairs by chunk for the same
// reason, inference.hpp:991-993).  Device-scope atomics are slow across the 8 XCDs, so the sort
// uses only LDS atomics: per-block LDS histograms -> per-(block, tile) offsets -> LDS-ranked
// scatter.  Order inside a tile is arbitrary; results do not depend on it.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSortChunk = 8192;    // item slots per block

__global__ void __launch_bounds__(256)
sort_hist_kernel(const ItemDesc* __restrict__ items, uint64_t n_slots, const uint32_t* __restrict__ n_dev, uint32_t T, uint32_t* __restrict__ H) {
    extern __shared__ uint32_t hist[];
    if (n_dev) n_slots = min(n_slots, (uint64_t)*n_dev);          // compacted list: only its first *n_dev slots are items
    if ((uint64_t)blockIdx.x * kSortChunk >= n_slots) return;      // (the later kernels skip this block's histogram as well)
    for (uint32_t t = threadIdx.x; t < T; t += 256) hist[t] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortChunk;
    for (uint32_t i = threadIdx.x; i < kSortChunk && base + i < n_slots; i += 256) {
        const uint32_t tile = items[base + i].tile;
        if (tile != kNoTile) atomicAdd(&hist[tile], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += 256) H[(size_t)blockIdx.x * T + t] = hist[t];
}

// per tile: exclusive running sum over blocks (in place), total per tile
__global__ void __launch_bounds__(256)
sort_colsum_kernel(uint32_t* __restrict__ H, uint32_t B, const uint32_t* __restrict__ n_dev, uint32_t T, uint32_t* __restrict__ total) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= T) return;
    if (n_dev) B = min(B, (uint32_t)(((uint64_t)*n_dev + kSortChunk - 1) / kSortChunk));
    uint32_t run = 0;
    for (uint32_t b = 0; b < B; ++b) { const uint32_t c = H[(size_t)b * T + t]; H[(size_t)b * T + t] = run; run += c; }
    total[t] = run;
}

// single block: exclusive scan of v[0..n) in place; v[n] = grand total
__global__ void __launch_bounds__(1024) sort_scan_kernel(uint32_t* __restrict__ v, uint32_t n) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < n ? v[i] : 0u;
        part[threadIdx.x] = x;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) v[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) v[n] = carry;
}

__global__ void __launch_bounds__(256)
sort_scatter_kernel(const ItemDesc* __restrict__ items, uint64_t n_slots, const uint32_t* __restrict__ n_dev, uint32_t T, const uint32_t* __restrict__ H,
                    const uint32_t* __restrict__ start, ItemDesc* __restrict__ sorted) {
    extern __shared__ uint32_t pos[];
    if (n_dev) n_slots = min(n_slots, (uint64_t)*n_dev);
    if ((uint64_t)blockIdx.x * kSortChunk >= n_slots) return;
    for (uint32_t t = threadIdx.x; t < T; t += 256) pos[t] = start[t] + H[(size_t)blockIdx.x * T + t];
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortChunk;
    for (uint32_t i = threadIdx.x; i < kSortChunk && base + i < n_slots; i += 256) {
        const ItemDesc d = items[base + i];
        if (d.tile != kNoTile) sorted[atomicAdd(&pos[d.tile], 1u)] = d;
    }
}

uint32_t sort_max_tiles() { return 36864; }   // LDS histogram: 4 B per tile, <= 144 KiB
size_t sort_hist_bytes(uint64_t n_slots, uint32_t T) { return ((n_slots + kSortChunk - 1) / kSortChunk) * (size_t)T * 4; }

void launch_sort_items(const LayerDev& L, uint64_t n_slots, const void* items, void* sorted, uint32_t* H,
                       uint32_t* start /*[n_tiles+1]*/, hipStream_t s, const uint32_t* n_dev) {
    if (n_slots == 0) return;
    const uint32_t T = L.n_tiles;
    if (T > sort_max_tiles()) fail("sort_items: too many tiles for the LDS histogram");
    const uint32_t B = (uint32_t)((n_slots + kSortChunk - 1) / kSortChunk);
    const size_t lds = (size_t)T * 4;
    if (lds > 48 * 1024) {   // per DEVICE attribute: set on every large launch (a per-thread cache would miss a second GPU)
        XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(sort_hist_kernel, dim3(B), dim3(256), lds, s, static_cast<const ItemDesc*>(items), n_slots, n_dev, T, H);
    hipLaunchKernelGGL(sort_colsum_kernel, dim3((T + 255) / 256), dim3(256), 0, s, H, B, n_dev, T, start);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, s, start, T);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(B), dim3(256), lds, s, static_cast<const ItemDesc*>(items), n_slots, n_dev, T, H,
                       start, static_cast<ItemDesc*>(sorted));
    XRL_LAUNCH_CHECK();
}

// Profiling aid: with XRL_STEP_MARKER=1 every predict_device call opens with this empty kernel, so that a rocprofv3 kernel trace can be
// cut into steps whatever kernels the layers run (scripts/pmc_traffic.py).  Never launched otherwise.
__global__ void step_marker_kernel() {}
void launch_step_marker(hipStream_t s) {
    hipLaunchKernelGGL(step_marker_kernel, dim3(1), dim3(64), 0, s);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Query ordering for a query-stationary layer (K1Q): counting sort of the QUERIES by the best parent of their beam
// (beam slot 0 = the parent whose children a query most likely keeps), written as a permutation.  K1Q then runs query perm[i] in
// launch slot i and hands every XCD a CONTIGUOUS range of slots (xrl_k1q.hip), so that the queries of one region of the tree -- which
// share most of their beam parents and, on topical data, many of their features -- request their (feature, parent) weight segments
// through the same L2 at about the same time.  The reference orders its (query, chunk) work by chunk for the same reason
// (inference.hpp:969-993).  Results do not depend on the order: every query writes to its own row of the output.
// Same scheme as the item sort: per-block LDS histograms, per-(block, key) offsets, LDS-ranked scatter.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kQSortChunk = 2048;   // queries per block

__device__ __forceinline__ uint32_t qsort_key(const uint32_t* __restrict__ p_idx, const uint32_t* __restrict__ p_cnt, uint32_t p_stride, uint32_t q, uint32_t T) {
    return p_cnt[q] ? min(p_idx[(size_t)q * p_stride], T - 1u) : T - 1u;
}

__global__ void __launch_bounds__(256)
qsort_hist_kernel(const uint32_t* __restrict__ p_idx, const uint32_t* __restrict__ p_cnt, uint32_t p_stride, uint32_t nrows, uint32_t T, uint32_t* __restrict__ H) {
    extern __shared__ uint32_t hist[];
    for (uint32_t t = threadIdx.x; t < T; t += 256) hist[t] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kQSortChunk;
    for (uint32_t i = threadIdx.x; i < kQSortChunk && base + i < nrows; i += 256) atomicAdd(&hist[qsort_key(p_idx, p_cnt, p_stride, base + i, T)], 1u);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += 256) H[(size_t)blockIdx.x * T + t] = hist[t];
}

__global__ void __launch_bounds__(256)
qsort_scatter_kernel(const uint32_t* __restrict__ p_idx, const uint32_t* __restrict__ p_cnt, uint32_t p_stride, uint32_t nrows, uint32_t T,
                     const uint32_t* __restrict__ H, const uint32_t* __restrict__ start, uint32_t* __restrict__ perm) {
    extern __shared__ uint32_t pos[];
    for (uint32_t t = threadIdx.x; t < T; t += 256) pos[t] = start[t] + H[(size_t)blockIdx.x * T + t];
    __syncthreads();
    const uint32_t base = blockIdx.x * kQSortChunk;
    for (uint32_t i = threadIdx.x; i < kQSortChunk && base + i < nrows; i += 256)
        perm[atomicAdd(&pos[qsort_key(p_idx, p_cnt, p_stride, base + i, T)], 1u)] = base + i;
}

uint32_t qsort_max_keys() { return 12288; }   // LDS histogram of 4 B per key: 48 KiB, no opt-in needed
size_t qsort_hist_bytes(uint32_t nrows, uint32_t T) { return ((size_t)(nrows + kQSortChunk - 1) / kQSortChunk) * (size_t)T * 4; }

void launch_sort_queries(BeamDev prev, uint32_t nrows, uint32_t n_keys, uint32_t* H, uint32_t* start /*[n_keys+1]*/, uint32_t* perm, hipStream_t s) {
    if (nrows == 0) return;
    if (n_keys == 0 || n_keys > qsort_max_keys()) fail("sort_queries: key range outside the LDS histogram");
    const uint32_t B = (nrows + kQSortChunk - 1) / kQSortChunk;
    const size_t lds = (size_t)n_keys * 4;
    hipLaunchKernelGGL(qsort_hist_kernel, dim3(B), dim3(256), lds, s, prev.idx, prev.cnt, prev.stride, nrows, n_keys, H);
    hipLaunchKernelGGL(sort_colsum_kernel, dim3((n_keys + 255) / 256), dim3(256), 0, s, H, B, nullptr, n_keys, start);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, s, start, n_keys);
    hipLaunchKernelGGL(qsort_scatter_kernel, dim3(B), dim3(256), lds, s, prev.idx, prev.cnt, prev.stride, nrows, n_keys, H, start, perm);
    XRL_LAUNCH_CHECK();
}

// XCD-aware block remap (blocks b, b+8, b+16, ... run on one XCD): give every XCD a CONTIGUOUS
// range of the tile-sorted work so a tile's data is fetched into one L2 only.  Bijective on [0, nb).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb) {
    const uint32_t xcd = b & 7u, q = nb >> 3, r = nb & 7u;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (b >> 3);
}

// ---------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------
struct K1Args {
    LayerDev L;
    QueriesDev X;
    const ItemDesc* items;
    const uint32_t* n_items;     // device count of (tile-sorted, all active) items, or nullptr: natural order
    float* cand;
    uint64_t n_slots;
    uint32_t row0, acc_stride;
    int pp_kind, pp_p, first_layer;
    int bias_first;              // sparse X, HASH_CHUNKED arithmetic: accumulators start at the bias product, nothing is added at the end
    int ablate;                  // debug: phase-skipping mask for timing ablations (0 in production)
    uint32_t lds_per_wave;       // bytes of dynamic LDS owned by each wavefront of a block
    uint32_t n_vblocks;          // number of wavefront-sized work blocks
    unsigned long long* phase;   // debug (ablate bit 6): per-phase cycle totals [prologue, fill, D1, D3, epilogue, waves]
    uint32_t* fb_out;            // pruning feedback: the launch's item count (a later stage of a bound-pruned layer) goes to this host-visible word
};

template <int G, int PPC, class ACC>
__device__ __forceinline__ void k1_epilogue(const K1Args& a, const ItemDesc& it, const TileDesc& td, int lig, ACC&& acc_at,
                                            bool add_bias) {
    // bias (sparse X: LAST, inference.hpp:806-811), transform in fp64, combine with the parent's
    // score (skipped on the first layer), write the child block
    if (it.tile == kNoTile || (a.ablate & 16)) return;
    float* __restrict__ out = a.cand + it.out_off;
    const float* __restrict__ bp = a.L.bias_prod + td.col_begin;
    for (uint32_t c = lig; c < td.ncols; c += G) {
        float acc = acc_at(c);
        if (add_bias) acc = __fadd_rn(acc, bp[c]);
        float v = (a.ablate & 8) ? acc : pp_transform<PPC>(a.pp_kind, a.pp_p, acc);
        if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
        out[c] = v;
    }
}

// ---- sparse queries: chunk_ops<csr, bin_search>, inference.hpp:769-813 --------------------------
// One wavefront = 64/G items, G lanes each.  Per step every lane fetches U query features
// (U*G consecutive features per item; all x loads, then all bitmap probes, are in flight together),
// hits are compacted IN FEATURE ORDER into a small per-item LDS FIFO (segmented ballot/popcount),
// then one lane per hit fetches the row extent and its first two entries (all hits of all items at
// once), and finally the rows are applied in order with the G lanes on distinct columns.
// NS = number of G-wide UNITS a tile row can span (NS*G >= widest tile of the layer).
struct __attribute__((packed, aligned(4))) RowExt { uint32_t start, end; };

template <int G, int NS> struct K1Cfg {
    static constexpr int W = 64 / G;                      // items per wavefront
#ifndef XRL_K1_FEAT
#define XRL_K1_FEAT 64
#endif
    static constexpr int U = (G >= 16) ? XRL_K1_FEAT / G : 8;     // query features per lane per step (XRL_K1_FEAT per item)
    static constexpr int H = (G > 32) ? 2 * G : 64;       // hit queue depth per item (>= G)
    static constexpr int UH = H * NS;                     // unit queue depth per item
#ifndef XRL_K1_P
#define XRL_K1_P 4
#endif
#ifndef XRL_K1_CLAMP
#define XRL_K1_CLAMP 1
#endif
#ifndef XRL_K1_NB
#define XRL_K1_NB 2
#endif
    static constexpr int P = XRL_K1_P;                    // units per register batch
    static constexpr int NB = XRL_K1_NB;                  // batches in the ring: NB-1 are loading while one is applied
    static constexpr int TAIL = (2 * NB - 1) * P;         // empty units readable past the longest queue
    static constexpr size_t lds_bytes(uint32_t acc_stride) {
        return (size_t)W * (UH + TAIL) * 8 + (size_t)W * H * 8 + (size_t)W * (acc_stride + G) * 4;
    }
};

#ifndef XRL_K1_WPE
#define XRL_K1_WPE 5
#endif
// LK = row lookup: 0 rank-bitmap {bits32, rank} (8 B / 32 features), 1 bucket table + binary search,
//      2 rank-bitmap {bits64, rank, extent of the word's first row} (16 B / 64 features; sparse tiles)
template <int G, int NS, int PPC, bool DENSE, int LK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XRL_K1_WPE, 8))) k1_kernel(K1Args a) {
    constexpr int W = K1Cfg<G, NS>::W, H = K1Cfg<G, NS>::H, UH = K1Cfg<G, NS>::UH, P = K1Cfg<G, NS>::P, U = K1Cfg<G, NS>::U,
                  NB = K1Cfg<G, NS>::NB, TAIL = K1Cfg<G, NS>::TAIL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    // the wavefronts of a block are fully independent: each owns a slice of the dynamic LDS
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t vblock = blockIdx.x * (blockDim.x >> 6) + wave;
    if (vblock >= a.n_vblocks) return;
    unsigned char* smem = smem_all + (size_t)wave * a.lds_per_wave;
    uint2* uq = reinterpret_cast<uint2*>(smem);                        // units {x value, entry start | count << 25}
    uint2* hq = reinterpret_cast<uint2*>(uq + W * (UH + TAIL));                 // hits  {x value, row slot}
    float* acc = reinterpret_cast<float*>(hq + W * H);
    const uint32_t acc_item = a.acc_stride + G;                        // + one private dummy slot per lane

    const int lane = threadIdx.x & 63;
    const int grp = lane / G, lig = lane % G;
#ifdef XRL_K1_PHASE_PROF   // debug build only: per-phase cycle accounting costs ~12 VGPRs
    const bool prof = a.phase != nullptr;
    unsigned long long t_last = prof ? __builtin_readcyclecounter() : 0ull, t_ph[5] = {0, 0, 0, 0, 0};
    auto tick = [&](int ph) { if (prof) { const unsigned long long t = __builtin_readcyclecounter(); t_ph[ph] += t - t_last; t_last = t; } };
#else
    auto tick = [](int) {};
#endif
    ItemDesc it = make_item(0u, kNoTile, 0u, 0.f, 0, 0u);
    if (a.n_items) {   // tile-sorted list: every XCD takes a contiguous run of tiles
        const uint32_t n = *a.n_items, nb = (n + W - 1) / W;
        if (a.fb_out && vblock == 0 && lane == 0) *a.fb_out = n;
        if (vblock >= nb) return;   // a compacted list (second phase of a pruned layer) usually fills a small part of the grid
        { const uint64_t slot = (uint64_t)xcd_remap(vblock, nb) * W + grp; if (slot < n) it = a.items[slot]; }
    } else {
        const uint64_t slot = (uint64_t)vblock * W + grp;
        if (slot < a.n_slots) it = a.items[slot];
    }
    const bool active = it.tile != kNoTile;
    TileDesc td{};
    uint64_t xe = 0, cur = 0;
    if (active) {
        td = a.L.tiles[it.tile];
        cur = it.x_begin; xe = it.x_begin + it.x_len;   // CSR queries; unused for dense ones
    }
    const uint32_t* __restrict__ rp = a.L.row_ext + td.rowptr_base;
    const Entry* __restrict__ ent = a.L.entries + td.ent_base;
    float* __restrict__ my_acc = acc + (size_t)grp * acc_item;
    uint2* __restrict__ my_hq = hq + (size_t)grp * H;
    uint2* __restrict__ my_uq = uq + (size_t)grp * (UH + TAIL);
    const uint32_t dummy = a.acc_stride + (uint32_t)lig;
    if (DENSE) {   // dense queries: bias FIRST (inference.hpp:824-830); bias_prod already holds 0.0f + bias*w
        const float* __restrict__ bp = a.L.bias_prod + td.col_begin;
        for (uint32_t c = lig; c < td.ncols; c += G) my_acc[c] = a.L.has_bias ? bp[c] : 0.0f;
    } else if (a.bias_first) {   // chunk_ops<csr, hash> (inference.hpp:716-722): 0.0 + bias * w first
        const float* __restrict__ bp = a.L.bias_prod + td.col_begin;
        for (uint32_t c = lig; c < td.ncols; c += G) my_acc[c] = a.L.has_bias ? bp[c] : 0.0f;
    } else if (!(a.ablate & 32)) {
        for (uint32_t c = lig; c < td.ncols; c += G) my_acc[c] = 0.0f;   // std::fill(..., 0.0), inference.hpp:964
    }
    wave_sync_lds();
    if (a.ablate & 2) cur = xe;

    const uint32_t* __restrict__ xi = a.X.col_idx;
    const float* __restrict__ xv = a.X.val;
    const BmWord* __restrict__ bm = a.L.bitmap + (LK != 0 ? 0ull : (uint64_t)(active ? it.tile : 0u) * a.L.nwords);
    const BmWord64* __restrict__ bm64 = a.L.bitmap64 + (LK != 2 ? 0ull : (uint64_t)(active ? it.tile : 0u) * a.L.nwords64);
    const uint32_t* __restrict__ bkt = LK == 1 ? a.L.bucket + (uint64_t)(active ? it.tile : 0u) * (a.L.bk_n + 1u) : nullptr;   // tile-relative row slots
    const uint32_t* __restrict__ ridx_t = a.L.row_idx + td.rowptr_base;                          // the tile's sorted row ids
    const unsigned long long below = (1ull << lig) - 1ull;
    const uint64_t xlast = xe > cur ? xe - 1 : 0;                      // a valid x index for clamped loads
    uint32_t nh = 0;                                                   // hits waiting in this item's queue
    tick(0);

    auto drain = [&]() {
        if (a.ablate & 4) { nh = 0; return; }
        tick(1);
        wave_sync_lds();
        // ---- D1: one lane per hit fetches the row extent and cuts the row into units of <= G entries,
        //      written in order (segmented scan of the unit counts when a row can span several units)
        uint32_t nu = 0;                                               // units queued for this item
        for (uint32_t h0 = 0; __any(h0 < nh); h0 += G) {
            const uint32_t h = h0 + lig;
            const bool ok = h < nh;
            const uint2 hv = my_hq[ok ? h : 0u];
            // queue value: the row slot -- or, with 64-feature bitmap words, the packed extent itself unless its length
            // field reads 0x7F, which marks "slot in the low bits" (kRowLookup; extents of 128-entry rows take that route)
            const bool need = ok && (LK != 2 || DENSE || (hv.y >> 25) == 0x7Fu);
            const uint32_t s = need ? ((LK == 2 && !DENSE) ? (hv.y & 0x1FFFFFFu) : hv.y) : 0u;
            const uint32_t rl = rp[s];                                 // unconditional (slot 0 when not needed): packed {offset, length - 1}
            const uint32_t rx = need ? rl : hv.y;
            const uint32_t rs = rx & 0x1FFFFFFu;
            const uint32_t len = ok ? (rx >> 25) + 1u : 0u;
            uint32_t cnt = (NS == 1) ? (len ? 1u : 0u) : min((len + G - 1) / G, (uint32_t)NS);
            uint32_t incl = cnt;
            if (G > 1) {
#pragma unroll
                for (int d = 1; d < G; d <<= 1) { const uint32_t y = __shfl_up(incl, d, G); if (lig >= d) incl += y; }
            }
            const uint32_t base = nu + incl - cnt;
#pragma unroll
            for (int k = 0; k < NS; ++k)
                if ((uint32_t)k < cnt) {
                    const uint32_t n_k = (k == NS - 1) ? len - (uint32_t)k * G : min(len - (uint32_t)k * G, (uint32_t)G);
                    my_uq[base + k] = make_uint2(hv.x, (rs + (uint32_t)k * G) | (n_k << 25));   // offsets < 2^25: xrl_model.cpp max_tile_entries
                }
            nu += (G > 1) ? __shfl(incl, G - 1, G) : incl;
        }
        // every item's queue is read up to the longest queue of the wavefront (+ the prefetch distance):
        // fill the difference with empty units
        uint32_t nu_max = nu;
#pragma unroll
        for (int d = G; d < 64; d <<= 1) nu_max = max(nu_max, (uint32_t)__shfl_xor((int)nu_max, d, 64));
        nu_max = __builtin_amdgcn_readfirstlane(nu_max);
        for (uint32_t j = nu + lig; j < nu_max + (uint32_t)TAIL; j += G) my_uq[j] = make_uint2(0u, 0u);
        wave_sync_lds();
        tick(2);
        // ---- D3: units in order.  Two register batches of P units are in flight: while batch A is
        //      applied the entries of batch B are already loading.  Every load is unconditional and
        //      unclamped (a load behind a per-lane branch makes hipcc wait vmcnt(0) before each one): the
        //      queue ends with 2P empty units and lanes past a unit's end read whatever follows the row
        //      (the entry array is padded) and add it to a private dummy slot.  The lanes of a unit hold
        //      distinct columns.  LDS operations of one wavefront execute in order, so only a compiler
        //      fence separates units.
        const uint2* __restrict__ uqp = my_uq;
        struct Batch { uint32_t xv[P], cn[P]; Entry e[P]; };
        auto load_batch = [&](const uint2* q, Batch& B) {
            uint32_t st[P];
#pragma unroll
            for (int p = 0; p < P; ++p) { const uint2 d = q[p]; B.xv[p] = d.x; st[p] = d.y & 0x1FFFFFFu; B.cn[p] = d.y >> 25; }
#pragma unroll
            for (int p = 0; p < P; ++p) {
#if XRL_K1_CLAMP   // lanes past the unit's end re-read its first entry (no extra cache lines) instead of running on
                B.e[p] = ent[st[p] + ((uint32_t)lig < B.cn[p] ? (uint32_t)lig : 0u)];
#else
                B.e[p] = ent[st[p] + (uint32_t)lig];
#endif
            }
        };
        auto apply_batch = [&](const Batch& B) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float v = __uint_as_float(B.xv[p]);
                const uint32_t ci = (uint32_t)lig < B.cn[p] ? B.e[p].col : dummy;
                my_acc[ci] = __fadd_rn(my_acc[ci], __fmul_rn(v, B.e[p].val));   // scalar * val, then add: no fma (inference.hpp:512-517)
                wave_sync_lds();
            }
        };
        Batch ring[NB];
#pragma unroll
        for (int b = 0; b < NB - 1; ++b) load_batch(uqp + b * P, ring[b]);
        for (uint32_t i0 = 0; i0 < nu_max; i0 += NB * P) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                load_batch(uqp + (b + NB - 1) * P, ring[(b + NB - 1) % NB]);
                apply_batch(ring[b]);
            }
            uqp += NB * P;
        }
        nh = 0;
        tick(3);
    };

    if (DENSE) {
        // chunk_ops<drm, bin_search>, inference.hpp:815-839: EVERY tile row (except the bias row, which is
        // the last one) is a hit with x value x[row feature]; rows go through the same unit queue.
        const float* __restrict__ xd = a.X.val + ((uint64_t)a.row0 + it.q) * a.X.cols;
        const uint32_t* __restrict__ ridx = a.L.row_idx + td.rowptr_base;
        uint32_t nr = active ? td.nrows : 0u;
        if (active && td.bias_slot != kNoBias) nr -= 1;
        for (uint32_t s0 = 0; __any(s0 < nr); s0 += H) {
            for (uint32_t j = lig; j < (uint32_t)H; j += G) {
                const uint32_t sidx = s0 + j;
                const bool ok = sidx < nr;
                const uint32_t f = ridx[ok ? sidx : 0u];
                const float xval = xd[f < a.X.cols ? f : 0u];
                if (ok) my_hq[j] = make_uint2(__float_as_uint(f < a.X.cols ? xval : 0.0f), sidx);
            }
            nh = s0 < nr ? min((uint32_t)H, nr - s0) : 0u;
            drain();
        }
        k1_epilogue<G, PPC>(a, it, td, lig, [&](uint32_t c) { return my_acc[c]; }, false);
        return;
    }
    uint32_t skip = 0;                 // u-slices of the current step already queued (after an overflow)
    while (__any(cur < xe)) {
        bool overflow = false;
        {
            // ---- load step: U*G consecutive features of the item
            uint32_t f[U]; float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t t = cur + (uint64_t)(u * G + lig);
                const bool ok = t < xe;
                const uint64_t tc = ok ? t : xlast;          // clamped: the load itself is unconditional
                const uint32_t fi = xi[tc];
                const float vi = xv[tc];
                f[u] = (ok && fi < a.L.w_rows) ? fi : 0xFFFFFFFFu;
                v[u] = vi;
            }
            // ---- row lookup: is feature f a row of the tile, and which slot
            bool hit[U]; uint32_t slot[U];      // slot: what goes into the hit queue (row slot; LK 2: extent or marked slot)
            if (LK == 0) {
                // rank-bitmap: one 8-byte load per probe
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool inr = f[u] != 0xFFFFFFFFu && !(a.ablate & 1);
                    const BmWord wi = bm[inr ? (f[u] >> 5) : 0u];
                    const uint32_t b = f[u] & 31u;
                    hit[u] = inr && ((wi.bits >> b) & 1u);
                    slot[u] = wi.rank + (uint32_t)__popc(wi.bits & ((1u << b) - 1u));
                }
            } else if (LK == 2) {
                // sparse tiles (few rows per word): one 16-byte load per probe returns the word AND the extent of its
                // first row, so only hits on a later row of the word go through the extent table in D1
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool inr = f[u] != 0xFFFFFFFFu;
                    const BmWord64 wi = bm64[inr ? (f[u] >> 6) : 0u];
                    const uint32_t b = f[u] & 63u;
                    const unsigned long long bits = ((unsigned long long)wi.hi << 32) | wi.lo;
                    hit[u] = inr && ((bits >> b) & 1ull);
                    const uint32_t before = (uint32_t)__popcll(bits & ((1ull << b) - 1ull));
                    slot[u] = before == 0u ? wi.ext0 : (0xFE000000u | (wi.rank + before));
                }
            } else {
                // layers whose bitmaps would not fit in HBM: bucket table over feature-id ranges (one 8-byte load), then
                // `bk_levels` branch-free binary-search steps over the tile's sorted row ids (the reference's lookup is a
                // binary search too, inference.hpp:786-803) and one load to confirm the match
                uint32_t hi[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool inr = f[u] != 0xFFFFFFFFu;
                    const RowExt be = *reinterpret_cast<const RowExt*>(bkt + (inr ? (f[u] >> a.L.bk_shift) : 0u));
                    slot[u] = be.start; hi[u] = inr ? be.end : be.start;
                }
                for (uint32_t lv = a.L.bk_levels; lv > 0; --lv) {
                    const uint32_t stp = 1u << (lv - 1);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t np = slot[u] + stp;
                        const bool in = np < hi[u];
                        const uint32_t r = ridx_t[in ? np : slot[u]];
                        slot[u] = (in && r <= f[u]) ? np : slot[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) hit[u] = slot[u] < hi[u] && ridx_t[slot[u]] == f[u];
            }
            // ---- queue the hits in feature order.  If an item's queue fills up the step is abandoned at
            //      slice `skip`, the queue is drained (outside this scope, so the step's registers are
            //      dead by then) and the same step is re-loaded and resumed from that slice.
            uint32_t done = skip;
            bool stopped = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long m = __ballot(hit[u]);
                const unsigned long long gm = (G == 64) ? m : ((m >> (grp * G)) & ((1ull << G) - 1ull));
                const uint32_t cnt = (uint32_t)__popcll(gm);
                if ((uint32_t)u >= done && !stopped) {
                    if (nh + cnt <= (uint32_t)H) {
                        if (hit[u]) my_hq[nh + (uint32_t)__popcll(gm & below)] = make_uint2(__float_as_uint(v[u]), slot[u]);
                        nh += cnt; done = u + 1;
                    } else {
                        stopped = true;
                    }
                }
            }
            if (done == (uint32_t)U) { if (cur < xe) cur += (uint64_t)U * G; skip = 0; }
            else { skip = done; overflow = true; }
        }
        if (__any(overflow)) drain();
    }
    drain();
    k1_epilogue<G, PPC>(a, it, td, lig, [&](uint32_t c) { return my_acc[c]; }, a.L.has_bias != 0 && !a.bias_first);
#ifdef XRL_K1_PHASE_PROF
    if (prof) {
        tick(4);
        if (lane == 0) { for (int i = 0; i < 5; ++i) atomicAdd(&a.phase[i], t_ph[i]); atomicAdd(&a.phase[5], 1ull); }
    }
#endif
}

template <class KERNEL>
static void launch_k1_any(KERNEL kernel, K1Args a, int W, size_t lds_wave, const K1Tune& tune, hipStream_t s) {
    lds_wave = (lds_wave + (size_t)std::max(0, tune.lds_pad) + 15) & ~(size_t)15;
    int wpb = (tune.wpb == 2 || tune.wpb == 4) ? tune.wpb : 1;
    while (wpb > 1 && lds_wave * wpb > 160 * 1024) wpb >>= 1;
    const size_t lds = lds_wave * wpb;
    if (lds > 160 * 1024) fail("k1: LDS request exceeds 160 KiB");
    if (lds > 48 * 1024)
        XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t vblocks = (a.n_slots + W - 1) / W;
    const uint64_t blocks = (vblocks + wpb - 1) / wpb;
    if (vblocks > 0x7FFFFFFFull) fail("k1: grid too large; lower max_batch_rows");
    a.lds_per_wave = (uint32_t)lds_wave; a.n_vblocks = (uint32_t)vblocks;
    hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(64 * wpb), lds, s, a);
    XRL_LAUNCH_CHECK();
}

static unsigned long long* g_phase_buf = nullptr;   // debug only (k1_ablate bit 6): one per process
unsigned long long* k1_phase_buffer() {
    if (!g_phase_buf) { XRL_HIP(hipMalloc(&g_phase_buf, 8 * 8)); XRL_HIP(hipMemset(g_phase_buf, 0, 64)); }
    return g_phase_buf;
}
void k1_phase_read(unsigned long long out[8], bool reset) {
    XRL_HIP(hipDeviceSynchronize());
    XRL_HIP(hipMemcpy(out, k1_phase_buffer(), 64, hipMemcpyDeviceToHost));
    if (reset) XRL_HIP(hipMemset(g_phase_buf, 0, 64));
}

int k1_auto_group(const LayerDev& L, const Layer& host, int dense) {
    // lanes per item: a 16- or 32-lane group whose NS slices cover the widest tile row
    (void)host; (void)dense;
    if (L.max_tile_cols <= 8) return 8;
    if (L.max_tile_cols <= 16) return 16;
    return 32;
}

void launch_k1(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items, const uint32_t* n_items,
               float* cand, int group, hipStream_t s) {
    if (P.nrows == 0) return;
    // K1T (densely held tile rows, accumulators in registers): measured faster on items in QUERY order (a bound-pruned layer's first stage: Amazon-670K
    // 1.010 -> 0.959 ms, Wiki10-31K 0.334 -> 0.286 ms) and slower on tile-sorted lists (Amazon-670K-hard 7.29 -> 8.48 ms: both kernels run at the L1-miss
    // request ceiling of ~80 G requests/s and a 384-byte dense row is 6 requests against ~4 for its entry list) -- tile_rows 1 = query-order launches only, 2 = all
    if ((P.tune.tile_rows >= 2 || (P.tune.tile_rows == 1 && !n_items)) && P.tune.ablate == 0 && k1t_serves(L, X)) { launch_k1t(L, P, X, items, n_items, cand, s); return; }
    K1Args a;
    a.L = L; a.X = X; a.items = static_cast<const ItemDesc*>(items); a.n_items = n_items; a.cand = cand;
    a.n_slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
    a.row0 = P.row0; a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.bias_first = P.bias_first;
    a.acc_stride = L.max_tile_cols | 1u;
    const int ablate = P.tune.ablate;
    a.ablate = ablate & 0xFF;
    // debug: bit 6 = per-phase cycle accounting; bits 8.. select one layer (value layer+1, 0 = every layer)
    a.phase = ((ablate & 64) && ((ablate >> 8) == 0 || (ablate >> 8) == P.layer + 1)) ? k1_phase_buffer() : nullptr;
    a.fb_out = (n_items && P.fb_host && P.layer >= 0 && P.layer < 16) ? P.fb_host + 32 + P.layer : nullptr;
    const int ppc = pp_class(P.pp);
#define XRL_K1_PP(GG, NN, DD, LL) do { if (ppc) launch_k1_any(&k1_kernel<GG, NN, 1, DD, LL>, a, 64 / GG, lds, P.tune, s); else launch_k1_any(&k1_kernel<GG, NN, 0, DD, LL>, a, 64 / GG, lds, P.tune, s); } while (0)
#define XRL_K1(GG, NN) do { \
        const size_t lds = K1Cfg<GG, NN>::lds_bytes(a.acc_stride); \
        if (X.dense) XRL_K1_PP(GG, NN, true, 0); \
        else if (L.bucket) XRL_K1_PP(GG, NN, false, 1); \
        else if (L.bitmap64) XRL_K1_PP(GG, NN, false, 2); \
        else XRL_K1_PP(GG, NN, false, 0); } while (0)
    // a tile row must fit NS units of `group` lanes; widen a (forced) group that is too narrow
    if (group < 1 || group > 64 || (group & (group - 1))) fail("k1: lanes-per-item must be a power of two in [1, 64]");
    auto max_ns = [](int g) { return g < 8 ? 1u : (g == 32 ? 4u : 2u); };
    auto units = [&](int g) { return (L.max_tile_cols + (uint32_t)g - 1) / (uint32_t)g; };
    while (group < 64 && units(group) > max_ns(group)) group <<= 1;
    const uint32_t ns = units(group);
    switch (group) {
    case 1: XRL_K1(1, 1); break;
    case 2: XRL_K1(2, 1); break;
    case 4: XRL_K1(4, 1); break;
    case 8: if (ns <= 1) XRL_K1(8, 1); else XRL_K1(8, 2); break;
    case 16: if (ns <= 1) XRL_K1(16, 1); else XRL_K1(16, 2); break;
    case 32: if (ns <= 1) XRL_K1(32, 1); else if (ns == 2) XRL_K1(32, 2); else if (ns == 3) XRL_K1(32, 3); else XRL_K1(32, 4); break;
    default: if (ns <= 1) XRL_K1(64, 1); else XRL_K1(64, 2); break;
    }
#undef XRL_K1
#undef XRL_K1_PP
}

// ---------------------------------------------------------------------------------------------
// K2: one wavefront per query.  Candidates are scanned in POSITION order; the running top-k list
// is kept sorted by (value desc, position asc), which is the comparator of sorted_csr
// (inference.hpp:1265-1273): a later candidate only displaces the current k-th if it is
// STRICTLY greater, and is inserted after every element that is >= it.
// ---------------------------------------------------------------------------------------------
struct K2Args {
    const uint32_t* chunk_col;
    const uint32_t* perm_inv;
    const uint32_t* p_idx; const uint32_t* p_cnt; uint32_t p_stride;
    const uint32_t* cand_off; const uint32_t* ncand; const float* cand;
    uint32_t* out_idx; float* out_val; uint32_t* out_cnt;
    uint32_t nrows, beam_in, cand_stride, k, out_stride;
    int implicit_root;
    // exact bound pruning (k2_topk_wave only): rank_limit > 0 restricts the selection to the candidates of the first rank_limit beam
    // slots and reports in done[q] whether that selection is already FINAL -- every candidate of a later slot scores at most its
    // parent's score (transform <= 1 times / <= 0 plus the parent's score) and would lose a tie by position, so once k selected
    // candidates score >= the next parent's score nothing can change.  skip_done: queries to leave untouched (second phase).
    const float* p_val;
    int mult;                    // the combiner multiplies (sigmoid, l{p}-hinge): a child of a parent with a NEGATIVE score (possible when an
                                 // earlier layer used another post-processor) lies in [score, 0], so the bound is max(score, 0); additive
                                 // combiners (log-*) add a transform <= 0: the bound is the score itself
    uint32_t rank_limit;
    uint32_t* done;
    const uint32_t* skip_done;
    const uint32_t* xok;         // [nrows] the pruning guard of every query (prune_guard_ok): 0 = never final before every candidate is scored
};

__device__ __forceinline__ uint32_t k2_child_id(const K2Args& a, uint64_t q, uint32_t pos) {
    // position -> (beam slot, child) -> original child id (reorder_prediction, inference.hpp:1776-1784)
    uint32_t parent = 0, off = 0;
    if (!a.implicit_root) {
        const uint32_t cnt = min(a.p_cnt[q], a.beam_in);
        uint32_t jj = 0;
        for (uint32_t j = 1; j < cnt; ++j) if (a.cand_off[q * a.beam_in + j] <= pos) jj = j; else break;
        off = a.cand_off[q * a.beam_in + jj];
        parent = a.p_idx[q * a.p_stride + jj];
    }
    const uint32_t child = a.chunk_col[parent] + (pos - off);
    return a.perm_inv ? a.perm_inv[child] : child;
}

__global__ void __launch_bounds__(64) k2_topk_reg(K2Args a) {   // k <= 64: lane i holds the i-th best
    const uint64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t n = a.ncand[q], k = a.k;
    const float* __restrict__ cv = a.cand + q * a.cand_stride;
    float lv = -INFINITY, th = -INFINITY;
    uint32_t lp = 0, m = 0;
    constexpr int KB = 4;     // candidate batches (64 each) fetched per iteration: KB loads in flight
    const uint32_t nlast = n ? n - 1 : 0;
    for (uint32_t base0 = 0; base0 < n; base0 += 64 * KB) {
        float vb[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b) { const uint32_t p = base0 + b * 64 + lane; vb[b] = cv[p < n ? p : nlast]; }   // unconditional, clamped
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const uint32_t base = base0 + b * 64;
            const uint32_t p = base + lane;
            const bool valid = p < n;
            const float v = vb[b];
            unsigned long long mask = __ballot(valid && (m < k || v > th));
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float vv = __shfl(v, l);
                if (m == k && !(vv > th)) continue;
                const int r = __popcll(__ballot((uint32_t)lane < m && lv >= vv));
                const float uv = __shfl_up(lv, 1);
                const uint32_t up = __shfl_up(lp, 1);
                if (lane > r) { lv = uv; lp = up; }
                else if (lane == r) { lv = vv; lp = base + l; }
                if (m < k) ++m;
                th = (m == k) ? __shfl(lv, (int)k - 1) : -INFINITY;
            }
        }
    }
    if ((uint32_t)lane < m) {
        a.out_idx[q * a.out_stride + lane] = k2_child_id(a, q, lp);
        a.out_val[q * a.out_stride + lane] = lv;
    }
    if (lane == 0) a.out_cnt[q] = m;
}

__global__ void __launch_bounds__(64) k2_topk_lds(K2Args a) {   // any k that fits LDS: sorted list in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lv = reinterpret_cast<float*>(smem);
    uint32_t* lp = reinterpret_cast<uint32_t*>(lv + a.k);
    const uint64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t n = a.ncand[q], k = a.k;
    const float* __restrict__ cv = a.cand + q * a.cand_stride;
    float th = -INFINITY;
    uint32_t m = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t p = base + lane;
        const bool valid = p < n;
        const float v = valid ? cv[p] : 0.f;
        unsigned long long mask = __ballot(valid && (m < k || v > th));
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float vv = __shfl(v, l);
            if (m == k && !(vv > th)) continue;
            uint32_t r = 0;
            for (uint32_t i0 = 0; i0 < m; i0 += 64) {
                const uint32_t i = i0 + lane;
                r += (uint32_t)__popcll(__ballot(i < m && lv[i] >= vv));
            }
            const uint32_t e = (m < k) ? m : k - 1;        // elements [r, e) move up by one
            for (uint32_t hi = e; hi > r;) {
                const uint32_t lo = (hi - r > 64) ? hi - 64 : r;
                const uint32_t i = lo + lane;
                const bool mv = i < hi;
                float tv = 0.f; uint32_t tp = 0;
                if (mv) { tv = lv[i]; tp = lp[i]; }
                wave_sync_lds();
                if (mv) { lv[i + 1] = tv; lp[i + 1] = tp; }
                wave_sync_lds();
                hi = lo;
            }
            if (lane == 0) { lv[r] = vv; lp[r] = base + l; }
            wave_sync_lds();
            if (m < k) ++m;
            th = (m == k) ? lv[k - 1] : -INFINITY;
        }
    }
    wave_sync_lds();
    for (uint32_t i = lane; i < m; i += 64) {
        a.out_idx[q * a.out_stride + i] = k2_child_id(a, q, lp[i]);
        a.out_val[q * a.out_stride + i] = lv[i];
    }
    if (lane == 0) a.out_cnt[q] = m;
}

// K2, register form (k <= 64, candidate rows of up to 64 * NS scores): the whole candidate row sits in registers
// (candidate p = r*64 + lane) and wave_topk (xrl_device.h) selects and ranks with ballot bisection instead of
// serial insertions.  Four queries (wavefronts) per workgroup.
template <int NS>
__global__ void __launch_bounds__(256) k2_topk_wave(K2Args a) {
    __shared__ uint2 sc_all[4 * 64];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q32 = blockIdx.x * 4u + wave;
    if (q32 >= a.nrows) return;
    const uint64_t q = q32;
    if (a.skip_done && a.skip_done[q]) return;
    uint32_t n = min(a.ncand[q], (uint32_t)(64 * NS));
    const uint32_t bcnt0 = a.implicit_root ? 1u : min(a.p_cnt[q], a.beam_in);
    // (both loads are issued up front, whether or not the query has that many parents: no dependent round trips later)
    const uint32_t rl = min(a.rank_limit, a.beam_in - 1u);
    const uint32_t lim_off = a.rank_limit ? a.cand_off[q * a.beam_in + rl] : 0u;
    const float ps_next = a.rank_limit ? a.p_val[q * a.p_stride + rl] : 0.0f;
    const bool limited = a.rank_limit != 0u && bcnt0 > a.rank_limit;
    if (limited) n = min(n, lim_off);
    const float* __restrict__ cv = a.cand + q * a.cand_stride;
    const uint32_t slast = a.cand_stride - 1u;                  // last float of the query's candidate row (the loads below do not wait for n)
    // the beam's block offsets and parents, one per lane (beams of up to 64 parents): in flight while the candidates are ranked,
    // so that mapping a winner's position back to its child needs no dependent loads afterwards
    const uint32_t bcnt = a.implicit_root ? 1u : min(a.p_cnt[q], a.beam_in);
    const bool lane_beam = !a.implicit_root && bcnt <= 64u;
    uint32_t b_off = 0xFFFFFFFFu, b_par = 0u, b_cc = 0u;
    if (lane_beam && (uint32_t)lane < bcnt) { b_off = a.cand_off[q * a.beam_in + lane]; b_par = a.p_idx[q * a.p_stride + lane]; b_cc = a.chunk_col[b_par]; }
    uint32_t key[NS], sbits[NS], pos[NS];
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        const uint32_t p = (uint32_t)r * 64u + (uint32_t)lane;
        const float v = cv[min(p, slast)];                         // unconditional, clamped to the row (positions >= n are masked below)
        sbits[r] = __float_as_uint(v); pos[r] = p;
        key[r] = p < n ? score_key(v) : 0u;
    }
    if (a.done) {   // (before the selection: it consumes the keys)
        bool d = true;
        // the k-th best >= the best any later slot can reach (a NaN parent score proves nothing: no pruning)
        if (limited) d = a.xok[q] != 0u && ps_next == ps_next && wave_count_ge<NS>(key, score_key(a.mult ? fmaxf(ps_next, 0.0f) : ps_next)) >= a.k;
        if (lane == 0) a.done[q] = d ? 1u : 0u;
    }
    uint32_t rank, sb, pp;
    const uint32_t kk = wave_topk<NS>(key, sbits, pos, a.k, sc_all + wave * 64u, lane, rank, sb, pp);
    uint32_t child;
    if (lane_beam) {
        uint32_t jj = 0;                                            // last beam slot whose block starts at or before the position
        for (uint32_t j = 1; j < bcnt; ++j) jj = ((uint32_t)__builtin_amdgcn_readlane((int)b_off, (int)j) <= pp) ? j : jj;
        const uint32_t off = (uint32_t)__shfl((int)b_off, (int)jj, 64), cc = (uint32_t)__shfl((int)b_cc, (int)jj, 64);
        child = cc + (pp - off);
        if ((uint32_t)lane < kk && a.perm_inv) child = a.perm_inv[child];
    } else {
        child = (uint32_t)lane < kk ? k2_child_id(a, q, pp) : 0u;
    }
    if ((uint32_t)lane < kk) {
        a.out_idx[q * a.out_stride + rank] = child;
        a.out_val[q * a.out_stride + rank] = __uint_as_float(sb);
    }
    if (lane == 0) a.out_cnt[q] = kk;
}

size_t k2_max_k() { return (160 * 1024) / 8; }

bool k2_wave_path(const LayerPlan& P) { return P.k <= 64 && P.cand_stride <= 64u * 32u; }

void launch_k2_topk(const LayerDev& L, const LayerPlan& P, BeamDev prev, const uint32_t* cand_off,
                    const uint32_t* ncand, const float* cand, uint32_t* out_idx, float* out_val,
                    uint32_t* out_cnt, uint32_t out_stride, hipStream_t s, uint32_t rank_limit, uint32_t limited_cands,
                    uint32_t* done, const uint32_t* skip_done, const uint32_t* xok) {
    if (P.nrows == 0) return;
    K2Args a;
    a.p_val = prev.val; a.rank_limit = rank_limit; a.done = done; a.skip_done = skip_done; a.xok = xok;
    if (done && !xok) fail("k2: bound pruning needs the per-query guard flags");
    a.mult = (P.pp.kind == PP_SIGMOID || P.pp.kind == PP_LP_HINGE) ? 1 : 0;
    if ((rank_limit || done || skip_done) && !k2_wave_path(P)) fail("k2: bound pruning needs the register top-k path");
    a.chunk_col = L.chunk_col; a.perm_inv = L.perm_inv;
    a.p_idx = prev.idx; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
    a.cand_off = cand_off; a.ncand = ncand; a.cand = cand;
    a.out_idx = out_idx; a.out_val = out_val; a.out_cnt = out_cnt;
    a.nrows = P.nrows; a.beam_in = P.beam_in; a.cand_stride = P.cand_stride; a.k = P.k; a.out_stride = out_stride;
    a.implicit_root = P.implicit_root;
    if (P.k == 0) fail("k2: only_topk / beam_size resolved to 0");
    // beyond the LDS kernel's reach (or forced, tests: k2_big_min_k): the segmented sort of xrl_topk_big.hip
    if (!rank_limit && !done && !skip_done && (P.k > k2_max_k() || (P.tune.k2_big_min_k > 0 && P.k >= (uint32_t)P.tune.k2_big_min_k))) {
        launch_k2_topk_big(L, P, prev, cand_off, ncand, cand, out_idx, out_val, out_cnt, out_stride, s);
        return;
    }
    if (k2_wave_path(P)) {
        // (a rank-limited selection looks at the first slots' candidates only: registers for that many)
        const uint32_t ns = ((rank_limit ? std::min(P.cand_stride, std::max(1u, limited_cands)) : P.cand_stride) + 63u) / 64u;
        const dim3 grid((P.nrows + 3u) / 4u), block(256);
        if (ns <= 1) hipLaunchKernelGGL(k2_topk_wave<1>, grid, block, 0, s, a);
        else if (ns <= 2) hipLaunchKernelGGL(k2_topk_wave<2>, grid, block, 0, s, a);
        else if (ns <= 4) hipLaunchKernelGGL(k2_topk_wave<4>, grid, block, 0, s, a);
        else if (ns <= 8) hipLaunchKernelGGL(k2_topk_wave<8>, grid, block, 0, s, a);
        else if (ns <= 13) hipLaunchKernelGGL(k2_topk_wave<13>, grid, block, 0, s, a);
        else if (ns <= 16) hipLaunchKernelGGL(k2_topk_wave<16>, grid, block, 0, s, a);
        else if (ns <= 24) hipLaunchKernelGGL(k2_topk_wave<24>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(k2_topk_wave<32>, grid, block, 0, s, a);
    } else if (P.k <= 64) {
        hipLaunchKernelGGL(k2_topk_reg, dim3(P.nrows), dim3(64), 0, s, a);
    } else {
        const size_t lds = (size_t)P.k * 8;
        if (P.k > k2_max_k()) fail("k2: only_topk/beam_size " + std::to_string(P.k) + " exceeds the device limit " + std::to_string(k2_max_k()));
        if (lds > 48 * 1024)   // per DEVICE attribute: set on every large launch
            XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k2_topk_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k2_topk_lds, dim3(P.nrows), dim3(64), lds, s, a);
    }
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Query-side helper for device-resident featurizers: [X_feat | X_emb] as one CSR, the query form of
// XR-Transformer's concat_model (TransformerMatcher.concat_features, pecos/xmc/xtransformer/matcher.py:864-890 followed by
// smat_util.hstack_csr): row r = the sparse features of row r, then dense_cols entries with column ids sparse_cols + j.
// One wavefront per row; the output row pointer is closed-form (in_ptr[r] + r * dense_cols).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
concat_csr_kernel(const uint64_t* __restrict__ in_ptr, const uint32_t* __restrict__ in_idx, const float* __restrict__ in_val,
                  const float* __restrict__ emb, uint32_t rows, uint32_t sparse_cols, uint32_t dense_cols, int normalize,
                  uint64_t* __restrict__ out_ptr, uint32_t* __restrict__ out_idx, float* __restrict__ out_val) {
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (r > rows) return;
    const uint64_t ob = in_ptr[r] + (uint64_t)r * dense_cols;
    if (lane == 0) out_ptr[r] = ob;
    if (r == rows) return;
    const uint64_t ib = in_ptr[r];
    const uint32_t n = (uint32_t)(in_ptr[r + 1] - ib);
    for (uint32_t t = lane; t < n; t += 64u) { out_idx[ob + t] = in_idx[ib + t]; out_val[ob + t] = in_val[ib + t]; }
    const float* __restrict__ e = emb + (uint64_t)r * dense_cols;
    // normalize != 0: sklearn.preprocessing.normalize(X_emb) (l2, rows; matcher.py:879-880) on the device: x / sqrt(sum x^2), rows of
    // norm < 10 eps left as they are.  The sum is a wavefront tree reduction, so values agree with numpy's to ~1e-7 relative, not bitwise.
    float scale = 1.0f;
    if (normalize) {
        float ss = 0.0f;
        for (uint32_t j = lane; j < dense_cols; j += 64u) ss += e[j] * e[j];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) ss += __shfl_xor(ss, d, 64);
        const float nrm = sqrtf(ss);
        scale = nrm < 10.0f * FLT_EPSILON ? 1.0f : nrm;
    }
    for (uint32_t j = lane; j < dense_cols; j += 64u) { out_idx[ob + n + j] = sparse_cols + j; out_val[ob + n + j] = normalize ? e[j] / scale : e[j]; }
}

void launch_concat_csr(const uint64_t* in_ptr, const uint32_t* in_idx, const float* in_val, const float* emb, uint32_t rows,
                       uint32_t sparse_cols, uint32_t dense_cols, int normalize, uint64_t* out_ptr, uint32_t* out_idx, float* out_val, hipStream_t s) {
    hipLaunchKernelGGL(concat_csr_kernel, dim3((rows + 1u + 3u) / 4u), dim3(256), 0, s, in_ptr, in_idx, in_val, emb, rows, sparse_cols,
                       dense_cols, normalize, out_ptr, out_idx, out_val);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// stats (not on the timed path): algorithmic bytes of the reference layout touched by a layer
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
stats_kernel(const float* __restrict__ chunk_alg, uint32_t nrows, uint32_t beam_in, int implicit_root,
             const uint32_t* __restrict__ p_idx, const uint32_t* __restrict__ p_cnt, uint32_t p_stride,
             const uint32_t* __restrict__ ncand, double* out2) {
    __shared__ double sb[256], sc[256];
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    double b = 0, c = 0;
    if (q < nrows) {
        const uint32_t cnt = implicit_root ? 1u : min(p_cnt[q], beam_in);
        for (uint32_t j = 0; j < cnt; ++j) b += chunk_alg[implicit_root ? 0u : p_idx[(size_t)q * p_stride + j]];
        c = ncand[q];
    }
    sb[threadIdx.x] = b; sc[threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sb[threadIdx.x] += sb[threadIdx.x + st]; sc[threadIdx.x] += sc[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&out2[0], sb[0]); atomicAdd(&out2[1], sc[0]); }
}

// matched work per (query, tile) item, counted by walking the item's query features against the tile's sorted row ids
// (independent of the row lookup structure the layer uses): out[0] items, [1] probes (= query features), [2] matched rows,
// [3] entries of the matched rows, [4] tile columns (scores written), [5] query features x tile columns (dense-format MACs)
__global__ void __launch_bounds__(256)
stats_items_kernel(LayerDev L, QueriesDev X, const ItemDesc* __restrict__ items, uint64_t n_slots, uint32_t row0, double* out6) {
    __shared__ double sh[6][256];
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (i < n_slots) {
        const ItemDesc it = items[i];
        if (it.tile != kNoTile) {
            const TileDesc td = L.tiles[it.tile];
            const uint32_t* __restrict__ ridx = L.row_idx + td.rowptr_base;
            const uint32_t* __restrict__ rext = L.row_ext + td.rowptr_base;
            uint32_t hits = 0, nx = 0; uint64_t ent = 0;
            if (X.dense) {
                nx = X.cols;
                const uint32_t nr = td.bias_slot != kNoBias ? td.nrows - 1 : td.nrows;
                hits = nr;
                for (uint32_t r = 0; r < nr; ++r) ent += (rext[r] >> 25) + 1u;
            } else {
                nx = it.x_len;
                uint32_t lo = 0;
                for (uint32_t t = 0; t < it.x_len; ++t) {
                    const uint32_t f = X.col_idx[it.x_begin + t];
                    uint32_t a = lo, b = td.nrows;
                    while (a < b) { const uint32_t mid = (a + b) >> 1; if (ridx[mid] < f) a = mid + 1; else b = mid; }
                    lo = a;
                    if (a < td.nrows && ridx[a] == f) { ++hits; ent += (rext[a] >> 25) + 1u; }
                }
            }
            v[0] = 1; v[1] = nx; v[2] = hits; v[3] = (double)ent; v[4] = td.ncols; v[5] = (double)nx * td.ncols;
        }
    }
    for (int k = 0; k < 6; ++k) sh[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) for (int k = 0; k < 6; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 6) atomicAdd(&out6[threadIdx.x], sh[threadIdx.x][0]);
}

void launch_stats(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev, const uint32_t* ncand, const void* items,
                  double* out8, hipStream_t s, uint64_t item_slots) {
    if (P.nrows == 0) return;
    if (ncand)   // (nullptr: only the items of a further item list of the same layer are added)
        hipLaunchKernelGGL(stats_kernel, dim3((P.nrows + 255) / 256), dim3(256), 0, s, L.chunk_alg_bytes, P.nrows,
                           P.beam_in, P.implicit_root, prev.idx, prev.cnt, prev.stride, ncand, out8);
    const uint64_t n_slots = item_slots ? item_slots : (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
    if (n_slots == 0) return;
    hipLaunchKernelGGL(stats_items_kernel, dim3((uint32_t)((n_slots + 255) / 256)), dim3(256), 0, s, L, X,
                       static_cast<const ItemDesc*>(items), n_slots, P.row0, out8 + 2);
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
