// K1T: the tile-format layer product on TILE ROWS held densely -- compute_sparse_predictions + chunk_ops<csr, bin_search>
// (inference.hpp:925-1007, 769-813) + bias + post-processor + combine (:506-518, 192-240, 1360-1384) for sparse queries.
//
// The tile format's own kernel (k1_kernel, xrl_kernels.hip) keeps a tile row as a list of {column, value} entries and applies a
// row to accumulators that live in LDS (the lanes of an item stride over the row's entries, each a read-modify-write of one LDS word).
// On a model the bound does not prune (Amazon-670K-hard: 4.9 M leaf items of ~47 matched rows x ~26 entries) that kernel is bound by
// vector-instruction issue: per matched row an extent lookup, a unit queue, a column select and an LDS read-add-write.
//
// Here every tile row that holds at least one weight is ALSO stored densely (LayerDev::wt): `wt_stride` = G * NR floats per row, column c
// of the tile at position c, kMissing (-0.0) where W has no entry, one all-missing pad row after every tile's rows.  A matched row is
// then ONE NR-dword load per lane (lane `lig` of the item's G lanes owns columns NR*lig .. NR*lig+NR-1: the G lanes read the row's
// G*NR*4 bytes contiguously) and NR multiply + NR add on accumulators held in REGISTERS: no extents, no unit queue, no LDS traffic but
// the hit queue.  With 288 GB of HBM the copy is cheap (Amazon-670K's leaf: 4.9 M rows x 384 B = 1.9 GB beside 1.0 GB of entries).
//
// Arithmetic: acc = fl32(acc + fl32(x * w)) per matched row in ascending feature order, separate multiply and add (no FMA), bias
// LAST (inference.hpp:806-811; HASH_CHUNKED: first, :716-722) -- the reference's order.  A missing cell multiplies by -0.0: for a
// finite x the product is a zero and leaves every reachable accumulator unchanged (accumulators start at +0.0 and can never become
// -0.0; same argument as the dense row format, xrl_model.h kMissing).  A drain that holds a NON-FINITE x (inf * -0.0 = NaN) runs
// the exact loop, which skips the cells whose bits are kMissing (an explicit -0.0 weight is stored as +0.0: identical for every x).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "xrl_device.h"
#include "xrl_kernels.h"
#include "xrl_items.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

namespace {

// block b of nb -> XCD b % 8 -> a contiguous eighth of the tile-sorted work (same mapping as k1_kernel)
__device__ __forceinline__ uint32_t xcd_remap_t(uint32_t b, uint32_t nb) {
    const uint32_t xcd = b & 7u, q = nb >> 3, r = nb & 7u;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (b >> 3);
}

struct K1TArgs {
    LayerDev L;
    QueriesDev X;
    const ItemDesc* items;
    const uint32_t* n_items;     // device count of (tile-sorted, all active) items, or nullptr: natural order
    float* cand;
    uint64_t n_slots;
    int pp_kind, pp_p, first_layer, bias_first;
    uint32_t n_vblocks;
    uint32_t* fb_out;            // pruning feedback: the launch's item count goes to this host-visible word
    uint32_t wt_bytes;           // BUF: size of the whole tile-row array (one buffer resource)
};

template <int NR> struct RowVec;
template <> struct RowVec<1> { float v[1]; };
template <> struct alignas(8) RowVec<2> { float v[2]; };
template <> struct RowVec<3> { float v[3]; };
template <> struct alignas(16) RowVec<4> { float v[4]; };

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NR> __device__ __forceinline__ RowVec<NR> row_load_buf(__amdgpu_buffer_rsrc_t rs, uint32_t voff);
template <> __device__ __forceinline__ RowVec<1> row_load_buf<1>(__amdgpu_buffer_rsrc_t rs, uint32_t voff) { return __builtin_bit_cast(RowVec<1>, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0)); }
template <> __device__ __forceinline__ RowVec<2> row_load_buf<2>(__amdgpu_buffer_rsrc_t rs, uint32_t voff) { return __builtin_bit_cast(RowVec<2>, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0)); }
template <> __device__ __forceinline__ RowVec<3> row_load_buf<3>(__amdgpu_buffer_rsrc_t rs, uint32_t voff) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)voff, 0, 0);
    RowVec<3> r; r.v[0] = __uint_as_float(t[0]); r.v[1] = __uint_as_float(t[1]); r.v[2] = __uint_as_float(t[2]); return r;
}
template <> __device__ __forceinline__ RowVec<4> row_load_buf<4>(__amdgpu_buffer_rsrc_t rs, uint32_t voff) { return __builtin_bit_cast(RowVec<4>, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0)); }

// G lanes per item (64 / G items per wavefront), NR columns per lane; LK = row lookup: 0 rank-bitmap {bits32, rank}, 2 {bits64, rank, -};
// BUF: the whole tile-row array is under 4 GiB and is addressed through ONE buffer resource with 32-bit byte offsets (a hit's queue word
// is its row's absolute offset: one vector add per row instead of a 64-bit address)
template <int G, int NR, int PPC, int LK, bool BUF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) k1t_kernel(K1TArgs a) {
    constexpr int W = 64 / G, H = 64, U = (G >= 16) ? 64 / G : 8, UNR = 8, STRIDE = G * NR;
    __shared__ uint2 hq_all[4 * W * (H + UNR)];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t vblock = blockIdx.x * 4u + wave;
    if (vblock >= a.n_vblocks) return;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, lig = lane % G;
    uint2* __restrict__ my_hq = hq_all + ((size_t)wave * W + grp) * (H + UNR);   // hits {x value, byte offset of the row in the tile's block}

    ItemDesc it = make_item(0u, kNoTile, 0u, 0.f, 0, 0u);
    if (a.n_items) {   // tile-sorted list: every XCD takes a contiguous run of tiles
        const uint32_t n = *a.n_items, nb = (n + W - 1) / W;
        if (a.fb_out && vblock == 0 && lane == 0) *a.fb_out = n;
        if (vblock >= nb) return;   // a compacted list (later stage of a pruned layer) usually fills a small part of the grid
        { const uint64_t slot = (uint64_t)xcd_remap_t(vblock, nb) * W + grp; if (slot < n) it = a.items[slot]; }
    } else {
        const uint64_t slot = (uint64_t)vblock * W + grp;
        if (slot < a.n_slots) it = a.items[slot];
    }
    const bool active = it.tile != kNoTile;
    TileDesc td{};
    uint64_t xe = 0, cur = 0, wbase = 0;
    if (active) {
        td = a.L.tiles[it.tile];
        wbase = a.L.wt_base[it.tile];
        cur = it.x_begin; xe = it.x_begin + it.x_len;
    }
    const char* __restrict__ wrow = reinterpret_cast<const char*>(a.L.wt + (BUF ? 0ull : wbase) + (uint32_t)(NR * lig));
    const uint32_t tbase = BUF ? (uint32_t)(wbase * 4ull) : 0u;          // BUF: queue words are offsets in the whole array
    const uint32_t lane_off = (uint32_t)(NR * lig * 4);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.L.wt), 0, (int)(BUF ? a.wt_bytes : 0u), 0x00020000);
    const uint32_t pad_off = tbase + td.nrows * (uint32_t)(STRIDE * 4);  // the tile's all-missing pad row
    float acc[NR];
    {
        const float* __restrict__ bp = a.L.bias_prod + td.col_begin;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const uint32_t c = (uint32_t)(NR * lig + k);
            acc[k] = (a.bias_first && a.L.has_bias && c < td.ncols) ? bp[c] : 0.0f;   // chunk_ops<csr, hash>: 0.0 + bias * w first
        }
    }
    const uint32_t* __restrict__ xi = a.X.col_idx;
    const float* __restrict__ xv = a.X.val;
    const BmWord* __restrict__ bm = a.L.bitmap + (LK != 0 ? 0ull : (uint64_t)(active ? it.tile : 0u) * a.L.nwords);
    const BmWord64* __restrict__ bm64 = a.L.bitmap64 + (LK != 2 ? 0ull : (uint64_t)(active ? it.tile : 0u) * a.L.nwords64);
    const unsigned long long below = (1ull << lig) - 1ull;
    const uint64_t xlast = xe > cur ? xe - 1 : 0;                       // a valid x index for clamped loads
    uint32_t nh = 0;                                                    // hits waiting in this item's queue
    bool nonfin = false;                                                // (wavefront-uniform) a queued hit carries a non-finite x

    auto drain = [&](auto exact_tag) {
        constexpr bool EX = decltype(exact_tag)::value;
        wave_sync_lds();
        uint32_t nh_max = nh;
#pragma unroll
        for (int d = G; d < 64; d <<= 1) nh_max = max(nh_max, (uint32_t)__shfl_xor((int)nh_max, d, 64));
        nh_max = __builtin_amdgcn_readfirstlane(nh_max);
        // every item's queue is read up to the longest one of the wavefront (rounded to the unroll): the rest multiplies +0.0 with the pad row
        for (uint32_t j = nh + (uint32_t)lig; j < ((nh_max + UNR - 1u) & ~(uint32_t)(UNR - 1)); j += G) my_hq[j] = make_uint2(0u, pad_off);
        wave_sync_lds();
#pragma unroll 1
        for (uint32_t h0 = 0; h0 < nh_max; h0 += UNR) {
            uint2 hv[UNR];
            RowVec<NR> w[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) hv[u] = my_hq[h0 + u];
#pragma unroll
            for (int u = 0; u < UNR; ++u) w[u] = BUF ? row_load_buf<NR>(wrs, hv[u].y + lane_off) : *reinterpret_cast<const RowVec<NR>*>(wrow + hv[u].y);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float x = __uint_as_float(hv[u].x);
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const float s = __fadd_rn(acc[k], __fmul_rn(x, w[u].v[k]));   // scalar * val, then add: no fma (inference.hpp:512-517)
                    acc[k] = (EX && __float_as_uint(w[u].v[k]) == kMissing) ? acc[k] : s;
                }
            }
        }
        nh = 0;
    };
    auto drain_any = [&]() {
        if (nonfin) drain(std::true_type{}); else drain(std::false_type{});
        nonfin = false;
    };

    uint32_t skip = 0;                 // u-slices of the current step already queued (after an overflow)
    while (__any(cur < xe)) {
        bool overflow = false;
        {
            // ---- load step: U*G consecutive features of the item
            uint32_t f[U]; float v[U];
            bool nf = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t t = cur + (uint64_t)(u * G + lig);
                const bool ok = t < xe;
                const uint64_t tc = ok ? t : xlast;          // clamped: the load itself is unconditional
                const uint32_t fi = xi[tc];
                const float vi = xv[tc];
                f[u] = (ok && fi < a.L.w_rows) ? fi : 0xFFFFFFFFu;
                v[u] = vi;
                nf = nf || (ok && (__float_as_uint(vi) & 0x7F800000u) == 0x7F800000u);
            }
            nonfin = nonfin || __any(nf);
            // ---- row lookup: is feature f a row of the tile, and which slot
            bool hit[U]; uint32_t off[U];
            if (LK == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool inr = f[u] != 0xFFFFFFFFu;
                    const BmWord wi = bm[inr ? (f[u] >> 5) : 0u];
                    const uint32_t b = f[u] & 31u;
                    hit[u] = inr && ((wi.bits >> b) & 1u);
                    off[u] = tbase + (wi.rank + (uint32_t)__popc(wi.bits & ((1u << b) - 1u))) * (uint32_t)(STRIDE * 4);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool inr = f[u] != 0xFFFFFFFFu;
                    const BmWord64 wi = bm64[inr ? (f[u] >> 6) : 0u];
                    const uint32_t b = f[u] & 63u;
                    const unsigned long long bits = ((unsigned long long)wi.hi << 32) | wi.lo;
                    hit[u] = inr && ((bits >> b) & 1ull);
                    off[u] = tbase + (wi.rank + (uint32_t)__popcll(bits & ((1ull << b) - 1ull))) * (uint32_t)(STRIDE * 4);
                }
            }
            // ---- queue the hits in feature order.  If an item's queue fills up the step is abandoned at slice `skip`, the queue is
            //      drained and the same step is re-loaded and resumed from that slice.
            uint32_t done = skip;
            bool stopped = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long m = __ballot(hit[u]);
                const unsigned long long gm = (G == 64) ? m : ((m >> (grp * G)) & ((1ull << G) - 1ull));
                const uint32_t cnt = (uint32_t)__popcll(gm);
                if ((uint32_t)u >= done && !stopped) {
                    if (nh + cnt <= (uint32_t)H) {
                        if (hit[u]) my_hq[nh + (uint32_t)__popcll(gm & below)] = make_uint2(__float_as_uint(v[u]), off[u]);
                        nh += cnt; done = u + 1;
                    } else {
                        stopped = true;
                    }
                }
            }
            if (done == (uint32_t)U) { if (cur < xe) cur += (uint64_t)U * G; skip = 0; }
            else { skip = done; overflow = true; }
        }
        if (__any(overflow)) drain_any();
    }
    drain_any();

    // ---- epilogue: bias (sparse X: LAST, inference.hpp:806-811), transform in fp64, combine with the parent's score, store
    if (!active) return;
    float* __restrict__ out = a.cand + it.out_off;
    const float* __restrict__ bp = a.L.bias_prod + td.col_begin;
    const bool add_bias = a.L.has_bias != 0 && !a.bias_first;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const uint32_t c = (uint32_t)(NR * lig + k);
        if (c < td.ncols) {
            float s = acc[k];
            if (add_bias) s = __fadd_rn(s, bp[c]);
            float v = pp_transform<PPC>(a.pp_kind, a.pp_p, s);
            if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
            out[c] = v;
        }
    }
}

// tile rows, dense: wt[(wt_base[t] / stride + r) * stride + col] = the weight of tile t's row r at tile column col (an explicit -0.0 as +0.0);
// the buffer is pre-filled with kMissing.  One workgroup per tile, 32 lanes per row.
__global__ void __launch_bounds__(256)
tile_rows_kernel(const TileDesc* __restrict__ tiles, const uint32_t* __restrict__ row_ext, const Entry* __restrict__ entries,
                 const uint64_t* __restrict__ wt_base, uint32_t stride, uint32_t* __restrict__ wt) {
    const TileDesc td = tiles[blockIdx.x];
    uint32_t* __restrict__ dst = wt + wt_base[blockIdx.x];
    const uint32_t* __restrict__ ext = row_ext + td.rowptr_base;
    const Entry* __restrict__ ent = entries + td.ent_base;
    const uint32_t sub = threadIdx.x >> 5, l = threadIdx.x & 31u;
    for (uint32_t r = sub; r < td.nrows; r += 8u) {
        const uint32_t e = ext[r], start = e & 0x1FFFFFFu, len = (e >> 25) + 1u;
        for (uint32_t i = l; i < len; i += 32u) {
            const Entry x = ent[start + i];
            const uint32_t b = __float_as_uint(x.val);
            dst[(uint64_t)r * stride + x.col] = b == kMissing ? 0u : b;
        }
    }
}

}  // namespace

void k1t_shape(uint32_t max_tile_cols, int& g, int& nr) {
    g = max_tile_cols <= 8 ? 8 : max_tile_cols <= 16 ? 16 : 32;
    nr = (int)((max_tile_cols + (uint32_t)g - 1) / (uint32_t)g);
    if (nr < 1) nr = 1;
}

void launch_tile_rows(const LayerDev& L, uint64_t total_floats, uint32_t* wt, hipStream_t s) {
    XRL_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(wt), (int)kMissing, total_floats, s));
    if (L.n_tiles) {
        hipLaunchKernelGGL(tile_rows_kernel, dim3(L.n_tiles), dim3(256), 0, s, L.tiles, L.row_ext, L.entries, L.wt_base, L.wt_stride, wt);
        XRL_LAUNCH_CHECK();
    }
}

bool k1t_serves(const LayerDev& L, const QueriesDev& X) { return L.wt != nullptr && !X.dense && (L.bitmap || L.bitmap64) && !L.bucket; }

void launch_k1t(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items, const uint32_t* n_items, float* cand, hipStream_t s) {
    if (P.nrows == 0) return;
    K1TArgs a;
    a.L = L; a.X = X; a.items = static_cast<const ItemDesc*>(items); a.n_items = n_items; a.cand = cand;
    a.n_slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.bias_first = P.bias_first;
    a.fb_out = (n_items && P.fb_host && P.layer >= 0 && P.layer < 16) ? P.fb_host + 32 + P.layer : nullptr;
    int g, nr;
    k1t_shape(L.max_tile_cols, g, nr);
    if ((uint32_t)(g * nr) != L.wt_stride || nr > 4) fail("k1t: the layer's tile rows were laid out for another shape");
    const uint64_t vblocks = (a.n_slots + (uint64_t)(64 / g) - 1) / (uint64_t)(64 / g);
    if (vblocks > 0x7FFFFFFFull) fail("k1t: grid too large; lower max_batch_rows");
    a.n_vblocks = (uint32_t)vblocks;
    const dim3 grid((uint32_t)((vblocks + 3) / 4)), block(256);
    const int ppc = pp_class(P.pp);
    const int lk = L.bitmap64 ? 2 : 0;
    const bool buf = L.wt_bytes != 0 && L.wt_bytes < 0xFFFFFF00ull;   // (gfx9 range-checks voffset against num_records: the array's bytes)
    a.wt_bytes = buf ? (uint32_t)L.wt_bytes : 0u;
#define XRL_K1T_B(GG, NN, BB) do { \
        if (ppc) { if (lk) hipLaunchKernelGGL((k1t_kernel<GG, NN, 1, 2, BB>), grid, block, 0, s, a); else hipLaunchKernelGGL((k1t_kernel<GG, NN, 1, 0, BB>), grid, block, 0, s, a); } \
        else     { if (lk) hipLaunchKernelGGL((k1t_kernel<GG, NN, 0, 2, BB>), grid, block, 0, s, a); else hipLaunchKernelGGL((k1t_kernel<GG, NN, 0, 0, BB>), grid, block, 0, s, a); } } while (0)
#define XRL_K1T(GG, NN) do { if (buf) XRL_K1T_B(GG, NN, true); else XRL_K1T_B(GG, NN, false); } while (0)
    if (g == 8) XRL_K1T(8, 1);
    else if (g == 16) XRL_K1T(16, 1);
    else if (nr == 1) XRL_K1T(32, 1);
    else if (nr == 2) XRL_K1T(32, 2);
    else if (nr == 3) XRL_K1T(32, 3);
    else XRL_K1T(32, 4);
#undef XRL_K1T
#undef XRL_K1T_B
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
