// Hand-written HIP kernels for gfx950 (CDNA4, 64-wide wavefronts) -- the XR-Linear beam-search
// hot path of pecos/core/xmc/inference.hpp, restated for the MI355X execution model.
//
//   K0 k0_prolongate   prolongate_predictions               inference.hpp:1155-1219
//   K1 k1_kernel       compute_sparse_predictions + chunk_ops + transform + combine
//                                                           inference.hpp:925-1007, 769-839, 506-518,
//                                                           1360-1384, PostProcessor :192-240
//   K2 k2_topk_*       sorted_csr + reorder_prediction      inference.hpp:1223-1298, 1919-1923
//   K3 k3_kernel       sparse_inner_products                matrix.hpp:1049-1060, 836-877
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
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "xrl_kernels.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

__device__ __forceinline__ void wave_sync_lds() {
    // LDS operations of one wavefront execute in program order; this only stops the compiler
    // from moving LDS accesses across the point (cross-lane RAW through LDS inside a wave).
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// post-processor (inference.hpp:192-240).  The reference lambdas take `const float&`:
//   sigmoid / log-sigmoid evaluate std::exp(float) (= expf) and continue in double;
//   l{p}-hinge keeps z in a FLOAT, then pow/exp in double.  Results are cast to float (:1369).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ref_expf(float x) { return (float)exp((double)x); }

// z^p for the integer p of l{p}-hinge.  z is a float, so z*z is EXACT in double (48-bit product) and
// z^3 = (z*z)*z, z^4 = (z*z)*(z*z) carry a single rounding: they are the correctly rounded powers,
// which is what glibc's pow returns (its error bound is < 1 ULP, correctly rounded in practice).
// Larger p fall back to pow().
__device__ __forceinline__ double hinge_pow(float zf, int p) {
    const double z = (double)zf;
    switch (p) {
    case 0: return 1.0;
    case 1: return z;
    case 2: return z * z;
    case 3: return (z * z) * z;
    case 4: { const double t = z * z; return t * t; }
    default: return pow(z, (double)p);
    }
}

__device__ __forceinline__ float pp_transform(int kind, int p, float v) {
    switch (kind) {
    case PP_SIGMOID: return (float)(1.0 / (1.0 + (double)ref_expf(-v)));
    case PP_LOG_SIGMOID: return (float)(-log(1.0 + (double)ref_expf(-v)));
    case PP_LP_HINGE: {
        const float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)exp(-hinge_pow(z, p));
    }
    case PP_LOG_LP_HINGE: {
        const float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)(-hinge_pow(z, p));
    }
    default: return v;
    }
}

__device__ __forceinline__ float pp_combine(int kind, float x, float parent) {
    switch (kind) {
    case PP_SIGMOID:
    case PP_LP_HINGE: return __fmul_rn(x, parent);        // std::multiplies<float>
    case PP_LOG_SIGMOID:
    case PP_LOG_LP_HINGE: return __fadd_rn(x, parent);    // std::plus<float>
    default: return x;
    }
}

// ---------------------------------------------------------------------------------------------
// K0: one thread per query
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k0_prolongate(const uint32_t* __restrict__ chunk_col, uint32_t nrows, uint32_t beam_in, int implicit_root,
              const uint32_t* __restrict__ p_idx, const uint32_t* __restrict__ p_cnt, uint32_t p_stride,
              uint32_t* __restrict__ cand_off, uint32_t* __restrict__ ncand) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= nrows) return;
    const uint32_t cnt = implicit_root ? 1u : min(p_cnt[q], beam_in);
    uint32_t off = 0;
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t parent = implicit_root ? 0u : p_idx[(size_t)q * p_stride + j];
        cand_off[(size_t)q * beam_in + j] = off;
        off += chunk_col[parent + 1] - chunk_col[parent];
    }
    ncand[q] = off;
}

void launch_k0_prolongate(const LayerDev& L, const LayerPlan& P, BeamDev prev, uint32_t* cand_off,
                          uint32_t* ncand, hipStream_t s) {
    if (P.nrows == 0) return;
    hipLaunchKernelGGL(k0_prolongate, dim3((P.nrows + 255) / 256), dim3(256), 0, s, L.chunk_col, P.nrows,
                       P.beam_in, P.implicit_root, prev.idx, prev.cnt, prev.stride, cand_off, ncand);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// item ordering: counting sort of the layer's (query, beam slot, tile) items by tile id, so that
// the wavefronts working on one tile run back to back on ONE XCD and find the tile's bitmap /
// row table / entries in that XCD's L2 (the reference sorts by chunk for the same reason,
// inference.hpp:991-993).  Order inside a tile is arbitrary (atomics); results do not depend on it.
// ---------------------------------------------------------------------------------------------
struct ItemArgs {
    const uint32_t* ptile;
    const uint32_t* p_idx; const uint32_t* p_cnt; uint32_t p_stride;
    uint32_t nrows, beam_in, TT;
    int implicit_root;
};

__device__ __forceinline__ bool decode_slot(const ItemArgs& a, uint64_t slot, uint32_t& q, uint32_t& j, uint32_t& tt,
                                            uint32_t& tile) {
    tt = (uint32_t)(slot % a.TT);
    const uint64_t r1 = slot / a.TT;
    j = (uint32_t)(r1 % a.beam_in);
    const uint64_t qq = r1 / a.beam_in;
    if (qq >= a.nrows) return false;
    q = (uint32_t)qq;
    uint32_t parent = 0;
    if (!a.implicit_root) {
        if (j >= min(a.p_cnt[q], a.beam_in)) return false;
        parent = a.p_idx[(uint64_t)q * a.p_stride + j];
    } else if (j != 0) {
        return false;
    }
    const uint32_t t0 = a.ptile[parent], t1 = a.ptile[parent + 1];
    if (tt >= t1 - t0) return false;
    tile = t0 + tt;
    return true;
}

__global__ void __launch_bounds__(256) sort_count_kernel(ItemArgs a, uint64_t slots, uint32_t* __restrict__ count) {
    const uint64_t slot = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (slot >= slots) return;
    uint32_t q, j, tt, tile;
    if (decode_slot(a, slot, q, j, tt, tile)) atomicAdd(&count[tile], 1u);
}

// single block: exclusive scan of count[0..n) in place -> start offsets; count[n] = total; fill[] = 0
__global__ void __launch_bounds__(1024) sort_scan_kernel(uint32_t* __restrict__ count, uint32_t* __restrict__ fill, uint32_t n) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? count[i] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) { count[i] = carry + part[threadIdx.x] - v; fill[i] = 0u; }
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) count[n] = carry;
}

__global__ void __launch_bounds__(256)
sort_scatter_kernel(ItemArgs a, uint64_t slots, const uint32_t* __restrict__ start, uint32_t* __restrict__ fill,
                    uint2* __restrict__ items) {
    const uint64_t slot = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (slot >= slots) return;
    uint32_t q, j, tt, tile;
    if (decode_slot(a, slot, q, j, tt, tile)) {
        const uint32_t pos = start[tile] + atomicAdd(&fill[tile], 1u);
        items[pos] = make_uint2(q, j | (tt << 16));
    }
}

static ItemArgs make_item_args(const LayerDev& L, const LayerPlan& P, const BeamDev& prev) {
    ItemArgs a;
    a.ptile = L.ptile; a.p_idx = prev.idx; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
    a.nrows = P.nrows; a.beam_in = P.beam_in; a.TT = L.max_tiles_per_parent; a.implicit_root = P.implicit_root;
    return a;
}

void launch_sort_items(const LayerDev& L, const LayerPlan& P, BeamDev prev, uint32_t* count /*[n_tiles+1]*/,
                       uint32_t* fill /*[n_tiles]*/, uint2* items, hipStream_t s) {
    if (P.nrows == 0) return;
    if (P.beam_in > 0xFFFFu || L.max_tiles_per_parent > 0xFFFFu) fail("sort_items: beam or tiles-per-parent exceed 16 bits");
    const ItemArgs a = make_item_args(L, P, prev);
    const uint64_t slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
    const uint64_t blocks = (slots + 255) / 256;
    if (blocks > 0x7FFFFFFFull) fail("sort_items: grid too large; lower max_batch_rows");
    XRL_HIP(hipMemsetAsync(count, 0, ((size_t)L.n_tiles + 1) * 4, s));
    hipLaunchKernelGGL(sort_count_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a, slots, count);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, s, count, fill, L.n_tiles);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a, slots, count, fill, items);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------
struct K1Args {
    LayerDev L;
    QueriesDev X;
    BeamDev prev;
    const uint32_t* cand_off;
    float* cand;
    const uint2* items;          // tile-sorted item list, or nullptr (natural order)
    const uint32_t* n_items;     // device count of valid items (sorted mode)
    uint32_t row0, nrows, beam_in, cand_stride, acc_stride;
    int pp_kind, pp_p, first_layer, implicit_root;
};

// XCD-aware block remap (blocks b, b+8, b+16, ... run on one XCD): give every XCD a CONTIGUOUS
// range of the (tile-sorted) work so a tile's data is fetched into one L2 only.  Bijective.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb) {
    const uint32_t xcd = b & 7u, q = nb >> 3, r = nb & 7u;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (b >> 3);
}

struct ItemCtx {
    bool active;
    uint32_t q, j, parent, tile;
    float pscore;
    TileDesc td;
};

template <int W>
__device__ __forceinline__ ItemCtx k1_item(const K1Args& a, int grp) {
    ItemCtx it{};
    uint64_t slot = (uint64_t)blockIdx.x * W + grp;
    ItemArgs ia;
    ia.ptile = a.L.ptile; ia.p_idx = a.prev.idx; ia.p_cnt = a.prev.cnt; ia.p_stride = a.prev.stride;
    ia.nrows = a.nrows; ia.beam_in = a.beam_in; ia.TT = a.L.max_tiles_per_parent; ia.implicit_root = a.implicit_root;
    uint32_t tt = 0;
    if (a.items) {
        // tile-sorted list: blocks beyond the valid range idle; the valid blocks are remapped so that
        // every XCD owns a contiguous run of tiles
        const uint32_t n_items = *a.n_items;
        const uint32_t nb = (n_items + W - 1) / W;
        slot = blockIdx.x < nb ? (uint64_t)xcd_remap(blockIdx.x, nb) * W + grp : (uint64_t)n_items;
        if (slot < n_items) {
            const uint2 e = a.items[slot];
            it.q = e.x; it.j = e.y & 0xFFFFu; tt = e.y >> 16;
            it.parent = a.implicit_root ? 0u : a.prev.idx[(uint64_t)it.q * a.prev.stride + it.j];
            it.tile = a.L.ptile[it.parent] + tt;
            it.active = true;
        }
    } else {
        it.active = decode_slot(ia, slot, it.q, it.j, tt, it.tile);
        if (it.active) it.parent = a.implicit_root ? 0u : a.prev.idx[(uint64_t)it.q * a.prev.stride + it.j];
    }
    it.pscore = 1.0f;
    if (it.active) {
        if (!a.implicit_root) it.pscore = a.prev.val[(uint64_t)it.q * a.prev.stride + it.j];
        it.td = a.L.tiles[it.tile];
    }
    return it;
}

template <int G, class ACC>
__device__ __forceinline__ void k1_epilogue(const K1Args& a, const ItemCtx& it, int lig, ACC&& acc_at) {
    // transform (fp64) + combine with the parent's score, write the child block
    if (!it.active) return;
    float* __restrict__ out = a.cand + (uint64_t)it.q * a.cand_stride + a.cand_off[(uint64_t)it.q * a.beam_in + it.j] +
                              (it.td.col_begin - a.L.chunk_col[it.parent]);
    for (uint32_t c = lig; c < it.td.ncols; c += G) {
        float v = pp_transform(a.pp_kind, a.pp_p, acc_at(c));
        if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
        out[c] = v;
    }
}

// ---- sparse queries: chunk_ops<csr, bin_search>, inference.hpp:769-813 --------------------------
template <int G, int U> struct K1Cfg {
    static constexpr int W = 64 / G;                                  // items per wavefront
    static constexpr int H = (2 * U * G > 16) ? 2 * U * G : 16;       // hit-FIFO depth per item
    static constexpr size_t lds_bytes(uint32_t acc_stride) { return (size_t)W * acc_stride * 4 + (size_t)W * H * (4 + 4 + 4 + 16); }
};

template <int G, int U>
__global__ void __launch_bounds__(64) k1_sparse_kernel(K1Args a) {
    constexpr int W = K1Cfg<G, U>::W, H = K1Cfg<G, U>::H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* fe = reinterpret_cast<uint4*>(smem);                        // first two entries of the hit row
    float* acc = reinterpret_cast<float*>(fe + W * H);
    float* fv = acc + (size_t)W * a.acc_stride;                        // x value of the hit
    uint32_t* fa = reinterpret_cast<uint32_t*>(fv + W * H);            // row slot, then row start
    uint32_t* fl = fa + W * H;                                         // row length

    const int lane = threadIdx.x;
    const int grp = lane / G, lig = lane % G;
    const ItemCtx it = k1_item<W>(a, grp);
    const uint32_t ncols = it.active ? it.td.ncols : 0u;
    const uint32_t* __restrict__ rp = a.L.row_ptr + it.td.rowptr_base;
    const Entry* __restrict__ ent = a.L.entries + it.td.ent_base;
    float* __restrict__ my_acc = acc + (size_t)grp * a.acc_stride;
    const uint32_t fbase = (uint32_t)grp * H;

    for (uint32_t c = lig; c < ncols; c += G) my_acc[c] = 0.0f;        // std::fill(..., 0.0), inference.hpp:964
    wave_sync_lds();

    uint64_t xe = 0, cur = 0;
    if (it.active) { const uint64_t qg = (uint64_t)a.row0 + it.q; cur = a.X.row_ptr[qg]; xe = a.X.row_ptr[qg + 1]; }
    const uint32_t* __restrict__ xi = a.X.col_idx;
    const float* __restrict__ xv = a.X.val;
    const BmWord* __restrict__ bm = a.L.bitmap + (uint64_t)it.tile * a.L.nwords;
    const unsigned long long below = (1ull << lig) - 1ull;

    do {
        // ---- fill: U*G consecutive features of the item per step; all x loads, then all bitmap
        //      probes are issued together; hits are compacted IN FEATURE ORDER into the FIFO
        uint32_t nh = 0;
        while (__any(cur < xe && nh + U * G <= (uint32_t)H)) {
            const bool can = (cur < xe) && (nh + U * G <= (uint32_t)H);
            uint32_t f[U]; float v[U]; bool ok[U]; BmWord w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t t = cur + (uint64_t)(u * G + lig);
                ok[u] = can && t < xe;
                f[u] = ok[u] ? xi[t] : 0xFFFFFFFFu;
                v[u] = ok[u] ? xv[t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = ok[u] && f[u] < a.L.w_rows;
                w[u] = ok[u] ? bm[f[u] >> 5] : BmWord{0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t b = f[u] & 31u;
                const bool hit = ok[u] && ((w[u].bits >> b) & 1u);
                const unsigned long long m = __ballot(hit);
                const unsigned long long gm = (G == 64) ? m : ((m >> (grp * G)) & ((1ull << G) - 1ull));
                if (hit) {
                    const uint32_t pos = fbase + nh + (uint32_t)__popcll(gm & below);
                    fv[pos] = v[u];
                    fa[pos] = w[u].rank + __popc(w[u].bits & ((1u << b) - 1u));
                }
                nh += (uint32_t)__popcll(gm);
            }
            if (can) cur += U * G;
        }
        wave_sync_lds();
        // ---- D1/D2: one lane per hit fetches the row extent and its first two entries (all hits of
        //      all items in flight at once -> two dependent memory latencies per batch, not per hit)
        for (uint32_t h = lig; h < nh; h += G) {
            const uint32_t s = fa[fbase + h];
            const uint32_t rs = rp[s], len = rp[s + 1] - rs;
            uint4 e2 = make_uint4(0u, 0u, 0u, 0u);
            if (len >= 2) { const Entry e0 = ent[rs], e1 = ent[rs + 1]; e2 = make_uint4(e0.col, __float_as_uint(e0.val), e1.col, __float_as_uint(e1.val)); }
            else if (len == 1) { const Entry e0 = ent[rs]; e2.x = e0.col; e2.y = __float_as_uint(e0.val); }
            fa[fbase + h] = rs; fl[fbase + h] = len; fe[fbase + h] = e2;
        }
        wave_sync_lds();
        // ---- D3: rows in feature order; inside a row the lanes take distinct columns
        for (uint32_t h = 0; __any(h < nh); ++h) {
            if (h < nh) {
                const float v = fv[fbase + h];
                const uint32_t len = fl[fbase + h];
                const uint32_t rs = fa[fbase + h];
                const uint4 e2 = fe[fbase + h];
                for (uint32_t e = lig; e < len; e += G) {
                    uint32_t col; float wv;
                    if (e == 0) { col = e2.x; wv = __uint_as_float(e2.y); }
                    else if (e == 1) { col = e2.z; wv = __uint_as_float(e2.w); }
                    else { const Entry en = ent[rs + e]; col = en.col; wv = en.val; }
                    // out[col] += scalar * val (inference.hpp:512-517): mul then add, no fma
                    my_acc[col] = __fadd_rn(my_acc[col], __fmul_rn(v, wv));
                }
            }
            wave_sync_lds();
        }
    } while (__any(cur < xe));
    // bias LAST (inference.hpp:806-811)
    if (it.active && it.td.bias_slot != kNoBias) {
        for (uint32_t e = rp[it.td.bias_slot] + lig; e < rp[it.td.bias_slot + 1]; e += G) {
            const Entry en = ent[e];
            my_acc[en.col] = __fadd_rn(my_acc[en.col], __fmul_rn(a.L.bias, en.val));
        }
    }
    wave_sync_lds();
    k1_epilogue<G>(a, it, lig, [&](uint32_t c) { return my_acc[c]; });
}

// ---- dense queries: chunk_ops<drm, bin_search>, inference.hpp:815-839 (bias FIRST, every row) -----
template <int G>
__global__ void __launch_bounds__(64) k1_dense_kernel(K1Args a) {
    constexpr int W = 64 / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* acc = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x;
    const int grp = lane / G, lig = lane % G;
    const ItemCtx it = k1_item<W>(a, grp);
    const uint32_t ncols = it.active ? it.td.ncols : 0u;
    const uint32_t* __restrict__ rp = a.L.row_ptr + it.td.rowptr_base;
    const Entry* __restrict__ ent = a.L.entries + it.td.ent_base;
    float* __restrict__ my_acc = acc + (size_t)grp * a.acc_stride;
    for (uint32_t c = lig; c < ncols; c += G) my_acc[c] = 0.0f;
    wave_sync_lds();
    const float* __restrict__ xd = a.X.val + ((uint64_t)a.row0 + it.q) * a.X.cols;
    const uint32_t* __restrict__ ridx = a.L.row_idx + (it.td.rowptr_base - it.tile);
    uint32_t nr = it.active ? it.td.nrows : 0u;
    if (it.active && it.td.bias_slot != kNoBias) {
        for (uint32_t e = rp[it.td.bias_slot] + lig; e < rp[it.td.bias_slot + 1]; e += G) {
            const Entry en = ent[e];
            my_acc[en.col] = __fadd_rn(my_acc[en.col], __fmul_rn(a.L.bias, en.val));
        }
        nr -= 1;
    }
    wave_sync_lds();
    for (uint32_t s = 0; __any(s < nr); ++s) {
        if (s < nr) {
            const uint32_t f = ridx[s];
            if (f < a.X.cols) {
                const float v = xd[f];
                for (uint32_t e = rp[s] + lig; e < rp[s + 1]; e += G) {
                    const Entry en = ent[e];
                    my_acc[en.col] = __fadd_rn(my_acc[en.col], __fmul_rn(v, en.val));
                }
            }
        }
        wave_sync_lds();
    }
    k1_epilogue<G>(a, it, lig, [&](uint32_t c) { return my_acc[c]; });
}

template <class KERNEL>
static void launch_k1_any(KERNEL kernel, const K1Args& a, uint64_t slots, int W, size_t lds, hipStream_t s) {
    if (lds > 160 * 1024) fail("k1: LDS request exceeds 160 KiB");
    if (lds > 48 * 1024)
        XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t blocks = (slots + W - 1) / W;
    if (blocks > 0x7FFFFFFFull) fail("k1: grid too large; lower max_batch_rows");
    hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(64), lds, s, a);
    XRL_LAUNCH_CHECK();
}

int k1_auto_group(const LayerDev& L, const Layer& host, int dense) {
    // lanes per item: enough to cover a typical tile row, and at least 8 so that the x reads of an
    // item coalesce into >= 32-byte segments and a wavefront's LDS stays small
    const double row_len = host.total_rows ? (double)host.nnz / (double)host.total_rows : 1.0;
    int g = dense ? 16 : 8;
    while (g < 64 && g < row_len * 0.75) g <<= 1;
    (void)L;
    return g;
}

void launch_k1(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev,
               const uint32_t* cand_off, float* cand, const uint2* items, const uint32_t* n_items, int group,
               hipStream_t s) {
    if (P.nrows == 0) return;
    K1Args a;
    a.L = L; a.X = X; a.prev = prev; a.cand_off = cand_off; a.cand = cand; a.items = items; a.n_items = n_items;
    a.row0 = P.row0; a.nrows = P.nrows; a.beam_in = P.beam_in; a.cand_stride = P.cand_stride;
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.implicit_root = P.implicit_root;
    a.acc_stride = L.max_tile_cols | 1u;
    const uint64_t slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
    if (X.dense) {
#define XRL_K1D(GG) case GG: launch_k1_any(&k1_dense_kernel<GG>, a, slots, 64 / GG, (size_t)(64 / GG) * a.acc_stride * 4, s); break;
        switch (group) { XRL_K1D(1) XRL_K1D(2) XRL_K1D(4) XRL_K1D(8) XRL_K1D(16) XRL_K1D(32) XRL_K1D(64)
        default: fail("k1: lanes-per-item must be a power of two in [1, 64]"); }
#undef XRL_K1D
    } else {
#define XRL_K1S(GG, UU) case GG: launch_k1_any(&k1_sparse_kernel<GG, UU>, a, slots, 64 / GG, K1Cfg<GG, UU>::lds_bytes(a.acc_stride), s); break;
        switch (group) { XRL_K1S(1, 8) XRL_K1S(2, 8) XRL_K1S(4, 4) XRL_K1S(8, 2) XRL_K1S(16, 2) XRL_K1S(32, 1) XRL_K1S(64, 1)
        default: fail("k1: lanes-per-item must be a power of two in [1, 64]"); }
#undef XRL_K1S
    }
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
            const int r = __popcll(__ballot((uint32_t)lane < m && lv >= vv));
            const float uv = __shfl_up(lv, 1);
            const uint32_t up = __shfl_up(lp, 1);
            if (lane > r) { lv = uv; lp = up; }
            else if (lane == r) { lv = vv; lp = base + l; }
            if (m < k) ++m;
            th = (m == k) ? __shfl(lv, (int)k - 1) : -INFINITY;
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

size_t k2_max_k() { return (160 * 1024) / 8; }

void launch_k2_topk(const LayerDev& L, const LayerPlan& P, BeamDev prev, const uint32_t* cand_off,
                    const uint32_t* ncand, const float* cand, uint32_t* out_idx, float* out_val,
                    uint32_t* out_cnt, uint32_t out_stride, hipStream_t s) {
    if (P.nrows == 0) return;
    K2Args a;
    a.chunk_col = L.chunk_col; a.perm_inv = L.perm_inv;
    a.p_idx = prev.idx; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
    a.cand_off = cand_off; a.ncand = ncand; a.cand = cand;
    a.out_idx = out_idx; a.out_val = out_val; a.out_cnt = out_cnt;
    a.nrows = P.nrows; a.beam_in = P.beam_in; a.cand_stride = P.cand_stride; a.k = P.k; a.out_stride = out_stride;
    a.implicit_root = P.implicit_root;
    if (P.k == 0) fail("k2: only_topk / beam_size resolved to 0");
    if (P.k <= 64) {
        hipLaunchKernelGGL(k2_topk_reg, dim3(P.nrows), dim3(64), 0, s, a);
    } else {
        const size_t lds = (size_t)P.k * 8;
        if (P.k > k2_max_k()) fail("k2: only_topk/beam_size " + std::to_string(P.k) + " exceeds the device limit " + std::to_string(k2_max_k()));
        static thread_local size_t configured = 0;
        if (lds > 48 * 1024 && lds > configured) {
            XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k2_topk_lds),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = lds;
        }
        hipLaunchKernelGGL(k2_topk_lds, dim3(P.nrows), dim3(64), lds, s, a);
    }
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

void launch_stats(const LayerDev& L, const LayerPlan& P, BeamDev prev, const uint32_t* ncand, double* out2,
                  hipStream_t s) {
    if (P.nrows == 0) return;
    hipLaunchKernelGGL(stats_kernel, dim3((P.nrows + 255) / 256), dim3(256), 0, s, L.chunk_alg_bytes, P.nrows,
                       P.beam_in, P.implicit_root, prev.idx, prev.cnt, prev.stride, ncand, out2);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// K3: sparse_inner_products, one thread per (row, col) pair, sequential fp32 accumulation in
// ascending index order (do_dot_product overloads, matrix.hpp:836-877)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k3_kernel(const uint64_t* __restrict__ x_ptr, const uint32_t* __restrict__ x_idx, const float* __restrict__ x_val,
          int x_dense, const uint64_t* __restrict__ w_ptr, const uint32_t* __restrict__ w_idx,
          const float* __restrict__ w_val, int w_dense, uint32_t dim, uint64_t len,
          const uint32_t* __restrict__ rows, const uint32_t* __restrict__ cols, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= len) return;
    const uint32_t r = rows[i], c = cols[i];
    float ret = 0.f;
    if (x_dense && w_dense) {
        const float* x = x_val + (uint64_t)r * dim; const float* w = w_val + (uint64_t)c * dim;
        for (uint32_t d = 0; d < dim; ++d) ret = __fadd_rn(ret, __fmul_rn(x[d], w[d]));
    } else if (x_dense) {
        const float* x = x_val + (uint64_t)r * dim;
        for (uint64_t s = w_ptr[c]; s < w_ptr[c + 1]; ++s) ret = __fadd_rn(ret, __fmul_rn(x[w_idx[s]], w_val[s]));
    } else if (w_dense) {
        const float* w = w_val + (uint64_t)c * dim;
        for (uint64_t s = x_ptr[r]; s < x_ptr[r + 1]; ++s) ret = __fadd_rn(ret, __fmul_rn(w[x_idx[s]], x_val[s]));
    } else {
        uint64_t s = x_ptr[r], se = x_ptr[r + 1], t = w_ptr[c], te = w_ptr[c + 1];
        while (s < se && t < te) {
            const uint32_t a = x_idx[s], b = w_idx[t];
            if (a == b) { ret = __fadd_rn(ret, __fmul_rn(x_val[s], w_val[t])); ++s; ++t; }
            else if (a < b) ++s;
            else ++t;
        }
    }
    out[i] = ret;
}

void launch_k3_inner_products(const uint64_t* x_ptr, const uint32_t* x_idx, const float* x_val, int x_dense,
                              const uint64_t* w_ptr, const uint32_t* w_idx, const float* w_val, int w_dense,
                              uint32_t dim, uint64_t len, const uint32_t* rows, const uint32_t* cols,
                              float* out, hipStream_t s) {
    if (len == 0) return;
    hipLaunchKernelGGL(k3_kernel, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, s, x_ptr, x_idx, x_val,
                       x_dense, w_ptr, w_idx, w_val, w_dense, dim, len, rows, cols, out);
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
