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

__device__ __forceinline__ float pp_transform(int kind, int p, float v) {
    switch (kind) {
    case PP_SIGMOID: return (float)(1.0 / (1.0 + (double)ref_expf(-v)));
    case PP_LOG_SIGMOID: return (float)(-log(1.0 + (double)ref_expf(-v)));
    case PP_LP_HINGE: {
        const float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)exp(-pow((double)z, (double)p));
    }
    case PP_LOG_LP_HINGE: {
        const float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)(-pow((double)z, (double)p));
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
// K1
// ---------------------------------------------------------------------------------------------
struct K1Args {
    LayerDev L;
    QueriesDev X;
    BeamDev prev;
    const uint32_t* cand_off;
    float* cand;
    uint32_t row0, nrows, beam_in, cand_stride, acc_stride;
    int pp_kind, pp_p, first_layer, implicit_root;
};

template <int G> struct K1Cfg {
    static constexpr int W = 64 / G;                     // items per wavefront
    static constexpr int H = (G < 4) ? 8 : 2 * G;        // FIFO depth per item (>= 2G)
};

template <int G, bool DENSE>
__global__ void __launch_bounds__(64) k1_kernel(K1Args a) {
    constexpr int W = K1Cfg<G>::W, H = K1Cfg<G>::H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* acc = reinterpret_cast<float*>(smem);
    const uint32_t acc_floats = (G == 1) ? a.acc_stride * 64u : a.acc_stride * (uint32_t)W;
    float* fv = acc + acc_floats;
    uint32_t* fs = reinterpret_cast<uint32_t*>(fv + W * H);

    const int lane = threadIdx.x;
    const int grp = lane / G, lig = lane % G;
    // accumulator / fifo addressing: G==1 keeps [c][lane] (bank = lane, conflict-free for any c)
    auto ACC = [&](uint32_t c) -> float& { return (G == 1) ? acc[c * 64u + lane] : acc[grp * a.acc_stride + c]; };
    auto FIDX = [&](uint32_t pos) -> uint32_t { return (G == 1) ? pos * 64u + lane : (uint32_t)grp * H + pos; };

    // ---- item decode: (query, beam slot, tile-in-parent), tile fastest
    const uint32_t TT = a.L.max_tiles_per_parent;
    const uint64_t slot = (uint64_t)blockIdx.x * W + grp;
    const uint32_t tt = (uint32_t)(slot % TT);
    const uint64_t r1 = slot / TT;
    const uint32_t j = (uint32_t)(r1 % a.beam_in);
    const uint64_t q = r1 / a.beam_in;
    bool active = q < a.nrows;
    uint32_t parent = 0;
    float pscore = 1.0f;
    if (active && !a.implicit_root) {
        active = j < min(a.prev.cnt[q], a.beam_in);
        if (active) {
            parent = a.prev.idx[q * a.prev.stride + j];
            pscore = a.prev.val[q * a.prev.stride + j];
        }
    } else if (active) {
        active = (j == 0);
    }
    TileDesc td{};
    uint32_t tile = 0;
    if (active) {
        const uint32_t t0 = a.L.ptile[parent], t1 = a.L.ptile[parent + 1];
        active = tt < t1 - t0;
        if (active) { tile = t0 + tt; td = a.L.tiles[tile]; }
    }
    const uint32_t ncols = active ? td.ncols : 0u;
    const uint32_t* __restrict__ rp = a.L.row_ptr + td.rowptr_base;
    const Entry* __restrict__ ent = a.L.entries + td.ent_base;

    for (uint32_t c = lig; c < ncols; c += G) ACC(c) = 0.0f;   // std::fill(..., 0.0), inference.hpp:964
    wave_sync_lds();

    const uint64_t qg = (uint64_t)a.row0 + q;
    if (DENSE) {
        // chunk_ops<drm, bin_search>, inference.hpp:815-839: bias FIRST, then every chunk row
        const float* __restrict__ xd = a.X.val + qg * a.X.cols;
        const uint32_t* __restrict__ ridx = a.L.row_idx + (td.rowptr_base - tile);
        uint32_t nr = active ? td.nrows : 0u;
        if (active && td.bias_slot != kNoBias) {
            for (uint32_t e = rp[td.bias_slot] + lig; e < rp[td.bias_slot + 1]; e += G) {
                const Entry en = ent[e];
                ACC(en.col) = __fadd_rn(ACC(en.col), __fmul_rn(a.L.bias, en.val));
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
                        ACC(en.col) = __fadd_rn(ACC(en.col), __fmul_rn(v, en.val));
                    }
                }
            }
            wave_sync_lds();
        }
    } else {
        // chunk_ops<csr, bin_search>, inference.hpp:769-813
        uint64_t xb = 0, xe = 0;
        if (active) { xb = a.X.row_ptr[qg]; xe = a.X.row_ptr[qg + 1]; }
        const uint32_t* __restrict__ xi = a.X.col_idx;
        const float* __restrict__ xv = a.X.val;
        const BmWord* __restrict__ bm = a.L.bitmap + (uint64_t)tile * a.L.nwords;
        uint64_t cur = xb;
        do {
            // ---- fill: probe G features per step, compact hits in feature order into the FIFO
            uint32_t nh = 0;
            while (__any(cur < xe && nh + G <= (uint32_t)H)) {
                const bool can = (cur < xe) && (nh + G <= (uint32_t)H);
                const uint64_t t = cur + lig;
                bool hit = false;
                uint32_t hslot = 0;
                float v = 0.f;
                if (can && t < xe) {
                    const uint32_t f = xi[t];
                    v = xv[t];
                    if (f < a.L.w_rows) {
                        const BmWord w = bm[f >> 5];
                        const uint32_t b = f & 31u;
                        if ((w.bits >> b) & 1u) {
                            hit = true;
                            hslot = w.rank + __popc(w.bits & ((1u << b) - 1u));
                        }
                    }
                }
                const unsigned long long m = __ballot(hit);
                const unsigned long long gm = (G == 64) ? m : ((m >> (grp * G)) & ((1ull << G) - 1ull));
                const uint32_t pos = nh + (uint32_t)__popcll(gm & ((1ull << lig) - 1ull));
                if (hit) { fv[FIDX(pos)] = v; fs[FIDX(pos)] = hslot; }
                nh += (uint32_t)__popcll(gm);
                if (can) cur += G;
            }
            wave_sync_lds();
            // ---- drain: rows in feature order; the G lanes stride over one row's entries
            for (uint32_t h = 0; __any(h < nh); ++h) {
                if (h < nh) {
                    const float v = fv[FIDX(h)];
                    const uint32_t s = fs[FIDX(h)];
                    for (uint32_t e = rp[s] + lig; e < rp[s + 1]; e += G) {
                        const Entry en = ent[e];
                        // out[col] += scalar * val (inference.hpp:512-517), mul then add, no fma
                        ACC(en.col) = __fadd_rn(ACC(en.col), __fmul_rn(v, en.val));
                    }
                }
                wave_sync_lds();
            }
        } while (__any(cur < xe));
        // bias LAST (inference.hpp:806-811)
        if (active && td.bias_slot != kNoBias) {
            for (uint32_t e = rp[td.bias_slot] + lig; e < rp[td.bias_slot + 1]; e += G) {
                const Entry en = ent[e];
                ACC(en.col) = __fadd_rn(ACC(en.col), __fmul_rn(a.L.bias, en.val));
            }
        }
        wave_sync_lds();
    }

    // ---- epilogue: transform (fp64) + combine with the parent's score, write the child block
    if (active) {
        float* __restrict__ out = a.cand + q * a.cand_stride + a.cand_off[q * a.beam_in + j] +
                                  (td.col_begin - a.L.chunk_col[parent]);
        for (uint32_t c = lig; c < ncols; c += G) {
            float v = pp_transform(a.pp_kind, a.pp_p, ACC(c));
            if (!a.first_layer) v = pp_combine(a.pp_kind, v, pscore);
            out[c] = v;
        }
    }
}

template <int G, bool DENSE>
static void launch_k1_inst(const K1Args& a, uint64_t slots, hipStream_t s) {
    constexpr int W = K1Cfg<G>::W, H = K1Cfg<G>::H;
    const uint32_t acc_floats = (G == 1) ? a.acc_stride * 64u : a.acc_stride * (uint32_t)W;
    const size_t lds = (size_t)acc_floats * 4 + (size_t)W * H * 8;
    if (lds > 160 * 1024) fail("k1: LDS request exceeds 160 KiB");
    static thread_local size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_kernel<G, DENSE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    const uint64_t blocks = (slots + W - 1) / W;
    if (blocks > 0x7FFFFFFFull) fail("k1: grid too large; lower max_batch_rows");
    hipLaunchKernelGGL((k1_kernel<G, DENSE>), dim3((uint32_t)blocks), dim3(64), lds, s, a);
    XRL_LAUNCH_CHECK();
}

int k1_auto_group(const LayerDev& L, const Layer& host, int dense) {
    // lanes per item ~ entries per tile row (power of two), raised until the accumulators of one
    // wavefront fit ~16 KiB of LDS so that several wavefronts share a CU.
    const double row_len = host.total_rows ? (double)host.nnz / (double)host.total_rows : 1.0;
    int g = 1;
    while (g < 64 && g < row_len * 0.75) g <<= 1;
    const uint32_t S = L.max_tile_cols | 1u;
    while (g < 64 && (size_t)(64 / g) * S * 4 > 16 * 1024) g <<= 1;
    (void)dense;
    return g;
}

void launch_k1(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, BeamDev prev,
               const uint32_t* cand_off, float* cand, int group, hipStream_t s) {
    if (P.nrows == 0) return;
    K1Args a;
    a.L = L; a.X = X; a.prev = prev; a.cand_off = cand_off; a.cand = cand;
    a.row0 = P.row0; a.nrows = P.nrows; a.beam_in = P.beam_in; a.cand_stride = P.cand_stride;
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.implicit_root = P.implicit_root;
    a.acc_stride = (group == 1) ? L.max_tile_cols : (L.max_tile_cols | 1u);
    const uint64_t slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
#define XRL_K1_CASE(GG) case GG: if (X.dense) launch_k1_inst<GG, true>(a, slots, s); else launch_k1_inst<GG, false>(a, slots, s); break;
    switch (group) {
        XRL_K1_CASE(1) XRL_K1_CASE(2) XRL_K1_CASE(4) XRL_K1_CASE(8) XRL_K1_CASE(16) XRL_K1_CASE(32) XRL_K1_CASE(64)
    default: fail("k1: lanes-per-item must be a power of two in [1, 64]");
    }
#undef XRL_K1_CASE
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
