// K1G: the dense-QUERY chunk products of a layer as a tiled, k-ordered SGEMM (chunk_ops<drm, bin_search>,
// inference.hpp:815-839, for dense-format layers).
//
// With dense X every (query, parent) item multiplies the query's D features into the parent's chunk: per item D x ncols
// multiply-adds over weights that every other query of the same parent needs too.  K1Q (query-stationary) streams the
// weight rows once PER QUERY -- BASELINE.json's dense-input config (D = 768, 3 M labels) then moves ~4 MB of weights per
// query through the L2.  K1G turns the loop nest around, the way a GEMM does:
//   * the layer's items are tile-sorted (counting sort of xrl_kernels.hip), so the queries that share a parent are adjacent;
//   * a workgroup owns ONE parent and up to QB of its queries: the parent's weight panel W[k0..k0+64, cols] and the queries'
//     X[q, k0..k0+64] are staged in LDS once per 64-feature step and every weight is reused QB times, every x value WP times;
//   * lane (cl, ql) holds an RQ x RC register tile of accumulators: queries {ql + 8 r}, column PAIRS {16 c2 + 2 cl, +1}; per 4
//     features it reads RQ float4 (queries) and 4 RC/2 float2 (weights of one feature for a column pair) from LDS --
//     conflict-free: 8 column lanes x 8 B are contiguous, 8 query lanes x 16 B cover distinct banks -- for 4 RQ RC
//     multiply-adds; adjacent columns in adjacent registers let v_pk_mul_f32 / v_pk_add_f32 work on pairs without moves.
// It is NOT an MFMA kernel: v_mfma_f32_* fuses the multiply and the add (one rounding), the reference rounds twice
// (`output[c] += x * w` compiled without FMA), and the round's contract is bit-identical label order.  Each accumulator
// therefore walks k in ascending order with a separate fp32 multiply and add (-ffp-contract=off), bias first -- exactly the
// reference's chain -- and what the matrix cores would have bought, operand reuse, comes from the LDS/register tiling.
// Roofline: 2 VALU lane-ops per multiply-add -> 39 T multiply-adds/s at 2.4 GHz (half the 157 TFLOP/s fp32 vector peak).
//
// Layers whose dense matrix holds kMissing cells (W has no entry there: sparse weight columns under dense X): with FINITE x a
// missing weight is staged as +0.0 (the product is +-0 and leaves every reachable accumulator unchanged), so the same 2-op loop
// serves them; a workgroup that holds a query with an inf / NaN (xguard_kernel flags them once per predict) takes the exact
// loop, which skips those cells like the reference's row walk does (select on the bit pattern, 4 lane-ops).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "xrl_device.h"
#include "xrl_kernels.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef XRL_K1G_UNROLL
#define XRL_K1G_UNROLL 2
#endif

struct alignas(16) ItemDescG {   // == ItemDesc of xrl_kernels.hip (K0 writes it)
    uint32_t q, tile, out_off; float pscore;
    uint64_t x_begin; uint32_t x_len, pad;
};
constexpr uint32_t kNoTileG = 0xFFFFFFFFu;

// KC features per LDS step (template parameter); the x panel's row stride in LDS is KC + 4 floats: 16-byte aligned rows on distinct banks

struct K1GArgs {
    LayerDev L; QueriesDev X;
    const ItemDescG* items;       // tile-sorted, all active
    const uint32_t* start;        // [n_tiles+1] first sorted item of every tile
    const uint32_t* blk_start;    // [n_tiles+1] first workgroup of every tile
    float* cand;
    const uint32_t* x_ok;         // [rows of the batch] 1 = every value of the query row is finite
    uint32_t row0;
    int pp_kind, pp_p, first_layer;
};

// one wavefront per dense query row: 1 = all values finite.  With finite x a missing weight may be multiplied as +0.0 -- the
// product is +-0 and leaves every reachable accumulator unchanged (it is never -0.0: it starts at bias_prod = (+0.0) + fl32(bias * w),
// which the model compiler computes exactly like that, so even a -0.0 product gives +0.0; and x*w + (-x*w) rounds to +0.0) --
// so K1G's inner loop needs no select; rows with an inf / NaN take the exact loop.  (A weight whose bits equal kMissing keeps
// the whole layer out of the dense format at load, xrl_model.cpp.)
// The same flag doubles as the guard of the exact bound pruning (prune_guard_ok, xrl_device.h): 1 only if, besides being finite, the
// row is small enough that no accumulator of any layer can overflow (wmax = the model's largest |weight| x max(1, |bias|)).
__global__ void __launch_bounds__(256) xguard_kernel(const uint64_t* __restrict__ row_ptr, const float* __restrict__ x, uint32_t rows, uint32_t cols,
                                                     uint32_t row0, float wmax, uint32_t* __restrict__ ok) {
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= rows) return;
    uint64_t b = ((uint64_t)row0 + r) * cols; uint32_t n = cols;
    if (row_ptr) { b = row_ptr[(uint64_t)row0 + r]; n = (uint32_t)(row_ptr[(uint64_t)row0 + r + 1] - b); }
    const float* __restrict__ p = x + b;
    uint32_t mx = 0u;
    for (uint32_t c = lane; c < n; c += 64u) mx = max(mx, __float_as_uint(p[c]) & 0x7FFFFFFFu);
    mx = wave_max_u32(mx);
    if (lane == 0) ok[r] = prune_guard_ok(mx, n, wmax) ? 1u : 0u;
}

void launch_xguard(const QueriesDev& X, uint32_t row0, uint32_t nrows, float wmax, uint32_t* ok, hipStream_t s) {
    if (nrows == 0) return;
    hipLaunchKernelGGL(xguard_kernel, dim3((nrows + 3u) / 4u), dim3(256), 0, s, X.dense ? nullptr : X.row_ptr, X.val, nrows, X.cols, row0, wmax, ok);
    XRL_LAUNCH_CHECK();
}

// per tile: number of workgroups = ceil(items / qb)   (exclusive-scanned afterwards)
__global__ void __launch_bounds__(256) k1g_count_blocks(const uint32_t* __restrict__ start, uint32_t n_tiles, uint32_t qb, uint32_t* __restrict__ blk) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_tiles) blk[t] = (start[t + 1] - start[t] + qb - 1u) / qb;
}

// single block: exclusive scan of v[0..n) in place; v[n] = grand total
__global__ void __launch_bounds__(1024) k1g_scan_kernel(uint32_t* __restrict__ v, uint32_t n) {
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

// RQ x RC: register tile of a lane (queries x columns); CS: wavefronts of the workgroup side by side along the COLUMNS (1 or 2; the
// other 4 / CS stack along the queries); KC: features per LDS step.
template <int RQ, int RC, int CS, int KC, int PPC>
__global__ void __launch_bounds__(256) k1g_kernel(K1GArgs a) {
    static_assert(RC % 2 == 0, "a lane owns column PAIRS");
    static_assert(CS == 1 || CS == 2 || CS == 4, "wavefronts along the columns");
    constexpr int LDK = KC + 4;
    constexpr int QW = 8 * RQ, QB = (4 / CS) * QW;            // queries per wavefront / workgroup
    constexpr int WCW = 8 * RC, WPC = WCW * CS;                // padded columns per wavefront / workgroup
    constexpr int RP = RC / 2;                                 // column pairs per lane
    constexpr int WLD = WPC;                                   // row stride of the weight panel in LDS (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                                           // [KC][WPC]: feature-major, columns contiguous (a lane reads COLUMN PAIRS: packed math without moves)
    float* sX = smem + KC * WLD;                                // [QB][LDK]
    uint32_t* sRow = reinterpret_cast<uint32_t*>(sX + QB * LDK);   // [QB] query row of every item of the workgroup
    uint32_t* sOut = sRow + QB;                                 // [QB] first candidate slot of the item's child block
    float* sPs = reinterpret_cast<float*>(sOut + QB);           // [QB] parent score
    __shared__ int s_exact;

    const uint32_t T = a.L.n_tiles, b = blockIdx.x;
    if (b >= a.blk_start[T]) return;
    uint32_t lo = 0, hi = T;                                    // largest t with blk_start[t] <= b (tiles without items share a start)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.blk_start[mid] <= b) lo = mid; else hi = mid; }
    const uint32_t t = lo, j = b - a.blk_start[lo];
    const uint32_t i0 = a.start[t] + j * (uint32_t)QB;
    const uint32_t nq = min((uint32_t)QB, a.start[t + 1] - i0);
    const TileDesc td = a.L.tiles[t];
    const uint32_t parent = a.L.tile_parent[t];
    const uint32_t gl = a.L.d_gp_log2, gmask = (1u << gl) - 1u;
    const uint32_t dt0 = a.L.d_ptile[parent], ndt = a.L.d_ptile[parent + 1] - dt0;
    const uint32_t WP = ndt << gl;                              // padded columns of this parent (<= WPC)
    const uint64_t wbase = (uint64_t)dt0 << gl;

    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, cl = lane & 7u, ql = lane >> 3;
    const uint32_t wq = wave % (uint32_t)(4 / CS), wc = wave / (uint32_t)(4 / CS);     // this wavefront's query block / column block
    if (tid == 0) s_exact = 0;
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)QB; i += 256u) {
        ItemDescG it{}; it.tile = kNoTileG;
        if (i < nq) it = a.items[i0 + i];
        sRow[i] = it.q; sOut[i] = it.out_off; sPs[i] = it.pscore;
        // the layer has cells without a weight AND this query holds an inf / NaN: the whole workgroup takes the exact loop
        if (i < nq && !a.L.d_full && !(a.x_ok && a.x_ok[it.q])) s_exact = 1;
    }

    // this lane's columns -> dense tile c >> gl, column c & gmask -> child (recomputed in the epilogue: not kept live across the loop)
    auto column = [&](int cc, uint32_t& child) -> bool {
        const uint32_t c = wc * (uint32_t)WCW + (uint32_t)(cc >> 1) * 16u + 2u * cl + (uint32_t)(cc & 1);   // lane cl owns the column pairs {16 c2 + 2 cl, +1} of its wavefront's block
        const uint32_t dt = c >> gl, within = c & gmask;
        bool v = c < WP;
        const uint32_t dtc = v ? dt0 + dt : 0u;
        const uint32_t cb = a.L.d_tcol[dtc], ce = a.L.d_tcol[dtc + 1];
        v = v && within < ce - cb;
        child = v ? cb + within : 0u;
        return v;
    };
    v2f acc[RQ][RP];
#pragma unroll
    for (int cc = 0; cc < RC; ++cc) {
        uint32_t child;
        const bool v = column(cc, child);
        const float b0 = (v && a.L.has_bias) ? a.L.bias_prod[child] : 0.0f;          // bias FIRST (inference.hpp:824-830)
#pragma unroll
        for (int r = 0; r < RQ; ++r) { if (cc & 1) acc[r][cc >> 1].y = b0; else acc[r][cc >> 1].x = b0; }
    }

    const uint32_t w_rows = a.L.w_rows, n_feat = a.L.has_bias ? w_rows - 1u : w_rows;
    const uint64_t ld = a.L.d_ld;
    const uint32_t* __restrict__ wd = a.L.wd;
    const float* __restrict__ xg = a.X.val;
    const uint32_t xcols = a.X.cols;
    const bool x16 = (xcols & 3u) == 0u && (reinterpret_cast<uintptr_t>(xg) & 15u) == 0u;
    const bool w16 = ((ld | wbase) & 3ull) == 0ull && (reinterpret_cast<uintptr_t>(wd) & 15u) == 0u;

    // Register-staged pipeline (issue early / write late): the global loads of step s+1 are in flight while step s is multiplied
    // out of LDS; one set of staging registers, written to LDS right after the barrier that retires step s.
    uint32_t padw = 0u;                                         // set below once the workgroup knows whether it runs the exact loop
    constexpr int WQ = (WPC + 3) / 4;                           // 16-byte column groups per feature row of the panel
    constexpr int WIT = (KC * WQ + 255) / 256;                  // weight-panel float4 per thread
    constexpr int XIT = QB * KC / 4 / 256;                      // query-panel float4 per thread
    static_assert(XIT >= 1 && XIT * 256 * 4 == QB * KC, "the query panel must divide among the threads");
    uint4 wreg[WIT]; float4 xreg[XIT];
    auto issue_loads = [&](uint32_t k0) {
        // weight panel: rows k0..k0+63, the parent's WP padded columns; thread e -> (feature k = e / (WPC/4), 4 columns from (e % (WPC/4)) * 4)
        // (dense tiles start at multiples of Gp >= ... columns and d_ld is a multiple of 32: 16-byte aligned whenever WPC >= 4)
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const uint32_t e = tid + (uint32_t)it * 256u;
            const uint32_t kq = e / (uint32_t)WQ, col = (e % (uint32_t)WQ) * 4u, f = k0 + kq;
            uint4 w = make_uint4(padw, padw, padw, padw);
            if (e < (uint32_t)(KC * WQ) && f < n_feat) {
                const uint32_t* __restrict__ src = wd + (uint64_t)f * ld + wbase + col;
                if (w16 && col + 4u <= WP) w = *reinterpret_cast<const uint4*>(src);
                else { if (col + 0u < WP) w.x = src[0]; if (col + 1u < WP) w.y = src[1]; if (col + 2u < WP) w.z = src[2]; if (col + 3u < WP) w.w = src[3]; }
            }
            if (padw == 0u) {                                     // fast loop: no entry -> +0.0 (finite x only, see xguard_kernel)
                w.x = w.x == kMissing ? 0u : w.x; w.y = w.y == kMissing ? 0u : w.y; w.z = w.z == kMissing ? 0u : w.z; w.w = w.w == kMissing ? 0u : w.w;
            }
            wreg[it] = w;
        }
        // query panel: X[row, k0..k0+63] of the workgroup's queries (coalesced along the features; 16 bytes per lane when the
        // rows are 16-byte aligned and the step lies inside the layer's features)
        const bool fast = x16 && k0 + (uint32_t)KC <= min(n_feat, xcols);
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const uint32_t e = tid + (uint32_t)it * 256u;
            const uint32_t qq = e / (uint32_t)(KC / 4), k4 = e % (uint32_t)(KC / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq < nq) {
                const float* __restrict__ xr = xg + ((uint64_t)a.row0 + sRow[qq]) * xcols;
                if (fast) v = *reinterpret_cast<const float4*>(xr + k0 + k4 * 4u);
                else {
                    const uint32_t f = k0 + k4 * 4u;
                    v.x = (f + 0u < n_feat && f + 0u < xcols) ? xr[f + 0u] : 0.0f;
                    v.y = (f + 1u < n_feat && f + 1u < xcols) ? xr[f + 1u] : 0.0f;
                    v.z = (f + 2u < n_feat && f + 2u < xcols) ? xr[f + 2u] : 0.0f;
                    v.w = (f + 3u < n_feat && f + 3u < xcols) ? xr[f + 3u] : 0.0f;
                }
            }
            xreg[it] = v;
        }
    };
    auto store_panels = [&]() {
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const uint32_t e = tid + (uint32_t)it * 256u;
            if (e < (uint32_t)(KC * WQ)) *reinterpret_cast<uint4*>(sW + ((size_t)(e / (uint32_t)WQ) * WLD + (size_t)(e % (uint32_t)WQ) * 4)) = wreg[it];   // sW[k][col]
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const uint32_t e = tid + (uint32_t)it * 256u;
            const uint32_t qq = e / (uint32_t)(KC / 4), k4 = e % (uint32_t)(KC / 4);
            *reinterpret_cast<float4*>(sX + qq * LDK + k4 * 4u) = xreg[it];
        }
    };

    __syncthreads();                                            // sRow / sOut / sPs / s_exact are visible
    const bool exact = s_exact != 0;                            // workgroup-uniform
    padw = exact ? kMissing : 0u;
    issue_loads(0u);
    for (uint32_t k0 = 0; k0 < n_feat; k0 += (uint32_t)KC) {
        __syncthreads();                                        // the previous step's readers are done
        store_panels();
        __syncthreads();
        if (k0 + (uint32_t)KC < n_feat) issue_loads(k0 + (uint32_t)KC);      // in flight during the arithmetic below
        // ---- RQ x RC register tile, 4 features per step; every accumulator takes its features in ascending order
        // LDS reads are software-pipelined: the weights of feature f+1 and the query values of the next 4 features are in flight
        // while feature f is multiplied (the last prefetches read the x rows' padding / the row after the weight panel: inside
        // the allocation, never used).
        auto multiply = [&](auto exact_tag) {
        constexpr bool EXACT = decltype(exact_tag)::value;
        const float* __restrict__ xl = sX + (size_t)(wq * QW + ql) * LDK;
        const float* __restrict__ wl = sW + (size_t)(wc * WCW + 2 * cl);
        float4 xv[RQ]; v2f wv[2][RP];
#pragma unroll
        for (int r = 0; r < RQ; ++r) xv[r] = *reinterpret_cast<const float4*>(xl + (size_t)r * 8 * LDK);
#pragma unroll
        for (int c2 = 0; c2 < RP; ++c2) wv[0][c2] = *reinterpret_cast<const v2f*>(wl + (size_t)c2 * 16);
#pragma unroll XRL_K1G_UNROLL
        for (int kk = 0; kk < KC; kk += 4) {
            float4 xn[RQ];
#pragma unroll
            for (int r = 0; r < RQ; ++r) xn[r] = *reinterpret_cast<const float4*>(xl + (size_t)r * 8 * LDK + kk + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // this lane's column pairs at feature kk+j+1: 8-byte reads, 8 lanes x 8 B contiguous (conflict-free), the other lanes broadcast
#pragma unroll
                for (int c2 = 0; c2 < RP; ++c2) wv[(j + 1) & 1][c2] = *reinterpret_cast<const v2f*>(wl + (size_t)(kk + j + 1) * WLD + (size_t)c2 * 16);
#pragma unroll
                for (int r = 0; r < RQ; ++r) {
                    const float x = j == 0 ? xv[r].x : j == 1 ? xv[r].y : j == 2 ? xv[r].z : xv[r].w;
                    const v2f x2 = {x, x};
#pragma unroll
                    for (int c2 = 0; c2 < RP; ++c2) {
                        const v2f w2 = wv[j & 1][c2];
                        const v2f s2 = acc[r][c2] + x2 * w2;      // built with -ffp-contract=off: v_pk_mul_f32, v_pk_add_f32 (multiply, round, add, round)
                        if (EXACT) {
                            acc[r][c2].x = (__float_as_uint(w2.x) == kMissing) ? acc[r][c2].x : s2.x;
                            acc[r][c2].y = (__float_as_uint(w2.y) == kMissing) ? acc[r][c2].y : s2.y;
                        } else acc[r][c2] = s2;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RQ; ++r) xv[r] = xn[r];
        }
        };
        if (exact) multiply(std::true_type{}); else multiply(std::false_type{});
    }
    // ---- transform, combine with the parent's score, write the child block
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < RC; ++cc) {
        uint32_t child;
        if (!column(cc, child)) continue;
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            const uint32_t qi = wq * (uint32_t)QW + (uint32_t)r * 8u + ql;
            if (qi >= nq) continue;
            float v = pp_transform<PPC>(a.pp_kind, a.pp_p, (cc & 1) ? acc[r][cc >> 1].y : acc[r][cc >> 1].x);
            if (!a.first_layer) v = pp_combine(a.pp_kind, v, sPs[qi]);
            a.cand[(size_t)sOut[qi] + (child - td.col_begin)] = v;
        }
    }
}

// padded columns a workgroup must cover, or 0 when K1G cannot serve the layer
uint32_t k1g_cols(const LayerDev& L) {
    if (!L.wd || !L.tile_parent || L.max_tiles_per_parent != 1) return 0;
    const uint32_t wp = L.d_max_tiles << L.d_gp_log2;
    return wp <= 128 ? wp : 0u;
}

template <int RQ, int RC, int CS, int KC> struct K1GShape {
    static constexpr uint32_t QB = (4 / CS) * 8 * RQ;
    static constexpr size_t lds() { return ((size_t)KC * 8 * RC * CS + (size_t)QB * (KC + 4) + 3 * (size_t)QB) * 4; }
};

template <int RQ, int RC, int CS, int KC>
static void k1g_go(K1GArgs& a, const LayerDev& L, const LayerPlan& P, uint32_t* blk_start, hipStream_t s) {
    constexpr uint32_t qb = K1GShape<RQ, RC, CS, KC>::QB;
    hipLaunchKernelGGL(k1g_count_blocks, dim3((L.n_tiles + 255u) / 256u), dim3(256), 0, s, a.start, L.n_tiles, qb, blk_start);
    hipLaunchKernelGGL(k1g_scan_kernel, dim3(1), dim3(1024), 0, s, blk_start, L.n_tiles);
    const uint64_t n_slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;
    const uint64_t blocks = (n_slots + qb - 1) / qb + L.n_tiles;      // every tile adds at most one partial workgroup
    if (blocks > 0x7FFFFFFFull) fail("k1g: grid too large; lower max_batch_rows");
    auto kern = pp_class(P.pp) ? &k1g_kernel<RQ, RC, CS, KC, 1> : &k1g_kernel<RQ, RC, CS, KC, 0>;
    constexpr size_t lds = K1GShape<RQ, RC, CS, KC>::lds();
    static_assert(lds <= 160 * 1024, "panel sizes exceed the LDS");
    if (lds > 48 * 1024) XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(256), lds, s, a);
}

void launch_k1g(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items_sorted, const uint32_t* start,
                uint32_t* blk_start, const uint32_t* x_ok, float* cand, hipStream_t s) {
    if (P.nrows == 0) return;
    const uint32_t wp = k1g_cols(L);
    if (wp == 0 || !X.dense) fail("k1g: layer not eligible");
    K1GArgs a;
    a.L = L; a.X = X; a.items = static_cast<const ItemDescG*>(items_sorted); a.start = start; a.blk_start = blk_start; a.cand = cand; a.x_ok = x_ok;
    a.row0 = P.row0; a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer;
    // shapes per class of padded parent width, chosen from the measurements in profiles/r02_k1g_shapes.txt; tune.k1g_variant = 1
    // (xrl_set_option "k1g_variant") runs the one-wavefront-per-64-queries shapes they replaced (A/B, tests)
    const bool alt = P.tune.k1g_variant == 1;
#define XRL_K1G(RQ, RC, CS, KC) k1g_go<RQ, RC, CS, KC>(a, L, P, blk_start, s)
    if (wp <= 16) { if (alt) XRL_K1G(4, 2, 1, 32); else XRL_K1G(4, 2, 1, 64); }
    else if (wp <= 32) { if (alt) XRL_K1G(4, 4, 1, 32); else XRL_K1G(4, 4, 1, 64); }
    else if (wp <= 64) { if (alt) XRL_K1G(2, 8, 1, 64); else XRL_K1G(4, 4, 2, 32); }
    else if (wp <= 96) { if (alt) XRL_K1G(2, 12, 1, 64); else XRL_K1G(4, 6, 2, 32); }
    else { if (alt) XRL_K1G(4, 8, 2, 64); else XRL_K1G(4, 8, 2, 32); }
#undef XRL_K1G
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
