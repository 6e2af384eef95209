// K1R: the (query, tile) chunk products of a tile-format layer, tile-RESIDENT -- the workgroup that owns a tile copies the
// tile's image (xrl_model.cpp) into LDS once and streams the tile's items through it.
//
//   reference: w_ops<chunked>::compute_sparse_predictions (inference.hpp:925-1007) with its sort-by-chunk (:991-993),
//              chunk_ops<csr, bin_search> (:769-813), add_scaled_chunk_row_to_output_block (:506-518),
//              transform + combine (:1360-1384, PostProcessor :192-240)
//
// Why.  K1 (xrl_kernels.hip) serves an item with ~210 L1->L2 requests (a rank-bitmap word per query feature, an extent per
// hit, the entry lines of every hit row) and sits on the fabric REQUEST-rate ceiling (profiles/r02_*): every request moves 64
// bytes of which 8-16 are used, and the ~600 items that visit a leaf tile per step each fetch the same lines again.  Here a
// tile's lookup structure and weights are read from HBM once per workgroup; per item only the descriptor, the query row and
// the output block touch global memory (~20 requests).
//
// Work decomposition.  Items are tile-sorted (launch_sort_items).  One workgroup = one tile (or 1/splits of its items);
// one wavefront = one item at a time, LANE == COLUMN, accumulators in registers (NS = 1 or 2 per lane: tiles of <= 64 / <= 128
// columns).  Per 64 query features: every lane probes one feature against the tile's rank-bitmap in LDS (one ds_read_b64 + the
// rank), hit lanes fetch their row descriptor (and, for rows held in entry form, their <= TS entries) -- all hits of the step
// in parallel -- then a SCALAR loop walks the hits in lane (= ascending feature) order:
//   dense row (len > T): every lane reads its column's weight (kMissing where the row has no entry), v_mul, v_add, select;
//   short row (len <= T): per entry, column and weight are broadcast from the hit lane (v_readlane), the one lane that owns
//                         the column adds.
// Each column therefore accumulates fl32(acc + fl32(x_f * w)) over its matched features in ascending feature order, which is
// the reference's order; an absent entry performs no operation (so explicit zeros in W and non-finite x behave as in the
// reference's row walk); bias last; transform in fp64; combine in fp32 -- bit-identical to K1.
#include <hip/hip_runtime.h>

#include "xrl_device.h"
#include "xrl_items.h"
#include "xrl_kernels.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

struct K1RArgs {
    const uint32_t* img; const uint64_t* img_off;   // tile images (LayerDev::img)
    const ItemDesc* items;                           // tile-sorted, all active
    const uint32_t* start;                           // [n_tiles + 1] first sorted item of every tile
    const uint32_t* xi; const float* xv;             // CSR queries
    float* cand;
    uint32_t w_rows, splits;
    int pp_kind, pp_p, first_layer, has_bias;
};

struct XChunk { uint32_t f; uint32_t v; };           // one query feature per lane: id (0xFFFFFFFF past the row's end), value bits

template <int NS, int TS, int PPC>
__global__ void __launch_bounds__(1024) k1r_kernel(K1RArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t t = blockIdx.x / a.splits, sp = blockIdx.x - t * a.splits;
    const uint32_t s0 = a.start[t], n_t = a.start[t + 1] - s0;
    const uint32_t b0 = s0 + (uint32_t)((uint64_t)n_t * sp / a.splits), b1 = s0 + (uint32_t)((uint64_t)n_t * (sp + 1u) / a.splits);
    if (b0 >= b1) return;                                               // uniform: no barrier has been reached yet
    const uint32_t nthreads = blockDim.x;
    const uint32_t nw = __builtin_amdgcn_readfirstlane(nthreads >> 6);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t* __restrict__ xi = a.xi;
    const float* __restrict__ xv = a.xv;

    const ItemDesc none = make_item(0u, kNoTile, 0u, 0.f, 0, 0u);
    auto ld_item = [&](uint32_t i) -> ItemDesc { return i < b1 ? a.items[i] : none; };   // uniform index: scalar loads
    // 64 query features of an item, one per lane; unconditional clamped loads (lanes past the row's end re-read element 0)
    auto ld_chunk = [&](const ItemDesc& d, uint32_t c0) -> XChunk {
        const uint32_t idx = c0 + (uint32_t)lane;
        const bool ok = idx < d.x_len;
        const uint64_t p = ok ? d.x_begin + idx : 0ull;
        XChunk r; r.f = xi[p]; r.v = __float_as_uint(xv[p]);
        if (!ok) r.f = 0xFFFFFFFFu;
        return r;
    };

    // the first items' descriptors and features are in flight during the image copy
    uint32_t i = b0 + wave;
    ItemDesc it = ld_item(i), it_n = ld_item(i + nw);
    XChunk A = ld_chunk(it, 0u), B = ld_chunk(it, 64u);
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.img + a.img_off[t]);   // images are 16-byte aligned, whole uint4s
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nq = src[0].x >> 2;
        uint32_t k = threadIdx.x;
        for (; k + 3u * nthreads < nq; k += 4u * nthreads) {            // four loads in flight per thread
            const uint4 q0 = src[k], q1 = src[k + nthreads], q2 = src[k + 2u * nthreads], q3 = src[k + 3u * nthreads];
            dst[k] = q0; dst[k + nthreads] = q1; dst[k + 2u * nthreads] = q2; dst[k + 3u * nthreads] = q3;
        }
        for (; k < nq; k += nthreads) dst[k] = src[k];
    }
    __syncthreads();
    const uint32_t ncols = __builtin_amdgcn_readfirstlane(smem[2]);
    const uint2* __restrict__ bm_bits = reinterpret_cast<const uint2*>(smem + 8);
    const uint16_t* __restrict__ bm_rank = reinterpret_cast<const uint16_t*>(smem + __builtin_amdgcn_readfirstlane(smem[3]));
    const uint32_t* __restrict__ rowdesc = smem + __builtin_amdgcn_readfirstlane(smem[4]);
    const float* __restrict__ t_bias = reinterpret_cast<const float*>(smem + __builtin_amdgcn_readfirstlane(smem[5]));
    const unsigned char* __restrict__ sbytes = reinterpret_cast<const unsigned char*>(smem);
    uint32_t coff[NS];                                                  // byte offset of this lane's columns inside a dense row (pad word past the tile's width)
#pragma unroll
    for (int r = 0; r < NS; ++r) coff[r] = min((uint32_t)(r * 64 + lane), ncols) * 4u;
    const uint32_t w_rows = a.w_rows;

    for (; i < b1; i += nw) {
        // ---- prefetch: descriptor two items ahead, the first 128 features of the next item
        const ItemDesc it_nn = ld_item(i + 2u * nw);
        const XChunk An = ld_chunk(it_n, 0u), Bn = ld_chunk(it_n, 64u);

        float acc[NS];
#pragma unroll
        for (int r = 0; r < NS; ++r) acc[r] = 0.0f;                     // std::fill(..., 0.0), inference.hpp:964
        const uint32_t x_len = it.x_len;
        for (uint32_t c0 = 0; c0 < x_len; c0 += 64u) {
            XChunk C; C.f = 0xFFFFFFFFu; C.v = 0u;
            if (c0 + 128u < x_len) C = ld_chunk(it, c0 + 128u);
            // ---- probe: is this lane's feature a row of the tile, and which
            const uint32_t f = A.f;
            const bool inr = f < w_rows;                                // also false on the padding lanes
            const uint32_t wq = inr ? (f >> 6) : 0u;
            const uint2 bw = bm_bits[wq];
            const uint32_t rk = bm_rank[wq];
            const unsigned long long b64 = ((unsigned long long)bw.y << 32) | bw.x;
            const uint32_t bpos = f & 63u;
            const bool hit = inr && ((b64 >> bpos) & 1ull);
            const uint32_t slot = rk + (uint32_t)__popcll(b64 & ((1ull << bpos) - 1ull));
            const uint32_t desc = rowdesc[hit ? slot : 0u];
            // rows held in entry form: the hit lane fetches its row's entries (all hits of the step at once)
            const bool is_short = hit && !(desc & 0x80000000u);
            const uint32_t len1 = is_short ? ((desc >> 24) & 7u) : 0u;  // entries - 1
            const uint2* __restrict__ ep = reinterpret_cast<const uint2*>(smem + (is_short ? (desc & 0xFFFFFFu) : 0u));   // pairs are 8-byte aligned
            uint2 e[TS];
#pragma unroll
            for (int k = 0; k < TS; ++k) e[k] = ep[min((uint32_t)k, len1)];
            // ---- hits in ascending feature order.  The dense-row weights of the NEXT hit are requested (unconditionally: offset 0
            //      when that hit is a short row or there is none) before the current hit is applied, so the LDS latency of a row
            //      hides behind the previous row's arithmetic.
            unsigned long long mask = __ballot(hit);
            if (mask) {
                uint32_t dn = (uint32_t)__builtin_amdgcn_readlane((int)desc, __ffsll((long long)mask) - 1);
                uint32_t wbn[NS];
                {
                    const unsigned char* __restrict__ row = sbytes + ((dn & 0x80000000u) ? (dn & 0xFFFFFFu) * 4u : 0u);
#pragma unroll
                    for (int r = 0; r < NS; ++r) wbn[r] = *reinterpret_cast<const uint32_t*>(row + coff[r]);
                }
                while (mask) {
                    const int h = __ffsll((long long)mask) - 1;
                    mask &= mask - 1ull;
                    const uint32_t d = dn;
                    uint32_t wb[NS];
#pragma unroll
                    for (int r = 0; r < NS; ++r) wb[r] = wbn[r];
                    {
                        const int hn = mask ? __ffsll((long long)mask) - 1 : h;
                        dn = (uint32_t)__builtin_amdgcn_readlane((int)desc, hn);
                        const unsigned char* __restrict__ row = sbytes + ((dn & 0x80000000u) ? (dn & 0xFFFFFFu) * 4u : 0u);
#pragma unroll
                        for (int r = 0; r < NS; ++r) wbn[r] = *reinterpret_cast<const uint32_t*>(row + coff[r]);
                    }
                    const float xs = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)A.v, h));
                    if (d & 0x80000000u) {
#pragma unroll
                        for (int r = 0; r < NS; ++r) {
                            // scalar * val, then add: no fma (inference.hpp:512-517); no entry -> no operation
                            const float s = __fadd_rn(acc[r], __fmul_rn(xs, __uint_as_float(wb[r])));
                            acc[r] = (wb[r] == kMissing) ? acc[r] : s;
                        }
                    } else {
                        const uint32_t l1 = (d >> 24) & 7u;
#pragma unroll
                        for (int k = 0; k < TS; ++k) {
                            if ((uint32_t)k <= l1) {
                                const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)e[k].x, h);
                                const float p = __fmul_rn(xs, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)e[k].y, h)));
                                if (NS == 1 || code < 64u) {
                                    const float s = __fadd_rn(acc[0], p);
                                    acc[0] = ((uint32_t)lane == code) ? s : acc[0];
                                } else {
                                    const float s = __fadd_rn(acc[NS - 1], p);
                                    acc[NS - 1] = ((uint32_t)lane + 64u == code) ? s : acc[NS - 1];
                                }
                            }
                        }
                    }
                }
            }
            A = B; B = C;
        }
        // ---- bias LAST (inference.hpp:806-811), transform in fp64, combine with the parent's score, write the child block
        float* __restrict__ out = a.cand + it.out_off;
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            const uint32_t c = (uint32_t)(r * 64 + lane);
            if (c < ncols) {
                float s = acc[r];
                if (a.has_bias) s = __fadd_rn(s, t_bias[c]);
                float v = pp_transform<PPC>(a.pp_kind, a.pp_p, s);
                if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
                out[c] = v;
            }
        }
        it = it_n; it_n = it_nn; A = An; B = Bn;
    }
}

bool k1r_eligible(const LayerDev& L) { return L.img != nullptr && L.max_tile_cols <= 128u && L.img_max_short <= kK1RMaxShort; }

void launch_k1r(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items_sorted, const uint32_t* start,
                float* cand, uint32_t splits, hipStream_t s) {
    if (P.nrows == 0) return;
    if (!k1r_eligible(L) || X.dense || X.nnz == 0) fail("k1r: layer / queries not eligible");
    K1RArgs a;
    a.img = L.img; a.img_off = L.img_off; a.items = static_cast<const ItemDesc*>(items_sorted); a.start = start;
    a.xi = X.col_idx; a.xv = X.val; a.cand = cand;
    a.w_rows = L.w_rows; a.splits = std::max(1u, splits);
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.has_bias = L.has_bias;
    const size_t lds = ((size_t)L.max_tile_img + 15) & ~(size_t)15;
    if (lds > 160 * 1024) fail("k1r: tile image exceeds the LDS");
    const uint64_t blocks = (uint64_t)L.n_tiles * a.splits;
    if (blocks > 0x7FFFFFFFull) fail("k1r: grid too large");
    // one workgroup per CU when the image takes more than half of the LDS (16 wavefronts); two of 8 wavefronts otherwise
    const uint32_t threads = lds > 80 * 1024 ? 1024u : 512u;
    const int ppc = pp_class(P.pp);
    const int ns = L.max_tile_cols <= 64u ? 1 : 2;
    const int ts = L.img_max_short <= 2u ? 2 : (L.img_max_short <= 4u ? 4 : 8);
#define XRL_K1R_GO(NN, TT, PP) do { \
        auto kern = &k1r_kernel<NN, TT, PP>; \
        if (lds > 48 * 1024) XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(threads), lds, s, a); } while (0)
#define XRL_K1R_T(NN, PP) do { if (ts == 2) XRL_K1R_GO(NN, 2, PP); else if (ts == 4) XRL_K1R_GO(NN, 4, PP); else XRL_K1R_GO(NN, 8, PP); } while (0)
#define XRL_K1R_N(PP) do { if (ns == 1) XRL_K1R_T(1, PP); else XRL_K1R_T(2, PP); } while (0)
    if (ppc) XRL_K1R_N(1); else XRL_K1R_N(0);
#undef XRL_K1R_N
#undef XRL_K1R_T
#undef XRL_K1R_GO
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
