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
// Work decomposition.  Items are tile-sorted (launch_sort_items).  One workgroup = a run of `ch` consecutive sorted items
// (usually of one tile; the image is re-loaded when the run crosses into the next tile); one wavefront = one item at a time,
// LANE == COLUMN PAIR (lane l owns columns 2l and 2l+1), accumulators in registers.  Per 64 query features: every lane probes
// one feature against the tile's rank-bitmap in LDS (one ds_read_b64 + the rank), hit lanes fetch their row descriptor (and,
// for rows held in entry form, their <= TS entries) -- all hits of the step in parallel -- and work out the NEXT hit lane and the
// LDS offset of their row; then a scalar loop walks the hits in lane (= ascending feature) order with three v_readlane per hit:
//   every hit:  the lanes read their pair of the row's weights (a dense row, or the all-zero row for a short one), v_pk_mul, v_pk_add;
//   short row:  additionally, per entry, column and weight are broadcast from the hit lane and the one lane that owns the column adds.
// The loop is kept poor in SCALAR instructions on purpose: the CU's single scalar unit serves all 16 wavefronts (a first version
// that derived the next hit and the row address with ~25 s_* instructions per hit ran at 17.5 ms, scalar-issue bound).
//
// Arithmetic: each column accumulates fl32(acc + fl32(x_f * w)) over its matched features in ascending feature order -- the
// reference's order.  A dense row holds +0.0 where it has no entry: for FINITE x that step is acc + (+-0) == acc (an accumulator
// that starts at +0.0 can never become -0.0), i.e. "no operation", exactly like the reference's skip; a 64-feature step that
// contains a non-finite x takes the exact loop instead (select on the row's column mask), so inf / NaN inputs still behave like
// the reference's row walk.  Bias last; transform in fp64; combine in fp32 -- bit-identical to K1.
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
    uint32_t w_rows, n_tiles, ch;                    // ch: sorted items per workgroup
    int pp_kind, pp_p, first_layer, has_bias;
    int bias_first;                                  // HASH_CHUNKED arithmetic: accumulators start at the bias product
};

struct XChunk { uint32_t f; uint32_t v; };           // one query feature per lane: id (0xFFFFFFFF past the row's end), value bits

template <int TS, int PPC>
__global__ void __launch_bounds__(1024) k1r_kernel(K1RArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // workgroup b walks the sorted items [b * ch, (b + 1) * ch): a run may cross tile boundaries (the image is re-loaded), and a
    // tile that serves many items is shared by as many workgroups as it takes -- the work per workgroup is bounded either way
    // (beams concentrate on few parents: on the Amazon-670K shape 10 of the 8192 leaf tiles receive 3/4 of the items)
    const uint32_t n_items = a.start[a.n_tiles];
    const uint32_t blk0 = blockIdx.x * a.ch;
    if (blk0 >= n_items) return;
    const uint32_t blk1 = min(n_items, blk0 + a.ch);
    const uint32_t nthreads = blockDim.x;
    const uint32_t nw = __builtin_amdgcn_readfirstlane(nthreads >> 6);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t* __restrict__ xi = a.xi;
    const float* __restrict__ xv = a.xv;
    const uint32_t w_rows = a.w_rows;
    const ItemDesc none = make_item(0u, kNoTile, 0u, 0.f, 0, 0u);
    const unsigned long long lane_bit = 1ull << lane;

  for (uint32_t pos = blk0; pos < blk1;) {
    const uint32_t t = __builtin_amdgcn_readfirstlane(a.items[pos].tile);
    const uint32_t b1 = __builtin_amdgcn_readfirstlane(min(blk1, a.start[t + 1]));   // end of this tile's run inside the block
    auto ld_item = [&](uint32_t i) -> ItemDesc { return i < b1 ? a.items[i] : none; };   // uniform index: scalar loads
    // 64 query features of an item, one per lane; unconditional clamped loads (lanes past the row's end re-read the row's first element)
    auto ld_chunk = [&](const ItemDesc& d, uint32_t c0) -> XChunk {
        const uint32_t idx = c0 + (uint32_t)lane;
        const bool ok = idx < d.x_len;
        const uint64_t p = d.x_begin + (ok ? idx : 0u);
        XChunk r; r.f = xi[p]; r.v = __float_as_uint(xv[p]);
        if (!ok) r.f = 0xFFFFFFFFu;
        return r;
    };

    // the first items' descriptors and features are in flight during the image copy
    uint32_t i = pos + wave;
    ItemDesc it = ld_item(i), it_n = ld_item(i + nw);
    XChunk A = ld_chunk(it, 0u), B = ld_chunk(it, 64u);
    __syncthreads();                                                    // the previous tile's readers are done
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.img + a.img_off[t]);   // images are 16-byte aligned, whole uint4s
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nq = (uint32_t)(a.img_off[t + 1] - a.img_off[t]) >> 2;
        uint32_t k = threadIdx.x;
        for (; k + 3u * nthreads < nq; k += 4u * nthreads) {            // four loads in flight per thread
            const uint4 q0 = src[k], q1 = src[k + nthreads], q2 = src[k + 2u * nthreads], q3 = src[k + 3u * nthreads];
            dst[k] = q0; dst[k + nthreads] = q1; dst[k + 2u * nthreads] = q2; dst[k + 3u * nthreads] = q3;
        }
        for (; k < nq; k += nthreads) dst[k] = src[k];
    }
    __syncthreads();
    const uint32_t ncols = __builtin_amdgcn_readfirstlane(smem[2]);
    const uint2* __restrict__ bm_bits = reinterpret_cast<const uint2*>(smem + 12);
    const uint16_t* __restrict__ bm_rank = reinterpret_cast<const uint16_t*>(smem + __builtin_amdgcn_readfirstlane(smem[3]));
    const uint32_t* __restrict__ rowdesc = smem + __builtin_amdgcn_readfirstlane(smem[4]);
    const float* __restrict__ t_bias = reinterpret_cast<const float*>(smem + __builtin_amdgcn_readfirstlane(smem[5]));
    const uint32_t zero_row = __builtin_amdgcn_readfirstlane(smem[8]) * 4u;          // byte offset of the all-zero row's first pair
    const unsigned char* __restrict__ sbytes = reinterpret_cast<const unsigned char*>(smem);
    const uint32_t npairs = (ncols + 1u) >> 1;
    const uint32_t poff = min((uint32_t)lane, npairs) * 8u;            // byte offset of this lane's pair inside a dense row (zero pair past the tile's width)

    for (; i < b1; i += nw) {
        // ---- prefetch: descriptor two items ahead, the first 128 features of the next item
        const ItemDesc it_nn = ld_item(i + 2u * nw);
        const XChunk An = ld_chunk(it_n, 0u), Bn = ld_chunk(it_n, 64u);

        float2 acc = make_float2(0.0f, 0.0f);                           // std::fill(..., 0.0), inference.hpp:964
        if (a.bias_first && a.has_bias) {                               // chunk_ops<csr, hash>, inference.hpp:716-722 (bias_prod = 0.0 + bias * w: never -0.0)
            const uint32_t cb = 2u * (uint32_t)lane;
            if (cb < ncols) acc.x = t_bias[cb];
            if (cb + 1u < ncols) acc.y = t_bias[cb + 1u];
        }
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
            const unsigned long long mask = __ballot(hit);
            // what the scalar loop reads per hit: the row's byte offset (zero row for a short one) with the "short" flag in bit 0,
            // and the NEXT hit lane (the last hit points at itself) with the row's length - 1 above it
            const uint32_t rowoff = is_short ? (zero_row | 1u) : (desc & 0x3FFFFFu) * 4u;
            const unsigned long long later = mask & ~(lane_bit | (lane_bit - 1ull));
            const uint32_t nxt = (later ? (uint32_t)__ffsll((long long)later) - 1u : (uint32_t)lane) | (len1 << 8);
            const bool nonfinite = __ballot(inr && (A.v & 0x7F800000u) == 0x7F800000u) != 0ull;
            if (mask) {
                int h = __ffsll((long long)mask) - 1;
                uint32_t rn = (uint32_t)__builtin_amdgcn_readlane((int)rowoff, h);
                uint2 wbn = *reinterpret_cast<const uint2*>(sbytes + (rn & ~3u) + poff);
                for (;;) {
                    // ---- this hit: x value, next hit, row weights (requested during the previous hit)
                    const float xs = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)A.v, h));
                    const uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)nxt, h);
                    const int hn = (int)(nx & 0xFFu);
                    const uint32_t r = rn;
                    const uint2 wb = wbn;
                    rn = (uint32_t)__builtin_amdgcn_readlane((int)rowoff, hn);
                    wbn = *reinterpret_cast<const uint2*>(sbytes + (rn & ~3u) + poff);   // the next hit's weights: in flight while this one is applied
                    // scalar * val, then add: no fma (inference.hpp:512-517)
                    const float s0 = __fadd_rn(acc.x, __fmul_rn(xs, __uint_as_float(wb.x)));
                    const float s1 = __fadd_rn(acc.y, __fmul_rn(xs, __uint_as_float(wb.y)));
                    if (!nonfinite) { acc.x = s0; acc.y = s1; }
                    else {
                        // exact: a column without an entry performs no operation (the row's column mask sits in the 4 words before its pairs)
                        const uint32_t mw = *reinterpret_cast<const uint32_t*>(sbytes + (r & ~3u) - 16u + (((uint32_t)lane >> 4) << 2));
                        const uint32_t mb = mw >> (((uint32_t)lane & 15u) << 1);
                        const bool in_row = (uint32_t)lane < npairs;
                        if (in_row && (mb & 1u)) acc.x = s0;
                        if (in_row && (mb & 2u)) acc.y = s1;
                    }
                    if (r & 1u) {
                        const uint32_t l1 = nx >> 8;
#pragma unroll
                        for (int k = 0; k < TS; ++k) {
                            if ((uint32_t)k <= l1) {
                                const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)e[k].x, h);
                                const float p = __fmul_rn(xs, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)e[k].y, h)));
                                const bool mine = (uint32_t)lane == (code >> 1);
                                if (code & 1u) { const float s = __fadd_rn(acc.y, p); acc.y = mine ? s : acc.y; }
                                else { const float s = __fadd_rn(acc.x, p); acc.x = mine ? s : acc.x; }
                            }
                        }
                    }
                    if (hn == h) break;
                    h = hn;
                }
            }
            A = B; B = C;
        }
        // ---- bias LAST (inference.hpp:806-811), transform in fp64, combine with the parent's score, write the child block
        float* __restrict__ out = a.cand + it.out_off;
        const uint32_t c = 2u * (uint32_t)lane;
        if (c < ncols) {
            float s = acc.x;
            if (a.has_bias && !a.bias_first) s = __fadd_rn(s, t_bias[c]);
            float v = pp_transform<PPC>(a.pp_kind, a.pp_p, s);
            if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
            out[c] = v;
        }
        if (c + 1u < ncols) {
            float s = acc.y;
            if (a.has_bias && !a.bias_first) s = __fadd_rn(s, t_bias[c + 1u]);
            float v = pp_transform<PPC>(a.pp_kind, a.pp_p, s);
            if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
            out[c + 1u] = v;
        }
        it = it_n; it_n = it_nn; A = An; B = Bn;
    }
    pos = b1;
  }
}

bool k1r_eligible(const LayerDev& L) { return L.img != nullptr && L.max_tile_cols <= 128u && L.img_max_short <= kK1RMaxShort; }

void launch_k1r(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items_sorted, const uint32_t* start,
                float* cand, uint32_t items_per_block, hipStream_t s) {
    if (P.nrows == 0) return;
    if (!k1r_eligible(L) || X.dense || X.nnz == 0) fail("k1r: layer / queries not eligible");
    K1RArgs a;
    a.img = L.img; a.img_off = L.img_off; a.items = static_cast<const ItemDesc*>(items_sorted); a.start = start;
    a.xi = X.col_idx; a.xv = X.val; a.cand = cand;
    a.w_rows = L.w_rows; a.n_tiles = L.n_tiles; a.ch = std::max(16u, items_per_block);
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.has_bias = L.has_bias; a.bias_first = P.bias_first;
    const size_t lds = ((size_t)L.max_tile_img + 15) & ~(size_t)15;
    if (lds > 160 * 1024) fail("k1r: tile image exceeds the LDS");
    const uint64_t n_slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;   // upper bound of the sorted item count (known on the device only)
    const uint64_t blocks = (n_slots + a.ch - 1) / a.ch;
    if (blocks > 0x7FFFFFFFull) fail("k1r: grid too large; lower max_batch_rows");
    // one workgroup per CU when the image takes more than half of the LDS (16 wavefronts); two of 8 wavefronts otherwise
    const uint32_t threads = lds > 80 * 1024 ? 1024u : 512u;
    const int ppc = pp_class(P.pp);
    const int ts = L.img_max_short <= 2u ? 2 : (L.img_max_short <= 4u ? 4 : 8);
#define XRL_K1R_GO(TT, PP) do { \
        auto kern = &k1r_kernel<TT, PP>; \
        if (lds > 48 * 1024) XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(threads), lds, s, a); } while (0)
#define XRL_K1R_T(PP) do { if (ts == 2) XRL_K1R_GO(2, PP); else if (ts == 4) XRL_K1R_GO(4, PP); else XRL_K1R_GO(8, PP); } while (0)
    if (ppc) XRL_K1R_T(1); else XRL_K1R_T(0);
#undef XRL_K1R_T
#undef XRL_K1R_GO
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
