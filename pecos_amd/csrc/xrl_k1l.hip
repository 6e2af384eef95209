// K1L: the (query, tile) chunk products of a tile-format layer, tile-resident like K1R (xrl_k1r.hip) but with K1's own work
// decomposition -- LANE == ENTRY, accumulators in LDS -- and FOUR items per wavefront.
//
//   reference: w_ops<chunked>::compute_sparse_predictions (inference.hpp:925-1007) with its sort-by-chunk (:991-993),
//              chunk_ops<csr, bin_search> (:769-813), add_scaled_chunk_row_to_output_block (:506-518),
//              transform + combine (:1360-1384, PostProcessor :192-240)
//
// Why a second tile-resident kernel.  On this chip the leaf is bound by VALU ISSUE (profiles/r03_k1r_experiments.txt: K1 720
// vector instructions per item at 65 % VALU-busy; K1R 971 per item -- slower although it fetches nothing twice).  The lever is
// instructions per item, and the way to fewer of them is to let one instruction serve several items: 16 lanes per item, 4 items
// per wavefront, for the lookups (16 query features per item per round), the row walk (a UNIT = up to 32 entries of one hit row, two
// per lane) and the epilogue alike.  Everything an item reads besides its query row -- rank-bitmap, row extents, entries, bias --
// comes from the tile's image in LDS (xrl_model.cpp k1l_build_image); the image is loaded once per run of tile-sorted items.
//
// Per group of 16 lanes (one item):
//   round   16 query features: probe the rank-bitmap (ds_read_b64 + rank), hit lanes read their row extent and cut the row into units
//           {x, first entry | count << 20}; a 16-lane prefix sum places the units in the item's queue IN FEATURE ORDER
//   drain   the queue unit by unit (all 4 groups in lockstep, up to the longest queue): lane l applies entries l and l + 16 of the
//           unit -- distinct columns of one row -- to the item's accumulators in LDS: fl32(acc + fl32(x * w)), no fma
//   end     bias (last, or first under HASH_CHUNKED), fp64 transform, fp32 combine, child block written
// Units are drained in queue order and LDS operations of a wavefront execute in order, so every column receives its matched
// features in ascending feature order: the reference's summation order, bit for bit (same as K1).
#include <hip/hip_runtime.h>

#include "xrl_device.h"
#include "xrl_items.h"
#include "xrl_kernels.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

struct K1LArgs {
    const uint32_t* img; const uint64_t* img_off;   // tile images (LayerDev::limg)
    const ItemDesc* items;                           // tile-sorted, all active
    const uint32_t* start;                           // [n_tiles + 1] first sorted item of every tile
    const uint32_t* xi; const float* xv;             // CSR queries
    float* cand;
    uint32_t w_rows, n_tiles, ch;                    // ch: sorted items per workgroup
    uint32_t img_cap;                                // bytes of LDS reserved for the image
    uint32_t uq_cap, uq_drain;                       // unit queue: entries per item, fill level that triggers a drain
    uint32_t acc_stride;                             // floats per item accumulator block
    uint32_t wave_bytes;                             // LDS bytes per wavefront (4 queues + 4 accumulator blocks)
    int pp_kind, pp_p, first_layer, has_bias, bias_first;
};

constexpr int kG = 16;            // lanes per item
constexpr int kUnit = 32;         // entries per unit (two per lane)

template <int PPC>
__global__ void __launch_bounds__(1024) k1l_kernel(K1LArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t n_items = a.start[a.n_tiles];
    const uint32_t blk0 = blockIdx.x * a.ch;
    if (blk0 >= n_items) return;
    const uint32_t blk1 = min(n_items, blk0 + a.ch);
    const uint32_t nthreads = blockDim.x;
    const uint32_t nw = __builtin_amdgcn_readfirstlane(nthreads >> 6);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u, grp = lane >> 4, lig = lane & 15u;
    const uint32_t* __restrict__ xi = a.xi;
    const float* __restrict__ xv = a.xv;
    const uint32_t w_rows = a.w_rows;
    // this wavefront's scratch: 4 unit queues {x bits, first entry | count << 20}, then 4 accumulator blocks
    unsigned char* wbase = reinterpret_cast<unsigned char*>(smem) + a.img_cap + (size_t)wave * a.wave_bytes;
    uint2* __restrict__ my_uq = reinterpret_cast<uint2*>(wbase) + (size_t)grp * a.uq_cap;
    float* __restrict__ my_acc = reinterpret_cast<float*>(wbase + (size_t)4 * a.uq_cap * 8) + (size_t)grp * a.acc_stride;
    unsigned char* __restrict__ my_acc_b = reinterpret_cast<unsigned char*>(my_acc);

  for (uint32_t pos = blk0; pos < blk1;) {
    const uint32_t t = __builtin_amdgcn_readfirstlane(a.items[pos].tile);
    const uint32_t b1 = __builtin_amdgcn_readfirstlane(min(blk1, a.start[t + 1]));   // end of this tile's run inside the block
    __syncthreads();                                                    // the previous tile's readers are done
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.img + a.img_off[t]);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nq = (uint32_t)(a.img_off[t + 1] - a.img_off[t]) >> 2;
        uint32_t k = threadIdx.x;
        for (; k + 3u * nthreads < nq; k += 4u * nthreads) {
            const uint4 q0 = src[k], q1 = src[k + nthreads], q2 = src[k + 2u * nthreads], q3 = src[k + 3u * nthreads];
            dst[k] = q0; dst[k + nthreads] = q1; dst[k + 2u * nthreads] = q2; dst[k + 3u * nthreads] = q3;
        }
        for (; k < nq; k += nthreads) dst[k] = src[k];
    }
    __syncthreads();
    const uint32_t ncols = __builtin_amdgcn_readfirstlane(smem[2]);
    const uint2* __restrict__ bm_bits = reinterpret_cast<const uint2*>(smem + 12);
    const uint16_t* __restrict__ bm_rank = reinterpret_cast<const uint16_t*>(smem + __builtin_amdgcn_readfirstlane(smem[3]));
    const uint32_t* __restrict__ rowext = smem + __builtin_amdgcn_readfirstlane(smem[4]);
    const float* __restrict__ t_bias = reinterpret_cast<const float*>(smem + __builtin_amdgcn_readfirstlane(smem[5]));
    const uint2* __restrict__ ent = reinterpret_cast<const uint2*>(smem + __builtin_amdgcn_readfirstlane(smem[7]));   // {column * 4, value bits}

    for (uint32_t i0 = pos + wave * 4u; i0 < b1; i0 += nw * 4u) {
        // ---- this group's item
        const uint32_t ii = i0 + grp;
        const bool active = ii < b1;
        ItemDesc it = make_item(0u, kNoTile, 0u, 0.f, 0, 0u);
        if (active) it = a.items[ii];
        const uint32_t x_len = it.x_len;
        const uint64_t x_begin = it.x_begin;
        // accumulators: +0.0 (std::fill, inference.hpp:964) -- or the bias product first (chunk_ops<csr, hash>, :716-722)
        for (uint32_t c = lig; c < ncols; c += kG) my_acc[c] = (a.bias_first && a.has_bias) ? t_bias[c] : 0.0f;
        uint32_t max_len = x_len;
        max_len = max(max_len, (uint32_t)__shfl_xor((int)max_len, 16, 64));
        max_len = max(max_len, (uint32_t)__shfl_xor((int)max_len, 32, 64));
        max_len = __builtin_amdgcn_readfirstlane(max_len);
        uint32_t nu = 0;                                                // units in this group's queue

        auto drain = [&]() {
            uint32_t steps = nu;
            steps = max(steps, (uint32_t)__shfl_xor((int)steps, 16, 64));
            steps = max(steps, (uint32_t)__shfl_xor((int)steps, 32, 64));
            steps = __builtin_amdgcn_readfirstlane(steps);
            wave_sync_lds();                                            // the queue writes are visible
            // unit s of every group in lockstep; descriptor and entries of unit s + 1 are requested before unit s is applied
            uint2 u = nu ? my_uq[0] : make_uint2(0u, 0u);
            uint32_t st = u.y & 0xFFFFFu;
            uint2 e0 = ent[st + lig], e1 = ent[st + lig + kG];
            for (uint32_t s = 0; s < steps; ++s) {
                const uint2 un = (s + 1u < nu) ? my_uq[s + 1u] : make_uint2(0u, 0u);
                const uint32_t stn = un.y & 0xFFFFFu;
                const uint2 e0n = ent[stn + lig], e1n = ent[stn + lig + kG];
                const uint32_t cnt = (s < nu) ? (u.y >> 20) : 0u;
                const float x = __uint_as_float(u.x);
                // scalar * val, then add: no fma (inference.hpp:512-517); the two entries of a lane hold distinct columns
                if (lig < cnt) {
                    float* p0 = reinterpret_cast<float*>(my_acc_b + e0.x);
                    const float s0 = __fadd_rn(*p0, __fmul_rn(x, __uint_as_float(e0.y)));
                    if (lig + kG < cnt) {
                        float* p1 = reinterpret_cast<float*>(my_acc_b + e1.x);
                        const float s1 = __fadd_rn(*p1, __fmul_rn(x, __uint_as_float(e1.y)));
                        *p1 = s1;
                    }
                    *p0 = s0;
                }
                wave_sync_lds();                                        // the next unit may touch the same columns
                u = un; e0 = e0n; e1 = e1n;
            }
            nu = 0;
        };

        // ---- rounds of 16 query features per item
        uint32_t f = 0xFFFFFFFFu, vb = 0u;
        {
            const bool ok = lig < x_len;
            const uint64_t p = x_begin + (ok ? lig : 0u);
            f = xi[p]; vb = __float_as_uint(xv[p]);
            if (!ok) f = 0xFFFFFFFFu;
        }
        for (uint32_t c0 = 0; c0 < max_len; c0 += kG) {
            uint32_t fn = 0xFFFFFFFFu, vn = 0u;                         // the next round's features: in flight during this round
            if (c0 + kG < max_len) {
                const uint32_t idx = c0 + kG + lig;
                const bool ok = idx < x_len;
                const uint64_t p = x_begin + (ok ? idx : 0u);
                fn = xi[p]; vn = __float_as_uint(xv[p]);
                if (!ok) fn = 0xFFFFFFFFu;
            }
            // probe: is this lane's feature a row of the tile, and which
            const bool inr = f < w_rows;                                // also false on the padding lanes
            const uint32_t wq = inr ? (f >> 6) : 0u;
            const uint2 bw = bm_bits[wq];
            const uint32_t rk = bm_rank[wq];
            const unsigned long long b64 = ((unsigned long long)bw.y << 32) | bw.x;
            const uint32_t bpos = f & 63u;
            const bool hit = inr && ((b64 >> bpos) & 1ull);
            const uint32_t slot = rk + (uint32_t)__popcll(b64 & ((1ull << bpos) - 1ull));
            const uint32_t rext = rowext[hit ? slot : 0u];
            const uint32_t rstart = rext & 0xFFFFFu, rlen = hit ? (rext >> 20) + 1u : 0u;
            const uint32_t cnt = (rlen + (uint32_t)kUnit - 1u) / (uint32_t)kUnit;      // units of this hit (0 without one)
            uint32_t incl = cnt;                                        // prefix sum over the group's 16 lanes = feature order
#pragma unroll
            for (int d = 1; d < kG; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, kG); if (lig >= (uint32_t)d) incl += y; }
            const uint32_t base = nu + incl - cnt;
            for (uint32_t k = 0; k < cnt; ++k)
                my_uq[base + k] = make_uint2(vb, (rstart + k * (uint32_t)kUnit) | (min((uint32_t)kUnit, rlen - k * (uint32_t)kUnit) << 20));
            nu += (uint32_t)__shfl((int)incl, kG - 1, kG);
            if (__any(nu > a.uq_drain)) drain();
            f = fn; vb = vn;
        }
        drain();
        // ---- bias LAST (inference.hpp:806-811), transform in fp64, combine with the parent's score, write the child block
        if (active) {
            float* __restrict__ out = a.cand + it.out_off;
            for (uint32_t c = lig; c < ncols; c += kG) {
                float s = my_acc[c];
                if (a.has_bias && !a.bias_first) s = __fadd_rn(s, t_bias[c]);
                float v = pp_transform<PPC>(a.pp_kind, a.pp_p, s);
                if (!a.first_layer) v = pp_combine(a.pp_kind, v, it.pscore);
                out[c] = v;
            }
        }
        wave_sync_lds();
    }
    pos = b1;
  }
}

bool k1l_eligible(const LayerDev& L) { return L.limg != nullptr && L.max_tile_cols <= 128u; }

void launch_k1l(const LayerDev& L, const LayerPlan& P, const QueriesDev& X, const void* items_sorted, const uint32_t* start,
                float* cand, uint32_t items_per_block, hipStream_t s) {
    if (P.nrows == 0) return;
    if (!k1l_eligible(L) || X.dense || X.nnz == 0) fail("k1l: layer / queries not eligible");
    K1LArgs a;
    a.img = L.limg; a.img_off = L.limg_off; a.items = static_cast<const ItemDesc*>(items_sorted); a.start = start;
    a.xi = X.col_idx; a.xv = X.val; a.cand = cand;
    a.w_rows = L.w_rows; a.n_tiles = L.n_tiles; a.ch = std::max(16u, items_per_block);
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.has_bias = L.has_bias; a.bias_first = P.bias_first;
    a.img_cap = (L.max_tile_limg + 15u) & ~15u;
    // a round adds at most 16 hits x ceil(widest row / 32) units to a queue; the queue is drained once it holds more than uq_drain
    const uint32_t per_round = 16u * ((L.max_tile_cols + (uint32_t)kUnit - 1u) / (uint32_t)kUnit);
    a.uq_drain = 64u; a.uq_cap = a.uq_drain + per_round + 2u;
    a.acc_stride = (L.max_tile_cols + 4u) | 1u;
    a.wave_bytes = (uint32_t)((4u * a.uq_cap * 8u + 4u * a.acc_stride * 4u + 15u) & ~15u);
    const size_t lds_total = 160 * 1024;
    if (a.img_cap + a.wave_bytes > lds_total) fail("k1l: tile image exceeds the LDS");
    uint32_t nwv = (uint32_t)std::min<size_t>(16, (lds_total - a.img_cap) / a.wave_bytes);
    if (nwv >= 4) nwv &= ~3u;                                            // one to four wavefronts per SIMD
    const size_t lds = (size_t)a.img_cap + (size_t)nwv * a.wave_bytes;
    const uint64_t n_slots = (uint64_t)P.nrows * P.beam_in * L.max_tiles_per_parent;   // upper bound of the sorted item count (known on the device only)
    const uint64_t blocks = (n_slots + a.ch - 1) / a.ch;
    if (blocks > 0x7FFFFFFFull) fail("k1l: grid too large; lower max_batch_rows");
    const int ppc = pp_class(P.pp);
    auto go = [&](auto kern) {
        if (lds > 48 * 1024) XRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(64u * nwv), lds, s, a);
    };
    if (ppc) go(&k1l_kernel<1>); else go(&k1l_kernel<0>);
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
