// K1Q: one whole beam-search layer in ONE kernel for layers held in the DENSE row format -- prolongate
// (inference.hpp:1155-1219), chunk products (:769-839, 506-518), post-processor + combine (:192-240, 1360-1384),
// top-k with the positional tie-break (:1223-1298) and the child re-ordering (:1919-1923) -- query-stationary:
// a wavefront owns one query, its lanes own the query's candidate columns.
//
// Why a second row format.  K1 (xrl_kernels.hip) looks every query feature up in a per-tile rank-bitmap and
// gathers the matching 8-byte entry rows; on MI355X it is bound by the NUMBER of cache lines those gathers
// request from the L2 (profiles/): a probe line per (feature, tile), an extent line and 1..5 entry
// lines per hit, three dependent loads deep.  For the narrow chunks of the upper tree levels (nr_splits = 16
// children per parent) the weights of one feature for one chunk are 64 bytes when stored densely --
//      wd[feature][dense tile * Gp + column]     (f32 bits; kMissing = -0.0 where W has no entry)
// -- so ONE independent load per (feature, chunk) replaces probe + extent + entries, half a cache line each,
// and because lane == column the accumulators live in registers: no LDS traffic, no compaction, 2 VALU
// instructions per (feature, 64 candidates).  It costs rows x padded-columns x 4 bytes of HBM per layer
// (Amazon-670K level 3: 4.4 GB), which is what 288 GB are for; layers that do not fit (the leaf) stay in
// the sparse tile format and run K0 -> K1 -> K2.
//
// Arithmetic is the reference's, bit for bit: per candidate column, fl32(acc + fl32(x_f * w)) over the
// query's features in ascending order.  A column WITHOUT an entry at feature f holds -0.0: for finite x the
// product is a zero and leaves the accumulator as it is (accumulators are never -0.0), so the fast loop treats
// it like any weight; a 64-feature chunk holding a non-finite x runs the exact loop, which skips such cells on
// the marker -- explicit zeros stored in W and non-finite x behave exactly as in the reference's sparse walk.
// Bias last (sparse X) / first (dense X); transform in fp64; combine in fp32.
//
// Round 4: the query's (feature, value) pairs arrive through SCALAR loads, weight rows are buffer resources
// (no 64-bit vector addressing), 8 wavefronts per SIMD; layers that run UNSTAGED ask the layer's PRESENCE words
// first and never request an empty (feature, parent) segment; one query in 64 reports to the pruning feedback.
//
// Wavefront layout: candidate u = r*64 + lane (r < NS registers) <-> slot u >> log2(Gp) = (beam rank j,
// dense tile tt of that parent), column u & (Gp-1).  u is also the candidate's POSITION in the reference's
// order (beam rank major, child order minor), which is what ties are broken by.
//
// This header holds the device code; the kernel instantiations are spread over xrl_k1q_n*.hip (one translation unit per candidate-register
// bucket and post-processor class, so that `make -j` compiles them side by side) and the host side lives in xrl_k1q.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "xrl_device.h"
#include "xrl_kernels.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

// what K1Q needs of one layer (a compact copy of LayerDev's dense-format fields + the layer's plan)
struct K1QLayer {
    const uint32_t* wd; uint64_t d_ld;
    const uint32_t* pres; uint32_t pres_words;     // presence words (LayerDev::pres) or nullptr
    const uint32_t* d_ptile; const uint32_t* d_tcol; const float* bias_prod; const uint32_t* perm_inv;
    uint32_t d_gp_log2, d_max_tiles, n_parents, w_rows;
    uint32_t beam_in, k, ns;          // ns: candidate registers per lane this layer needs
    int has_bias, pp_kind, pp_p, first_layer, implicit_root;
    int prune;                        // exact bound pruning: score the first candidate register before requesting the others' weights
    int bias_first;                   // sparse X, HASH_CHUNKED arithmetic (inference.hpp:705-735): bias before the features, like dense X
    int layer_id;                     // index in the chain: the layer's slot in the pruning feedback counters
    int regular;                      // LayerDev::d_regular: dense tile = parent, first child = parent << d_gp_log2 (no d_ptile / d_tcol lookups)
};
constexpr int kK1QMaxLayers = 8;

struct K1QArgs {
    K1QLayer layer[kK1QMaxLayers];
    int n_layers;                     // consecutive dense-format layers run back to back by the same wavefront: the beam stays in LDS
    int fuse01;                       // layers 0 and 1 share one walk over the query's features: 1 = k1q_layer01 (a load per level), 2 = k1q_layer01m (merged rows: one load)
    const uint32_t* wd01; uint32_t wd01_c1;   // LayerDev::wd01 of the root layer
    QueriesDev X;
    const uint32_t* p_idx; const float* p_val; const uint32_t* p_cnt; uint32_t p_stride;
    uint32_t* out_idx; float* out_val; uint32_t* out_cnt; uint32_t out_stride;
    uint32_t row0, nrows;
    float prune_wmax;                 // the model's largest |weight| x max(1, |bias|): the pruning guard (prune_guard_ok, xrl_device.h)
    uint32_t* out_xok;                // non-null: the guard flag of every query is also written here (for a pruned tile-format layer that follows)
    const uint32_t* qperm; uint32_t xcd_per;   // sorted launch (launch_sort_queries): launch slot -> query; workgroups per XCD (block b runs on XCD b % 8 and takes
                                      // slot group (b % 8) * xcd_per + b / 8: every XCD walks a contiguous range of the sorted queries)
    uint32_t* fb_dev; uint32_t* fb_host;   // pruning feedback: sampled counters per layer {staged queries, of them: second pass needed} (device atomics);
                                      // the first wavefront of a launch copies what earlier launches counted to the host-visible words
};

// Weight rows as raw buffer loads.
// BIGW = false (every matrix of the launch < 4 GiB; round 5): the layer's whole matrix is ONE loop-invariant buffer resource and the
// feature's row is selected by the instruction's SCALAR offset f * row_bytes -- one scalar instruction (s_mul_i32) per row.  gfx9-class
// hardware range-checks a raw buffer access as  voffset >= num_records - soffset,  so num_records is the MATRIX's bytes: a lane offset of
// 0xFFFFFFF0 (how the presence path switches a lane off) still reads 0 without a memory request, any other lane offset used here stays
// inside its row.
// BIGW = true: the row's 64-bit base is rebuilt per feature (s_mul_hi, s_mul, s_add, s_addc, s_and), num_records = one row.
// Round 4 used the second form everywhere, with a clamp: 6 scalar instructions per weight load against 2 vector ones -- and a CU has ONE
// scalar unit (profiles/r05_k1q_scalar.md).
template <bool BIGW>
__device__ __forceinline__ uint32_t k1q_load_w(const uint32_t* wd, uint32_t row_bytes, uint32_t rows, uint32_t f, uint32_t voff) {
    if (BIGW) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(wd) + (uint64_t)f * row_bytes), 0, (int)row_bytes, 0x00020000);
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(wd), 0, (int)(rows * row_bytes), 0x00020000);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)(f * row_bytes), 0);
}

#ifndef XRL_K1Q_U1
#define XRL_K1Q_U1 16
#endif
#ifndef XRL_K1Q_U3
#define XRL_K1Q_U3 8
#endif
#ifndef XRL_K1Q_U2
#define XRL_K1Q_U2 XRL_K1Q_U3
#endif
#ifndef XRL_K1Q_RL
#define XRL_K1Q_RL 0     // 1: a batch's (feature id, value) pairs come from the chunk's per-lane registers (v_readlane) instead of scalar loads
#endif
template <int NS> struct K1QCfg {
    // weight rows (features) whose loads are in flight together: U * NS loads per lane
    static constexpr int U = NS <= 1 ? XRL_K1Q_U1 : NS <= 2 ? XRL_K1Q_U2 : NS <= 3 ? XRL_K1Q_U3 : NS <= 12 ? 4 : 2;
};


// PRESENCE MASKS (round 6).  Layers that run UNSTAGED on sparse X ask the layer's presence words before requesting weight segments (a third to
// a half of the (feature, parent) segments of such layers are empty).  Round 4 / 5 asked per batch, lane = candidate: a presence load, a wait,
// then the weight load -- two dependent round trips per batch of 8 features.  Now the question is asked once per 64-feature CHUNK with the
// lanes turned around: lane = chunk feature, and for each of the <= 4 beam slots of each candidate register (dense tiles of >= 16 columns) ONE
// load returns the presence word of (this lane's feature, the slot's tile) -- 4 independent loads per register, all lanes in flight together.
// Result: one bit per (register, slot) in a per-lane mask; inside the batch loop a feature's mask is one v_readlane and a lane whose slot holds
// no weight at the feature switches its weight load off (offset outside the resource: 0.0, no memory request).  A batch is ONE round trip.
// A function of its own (noinline): the unstaged kernels have no scalar registers to spare, and the 4 NR tile ids + offsets + shifts are
// allocated in the callee's frame instead of being spilled around the hot loop.
//   pres_rs: the layer's presence array as one buffer resource (pres_all bytes); fvo: this lane's feature x bytes per presence row; t0..t2: per lane,
//   the dense tile of the lane's candidate in registers RB, RB+1, RB+2 (slot q's tile sits in lane q << gl); Q = 64 >> gl slots per register
// Rows of <= 16 presence words (layers of <= 512 dense tiles: Amazon-670K's levels 2 and 3) are fetched WHOLE, one 64-byte row per lane in <= 4
// 16-byte loads (a lane's four loads share one cache line: 64 line requests per chunk, where one 4-byte load per slot asked for 64 x 12),
// and turned around through a wavefront-private LDS stage (word-major: conflict-free both ways) so that slot q's word -- the same word index
// for every lane -- is one ds_read.  Wider rows fall back to one 4-byte load per slot.
constexpr uint32_t kPresStageWords = 16;
__device__ __forceinline__ uint32_t k1q_presence_mask(__amdgpu_buffer_rsrc_t pres_rs, uint32_t pres_all, uint32_t pres_words, uint32_t fvo, uint32_t t0, uint32_t t1, uint32_t t2,
                                                      uint32_t gl, int nr, uint32_t* stage, int lane) {
    const uint32_t Q = 64u >> gl;
    uint32_t pm = 0u;
    const bool staged_rows = pres_words <= kPresStageWords;
    if (staged_rows) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((uint32_t)i * 4u < pres_words) {
                // (a row shorter than the 16 bytes asked for runs on into the next row -- or, at the array's end, out of range: zeros; those words are never selected)
                const u4 v = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(pres_rs, (int)(fvo + 16u * (uint32_t)i), 0, 0));
                stage[(4 * i + 0) * 64 + lane] = v.x; stage[(4 * i + 1) * 64 + lane] = v.y; stage[(4 * i + 2) * 64 + lane] = v.z; stage[(4 * i + 3) * 64 + lane] = v.w;
            }
        wave_sync_lds();
    }
    for (int r = 0; r < nr; ++r) {
        const uint32_t tiles = r == 0 ? t0 : r == 1 ? t1 : t2;
        uint32_t pw[4], tq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tq[q] = (uint32_t)__builtin_amdgcn_readlane((int)tiles, (int)(((uint32_t)q << gl) & 63u));
            if (staged_rows) pw[q] = stage[(min(tq[q] >> 5, kPresStageWords - 1u)) * 64u + (uint32_t)lane];
            // (slots a register does not have -- 32-column tiles: 2 per register -- address outside the resource: no request, bit 0;
            //  gfx950 treats voffset + soffset >= num_records as out of range: scripts/buffer_range_probe.hip)
            else pw[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pres_rs, (int)fvo, (int)((uint32_t)q < Q ? (tq[q] >> 5) * 4u : pres_all), 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) pm |= ((uint32_t)q < Q ? ((pw[q] >> (tq[q] & 31u)) & 1u) : 0u) << (r * 4 + q);
    }
    if (staged_rows) wave_sync_lds();                                  // (the stage is rewritten by the next chunk)
    return pm;
}

// One layer for one query (one wavefront): beam in s_bidx / s_bval[0..cnt) -> beam out in the same arrays; returns the new count.
// BIASF: the accumulators start at the bias product (dense X; sparse X under HASH_CHUNKED, inference.hpp:716-722) instead of receiving it
// last -- a compile-time switch: as a run-time one it cost the widest kernel 8 VGPRs and a wavefront per SIMD
// PRES: the instantiation carries the presence-word path (layers that run UNSTAGED on sparse X -- every beam parent's segments are
// requested, a third to a half of them empty); the staged default does not pay its registers (4.15 vs 4.49 ms on Amazon-670K)
template <int NS, int PPC, bool DENSEX, bool BIASF, bool PRES, bool BIGW>
__device__ __forceinline__ uint32_t k1q_layer(const K1QLayer& Ly, const QueriesDev& X, uint64_t xrow, uint32_t cnt_in,
                                               uint32_t* s_bidx, float* s_bval, uint2* sc, int lane, float wmax, uint32_t& fbm, uint32_t* pstage) {
    // ---- prolongate: which (parent, dense tile, column) does each of this lane's candidates stand for
    const uint32_t gl = Ly.d_gp_log2, gmask = (1u << gl) - 1u, TT = Ly.d_max_tiles;
    const uint32_t cnt = Ly.implicit_root ? 1u : min(cnt_in, Ly.beam_in);
    uint32_t woff[NS], child[NS]; float ps[NS], acc[NS]; bool valid[NS];
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        const uint32_t u = (uint32_t)r * 64u + (uint32_t)lane;
        const uint32_t slot = u >> gl, col = u & gmask;
        const uint32_t j = TT == 1u ? slot : slot / TT;
        const uint32_t tt = TT == 1u ? 0u : slot - j * TT;
        bool v = j < cnt;
        uint32_t parent = 0; float pscore = 1.0f;
        if (!Ly.implicit_root) { parent = s_bidx[v ? j : 0u]; pscore = s_bval[v ? j : 0u]; }
        v = v && parent < Ly.n_parents;
        if (!v) parent = 0;
        uint32_t dtc, cb;
        if (Ly.regular) { dtc = parent; cb = parent << gl; }            // one full tile per parent: nothing to look up (and no dependent round trips before the first weight load)
        else {
            const uint32_t dt = Ly.d_ptile[parent] + tt;
            v = v && dt < Ly.d_ptile[parent + 1];
            dtc = v ? dt : 0u;
            cb = Ly.d_tcol[dtc];
            const uint32_t ce = Ly.d_tcol[dtc + 1];
            v = v && col < ce - cb;
        }
        woff[r] = v ? ((dtc << gl) + col) * 4u : 0u;                   // BYTE offset inside a feature row (d_ld < 2^30)
        child[r] = v ? cb + col : 0u;
        ps[r] = pscore; valid[r] = v;
        // dense queries: bias FIRST (inference.hpp:824-830); bias_prod holds fl32(bias * w) or +0.0
        acc[r] = (BIASF && Ly.has_bias) ? Ly.bias_prod[child[r]] : 0.0f;
    }
    // bound pruning: score of the first beam parent that has no candidate in register 0 (the beam is sorted best first)
    const uint32_t j_next = (64u >> gl) / TT;
    const bool prune_next_ok = Ly.prune && !Ly.first_layer && !Ly.implicit_root && Ly.pp_kind != PP_NOOP;
    const bool prune_all_in_first = j_next >= cnt;                    // every parent's candidates sit in register 0 already
    // (a multiplying combiner keeps a child of a parent with a NEGATIVE score -- possible when an earlier layer used another
    //  post-processor -- inside [score, 0]: the bound is then max(score, 0); the adding ones add a transform <= 0)
    float ps_next = (prune_next_ok && !prune_all_in_first) ? s_bval[j_next] : 0.0f;
    if (Ly.pp_kind == PP_SIGMOID || Ly.pp_kind == PP_LP_HINGE) ps_next = fmaxf(ps_next, 0.0f);
    if (!(ps_next == ps_next)) ps_next = INFINITY;                    // a NaN parent score proves nothing: no pruning
    wave_sync_lds();                                                   // the beam has been read: the arrays may be overwritten below

    const uint32_t* __restrict__ wd = Ly.wd;
    const uint32_t ld = (uint32_t)(Ly.d_ld * 4u);                       // bytes per feature row
    const uint32_t w_rows = Ly.w_rows;
    // presence words: dense tile of a lane's byte offset = woff >> (gl + 2); its word's byte offset in the presence row = (tile >> 5) * 4
    const uint32_t* __restrict__ pres = Ly.pres;
    const uint32_t pres_bytes = Ly.pres_words * 4u, dt_shift = gl + 2u;
    const __amdgpu_buffer_rsrc_t pres_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(PRES && pres ? pres : wd), 0, (int)(PRES && pres ? (w_rows + 1u) * pres_bytes : 0u), 0x00020000);
    uint32_t xmx = 0u, xn = 0u;                                        // pruning guard: largest |x| bits this lane has seen, features of the query

    // One pass over the query's features for the candidate registers [RB, RE): U features per batch, their U*(RE-RB) weight loads issued
    // together, then applied in feature order.  Round 4 ("K1Q diet"):
    //  * the (feature id, value) pairs of the query are read with SCALAR loads (the row is wavefront-uniform): no v_readlane broadcast;
    //  * the weight row of a feature is a BUFFER resource (base = wd + f * ld, in SGPRs, rebuilt per feature with scalar arithmetic),
    //    the lane's 32-bit byte offset its VGPR operand: no 64-bit vector address arithmetic;
    //  * cells without a weight hold -0.0 (kMissing): with finite x the fast loop is `acc + x * w` for every cell (2 vector instructions
    //    per (feature, register), packed in pairs by the compiler); a 64-feature chunk that holds a NON-FINITE x, and the last rows of X
    //    (whose tail batch may not read past the array), run the exact loop, which skips cells on the marker like the reference's row walk.
    auto pass = [&](auto rb_tag, auto re_tag) {
        constexpr int RB = decltype(rb_tag)::value, RE = decltype(re_tag)::value, NR = RE - RB;
        constexpr int UU = K1QCfg<NR>::U;
        // presence masks (k1q_presence_mask): one bit per (register of this pass, beam slot), per lane = feature of the current chunk
        const bool pmask_ok = PRES && pres != nullptr && NR <= 3;
        uint32_t pm = 0u, mybit[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) mybit[r] = 1u << (((uint32_t)r * 4u + ((uint32_t)lane >> gl)) & 31u);
        auto body_f = [&](auto exact_tag, auto&& getf, auto&& getx, uint32_t tl) {   // getf(u) / getx(u): feature id and value of the batch's u-th feature; tl: its slot in the 64-feature chunk
            constexpr bool EX = decltype(exact_tag)::value;
            uint32_t wb[UU][NR];
            if (PRES && !EX && pmask_ok) {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const uint32_t fu = getf(u);
                    const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)pm, (int)((tl + (uint32_t)u) & 63u));
#pragma unroll
                    for (int r = 0; r < NR; ++r) wb[u][r] = k1q_load_w<BIGW>(wd, ld, w_rows + 1u, fu, (sm & mybit[r]) ? woff[RB + r] : 0xFFFFFFF0u);
                }
            } else {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    // (the feature id is <= w_rows here: padding slots and features outside the layer name the all-missing row the model compiler appends)
                    const uint32_t fu = getf(u);
#pragma unroll
                    for (int r = 0; r < NR; ++r) wb[u][r] = k1q_load_w<BIGW>(wd, ld, w_rows + 1u, fu, woff[RB + r]);
                }
            }
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                const float xu = getx(u);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    // scalar * val, then add: no fma (inference.hpp:512-517); no entry -> no operation
                    const float sm = __fadd_rn(acc[RB + r], __fmul_rn(xu, __uint_as_float(wb[u][r])));
                    acc[RB + r] = (EX && wb[u][r] == kMissing) ? acc[RB + r] : sm;
                }
            }
        };
        auto body = [&](auto exact_tag, const uint32_t (&fs)[UU], const float (&xs)[UU], uint32_t tl) {
            body_f(exact_tag, [&](int u) { return fs[u]; }, [&](int u) { return xs[u]; }, tl);
        };
        // the query's (feature, value) pairs: its CSR row (chunk_ops<csr, bin_search>, inference.hpp:769-813: ascending features), or every
        // chunk row except the bias row with x gathered by row id (chunk_ops<drm, bin_search>, :815-839)
        const uint32_t* __restrict__ fsrc = nullptr; const float* __restrict__ vsrc; uint32_t n; uint64_t room;
        if (DENSEX) {
            vsrc = X.val + xrow * X.cols;
            n = Ly.has_bias ? w_rows - 1u : w_rows;
            room = ((uint64_t)X.rows - xrow) * X.cols;                    // floats readable from vsrc[0]
        } else {
            const uint64_t xb = X.row_ptr[xrow];
            n = (uint32_t)(X.row_ptr[xrow + 1] - xb);
            fsrc = X.col_idx + xb; vsrc = X.val + xb;
            room = X.nnz - xb;
        }
        xn = n;
        constexpr uint32_t CH = 64u;
        for (uint32_t t0 = 0; t0 < n; t0 += CH) {
            const uint32_t nc = min(CH, n - t0);
            // the chunk's values once per lane: the pruning guard's maximum, and "is every value finite"
            const uint32_t xraw = (uint32_t)lane < nc ? __float_as_uint(vsrc[t0 + (uint32_t)lane]) : 0u;
            const uint32_t vb = xraw & 0x7FFFFFFFu;
            xmx = max(xmx, vb);
            // the fast loops take a feature id as it is (no clamp per feature): a chunk that holds an id beyond the layer's rows takes the exact
            // loop below, which sends such features to the all-missing row like the reference's row lookup finds nothing for them
            const uint32_t ib = (!DENSEX && (uint32_t)lane < nc) ? fsrc[t0 + (uint32_t)lane] : 0u;
            const bool nonfinite = __ballot(vb >= 0x7F800000u || ib > w_rows) != 0ull;
            if (pmask_ok && !nonfinite)   // (ib <= w_rows here; lanes past the chunk's end hold feature 0: their masks are never applied to a weight that counts)
                pm = k1q_presence_mask(pres_rs, (w_rows + 1u) * pres_bytes, Ly.pres_words, ib * pres_bytes, woff[RB] >> dt_shift, woff[RB + (NR > 1 ? 1 : 0)] >> dt_shift,
                                       woff[RB + (NR > 2 ? 2 : 0)] >> dt_shift, gl, NR < 3 ? NR : 3, pstage, lane);
            for (uint32_t t = t0; t < t0 + nc; t += (uint32_t)UU) {
                uint32_t fs[UU]; float xs[UU];
                if (!nonfinite && t + (uint32_t)UU <= t0 + nc) {             // a full batch: plain uniform loads
                    if (XRL_K1Q_RL && !DENSEX) {
                        // the batch's (feature id, value) pairs straight from the chunk's per-lane registers, each at its point of use: no
                        // scalar-load round trip per batch and no SGPR arrays alive across the loads
                        const int tl = (int)(t - t0);
                        body_f(std::false_type{}, [&](int u) { return (uint32_t)__builtin_amdgcn_readlane((int)ib, tl + u); },
                               [&](int u) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)xraw, tl + u)); }, (uint32_t)tl);
                    } else {
#pragma unroll
                        for (int u = 0; u < UU; ++u) { fs[u] = DENSEX ? t + (uint32_t)u : fsrc[t + (uint32_t)u]; xs[u] = vsrc[t + (uint32_t)u]; }
                        body(std::false_type{}, fs, xs, t - t0);
                    }
                } else if (!nonfinite && (uint64_t)t + (uint32_t)UU <= room) {  // the row's tail: the loads run on into the next row, the slots past the end are neutralised
#pragma unroll
                    for (int u = 0; u < UU; ++u) {
                        const bool ok = t + (uint32_t)u < n;
                        const uint32_t f = DENSEX ? t + (uint32_t)u : fsrc[t + (uint32_t)u];
                        const float x = vsrc[t + (uint32_t)u];
                        fs[u] = ok ? f : w_rows; xs[u] = ok ? x : 0.0f;
                    }
                    body(std::false_type{}, fs, xs, t - t0);
                } else {                                                      // non-finite x in the chunk, or the end of the X arrays: clamped loads, exact loop
#pragma unroll
                    for (int u = 0; u < UU; ++u) {
                        const bool ok = t + (uint32_t)u < n;
                        const uint32_t ic = ok ? t + (uint32_t)u : n - 1u;
                        const uint32_t f = DENSEX ? ic : fsrc[ic];
                        const float x = vsrc[ic];
                        fs[u] = ok ? min(f, w_rows) : w_rows; xs[u] = ok ? x : 0.0f;
                    }
                    body(std::true_type{}, fs, xs, t - t0);
                }
            }
        }
    };
    // bias last (sparse X, inference.hpp:806-811), transform in fp64, combine with the parent's score
    uint32_t key[NS], sbits[NS];
    auto finish = [&](int r) -> float {
        float sm = acc[r];
        if (!BIASF && Ly.has_bias) sm = __fadd_rn(sm, Ly.bias_prod[child[r]]);
        float v = pp_transform<PPC>(Ly.pp_kind, Ly.pp_p, sm);
        if (!Ly.first_layer) v = pp_combine(Ly.pp_kind, v, ps[r]);
        sbits[r] = __float_as_uint(v);
        key[r] = valid[r] ? score_key(v) : 0u;
        return v;
    };
    // EXACT bound pruning (option prune): with a combiner a child's score is <= its parent's (transform <= 1 times, or <= 0 plus, the
    // parent's score) and a later candidate loses ties by position -- so when k candidates of the FIRST register (the best beam
    // parents) already score >= the score of the first parent outside it, no candidate of the other registers can enter the top-k:
    // their weight rows are never requested.  Otherwise a second pass scores them; either way the selection below is the reference's.
    bool staged = false;
    if (NS > 1 && prune_next_ok) {
        staged = true;
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        const float v0 = finish(0);
        const uint32_t cge = (uint32_t)__popcll(__ballot(valid[0] && v0 >= ps_next));
        // (the guard: a query that could produce a NaN score -- non-finite or huge x, non-finite weights -- is never pruned)
        const bool xok = prune_guard_ok(wave_max_u32(xmx), xn, wmax);
        const bool second = !prune_all_in_first && (cge < Ly.k || !xok);
#ifndef XRL_K1Q_NOFB
        if (!prune_all_in_first) fbm |= (1u | (second ? 0x10000u : 0u)) << Ly.layer_id;   // pruning feedback: staged here / second pass needed (counted once, at the end of the kernel)
#endif
        if (second) {
            pass(std::integral_constant<int, (NS > 1 ? 1 : 0)>{}, std::integral_constant<int, NS>{});
#pragma unroll
            for (int r = 1; r < NS; ++r) finish(r);
        } else {
#pragma unroll
            for (int r = 1; r < NS; ++r) { key[r] = 0u; sbits[r] = 0u; }
        }
    }
    if (!staged) {
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NS>{});
#pragma unroll
        for (int r = 0; r < NS; ++r) finish(r);
    }
    // ---- top-k (value desc, position asc) and reorder_prediction: the next beam, best first
    uint32_t rank, sb, ch;
    const uint32_t kk = wave_topk<NS>(key, sbits, child, Ly.k, sc, lane, rank, sb, ch);
    if ((uint32_t)lane < kk) {
        s_bidx[rank] = Ly.perm_inv ? Ly.perm_inv[ch] : ch;
        s_bval[rank] = __uint_as_float(sb);
    }
    wave_sync_lds();
    return kk;
}

// Levels 0 and 1 in ONE pass over the query's features (sparse X, fused launches).  When the root layer keeps every one of its K0
// children (K0 <= its k), the next layer always evaluates ALL of their children: which weight columns level 1 reads does not depend
// on level 0's scores, only the ORDER of the parents (= candidate positions, the tie-break) and the parents' scores do.  So both
// layers' accumulators are filled by the same feature walk -- lanes [0, K0) hold level 0's columns, every lane holds one level-1
// candidate of the parents taken in column order -- and afterwards level 0 is ranked, the level-1 scores are combined with their
// parent's score and moved to the lane their reference position names (parents in rank order), where the usual top-k runs.
// Saves one of the two latency-bound feature walks of the narrow top levels.
template <int PPC, bool BIASF, bool BIGW>
__device__ __forceinline__ uint32_t k1q_layer01(const K1QLayer& L0, const K1QLayer& L1, const QueriesDev& X, uint64_t xrow,
                                                 uint32_t* s_bidx, float* s_bval, uint2* sc, int lane) {
#ifndef XRL_K1Q_U01
#define XRL_K1Q_U01 8
#endif
    constexpr int UU = XRL_K1Q_U01;
    // ---- level 0: lane c < K0 <-> child c of the root (one dense tile at offset 0)
    const uint32_t K0 = L0.d_tcol[1] - L0.d_tcol[0];
    const bool v0 = (uint32_t)lane < K0;
    const uint32_t woff0 = v0 ? (uint32_t)lane * 4u : 0u;
    const uint32_t child0 = v0 ? L0.d_tcol[0] + (uint32_t)lane : 0u;
    const uint32_t orig0 = v0 ? (L0.perm_inv ? L0.perm_inv[child0] : child0) : 0xFFFFFFFFu;
    float acc0 = (BIASF && L0.has_bias && v0) ? L0.bias_prod[child0] : 0.0f;
    // ---- level 1: candidate u = lane of the parents in COLUMN order (virtual beam slot j = level-0 column j)
    const uint32_t gl = L1.d_gp_log2, gmask = (1u << gl) - 1u, TT = L1.d_max_tiles;
    const uint32_t slot = (uint32_t)lane >> gl, col = (uint32_t)lane & gmask;
    const uint32_t j = TT == 1u ? slot : slot / TT, tt = TT == 1u ? 0u : slot - j * TT;
    bool v1 = j < K0;
    uint32_t parent = (uint32_t)__shfl((int)orig0, (int)(v1 ? j : 0u), 64);
    v1 = v1 && parent < L1.n_parents;
    if (!v1) parent = 0;
    const uint32_t dt = L1.d_ptile[parent] + tt;
    v1 = v1 && dt < L1.d_ptile[parent + 1];
    const uint32_t dtc = v1 ? dt : 0u;
    const uint32_t cb = L1.d_tcol[dtc], ce = L1.d_tcol[dtc + 1];
    v1 = v1 && col < ce - cb;
    const uint32_t woff1 = v1 ? ((dtc << gl) + col) * 4u : 0u;
    const uint32_t child1 = v1 ? cb + col : 0u;
    float acc1 = (BIASF && L1.has_bias && v1) ? L1.bias_prod[child1] : 0.0f;

    // ---- one walk over the query's features, UU at a time: 2 * UU loads in flight (scalar feature loads, buffer-resource rows, fast /
    //      exact loops: see k1q_layer's pass)
    const uint32_t* __restrict__ wd0 = L0.wd; const uint32_t* __restrict__ wd1 = L1.wd;
    const uint32_t ld0 = (uint32_t)(L0.d_ld * 4u), ld1 = (uint32_t)(L1.d_ld * 4u);   // bytes per feature row
    const uint32_t wr = L0.w_rows;                                      // == L1.w_rows (launch_k1q fuses the two levels only then)
    const uint64_t xb = X.row_ptr[xrow];
    const uint32_t xl = (uint32_t)(X.row_ptr[xrow + 1] - xb);
    const uint32_t* __restrict__ xi = X.col_idx + xb;
    const float* __restrict__ xv = X.val + xb;
    const uint64_t room = X.nnz - xb;
    auto body = [&](auto exact_tag, const uint32_t (&fs)[UU], const float (&xs)[UU]) {
        constexpr bool EX = decltype(exact_tag)::value;
        uint32_t w0[UU], w1[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            w0[u] = k1q_load_w<BIGW>(wd0, ld0, wr + 1u, fs[u], woff0);
            w1[u] = k1q_load_w<BIGW>(wd1, ld1, wr + 1u, fs[u], woff1);
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const float s0 = __fadd_rn(acc0, __fmul_rn(xs[u], __uint_as_float(w0[u])));
            acc0 = (EX && w0[u] == kMissing) ? acc0 : s0;
            const float s1 = __fadd_rn(acc1, __fmul_rn(xs[u], __uint_as_float(w1[u])));
            acc1 = (EX && w1[u] == kMissing) ? acc1 : s1;
        }
    };
    for (uint32_t t0 = 0; t0 < xl; t0 += 64u) {
        const uint32_t nc = min(64u, xl - t0);
        const uint32_t vb = (uint32_t)lane < nc ? (__float_as_uint(xv[t0 + (uint32_t)lane]) & 0x7FFFFFFFu) : 0u;
        const uint32_t ib = (uint32_t)lane < nc ? xi[t0 + (uint32_t)lane] : 0u;
        const bool nonfinite = __ballot(vb >= 0x7F800000u || ib > wr) != 0ull;   // (or a feature id beyond the layers' rows: the exact loop clamps it, see k1q_layer)
        for (uint32_t t = t0; t < t0 + nc; t += (uint32_t)UU) {
            uint32_t fs[UU]; float xs[UU];
            if (!nonfinite && t + (uint32_t)UU <= t0 + nc) {
#pragma unroll
                for (int u = 0; u < UU; ++u) { fs[u] = xi[t + (uint32_t)u]; xs[u] = xv[t + (uint32_t)u]; }
                body(std::false_type{}, fs, xs);
            } else if (!nonfinite && (uint64_t)t + (uint32_t)UU <= room) {
#pragma unroll
                for (int u = 0; u < UU; ++u) { const bool ok = t + (uint32_t)u < xl; const uint32_t f = xi[t + (uint32_t)u]; const float x = xv[t + (uint32_t)u]; fs[u] = ok ? f : wr; xs[u] = ok ? x : 0.0f; }
                body(std::false_type{}, fs, xs);
            } else {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const bool ok = t + (uint32_t)u < xl; const uint32_t ic = ok ? t + (uint32_t)u : xl - 1u;
                    const uint32_t f = xi[ic]; const float x = xv[ic];
                    fs[u] = ok ? min(f, wr) : wr; xs[u] = ok ? x : 0.0f;
                }
                body(std::true_type{}, fs, xs);
            }
        }
    }
    // ---- level 0: bias, transform (first layer: no combine), rank of every node in (value desc, position asc) order
    if (!BIASF && L0.has_bias && v0) acc0 = __fadd_rn(acc0, L0.bias_prod[child0]);
    const float s0v = pp_transform<PPC>(L0.pp_kind, L0.pp_p, acc0);
    const uint32_t k0key = v0 ? score_key(s0v) : 0u;
    uint32_t rank0 = 0;
    for (uint32_t c = 0; c < K0; ++c) {
        const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)k0key, (int)c);
        rank0 += (kc > k0key || (kc == k0key && c < (uint32_t)lane)) ? 1u : 0u;
    }
    // ---- level 1: bias, transform, combine with the parent's score; candidate position = (rank of the parent, tile, column)
    if (!BIASF && L1.has_bias && v1) acc1 = __fadd_rn(acc1, L1.bias_prod[child1]);
    float s1v = pp_transform<PPC>(L1.pp_kind, L1.pp_p, acc1);
    const float psv = __shfl(s0v, (int)(j < K0 ? j : 0u), 64);
    if (!L1.first_layer) s1v = pp_combine(L1.pp_kind, s1v, psv);
    const uint32_t prank = (uint32_t)__shfl((int)rank0, (int)(j < K0 ? j : 0u), 64);
    const uint32_t position = (((prank * TT) + tt) << gl) + col;       // < 64: one candidate register
    // move every candidate to the lane its position names (slots no candidate names stay marked empty)
    sc[lane] = make_uint2(0u, 0xFFFFFFFFu);
    wave_sync_lds();
    if (v1) sc[position] = make_uint2(__float_as_uint(s1v), child1);
    wave_sync_lds();
    const uint2 mine = sc[lane];
    wave_sync_lds();
    uint32_t key[1], sbits[1], payload[1];
    sbits[0] = mine.x; payload[0] = mine.y;
    key[0] = mine.y != 0xFFFFFFFFu ? score_key(__uint_as_float(mine.x)) : 0u;
    uint32_t rank, sb, ch;
    const uint32_t kk = wave_topk<1>(key, sbits, payload, L1.k, sc, lane, rank, sb, ch);
    if ((uint32_t)lane < kk) {
        s_bidx[rank] = L1.perm_inv ? L1.perm_inv[ch] : ch;
        s_bval[rank] = __uint_as_float(sb);
    }
    wave_sync_lds();
    return kk;
}


// The same with ONE load per feature: the root layer carries both levels' dense rows side by side (LayerDev::wd01: 64 columns per feature,
// level 1's row at its own offsets in [0, c1), level 0's K0 columns behind it), lanes [0, c1) hold the level-1 candidates and lanes
// [c1, c1 + K0) the level-0 columns -- one accumulator, one transform pass when the two levels share their post-processor.  Half the loads,
// multiplies and adds of k1q_layer01, one fp64 transform instead of two.
template <int PPC, bool BIASF>
__device__ __forceinline__ uint32_t k1q_layer01m(const K1QLayer& L0, const K1QLayer& L1, const uint32_t* __restrict__ wd01, uint32_t c1, const QueriesDev& X, uint64_t xrow,
                                                  uint32_t* s_bidx, float* s_bval, uint2* sc, int lane) {
    constexpr int UU = 16;
    const uint32_t K0 = L0.d_tcol[1] - L0.d_tcol[0];
    // ---- level 0: lane c1 + c <-> child c of the root
    const uint32_t l0 = (uint32_t)lane - c1;
    const bool v0 = l0 < K0;
    const uint32_t child0 = v0 ? L0.d_tcol[0] + l0 : 0u;
    const uint32_t orig0 = v0 ? (L0.perm_inv ? L0.perm_inv[child0] : child0) : 0xFFFFFFFFu;
    // ---- level 1: candidate u = lane < c1 of the parents taken in COLUMN order (virtual beam slot j = level-0 column j)
    const uint32_t gl = L1.d_gp_log2, gmask = (1u << gl) - 1u, TT = L1.d_max_tiles;
    const uint32_t slot = (uint32_t)lane >> gl, col = (uint32_t)lane & gmask;
    const uint32_t j = TT == 1u ? slot : slot / TT, tt = TT == 1u ? 0u : slot - j * TT;
    bool v1 = (uint32_t)lane < c1 && j < K0;
    uint32_t parent = (uint32_t)__shfl((int)orig0, (int)(c1 + (v1 ? j : 0u)), 64);
    v1 = v1 && parent < L1.n_parents;
    if (!v1) parent = 0;
    uint32_t dtc, cb;
    if (L1.regular) { dtc = parent; cb = parent << gl; }
    else {
        const uint32_t dt = L1.d_ptile[parent] + tt;
        v1 = v1 && dt < L1.d_ptile[parent + 1];
        dtc = v1 ? dt : 0u;
        cb = L1.d_tcol[dtc];
        const uint32_t ce = L1.d_tcol[dtc + 1];
        v1 = v1 && col < ce - cb;
    }
    const uint32_t child1 = v1 ? cb + col : 0u;
    // this lane's column of the merged row; lanes that hold neither a level-1 candidate nor a level-0 column address outside the resource (no request)
    const uint32_t woff = v1 ? ((dtc << gl) + col) * 4u : v0 ? (c1 + l0) * 4u : 0xFFFFFFF0u;
    const float bp1 = (v1 && L1.has_bias) ? L1.bias_prod[child1] : 0.0f, bp0 = (v0 && L0.has_bias) ? L0.bias_prod[child0] : 0.0f;
    const float bp = v1 ? bp1 : bp0;
    float acc = BIASF ? bp : 0.0f;

    const uint32_t wr = L0.w_rows;                                      // == L1.w_rows
    const uint64_t xb = X.row_ptr[xrow];
    const uint32_t xl = (uint32_t)(X.row_ptr[xrow + 1] - xb);
    const uint32_t* __restrict__ xi = X.col_idx + xb;
    const float* __restrict__ xv = X.val + xb;
    // One 64-feature chunk at a time in the lanes (lane t holds feature t0 + t); a batch takes its UU (feature id, value) pairs by v_readlane at
    // the point of use -- no scalar loads, no register arrays -- and slots past the row's end name the all-missing row with x = 0, so the last
    // batch needs no path of its own.  A chunk with a non-finite x or an id beyond the layers' rows runs the exact loop (skips cells on the marker).
    auto body = [&](auto exact_tag, uint32_t ib, uint32_t xr, uint32_t tl) {
        constexpr bool EX = decltype(exact_tag)::value;
        uint32_t w[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) w[u] = k1q_load_w<false>(wd01, 256u, wr + 1u, (uint32_t)__builtin_amdgcn_readlane((int)ib, (int)(tl + (uint32_t)u)), woff);
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const float x = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)xr, (int)(tl + (uint32_t)u)));
            const float sm = __fadd_rn(acc, __fmul_rn(x, __uint_as_float(w[u])));
            acc = (EX && w[u] == kMissing) ? acc : sm;
        }
    };
    for (uint32_t t0 = 0; t0 < xl; t0 += 64u) {
        const uint32_t nc = min(64u, xl - t0);
        const uint32_t xr = (uint32_t)lane < nc ? __float_as_uint(xv[t0 + (uint32_t)lane]) : 0u;
        const uint32_t ir = (uint32_t)lane < nc ? xi[t0 + (uint32_t)lane] : wr;
        const bool nonfinite = __ballot((xr & 0x7FFFFFFFu) >= 0x7F800000u || ir > wr) != 0ull;
        const uint32_t ib = min(ir, wr);                                  // (features beyond the layers' rows: the all-missing row, like the reference's lookup finds nothing)
        for (uint32_t tl = 0; tl < nc; tl += (uint32_t)UU) {
            if (!nonfinite) body(std::false_type{}, ib, xr, tl);
            else body(std::true_type{}, ib, xr, tl);
        }
    }
    // ---- bias last, transform (one pass when both levels share the post-processor)
    if (!BIASF) acc = __fadd_rn(acc, bp);      // (bp is +0.0 where the level has no bias or the column no bias entry: leaves the accumulator as it is, like the `if` of k1q_layer01 -- accumulators are never -0.0)
    float sv;
    if (L0.pp_kind == L1.pp_kind && L0.pp_p == L1.pp_p) sv = pp_transform<PPC>(L1.pp_kind, L1.pp_p, acc);
    else { const float a1 = pp_transform<PPC>(L1.pp_kind, L1.pp_p, acc), a0 = pp_transform<PPC>(L0.pp_kind, L0.pp_p, acc); sv = v1 ? a1 : a0; }
    // ---- level 0 (first layer: no combine): rank of every node in (value desc, position asc) order, on lanes [c1, c1 + K0)
    const uint32_t k0key = v0 ? score_key(sv) : 0u;
    uint32_t rank0 = 0;
    for (uint32_t c = 0; c < K0; ++c) {
        const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)k0key, (int)(c1 + c));
        rank0 += (kc > k0key || (kc == k0key && c < l0)) ? 1u : 0u;
    }
    // ---- level 1: combine with the parent's score; candidate position = (rank of the parent, tile, column)
    const uint32_t jsrc = c1 + (j < K0 ? j : 0u);
    const float psv = __shfl(sv, (int)jsrc, 64);
    float s1v = sv;
    if (!L1.first_layer) s1v = pp_combine(L1.pp_kind, sv, psv);
    const uint32_t prank = (uint32_t)__shfl((int)rank0, (int)jsrc, 64);
    const uint32_t position = (((prank * TT) + tt) << gl) + col;       // < c1 <= 64: one candidate register
    sc[lane] = make_uint2(0u, 0xFFFFFFFFu);
    wave_sync_lds();
    if (v1) sc[position] = make_uint2(__float_as_uint(s1v), child1);
    wave_sync_lds();
    const uint2 mine = sc[lane];
    wave_sync_lds();
    uint32_t key[1], sbits[1], payload[1];
    sbits[0] = mine.x; payload[0] = mine.y;
    key[0] = mine.y != 0xFFFFFFFFu ? score_key(__uint_as_float(mine.x)) : 0u;
    uint32_t rank, sb, ch;
    const uint32_t kk = wave_topk<1>(key, sbits, payload, L1.k, sc, lane, rank, sb, ch);
    if ((uint32_t)lane < kk) {
        s_bidx[rank] = L1.perm_inv ? L1.perm_inv[ch] : ch;
        s_bval[rank] = __uint_as_float(sb);
    }
    wave_sync_lds();
    return kk;
}

// MULTI = false: exactly one layer (layer[0]); the layer loop and its run-time descriptor indexing cost ~20 VGPRs, which the
// single-layer launches (wide layers, k1q_fuse = 0) do not pay.
// The fused kernel of narrow layers is compiled for 7 wavefronts per SIMD (72 VGPRs instead of the 74 the compiler settles on,
// no spills; the exp-family post-processors would spill and keep the default): measured 6.57 vs 6.73 ms on Amazon-670K's levels 0-3; 8 (64 VGPRs, 8 spilled) loses, and so does any target
// on the wide single-layer kernels.
template <int NSMAX, int PPC, bool DENSEX, bool MULTI, bool BIASF, bool PRES, bool BIGW>
#ifndef XRL_K1Q_WPE
#define XRL_K1Q_WPE 8    // round 4: 8 wavefronts per SIMD (64 VGPRs) -- the buffer-resource loads need no 64-bit vector addresses; 4.42 -> 4.14 ms on Amazon-670K
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NSMAX <= 3 && PPC == 0) ? XRL_K1Q_WPE : 1, 8))) k1q_kernel(K1QArgs a) {
    __shared__ uint2 sc_all[4 * 64];
    __shared__ uint32_t bidx_all[4 * 64];
    __shared__ float bval_all[4 * 64];
    __shared__ uint32_t pstage_all[PRES ? 4 * kPresStageWords * 64 : 1];   // presence rows of the current chunk, word-major, per wavefront (k1q_presence_mask)
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: scalar control flow below
    uint32_t q = blockIdx.x * 4u + wave;
    if (a.qperm) {
        // queries sorted by the best parent of their beam: this XCD's workgroups take neighbouring queries one after the other, which
        // share most of their (feature, parent) weight segments -- L2 hits instead of one fabric request per XCD and query
        const uint32_t slot = ((blockIdx.x & 7u) * a.xcd_per + (blockIdx.x >> 3)) * 4u + wave;
        if (slot >= a.nrows) return;
        q = __builtin_amdgcn_readfirstlane(a.qperm[slot]);
    }
    if (q >= a.nrows) return;
    uint2* sc = sc_all + wave * 64u;
    uint32_t* s_bidx = bidx_all + wave * 64u; float* s_bval = bval_all + wave * 64u;
    uint32_t* pstage = pstage_all + (PRES ? wave * (kPresStageWords * 64u) : 0u);

    // incoming beam -> LDS (k <= 64 entries); the implicit root needs none
    uint32_t cnt = 1;
    if (!a.layer[0].implicit_root) {
        cnt = min(a.p_cnt[q], a.layer[0].beam_in);
        if ((uint32_t)lane < cnt) { s_bidx[lane] = a.p_idx[(size_t)q * a.p_stride + lane]; s_bval[lane] = a.p_val[(size_t)q * a.p_stride + lane]; }
    }
    wave_sync_lds();
    const uint64_t xrow = (uint64_t)a.row0 + q;
    uint32_t fbm = 0u;                                                 // pruning feedback: bit l = layer l ran staged, bit 16 + l = its second pass was needed
    int l_first = 0;
    if (MULTI && !DENSEX && a.fuse01 == 2) { cnt = k1q_layer01m<PPC, BIASF>(a.layer[0], a.layer[1], a.wd01, a.wd01_c1, a.X, xrow, s_bidx, s_bval, sc, lane); l_first = 2; }
    else if (MULTI && !DENSEX && a.fuse01) { cnt = k1q_layer01<PPC, BIASF, BIGW>(a.layer[0], a.layer[1], a.X, xrow, s_bidx, s_bval, sc, lane); l_first = 2; }
    for (int l = l_first; l < (MULTI ? a.n_layers : 1); ++l) {
        const K1QLayer& Ly = a.layer[l];
        const uint32_t ns = Ly.ns;
        // every layer runs the body compiled for ITS register count (a narrower layer does not pay for the widest one's loads)
        if (ns <= 1) cnt = k1q_layer<1, PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else if (NSMAX >= 2 && ns <= 2) cnt = k1q_layer<(NSMAX >= 2 ? 2 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else if (NSMAX >= 3 && ns <= 3) cnt = k1q_layer<(NSMAX >= 3 ? 3 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else if (NSMAX >= 4 && ns <= 4) cnt = k1q_layer<(NSMAX >= 4 ? 4 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else if (NSMAX >= 6 && ns <= 6) cnt = k1q_layer<(NSMAX >= 6 ? 6 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else if (NSMAX >= 8 && ns <= 8) cnt = k1q_layer<(NSMAX >= 8 ? 8 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else if (NSMAX >= 12 && ns <= 12) cnt = k1q_layer<(NSMAX >= 12 ? 12 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
        else cnt = k1q_layer<(NSMAX >= 16 ? 16 : 1), PPC, DENSEX, BIASF, PRES, BIGW>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm, pstage);
    }
    if ((uint32_t)lane < cnt) {
        const size_t o = (size_t)q * a.out_stride + (uint32_t)lane;
        a.out_idx[o] = s_bidx[lane];
        a.out_val[o] = s_bval[lane];
    }
    if (lane == 0) a.out_cnt[q] = cnt;
    // pruning feedback (Model::fb_*): one query in 64 adds its layers' outcomes to the device counters; query 0 publishes what the
    // EARLIER launches counted to the host-visible words (read by the host at the start of a later predict, without synchronisation)
#ifndef XRL_K1Q_NOFB
    if (a.fb_dev && (q & 63u) == 0u) {
        if (q == 0u && a.fb_host && lane < 32) a.fb_host[lane] = a.fb_dev[lane];
        if (lane < 16 && ((fbm >> lane) & 1u)) { atomicAdd(&a.fb_dev[2 * lane], 1u); if ((fbm >> (16 + lane)) & 1u) atomicAdd(&a.fb_dev[2 * lane + 1], 1u); }
    }
#endif
    if (a.out_xok) {   // the pruning guard of this query, for a bound-pruned tile-format layer that follows (its K2 decides there)
        uint32_t mx = 0u, n;
        if (DENSEX) {
            n = a.X.cols;
            const float* __restrict__ xd = a.X.val + xrow * a.X.cols;
            for (uint32_t c = (uint32_t)lane; c < n; c += 64u) mx = max(mx, __float_as_uint(xd[c]) & 0x7FFFFFFFu);
        } else {
            const uint64_t xb = a.X.row_ptr[xrow];
            n = (uint32_t)(a.X.row_ptr[xrow + 1] - xb);
            for (uint32_t c = (uint32_t)lane; c < n; c += 64u) mx = max(mx, __float_as_uint(a.X.val[xb + c]) & 0x7FFFFFFFu);
        }
        mx = wave_max_u32(mx);
        if (lane == 0) a.out_xok[q] = prune_guard_ok(mx, n, a.prune_wmax) ? 1u : 0u;
    }
}

// one translation unit's share of the instantiations: k1q_kernel<NN, PP, ...> for every (DENSEX, MULTI, BIASF, PRES, BIGW) the host side can ask for
struct K1QVariant { bool dense_x, multi, bias_first, pres, big; };
template <int NN, int PP, bool ALLOW_MULTI>
inline void k1q_launch_variant(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v) {
    const dim3 block(256);
    constexpr bool kPresOk = NN <= 3;      // PRES instantiations exist for the narrow kernels only (sparse X, <= 3 registers)
#define XRL_K1Q_B(DX, MM, BF, PR) do { if (v.big) hipLaunchKernelGGL((k1q_kernel<NN, PP, DX, MM, BF, PR, true>), grid, block, 0, s, a); \
                                       else hipLaunchKernelGGL((k1q_kernel<NN, PP, DX, MM, BF, PR, false>), grid, block, 0, s, a); } while (0)
#define XRL_K1Q_P(DX, MM, BF) do { if (v.pres && !(DX) && kPresOk) XRL_K1Q_B(DX, MM, BF, (!(DX) && kPresOk)); else XRL_K1Q_B(DX, MM, BF, false); } while (0)
#define XRL_K1Q_M(MM) do { if (v.dense_x) XRL_K1Q_P(true, MM, true); else if (v.bias_first) XRL_K1Q_P(false, MM, true); else XRL_K1Q_P(false, MM, false); } while (0)
    if (ALLOW_MULTI && v.multi) XRL_K1Q_M(ALLOW_MULTI); else XRL_K1Q_M(false);
#undef XRL_K1Q_M
#undef XRL_K1Q_P
#undef XRL_K1Q_B
}
// (defined in xrl_k1q_n1p0.hip ... xrl_k1q_n16.hip)
void k1q_launch_n1p0(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v);
void k1q_launch_n1p1(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v);
void k1q_launch_n3p0(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v);
void k1q_launch_n3p1(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v);
void k1q_launch_n6(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v, int ppc);
void k1q_launch_n16(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v, int ppc);

}  // namespace xrl
