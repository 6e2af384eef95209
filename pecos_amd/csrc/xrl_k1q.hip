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
};
constexpr int kK1QMaxLayers = 8;

struct K1QArgs {
    K1QLayer layer[kK1QMaxLayers];
    int n_layers;                     // consecutive dense-format layers run back to back by the same wavefront: the beam stays in LDS
    int fuse01;                       // layers 0 and 1 share one walk over the query's features (k1q_layer01)
    QueriesDev X;
    const uint32_t* p_idx; const float* p_val; const uint32_t* p_cnt; uint32_t p_stride;
    uint32_t* out_idx; float* out_val; uint32_t* out_cnt; uint32_t out_stride;
    uint32_t row0, nrows;
    float prune_wmax;                 // the model's largest |weight| x max(1, |bias|): the pruning guard (prune_guard_ok, xrl_device.h)
    uint32_t* out_xok;                // non-null: the guard flag of every query is also written here (for a pruned tile-format layer that follows)
    uint32_t* fb_dev; uint32_t* fb_host;   // pruning feedback: sampled counters per layer {staged queries, of them: second pass needed} (device atomics);
                                      // the first wavefront of a launch copies what earlier launches counted to the host-visible words
};

// the weight row of feature f as a raw buffer resource: base = wd + f * ld (scalar arithmetic), num_records = the row's bytes (a lane
// offset past the row reads 0, never memory), dword 3 = the untyped 32-bit format word of gfx9-class buffer descriptors
// (row_bytes = d_ld * 4 < 2^32: one 32 x 32 -> 64-bit scalar multiply per row)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k1q_row_rsrc(const uint32_t* wd, uint32_t row_bytes, uint32_t f) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(wd) + (uint64_t)f * row_bytes), 0, (int)row_bytes, 0x00020000);
}

// (Measured and rejected, profiles/r04_k1q_diet.md: ONE resource per layer with the row selected by the instruction's scalar offset --
//  2 scalar instructions per row instead of 6 for matrices under 4 GB -- needs both forms in the kernel (level 3 is 4.4 GB); the second
//  code path cost 168 SGPR spills at 8 wavefronts per SIMD: 4.64 vs 4.13 ms on Amazon-670K.)
#ifndef XRL_K1Q_U1
#define XRL_K1Q_U1 16
#endif
#ifndef XRL_K1Q_U3
#define XRL_K1Q_U3 8
#endif
template <int NS> struct K1QCfg {
    // weight rows (features) whose loads are in flight together: U * NS loads per lane
    static constexpr int U = NS <= 1 ? XRL_K1Q_U1 : NS <= 3 ? XRL_K1Q_U3 : NS <= 12 ? 4 : 2;
};

// One layer for one query (one wavefront): beam in s_bidx / s_bval[0..cnt) -> beam out in the same arrays; returns the new count.
// BIASF: the accumulators start at the bias product (dense X; sparse X under HASH_CHUNKED, inference.hpp:716-722) instead of receiving it
// last -- a compile-time switch: as a run-time one it cost the widest kernel 8 VGPRs and a wavefront per SIMD
// PRES: the instantiation carries the presence-word path (layers that run UNSTAGED on sparse X -- every beam parent's segments are
// requested, a third to a half of them empty); the staged default does not pay its registers (4.15 vs 4.49 ms on Amazon-670K)
template <int NS, int PPC, bool DENSEX, bool BIASF, bool PRES>
__device__ __forceinline__ uint32_t k1q_layer(const K1QLayer& Ly, const QueriesDev& X, uint64_t xrow, uint32_t cnt_in,
                                               uint32_t* s_bidx, float* s_bval, uint2* sc, int lane, float wmax, uint32_t& fbm) {
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
        const uint32_t dt = Ly.d_ptile[parent] + tt;
        v = v && dt < Ly.d_ptile[parent + 1];
        const uint32_t dtc = v ? dt : 0u;
        const uint32_t cb = Ly.d_tcol[dtc], ce = Ly.d_tcol[dtc + 1];
        v = v && col < ce - cb;
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
    const uint32_t pres_bytes = Ly.pres_words * 4u, dt_shift = gl + 2u, pw_shift = gl + 2u + 5u - 2u;
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
        auto body = [&](auto exact_tag, const uint32_t (&fs)[UU], const float (&xs)[UU]) {
            constexpr bool EX = decltype(exact_tag)::value;
            uint32_t wb[UU][NR];
            if (PRES && !EX && pres != nullptr) {
                // PRESENCE: first the word that says whether this lane's dense tile holds any weight at the feature (one small row per
                // feature, mostly L2-resident), then the weight load with the lane's offset -- or an offset outside the resource for an
                // empty tile: such lanes read 0.0 without a memory request.  A 64-byte segment none of whose lanes asks is never fetched.
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const __amdgpu_buffer_rsrc_t ps = k1q_row_rsrc(pres, pres_bytes, min(fs[u], w_rows));
#pragma unroll
                    for (int r = 0; r < NR; ++r) wb[u][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ps, (int)((woff[RB + r] >> pw_shift) & ~3u), 0, 0);
                }
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const __amdgpu_buffer_rsrc_t rs = k1q_row_rsrc(wd, ld, min(fs[u], w_rows));
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const bool present = ((wb[u][r] >> ((woff[RB + r] >> dt_shift) & 31u)) & 1u) != 0u;
                        wb[u][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(present ? woff[RB + r] : 0xFFFFFFF0u), 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    // features outside the layer (and padding slots: fs = w_rows) read the all-missing row the model compiler appends
                    const __amdgpu_buffer_rsrc_t rs = k1q_row_rsrc(wd, ld, min(fs[u], w_rows));
#pragma unroll
                    for (int r = 0; r < NR; ++r) wb[u][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)woff[RB + r], 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < UU; ++u) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    // scalar * val, then add: no fma (inference.hpp:512-517); no entry -> no operation
                    const float sm = __fadd_rn(acc[RB + r], __fmul_rn(xs[u], __uint_as_float(wb[u][r])));
                    acc[RB + r] = (EX && wb[u][r] == kMissing) ? acc[RB + r] : sm;
                }
            }
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
        for (uint32_t t0 = 0; t0 < n; t0 += 64u) {
            const uint32_t nc = min(64u, n - t0);
            // the chunk's values once per lane: the pruning guard's maximum, and "is every value finite"
            const uint32_t vb = (uint32_t)lane < nc ? (__float_as_uint(vsrc[t0 + (uint32_t)lane]) & 0x7FFFFFFFu) : 0u;
            xmx = max(xmx, vb);
            const bool nonfinite = __ballot(vb >= 0x7F800000u) != 0ull;
            for (uint32_t t = t0; t < t0 + nc; t += (uint32_t)UU) {
                uint32_t fs[UU]; float xs[UU];
                if (!nonfinite && t + (uint32_t)UU <= t0 + nc) {             // a full batch: plain uniform loads
#pragma unroll
                    for (int u = 0; u < UU; ++u) { fs[u] = DENSEX ? t + (uint32_t)u : fsrc[t + (uint32_t)u]; xs[u] = vsrc[t + (uint32_t)u]; }
                    body(std::false_type{}, fs, xs);
                } else if (!nonfinite && (uint64_t)t + (uint32_t)UU <= room) {  // the row's tail: the loads run on into the next row, the slots past the end are neutralised
#pragma unroll
                    for (int u = 0; u < UU; ++u) {
                        const bool ok = t + (uint32_t)u < n;
                        const uint32_t f = DENSEX ? t + (uint32_t)u : fsrc[t + (uint32_t)u];
                        const float x = vsrc[t + (uint32_t)u];
                        fs[u] = ok ? f : w_rows; xs[u] = ok ? x : 0.0f;
                    }
                    body(std::false_type{}, fs, xs);
                } else {                                                      // non-finite x in the chunk, or the end of the X arrays: clamped loads, exact loop
#pragma unroll
                    for (int u = 0; u < UU; ++u) {
                        const bool ok = t + (uint32_t)u < n;
                        const uint32_t ic = ok ? t + (uint32_t)u : n - 1u;
                        const uint32_t f = DENSEX ? ic : fsrc[ic];
                        const float x = vsrc[ic];
                        fs[u] = ok ? f : w_rows; xs[u] = ok ? x : 0.0f;
                    }
                    body(std::true_type{}, fs, xs);
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
template <int PPC, bool BIASF>
__device__ __forceinline__ uint32_t k1q_layer01(const K1QLayer& L0, const K1QLayer& L1, const QueriesDev& X, uint64_t xrow,
                                                 uint32_t* s_bidx, float* s_bval, uint2* sc, int lane) {
    constexpr int UU = 8;
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
    const uint32_t wr0 = L0.w_rows, wr1 = L1.w_rows;
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
            w0[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(k1q_row_rsrc(wd0, ld0, min(fs[u], wr0)), (int)woff0, 0, 0);
            w1[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(k1q_row_rsrc(wd1, ld1, min(fs[u], wr1)), (int)woff1, 0, 0);
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
        const bool nonfinite = __ballot(vb >= 0x7F800000u) != 0ull;
        for (uint32_t t = t0; t < t0 + nc; t += (uint32_t)UU) {
            uint32_t fs[UU]; float xs[UU];
            if (!nonfinite && t + (uint32_t)UU <= t0 + nc) {
#pragma unroll
                for (int u = 0; u < UU; ++u) { fs[u] = xi[t + (uint32_t)u]; xs[u] = xv[t + (uint32_t)u]; }
                body(std::false_type{}, fs, xs);
            } else if (!nonfinite && (uint64_t)t + (uint32_t)UU <= room) {
#pragma unroll
                for (int u = 0; u < UU; ++u) { const bool ok = t + (uint32_t)u < xl; const uint32_t f = xi[t + (uint32_t)u]; const float x = xv[t + (uint32_t)u]; fs[u] = ok ? f : 0xFFFFFFFFu; xs[u] = ok ? x : 0.0f; }
                body(std::false_type{}, fs, xs);
            } else {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const bool ok = t + (uint32_t)u < xl; const uint32_t ic = ok ? t + (uint32_t)u : xl - 1u;
                    const uint32_t f = xi[ic]; const float x = xv[ic];
                    fs[u] = ok ? f : 0xFFFFFFFFu; xs[u] = ok ? x : 0.0f;
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

// MULTI = false: exactly one layer (layer[0]); the layer loop and its run-time descriptor indexing cost ~20 VGPRs, which the
// single-layer launches (wide layers, k1q_fuse = 0) do not pay.
// The fused kernel of narrow layers is compiled for 7 wavefronts per SIMD (72 VGPRs instead of the 74 the compiler settles on,
// no spills; the exp-family post-processors would spill and keep the default): measured 6.57 vs 6.73 ms on Amazon-670K's levels 0-3; 8 (64 VGPRs, 8 spilled) loses, and so does any target
// on the wide single-layer kernels.
template <int NSMAX, int PPC, bool DENSEX, bool MULTI, bool BIASF, bool PRES>
#ifndef XRL_K1Q_WPE
#define XRL_K1Q_WPE 8    // round 4: 8 wavefronts per SIMD (64 VGPRs) -- the buffer-resource loads need no 64-bit vector addresses; 4.42 -> 4.14 ms on Amazon-670K
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MULTI && NSMAX <= 3 && PPC == 0) ? XRL_K1Q_WPE : 1, 8))) k1q_kernel(K1QArgs a) {
    __shared__ uint2 sc_all[4 * 64];
    __shared__ uint32_t bidx_all[4 * 64];
    __shared__ float bval_all[4 * 64];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: scalar control flow below
    const uint32_t q = blockIdx.x * 4u + wave;
    if (q >= a.nrows) return;
    uint2* sc = sc_all + wave * 64u;
    uint32_t* s_bidx = bidx_all + wave * 64u; float* s_bval = bval_all + wave * 64u;

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
    if (MULTI && !DENSEX && a.fuse01) { cnt = k1q_layer01<PPC, BIASF>(a.layer[0], a.layer[1], a.X, xrow, s_bidx, s_bval, sc, lane); l_first = 2; }
    for (int l = l_first; l < (MULTI ? a.n_layers : 1); ++l) {
        const K1QLayer& Ly = a.layer[l];
        const uint32_t ns = Ly.ns;
        // every layer runs the body compiled for ITS register count (a narrower layer does not pay for the widest one's loads)
        if (ns <= 1) cnt = k1q_layer<1, PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else if (NSMAX >= 2 && ns <= 2) cnt = k1q_layer<(NSMAX >= 2 ? 2 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else if (NSMAX >= 3 && ns <= 3) cnt = k1q_layer<(NSMAX >= 3 ? 3 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else if (NSMAX >= 4 && ns <= 4) cnt = k1q_layer<(NSMAX >= 4 ? 4 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else if (NSMAX >= 6 && ns <= 6) cnt = k1q_layer<(NSMAX >= 6 ? 6 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else if (NSMAX >= 8 && ns <= 8) cnt = k1q_layer<(NSMAX >= 8 ? 8 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else if (NSMAX >= 12 && ns <= 12) cnt = k1q_layer<(NSMAX >= 12 ? 12 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
        else cnt = k1q_layer<(NSMAX >= 16 ? 16 : 1), PPC, DENSEX, BIASF, PRES>(Ly, a.X, xrow, cnt, s_bidx, s_bval, sc, lane, a.prune_wmax, fbm);
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

// registers per lane a layer needs with `beam_in` parents per query, or 0 when K1Q cannot serve it
uint32_t k1q_regs(const LayerDev& L, uint32_t beam_in, uint32_t k, bool dense_x) {
    if (!dense_x && !L.d_sparse_ok) return 0;                         // sparse X, wide parents, near-empty segments: the tile format wins
    if (!L.wd || k == 0 || k > 64 || beam_in > 64) return 0;      // the beam lives in 64-entry LDS arrays on its way through the layers
    const uint64_t cands = ((uint64_t)beam_in * L.d_max_tiles) << L.d_gp_log2;
    const uint64_t ns = (cands + 63) / 64;
    // (17..24 registers were tried for Wiki10-31K's leaf -- beam 20 x 64 children -- and lost to the tile kernels, 4.2 vs 2.7 ms:
    //  with 418 features per query and 0.3 % dense columns the dense format reads two lines per (feature, parent) for nothing)
    return ns <= 16 ? (uint32_t)std::max<uint64_t>(1, ns) : 0u;
}

static bool k1q_fuse01_enabled() {   // XRL_K1Q_FUSE01=0: levels 0 and 1 take separate feature walks (A/B, tests)
    const char* e = std::getenv("XRL_K1Q_FUSE01");
    return !(e && e[0] == '0');
}
static uint32_t k1q_bucket(uint32_t ns) { return ns <= 1 ? 1 : ns <= 2 ? 2 : ns <= 3 ? 3 : ns <= 4 ? 4 : ns <= 6 ? 6 : ns <= 8 ? 8 : ns <= 12 ? 12 : 16; }
static uint32_t k1q_kernel_bucket(uint32_t ns) { return ns <= 1 ? 1 : ns <= 3 ? 3 : ns <= 6 ? 6 : 16; }   // kernels are compiled for these maxima

// n consecutive dense-format layers (n <= kK1QMaxLayers) in ONE launch: previous beam in, the last layer's beam out
void launch_k1q(const LayerDev* const* Ls, const LayerPlan* Ps, int n, const QueriesDev& X, BeamDev prev, uint32_t* out_idx, float* out_val,
                uint32_t* out_cnt, uint32_t out_stride, hipStream_t s, float prune_wmax, uint32_t* out_xok) {
    if (n <= 0 || n > kK1QMaxLayers) fail("k1q: bad layer count");
    if (n > 1) for (int l = 0; l < n; ++l) if (k1q_regs(*Ls[l], Ps[l].beam_in, Ps[l].k, true) > 3) fail("k1q: only layers of <= 3 candidate registers can share a launch");
    if (Ps[0].nrows == 0) return;
    K1QArgs a;
    uint32_t nsmax = 1; int ppc = 0;
    for (int l = 0; l < n; ++l) {
        const LayerDev& L = *Ls[l]; const LayerPlan& P = Ps[l];
        const uint32_t ns = k1q_regs(L, P.beam_in, P.k, true);      // capacity check only; whether sparse X SHOULD use the format is the caller's policy
        if (ns == 0) fail("k1q: layer not eligible");
        K1QLayer& y = a.layer[l];
        y.wd = L.wd; y.d_ld = L.d_ld; y.pres = (!X.dense && !P.tune.ablate && (P.tune.pres_mode == 2 || (P.tune.pres_mode == 1 && !P.prune))) ? L.pres : nullptr; y.pres_words = L.pres_words;   // presence words: layers that run unstaged
        y.d_ptile = L.d_ptile; y.d_tcol = L.d_tcol; y.bias_prod = L.bias_prod; y.perm_inv = L.perm_inv;
        y.d_gp_log2 = L.d_gp_log2; y.d_max_tiles = L.d_max_tiles; y.n_parents = L.n_parents; y.w_rows = L.w_rows;
        y.beam_in = P.beam_in; y.k = P.k; y.ns = k1q_bucket(ns);
        y.has_bias = L.has_bias; y.pp_kind = P.pp.kind; y.pp_p = P.pp.p; y.first_layer = P.first_layer; y.implicit_root = P.implicit_root; y.bias_first = P.bias_first; y.prune = P.prune;
        y.layer_id = (P.layer >= 0 && P.layer < 16) ? P.layer : 0;
        nsmax = std::max(nsmax, y.ns); ppc |= pp_class(P.pp);
    }
    a.n_layers = n; a.X = X;
    // root + next level in one feature walk: the root keeps all of its children (so level 1 always evaluates all of theirs), both sit
    // in one candidate register, sparse X
    a.fuse01 = 0;
    if (n >= 2 && !X.dense && Ps[0].implicit_root && Ps[0].first_layer && Ls[0]->n_parents == 1 && Ls[0]->d_max_tiles == 1 && Ps[0].tune.ablate == 0) {
        const uint32_t K0 = Ls[0]->n_children;
        const uint64_t c1 = ((uint64_t)K0 * Ls[1]->d_max_tiles) << Ls[1]->d_gp_log2;
        if (K0 >= 1 && K0 <= 64 && K0 <= Ps[0].k && Ps[1].beam_in >= K0 && c1 <= 64 && a.layer[0].ns == 1 && a.layer[1].ns == 1 && Ps[1].k <= 64) a.fuse01 = k1q_fuse01_enabled() ? 1 : 0;
    }
    a.p_idx = prev.idx; a.p_val = prev.val; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
    a.out_idx = out_idx; a.out_val = out_val; a.out_cnt = out_cnt; a.out_stride = out_stride;
    a.row0 = Ps[0].row0; a.nrows = Ps[0].nrows;
    a.prune_wmax = prune_wmax; a.out_xok = out_xok;
    a.fb_dev = Ps[0].fb_dev; a.fb_host = Ps[0].fb_host;
    const dim3 grid((a.nrows + 3u) / 4u), block(256);
    const bool bias_first = Ps[0].bias_first != 0;
    // PRES instantiations exist for the narrow kernels only (sparse X, <= 3 registers: the layers of 16-children parents the presence
    // words are built for); a launch takes one when any of its layers carries presence words this time (launch_k1q's caller decides: layers
    // that run unstaged)
    bool any_pres = false;
    for (int l = 0; l < n; ++l) any_pres = any_pres || a.layer[l].pres != nullptr;
#define XRL_K1Q_P(NN, PP, DX, MM, BF) do { if (any_pres && !(DX) && (NN) <= 3) hipLaunchKernelGGL((k1q_kernel<NN, PP, DX, MM, BF, (!(DX) && (NN) <= 3)>), grid, block, 0, s, a); \
                                           else hipLaunchKernelGGL((k1q_kernel<NN, PP, DX, MM, BF, false>), grid, block, 0, s, a); } while (0)
#define XRL_K1Q_M(NN, MM) do { \
        if (X.dense) { if (ppc) XRL_K1Q_P(NN, 1, true, MM, true); else XRL_K1Q_P(NN, 0, true, MM, true); } \
        else if (bias_first) { if (ppc) XRL_K1Q_P(NN, 1, false, MM, true); else XRL_K1Q_P(NN, 0, false, MM, true); } \
        else { if (ppc) XRL_K1Q_P(NN, 1, false, MM, false); else XRL_K1Q_P(NN, 0, false, MM, false); } } while (0)
#define XRL_K1Q(NN) do { if (n > 1) XRL_K1Q_M(NN, true); else XRL_K1Q_M(NN, false); } while (0)
    switch (k1q_kernel_bucket(nsmax)) {
    case 1: XRL_K1Q(1); break;
    case 3: XRL_K1Q(3); break;
    case 6: XRL_K1Q_M(6, false); break;      // only narrow layers (<= 3 registers) are fused (xrl_predict.cpp)
    default: XRL_K1Q_M(16, false); break;
    }
#undef XRL_K1Q_P
#undef XRL_K1Q
#undef XRL_K1Q_M
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Model compiler, device side: presence words of the dense row format (LayerDev::pres).  One wavefront per feature row, 64 columns
// per step: a ballot of "holds a weight" is folded into one bit per dense tile (a tile = 2^gl <= 32 columns).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
presence_kernel(const uint32_t* __restrict__ wd, uint64_t ld, uint32_t rows, uint32_t gl, uint32_t n_tiles, uint32_t pw, uint32_t* __restrict__ pres) {
    const uint32_t f = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (f >= rows) return;
    const uint32_t* __restrict__ row = wd + (uint64_t)f * ld;
    const uint32_t gp = 1u << gl, tpc = 64u >> gl;                   // tiles per 64-column step
    uint32_t word = 0u, wi = 0u, filled = 0u;
    for (uint32_t c0 = 0; c0 < (n_tiles << gl); c0 += 64u) {
        const uint32_t c = c0 + lane;
        const bool nz = c < (uint32_t)ld && row[c] != kMissing;
        const unsigned long long m = __ballot(nz);
        // lane t < tpc: does tile (c0 >> gl) + t hold a weight
        const unsigned long long seg = gp == 64u ? m : ((m >> (lane < tpc ? lane * gp : 0u)) & ((1ull << gp) - 1ull));
        const unsigned long long tb = __ballot(lane < tpc && seg != 0ull);
        word |= (uint32_t)tb << filled; filled += tpc;
        if (filled == 32u) { if (lane == 0) pres[(uint64_t)f * pw + wi] = word; word = 0u; filled = 0u; ++wi; }
    }
    if (lane == 0) { if (filled) pres[(uint64_t)f * pw + wi++] = word; for (; wi < pw; ++wi) pres[(uint64_t)f * pw + wi] = 0u; }
}

void launch_presence(const uint32_t* wd, uint64_t ld, uint32_t rows, uint32_t gp_log2, uint32_t n_tiles, uint32_t pres_words, uint32_t* pres, hipStream_t s) {
    if (rows == 0) return;
    hipLaunchKernelGGL(presence_kernel, dim3((rows + 3u) / 4u), dim3(256), 0, s, wd, ld, rows, gp_log2, n_tiles, pres_words, pres);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Model compiler, device side: scatter CSC weight columns into the dense row format.
//   wd[row * ld + dst_off[c]] = W[row, src_col[c]]   for every (rearranged) child column c
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
densify_kernel(const uint64_t* __restrict__ col_ptr, const uint32_t* __restrict__ row_idx, const float* __restrict__ val,
               const uint32_t* __restrict__ src_col, const uint32_t* __restrict__ dst_off, uint32_t n_children,
               uint64_t ld, uint32_t* __restrict__ wd) {
    // one wavefront per column: lanes stride over the column's entries
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c >= n_children) return;
    const uint32_t oc = src_col[c], off = dst_off[c];
    const uint64_t e0 = col_ptr[oc], e1 = col_ptr[oc + 1];
    // (an explicit -0.0 would read as "no entry": stored as +0.0 -- x * (+-0.0) leaves an accumulator unchanged either way)
    for (uint64_t e = e0 + (threadIdx.x & 63u); e < e1; e += 64u) { const uint32_t b = __float_as_uint(val[e]); wd[(uint64_t)row_idx[e] * ld + off] = b == kMissing ? 0u : b; }
}

void launch_densify(const uint64_t* col_ptr, const uint32_t* row_idx, const float* val, const uint32_t* src_col,
                    const uint32_t* dst_off, uint32_t n_children, uint32_t w_rows, uint64_t ld, uint32_t* wd, hipStream_t s) {
    // w_rows + 1 rows: the extra last row stays all-kMissing (K1Q sends out-of-range features there)
    XRL_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(wd), (int)kMissing, ((size_t)w_rows + 1) * ld, s));
    if (n_children) {
        hipLaunchKernelGGL(densify_kernel, dim3((n_children + 3u) / 4u), dim3(256), 0, s, col_ptr, row_idx, val, src_col, dst_off,
                           n_children, ld, wd);
        XRL_LAUNCH_CHECK();
    }
}

}  // namespace xrl
