// K1Q, host side: eligibility, launch dispatch, and the device-side model-compiler kernels of the dense row format (presence words,
// densify).  The kernel itself is xrl_k1q_impl.h; its instantiations are compiled in xrl_k1q_n*.hip.
#include "xrl_k1q_impl.h"

namespace xrl {

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

static int k1q_fuse01_enabled() {   // XRL_K1Q_FUSE01=0: levels 0 and 1 take separate feature walks; 1: one walk, a load per level; default 2: one walk, ONE load (merged rows) (A/B, tests)
    const char* e = std::getenv("XRL_K1Q_FUSE01");
    return (e && e[0] == '0') ? 0 : (e && e[0] == '1') ? 1 : 2;
}
static uint32_t k1q_bucket(uint32_t ns) { return ns <= 1 ? 1 : ns <= 2 ? 2 : ns <= 3 ? 3 : ns <= 4 ? 4 : ns <= 6 ? 6 : ns <= 8 ? 8 : ns <= 12 ? 12 : 16; }
static uint32_t k1q_kernel_bucket(uint32_t ns) { return ns <= 1 ? 1 : ns <= 3 ? 3 : ns <= 6 ? 6 : 16; }   // kernels are compiled for these maxima

// n consecutive dense-format layers (n <= kK1QMaxLayers) in ONE launch: previous beam in, the last layer's beam out
void launch_k1q(const LayerDev* const* Ls, const LayerPlan* Ps, int n, const QueriesDev& X, BeamDev prev, uint32_t* out_idx, float* out_val,
                uint32_t* out_cnt, uint32_t out_stride, hipStream_t s, float prune_wmax, uint32_t* out_xok, const uint32_t* qperm) {
    if (n <= 0 || n > kK1QMaxLayers) fail("k1q: bad layer count");
    if (n > 1) for (int l = 0; l < n; ++l) if (k1q_regs(*Ls[l], Ps[l].beam_in, Ps[l].k, true) > 3) fail("k1q: only layers of <= 3 candidate registers can share a launch");
    if (Ps[0].nrows == 0) return;
    K1QArgs a{};
    uint32_t nsmax = 1; int ppc = 0;
    for (int l = 0; l < n; ++l) {
        const LayerDev& L = *Ls[l]; const LayerPlan& P = Ps[l];
        const uint32_t ns = k1q_regs(L, P.beam_in, P.k, true);      // capacity check only; whether sparse X SHOULD use the format is the caller's policy
        if (ns == 0) fail("k1q: layer not eligible");
        K1QLayer& y = a.layer[l];
        y.wd = L.wd; y.d_ld = L.d_ld;
        // presence words: layers that run unstaged (or every layer that has them: presence = 2); the masks need dense tiles of >= 16 columns (<= 4 beam slots per candidate register)
        y.pres = (!X.dense && !P.tune.ablate && L.d_gp_log2 >= 4 && (P.tune.pres_mode == 2 || (P.tune.pres_mode == 1 && !P.prune))) ? L.pres : nullptr;
        y.pres_words = L.pres_words;
        y.d_ptile = L.d_ptile; y.d_tcol = L.d_tcol; y.bias_prod = L.bias_prod; y.perm_inv = L.perm_inv;
        y.d_gp_log2 = L.d_gp_log2; y.d_max_tiles = L.d_max_tiles; y.n_parents = L.n_parents; y.w_rows = L.w_rows;
        y.beam_in = P.beam_in; y.k = P.k; y.ns = k1q_bucket(ns);
        y.has_bias = L.has_bias; y.pp_kind = P.pp.kind; y.pp_p = P.pp.p; y.first_layer = P.first_layer; y.implicit_root = P.implicit_root; y.bias_first = P.bias_first; y.prune = P.prune;
        y.layer_id = (P.layer >= 0 && P.layer < 16) ? P.layer : 0;
        y.regular = L.d_regular;
        nsmax = std::max(nsmax, y.ns); ppc |= pp_class(P.pp);
    }
    a.n_layers = n; a.X = X;
    // root + next level in one feature walk: the root keeps all of its children (so level 1 always evaluates all of theirs), both sit
    // in one candidate register, sparse X
    a.fuse01 = 0;
    if (n >= 2 && !X.dense && Ps[0].implicit_root && Ps[0].first_layer && Ls[0]->n_parents == 1 && Ls[0]->d_max_tiles == 1 && Ps[0].tune.ablate == 0) {
        const uint32_t K0 = Ls[0]->n_children;
        const uint64_t c1 = ((uint64_t)K0 * Ls[1]->d_max_tiles) << Ls[1]->d_gp_log2;
        if (K0 >= 1 && K0 <= 64 && K0 <= Ps[0].k && Ps[1].beam_in >= K0 && c1 <= 64 && Ls[0]->w_rows == Ls[1]->w_rows && a.layer[0].ns == 1 && a.layer[1].ns == 1 && Ps[1].k <= 64) a.fuse01 = k1q_fuse01_enabled();
        // one load per feature for both levels when the model carries their merged matrix (finalize_model) and the candidate layout is the one it was built for
        if (a.fuse01 == 2 && !(Ls[0]->wd01 && Ls[0]->wd01_c1 == c1 && c1 + K0 <= 64)) a.fuse01 = 1;
        a.wd01 = Ls[0]->wd01; a.wd01_c1 = Ls[0]->wd01_c1;
    }
    a.p_idx = prev.idx; a.p_val = prev.val; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
    a.out_idx = out_idx; a.out_val = out_val; a.out_cnt = out_cnt; a.out_stride = out_stride;
    a.row0 = Ps[0].row0; a.nrows = Ps[0].nrows;
    a.prune_wmax = prune_wmax; a.out_xok = out_xok;
    a.fb_dev = Ps[0].fb_dev; a.fb_host = Ps[0].fb_host;
    const uint32_t n_wg = (a.nrows + 3u) / 4u;
    a.qperm = qperm; a.xcd_per = (n_wg + 7u) / 8u;
    const dim3 grid(qperm ? a.xcd_per * 8u : n_wg);
    const bool bias_first = Ps[0].bias_first != 0;
    // PRES instantiations exist for the narrow kernels only (sparse X, <= 3 registers: the layers of 16-children parents the presence
    // words are built for); a launch takes one when any of its layers carries presence words this time (launch_k1q's caller decides: layers
    // that run unstaged).  BIGW: some matrix of the launch reaches 4 GiB -- a row's byte offset then needs the 64-bit form (k1q_load_w)
    K1QVariant v{};
    v.dense_x = X.dense != 0; v.multi = n > 1; v.bias_first = bias_first;
    for (int l = 0; l < n; ++l) {
        v.pres = v.pres || a.layer[l].pres != nullptr;
        v.big = v.big || ((uint64_t)Ls[l]->w_rows + 1) * Ls[l]->d_ld * 4ull >= 0xFFFFFFF0ull;   // (0xFFFFFFF0 is the lane offset that switches a lane off: it must stay outside the resource)
    }
    switch (k1q_kernel_bucket(nsmax)) {
    case 1: if (ppc) k1q_launch_n1p1(a, grid, s, v); else k1q_launch_n1p0(a, grid, s, v); break;
    case 3: if (ppc) k1q_launch_n3p1(a, grid, s, v); else k1q_launch_n3p0(a, grid, s, v); break;
    case 6: v.multi = false; k1q_launch_n6(a, grid, s, v, ppc); break;      // only narrow layers (<= 3 registers) are fused (xrl_predict.cpp)
    default: v.multi = false; k1q_launch_n16(a, grid, s, v, ppc); break;
    }
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

// levels 0 and 1 side by side: out[row][c] = wd1[row][c] (c < c1), out[row][c1 + j] = wd0[row][j] (j < k0), kMissing elsewhere; 64 columns per row
__global__ void __launch_bounds__(256)
merge01_kernel(const uint32_t* __restrict__ wd0, uint64_t ld0, uint32_t k0, const uint32_t* __restrict__ wd1, uint64_t ld1, uint32_t c1, uint32_t rows, uint32_t* __restrict__ out) {
    const uint32_t f = blockIdx.x * 4u + (threadIdx.x >> 6), c = threadIdx.x & 63u;
    if (f >= rows) return;
    uint32_t v = kMissing;
    if (c < c1) v = wd1[(uint64_t)f * ld1 + c];
    else if (c - c1 < k0) v = wd0[(uint64_t)f * ld0 + (c - c1)];
    out[(uint64_t)f * 64u + c] = v;
}
void launch_merge01(const uint32_t* wd0, uint64_t ld0, uint32_t k0, const uint32_t* wd1, uint64_t ld1, uint32_t c1, uint32_t rows, uint32_t* out, hipStream_t s) {
    if (rows == 0) return;
    hipLaunchKernelGGL(merge01_kernel, dim3((rows + 3u) / 4u), dim3(256), 0, s, wd0, ld0, k0, wd1, ld1, c1, rows, out);
    XRL_LAUNCH_CHECK();
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
