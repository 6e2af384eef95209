// Pair kernels: one (query row, weight column) inner product per group of PG = 16 lanes, against W in CSC form.
//
//   K3  k3_pairs_kernel      sparse_inner_products                      matrix.hpp:1049-1060, do_dot_product :836-877
//   K4  k4_selected_kernel   predict_on_selected_outputs, one layer     inference.hpp:1018-1078, 1302-1358
//   K1C k1c_csc_kernel       the CSC route of a whole layer (w_ops<csc_t>::compute_sparse_predictions,
//                            inference.hpp:1081-1149) over the candidates K0 laid out -- what the reference's
//                            single-layer API runs (libpecos.cpp:201-235 builds MLModel<csc_t>)
//
// Arithmetic contract of the CSC route (vector_ops::inner_product, inference.hpp:1018-1078), bit for bit:
//   sparse X:  res = 0;  res += fl32(bias * w_bias) if the column's last entry is the bias row;  res += dot
//              where dot = 0, then dot = fl32(dot + fl32(x_f * w_f)) over the matching indices in ASCENDING order
//              (do_dot_product, matrix.hpp:836-859: summed separately, then added -- unlike the chunked route)
//   dense X:   bias > 0:  res = fl32(bias * w_bias) first, then res = fl32(res + fl32(x[idx] * w)) over the non-bias
//              entries in order;  bias <= 0: the same chain over all entries
//
// Round 1 ran these as one THREAD per pair (64 divergent walks per wavefront, every load uncoalesced).  Here the 16 lanes of
// a group read 16 consecutive entries of the SHORTER index list per step (coalesced 64-byte segments), each lane binary-
// searches its index in the longer list, and the matches' products -- computed in parallel -- are folded into the running
// sum in lane order, which is ascending index order: the same chain of fp32 additions the reference performs.
#include <hip/hip_runtime.h>

#include "xrl_device.h"
#include "xrl_kernels.h"

namespace xrl {

#define XRL_LAUNCH_CHECK() XRL_HIP(hipGetLastError())

constexpr int PG = 16;            // lanes per pair
constexpr int PAIRS_PER_BLOCK = 256 / PG;

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// acc = fl32(acc + p_b) for the lanes b of this group whose bit is set in gm, in lane order
__device__ __forceinline__ float fold_in_lane_order(float acc, float prod, uint32_t gm, int gbase) {
    float p[PG];
#pragma unroll
    for (int b = 0; b < PG; ++b) p[b] = __shfl(prod, gbase + b, 64);   // independent of the chain: all in flight together
#pragma unroll
    for (int b = 0; b < PG; ++b) { const float s = __fadd_rn(acc, p[b]); acc = ((gm >> b) & 1u) ? s : acc; }
    return acc;
}

// do_dot_product(sparse, sparse), matrix.hpp:836-859: matches in ascending index order, ret starts at 0
__device__ __forceinline__ float dot_sparse_sparse(const uint32_t* __restrict__ xi, const float* __restrict__ xv, uint32_t xn,
                                                   const uint32_t* __restrict__ wi, const float* __restrict__ wv, uint32_t wn,
                                                   int lig, int gbase) {
    // stream the shorter list, search the longer one (the set of matches and their order do not depend on the choice)
    const bool sx = xn <= wn;
    const uint32_t* __restrict__ ai = sx ? xi : wi; const float* __restrict__ av = sx ? xv : wv; const uint32_t an = sx ? xn : wn;
    const uint32_t* __restrict__ bi = sx ? wi : xi; const float* __restrict__ bv = sx ? wv : xv; const uint32_t bn = sx ? wn : xn;
    float dot = 0.0f;
    for (uint32_t c0 = 0; c0 < an; c0 += PG) {
        const uint32_t t = c0 + (uint32_t)lig;
        const bool ok = t < an;
        const uint32_t key = ai[ok ? t : 0u];
        const float a = av[ok ? t : 0u];
        const uint32_t pos = ok && bn ? lower_bound_u32(bi, bn, key) : bn;
        const bool hit = pos < bn && bi[pos] == key;
        const float prod = hit ? __fmul_rn(a, bv[pos]) : 0.0f;
        const uint32_t gm = (uint32_t)(__ballot(hit) >> gbase) & 0xFFFFu;
        if (gm) dot = fold_in_lane_order(dot, prod, gm, gbase);
    }
    return dot;
}

// res = fl32(res + fl32(x[idx_s] * w_s)) over s in [0, n) in order: do_dot_product(dense, sparse) / the dense-X bias-first loop
__device__ __forceinline__ float chain_dense_x(float res, const float* __restrict__ x, uint32_t x_cols, const uint32_t* __restrict__ wi,
                                               const float* __restrict__ wv, uint32_t n, int lig, int gbase) {
    for (uint32_t c0 = 0; c0 < n; c0 += PG) {
        const uint32_t t = c0 + (uint32_t)lig;
        const bool ok = t < n;
        const uint32_t f = wi[ok ? t : 0u];
        const float prod = ok ? __fmul_rn(f < x_cols ? x[f] : 0.0f, wv[t]) : 0.0f;
        const uint32_t cnt = min((uint32_t)PG, n - c0);
        res = fold_in_lane_order(res, prod, cnt >= 16 ? 0xFFFFu : ((1u << cnt) - 1u), gbase);
    }
    return res;
}

struct CscDev { const uint64_t* col_ptr; const uint32_t* row_idx; const float* val; uint32_t w_rows; float bias; };

// vector_ops::inner_product for column j (original column id) against query row q
__device__ __forceinline__ float csc_route_product(const CscDev& W, const QueriesDev& X, uint64_t q, uint32_t j, int lig, int gbase) {
    const uint64_t cb = W.col_ptr[j], ce = W.col_ptr[j + 1];
    const uint32_t wn = (uint32_t)(ce - cb);
    const uint32_t* __restrict__ wi = W.row_idx + cb; const float* __restrict__ wv = W.val + cb;
    const bool use_bias = W.bias > 0.0f;
    const bool has_b = use_bias && wn > 0 && wi[wn - 1] == W.w_rows - 1;
    float res = 0.0f;
    if (has_b) res = __fadd_rn(res, __fmul_rn(W.bias, wv[wn - 1]));
    if (X.dense) {
        const float* __restrict__ x = X.val + q * X.cols;
        return chain_dense_x(res, x, X.cols, wi, wv, (use_bias && has_b) ? wn - 1 : wn, lig, gbase);
    }
    const uint64_t xb = X.row_ptr[q];
    const uint32_t xn = (uint32_t)(X.row_ptr[q + 1] - xb);
    return __fadd_rn(res, dot_sparse_sparse(X.col_idx + xb, X.val + xb, xn, wi, wv, wn, lig, gbase));
}

// ---------------------------------------------------------------------------------------------
// K4
// ---------------------------------------------------------------------------------------------
struct K4Args {
    CscDev W; QueriesDev X;
    const uint32_t* pair_q; const uint32_t* node; const uint32_t* ppos;
    const uint64_t* prev_off;    // [rows+1] offsets of the previous layer's per-query lists
    const float* prev_val;
    float* out_val;
    uint64_t n_pairs;
    int pp_kind, pp_p, first_layer;
};

template <int PPC>
__global__ void __launch_bounds__(256) k4_selected_kernel(K4Args a) {
    const int lane = threadIdx.x & 63, lig = lane % PG, gbase = lane - lig;
    const uint64_t i = (uint64_t)blockIdx.x * PAIRS_PER_BLOCK + threadIdx.x / PG;
    if (i >= a.n_pairs) return;
    const uint32_t q = a.pair_q[i];
    const float res = csc_route_product(a.W, a.X, q, a.node[i], lig, gbase);
    if (lig == 0) {
        float v = pp_transform<PPC>(a.pp_kind, a.pp_p, res);
        if (!a.first_layer) v = pp_combine(a.pp_kind, v, a.prev_val[a.prev_off[q] + a.ppos[i]]);
        a.out_val[i] = v;
    }
}

void launch_k4_selected(const uint64_t* col_ptr, const uint32_t* row_idx, const float* val, uint32_t w_rows, float bias,
                        const QueriesDev& X, const uint32_t* pair_q, const uint32_t* node, const uint32_t* ppos,
                        const uint64_t* prev_off, const float* prev_val, float* out_val, uint64_t n_pairs,
                        const PostProc& pp, int first_layer, hipStream_t s) {
    if (n_pairs == 0) return;
    K4Args a;
    a.W = CscDev{col_ptr, row_idx, val, w_rows, bias}; a.X = X; a.pair_q = pair_q; a.node = node; a.ppos = ppos;
    a.prev_off = prev_off; a.prev_val = prev_val; a.out_val = out_val; a.n_pairs = n_pairs;
    a.pp_kind = pp.kind; a.pp_p = pp.p; a.first_layer = first_layer;
    const uint64_t blocks = (n_pairs + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK;
    if (blocks > 0x7FFFFFFFull) fail("k4: too many (query, label) pairs in one call");
    if (pp_class(pp)) hipLaunchKernelGGL(k4_selected_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k4_selected_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// K1C: every candidate K0 laid out for a layer, through the CSC route
// ---------------------------------------------------------------------------------------------
struct K1CArgs {
    CscDev W; QueriesDev X;
    const uint32_t* chunk_col; const uint32_t* perm_inv;
    const uint32_t* p_idx; const float* p_val; const uint32_t* p_cnt; uint32_t p_stride;
    const uint32_t* cand_off; const uint32_t* ncand; float* cand;
    uint32_t row0, nrows, beam_in, cand_stride;
    int pp_kind, pp_p, first_layer, implicit_root;
};

template <int PPC>
__global__ void __launch_bounds__(256) k1c_csc_kernel(K1CArgs a) {
    const int lane = threadIdx.x & 63, lig = lane % PG, gbase = lane - lig;
    const uint64_t g = (uint64_t)blockIdx.x * PAIRS_PER_BLOCK + threadIdx.x / PG;
    const uint64_t q = g / a.cand_stride;
    if (q >= a.nrows) return;
    const uint32_t pos = (uint32_t)(g - q * a.cand_stride);
    if (pos >= a.ncand[q]) return;
    // position -> (beam slot, child): prolongate's layout (K0)
    uint32_t parent = 0, off = 0; float pscore = 1.0f;
    if (!a.implicit_root) {
        const uint32_t cnt = min(a.p_cnt[q], a.beam_in);
        uint32_t jj = 0;
        for (uint32_t j = 1; j < cnt; ++j) if (a.cand_off[q * a.beam_in + j] <= pos) jj = j; else break;
        off = a.cand_off[q * a.beam_in + jj];
        parent = a.p_idx[q * a.p_stride + jj]; pscore = a.p_val[q * a.p_stride + jj];
    }
    const uint32_t child = a.chunk_col[parent] + (pos - off);
    const uint32_t col = a.perm_inv ? a.perm_inv[child] : child;         // W's own column id
    const float res = csc_route_product(a.W, a.X, (uint64_t)a.row0 + q, col, lig, gbase);
    if (lig == 0) {
        float v = pp_transform<PPC>(a.pp_kind, a.pp_p, res);
        if (!a.first_layer) v = pp_combine(a.pp_kind, v, pscore);
        a.cand[q * a.cand_stride + pos] = v;
    }
}

void launch_k1c_csc(const LayerDev& L, const uint64_t* col_ptr, const uint32_t* row_idx, const float* val, const LayerPlan& P,
                    const QueriesDev& X, BeamDev prev, const uint32_t* cand_off, const uint32_t* ncand, float* cand, hipStream_t s) {
    if (P.nrows == 0) return;
    K1CArgs a;
    a.W = CscDev{col_ptr, row_idx, val, L.w_rows, L.bias}; a.X = X;
    a.chunk_col = L.chunk_col; a.perm_inv = L.perm_inv;
    a.p_idx = prev.idx; a.p_val = prev.val; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
    a.cand_off = cand_off; a.ncand = ncand; a.cand = cand;
    a.row0 = P.row0; a.nrows = P.nrows; a.beam_in = P.beam_in; a.cand_stride = P.cand_stride;
    a.pp_kind = P.pp.kind; a.pp_p = P.pp.p; a.first_layer = P.first_layer; a.implicit_root = P.implicit_root;
    const uint64_t groups = (uint64_t)P.nrows * P.cand_stride;
    const uint64_t blocks = (groups + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK;
    if (blocks > 0x7FFFFFFFull) fail("k1c: grid too large; lower max_batch_rows");
    if (pp_class(P.pp)) hipLaunchKernelGGL(k1c_csc_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k1c_csc_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    XRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// K3: sparse_inner_products, the four layout combinations (do_dot_product overloads, matrix.hpp:836-877)
// ---------------------------------------------------------------------------------------------
struct K3Args {
    const uint64_t* x_ptr; const uint32_t* x_idx; const float* x_val; int x_dense;
    const uint64_t* w_ptr; const uint32_t* w_idx; const float* w_val; int w_dense;
    uint32_t dim; uint64_t len;
    const uint32_t* rows; const uint32_t* cols; float* out;
};

__global__ void __launch_bounds__(256) k3_pairs_kernel(K3Args a) {
    const int lane = threadIdx.x & 63, lig = lane % PG, gbase = lane - lig;
    const uint64_t i = (uint64_t)blockIdx.x * PAIRS_PER_BLOCK + threadIdx.x / PG;
    if (i >= a.len) return;
    const uint32_t r = a.rows[i], c = a.cols[i];
    float ret = 0.0f;
    if (a.x_dense && a.w_dense) {          // :861-868  ret += x[d] * w[d]
        const float* __restrict__ x = a.x_val + (uint64_t)r * a.dim; const float* __restrict__ w = a.w_val + (uint64_t)c * a.dim;
        for (uint32_t c0 = 0; c0 < a.dim; c0 += PG) {
            const uint32_t t = c0 + (uint32_t)lig; const bool ok = t < a.dim;
            const float prod = ok ? __fmul_rn(x[t], w[t]) : 0.0f;
            const uint32_t cnt = min((uint32_t)PG, a.dim - c0);
            ret = fold_in_lane_order(ret, prod, cnt >= 16 ? 0xFFFFu : ((1u << cnt) - 1u), gbase);
        }
    } else if (a.x_dense) {                // :870-877  ret += x[w.idx[s]] * w.val[s]
        const uint64_t wb = a.w_ptr[c];
        ret = chain_dense_x(0.0f, a.x_val + (uint64_t)r * a.dim, 0xFFFFFFFFu, a.w_idx + wb, a.w_val + wb, (uint32_t)(a.w_ptr[c + 1] - wb), lig, gbase);
    } else if (a.w_dense) {                // mirrored: ret += w[x.idx[s]] * x.val[s]
        const uint64_t xb = a.x_ptr[r];
        ret = chain_dense_x(0.0f, a.w_val + (uint64_t)c * a.dim, 0xFFFFFFFFu, a.x_idx + xb, a.x_val + xb, (uint32_t)(a.x_ptr[r + 1] - xb), lig, gbase);
    } else {                               // :836-859
        const uint64_t xb = a.x_ptr[r], wb = a.w_ptr[c];
        ret = dot_sparse_sparse(a.x_idx + xb, a.x_val + xb, (uint32_t)(a.x_ptr[r + 1] - xb), a.w_idx + wb, a.w_val + wb,
                                (uint32_t)(a.w_ptr[c + 1] - wb), lig, gbase);
    }
    if (lig == 0) a.out[i] = ret;
}

void launch_k3_inner_products(const uint64_t* x_ptr, const uint32_t* x_idx, const float* x_val, int x_dense,
                              const uint64_t* w_ptr, const uint32_t* w_idx, const float* w_val, int w_dense,
                              uint32_t dim, uint64_t len, const uint32_t* rows, const uint32_t* cols,
                              float* out, hipStream_t s) {
    if (len == 0) return;
    K3Args a{x_ptr, x_idx, x_val, x_dense, w_ptr, w_idx, w_val, w_dense, dim, len, rows, cols, out};
    const uint64_t blocks = (len + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK;
    if (blocks > 0x7FFFFFFFull) fail("k3: too many pairs in one call");
    hipLaunchKernelGGL(k3_pairs_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    XRL_LAUNCH_CHECK();
}

}  // namespace xrl
