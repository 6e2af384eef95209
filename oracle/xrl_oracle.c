/*
 * xrl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's XR-Linear batch-inference
 * path (amzn/pecos @ 2024-10-20).  It exists only so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check the HIP path; nothing in pecos_amd/ may link,
 * import or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_pinned.py checks this file against
 *   (1) the reference's own golden predictions (test/tst-data/xmc/xlinear/*.npz) on models the
 *       reference itself trained (fixtures + generating script under tests/golden/), and
 *   (2) outputs of the real reference compiled from /root/reference (oracle/_ref) on seeded
 *       synthetic models, bit-for-bit (indices and fp32 scores).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * The arithmetic contract (order of every fp32 operation) is the one of the reference's default
 * layout, sparse X x BINARY_SEARCH_CHUNKED W:
 *   pecos/core/xmc/inference.hpp:769-813 (intersection walk, bias LAST)
 *   pecos/core/xmc/inference.hpp:506-518 (out[c] += scalar * val; separate mul and add)
 * and for dense X x BINARY_SEARCH_CHUNKED W:
 *   pecos/core/xmc/inference.hpp:815-839 (bias FIRST, then every chunk row)
 *
 * Build: see oracle/Makefile (-ffp-contract=off: the reference .so carries no FMA on this path).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t rows, cols;
    const uint64_t* col_ptr;
    const uint32_t* row_idx;
    const float* val;
} orc_csc_t; /* mirrors ScipyCscF32, pecos/core/utils/matrix.hpp:56-62 */

/* post-processor kinds, pecos/core/xmc/inference.hpp:192-240 */
enum { ORC_PP_NOOP = 0, ORC_PP_SIGMOID = 1, ORC_PP_LOG_SIGMOID = 2, ORC_PP_LP_HINGE = 3, ORC_PP_LOG_LP_HINGE = 4 };

typedef struct {
    orc_csc_t W;     /* (D [+1 if bias>0]) x K_l */
    orc_csc_t C;     /* K_l x K_{l-1}; only the pattern is used */
    float bias;      /* MLModelMetadata.bias, inference.hpp:120-157 */
    uint32_t only_topk;
    int32_t pp_kind;
    int32_t pp_p;
} orc_layer_t;

/* inference.hpp:192-240.  NB the reference lambdas take `const float& v`:
 *   sigmoid      1.0 / (1.0 + std::exp(-v))      -> std::exp(float) == expf, rest in double
 *   log-sigmoid  -std::log(1.0 + std::exp(-v))   -> expf, then double log
 *   l{p}-hinge   T z = std::max(0.0, 1.0 - v); std::exp(-std::pow(z, p))  (z is a FLOAT variable,
 *                pow(float,size_t) promotes to double, exp in double)
 *   log-l{p}-hinge  -std::pow(z, p)
 * and transform_matrix_csr casts the result to float (inference.hpp:1369). */
static float orc_transform(int kind, int p, float v) {
    switch (kind) {
    case ORC_PP_SIGMOID: return (float)(1.0 / (1.0 + (double)expf(-v)));
    case ORC_PP_LOG_SIGMOID: return (float)(-log(1.0 + (double)expf(-v)));
    case ORC_PP_LP_HINGE: {
        float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)exp(-pow((double)z, (double)p));
    }
    case ORC_PP_LOG_LP_HINGE: {
        float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)(-pow((double)z, (double)p));
    }
    default: return v;
    }
}

/* inference.hpp:1373-1384 + the Combiner of each post-processor (std::multiplies<float> /
 * std::plus<float> / noop keeps x). */
static float orc_combine(int kind, float x, float parent) {
    switch (kind) {
    case ORC_PP_SIGMOID:
    case ORC_PP_LP_HINGE: return x * parent;
    case ORC_PP_LOG_SIGMOID:
    case ORC_PP_LOG_LP_HINGE: return x + parent;
    default: return x;
    }
}

typedef struct { float val; uint32_t pos; uint32_t id; } orc_cand_t;

/* comparator of sorted_csr, inference.hpp:1265-1273: value descending, ties -> smaller position */
static int orc_cand_cmp(const void* a, const void* b) {
    const orc_cand_t* x = (const orc_cand_t*)a;
    const orc_cand_t* y = (const orc_cand_t*)b;
    if (x->val == y->val) return (x->pos > y->pos) - (x->pos < y->pos);
    return (x->val > y->val) ? -1 : 1;
}

/*
 * One layer of MLModel::predict_internal (inference.hpp:2029-2080) for all queries.
 *   X: CSR (x_dense == NULL) or row-major dense (x_dense != NULL, x_cols wide).
 *   prev_*: previous beam, fixed stride `prev_stride` per query (cnt entries valid);
 *           prev == NULL means the all-ones N x 1 start (inference.hpp:2462-2463).
 *   out_*:  new beam, stride `out_stride` (>= k).
 * Returns 0, or -1 on allocation failure / bad shape.
 */
/* weight_matrix_type HASH_CHUNKED with sparse X: chunk_ops<csr, hash_chunked> (inference.hpp:705-735) adds the bias row FIRST
 * and then the query's features in ascending order (the hash map only replaces the row lookup); set by the tests that pin
 * that layout.  Dense X under the hash layout walks the hash map in ITS order (:737-768) -- not restated. */
/* thread-local: two OracleModel instances of different layouts driven from different threads -- pytest-xdist workers, bench.py's
 * threaded parity check -- each set and read their own flag; the restatement itself is single-threaded. */
static __thread int orc_hash_arith = 0;
void orc_set_hash_arith(int on) { orc_hash_arith = on; }

int orc_layer_predict(const orc_layer_t* L, uint32_t n_rows,
                      const uint64_t* x_indptr, const uint32_t* x_idx, const float* x_val,
                      const float* x_dense, uint32_t x_cols,
                      const uint32_t* prev_idx, const float* prev_val, const uint32_t* prev_cnt,
                      uint32_t prev_stride, int no_prev_pred, uint32_t k, int pp_kind, int pp_p,
                      uint32_t* out_idx, float* out_val, uint32_t* out_cnt, uint32_t out_stride) {
    const orc_csc_t* W = &L->W;
    const orc_csc_t* C = &L->C;
    const int use_bias = L->bias > 0.0f;           /* LayerData::init, inference.hpp:1850 */
    const uint32_t bias_row = W->rows - 1;         /* check_bias_explicit, inference.hpp:500-502 */
    if (k > out_stride) return -1;

    /* dense scatter of the query (touched flags so that explicit zeros still "match") */
    float* xs = (float*)calloc((size_t)W->rows + 1, sizeof(float));
    uint8_t* touched = (uint8_t*)calloc((size_t)W->rows + 1, 1);
    size_t cap = 1024;
    orc_cand_t* cand = (orc_cand_t*)malloc(cap * sizeof(orc_cand_t));
    if (!xs || !touched || !cand) { free(xs); free(touched); free(cand); return -1; }

    for (uint32_t q = 0; q < n_rows; ++q) {
        const float* xd = x_dense ? x_dense + (size_t)q * x_cols : NULL;
        if (!xd) {
            for (uint64_t t = x_indptr[q]; t < x_indptr[q + 1]; ++t)
                if (x_idx[t] < W->rows) { xs[x_idx[t]] = x_val[t]; touched[x_idx[t]] = 1; }
        }
        uint32_t one_idx = 0; float one_val = 1.0f;
        const uint32_t* p_idx = prev_idx ? prev_idx + (size_t)q * prev_stride : &one_idx;
        const float* p_val = prev_val ? prev_val + (size_t)q * prev_stride : &one_val;
        const uint32_t p_cnt = prev_cnt ? prev_cnt[q] : 1;

        /* prolongate_predictions, inference.hpp:1155-1219: beam order, then C's stored order */
        size_t n = 0;
        for (uint32_t b = 0; b < p_cnt; ++b) {
            const uint32_t parent = p_idx[b];
            if (parent >= C->cols) { free(xs); free(touched); free(cand); return -1; }
            for (uint64_t c = C->col_ptr[parent]; c < C->col_ptr[parent + 1]; ++c) {
                const uint32_t j = C->row_idx[c];
                if (n == cap) {
                    cap *= 2;
                    orc_cand_t* nc = (orc_cand_t*)realloc(cand, cap * sizeof(orc_cand_t));
                    if (!nc) { free(xs); free(touched); free(cand); return -1; }
                    cand = nc;
                }
                float acc = 0.0f;
                const uint64_t cb = W->col_ptr[j], ce = W->col_ptr[j + 1];
                const int has_bias = use_bias && ce > cb && W->row_idx[ce - 1] == bias_row;
                const uint64_t ce_nb = has_bias ? ce - 1 : ce;
                if (xd) {
                    /* chunk_ops<drm, bin_search>, inference.hpp:815-839: bias first, every row */
                    if (has_bias) { float pr = L->bias * W->val[ce - 1]; acc = acc + pr; }
                    for (uint64_t e = cb; e < ce_nb; ++e) {
                        float pr = xd[W->row_idx[e]] * W->val[e];
                        acc = acc + pr;
                    }
                } else {
                    /* chunk_ops<csr, bin_search>, inference.hpp:769-813: matched rows ascending,
                     * out += x_f * w (mul then add, :512-517); bias last (:806-811) -- or first, hash layout (:716-722) */
                    if (has_bias && orc_hash_arith) { float pr = L->bias * W->val[ce - 1]; acc = acc + pr; }
                    for (uint64_t e = cb; e < ce_nb; ++e) {
                        const uint32_t f = W->row_idx[e];
                        if (touched[f]) { float pr = xs[f] * W->val[e]; acc = acc + pr; }
                    }
                    if (has_bias && !orc_hash_arith) { float pr = L->bias * W->val[ce - 1]; acc = acc + pr; }
                }
                float v = orc_transform(pp_kind, pp_p, acc);            /* inference.hpp:1360-1371 */
                if (!no_prev_pred) v = orc_combine(pp_kind, v, p_val[b]); /* inference.hpp:2071-2073 */
                cand[n].val = v; cand[n].pos = (uint32_t)n; cand[n].id = j;
                ++n;
            }
        }
        /* sorted_csr, inference.hpp:1223-1298: rows shorter than k stay short */
        qsort(cand, n, sizeof(orc_cand_t), orc_cand_cmp);
        const uint32_t m = (uint32_t)(n < k ? n : k);
        for (uint32_t i = 0; i < m; ++i) {
            out_idx[(size_t)q * out_stride + i] = cand[i].id;
            out_val[(size_t)q * out_stride + i] = cand[i].val;
        }
        out_cnt[q] = m;

        if (!xd) {
            for (uint64_t t = x_indptr[q]; t < x_indptr[q + 1]; ++t)
                if (x_idx[t] < W->rows) { xs[x_idx[t]] = 0.0f; touched[x_idx[t]] = 0; }
        }
    }
    free(xs); free(touched); free(cand);
    return 0;
}

/*
 * HierarchicalMLModel::predict, inference.hpp:2446-2488.
 * beam / topk / pp_kind follow the C ABI's override semantics: 0 (or pp_kind < 0) means
 * "use the layer's own param.json value" (inference.hpp:2055-2058).
 * Output stride is `out_stride`; if layer_trace_* are non-NULL they receive every layer's beam
 * (depth x n_rows x out_stride), for kernel debugging.
 */
int orc_predict(const orc_layer_t* layers, uint32_t depth, uint32_t n_rows,
                const uint64_t* x_indptr, const uint32_t* x_idx, const float* x_val,
                const float* x_dense, uint32_t x_cols,
                uint32_t beam, uint32_t topk, int pp_kind, int pp_p,
                uint32_t* out_idx, float* out_val, uint32_t* out_cnt, uint32_t out_stride,
                uint32_t* trace_idx, float* trace_val, uint32_t* trace_cnt) {
    uint32_t stride = out_stride;
    for (uint32_t l = 0; l < depth; ++l) {
        uint32_t kl = (l == depth - 1) ? topk : beam;
        if (kl == 0) kl = layers[l].only_topk;
        if (kl > stride) stride = kl;
    }
    const size_t cells = (size_t)n_rows * stride;
    uint32_t* a_idx = (uint32_t*)malloc((cells + 1) * 4); float* a_val = (float*)malloc((cells + 1) * 4);
    uint32_t* b_idx = (uint32_t*)malloc((cells + 1) * 4); float* b_val = (float*)malloc((cells + 1) * 4);
    uint32_t* a_cnt = (uint32_t*)malloc(((size_t)n_rows + 1) * 4);
    uint32_t* b_cnt = (uint32_t*)malloc(((size_t)n_rows + 1) * 4);
    int rc = (a_idx && a_val && b_idx && b_val && a_cnt && b_cnt) ? 0 : -1;
    const uint32_t *pi = NULL, *pc = NULL; const float* pv = NULL;
    for (uint32_t l = 0; l < depth && rc == 0; ++l) {
        uint32_t kl = (l == depth - 1) ? topk : beam;              /* inference.hpp:2471 */
        if (kl == 0) kl = layers[l].only_topk;                     /* inference.hpp:2055 */
        const int kind = pp_kind >= 0 ? pp_kind : layers[l].pp_kind;
        const int p = pp_kind >= 0 ? pp_p : layers[l].pp_p;
        uint32_t* oi = (l & 1) ? b_idx : a_idx; float* ov = (l & 1) ? b_val : a_val;
        uint32_t* oc = (l & 1) ? b_cnt : a_cnt;
        rc = orc_layer_predict(&layers[l], n_rows, x_indptr, x_idx, x_val, x_dense, x_cols,
                               pi, pv, pc, stride, l == 0, kl, kind, p, oi, ov, oc, stride);
        if (rc == 0 && trace_idx) {
            for (uint32_t q = 0; q < n_rows; ++q) {
                trace_cnt[(size_t)l * n_rows + q] = oc[q];
                for (uint32_t i = 0; i < oc[q] && i < out_stride; ++i) {
                    trace_idx[((size_t)l * n_rows + q) * out_stride + i] = oi[(size_t)q * stride + i];
                    trace_val[((size_t)l * n_rows + q) * out_stride + i] = ov[(size_t)q * stride + i];
                }
            }
        }
        pi = oi; pv = ov; pc = oc;
    }
    if (rc == 0) {
        for (uint32_t q = 0; q < n_rows; ++q) {
            uint32_t m = depth ? pc[q] : 0;
            if (m > out_stride) m = out_stride;
            out_cnt[q] = m;
            for (uint32_t i = 0; i < m; ++i) {
                out_idx[(size_t)q * out_stride + i] = pi[(size_t)q * stride + i];
                out_val[(size_t)q * out_stride + i] = pv[(size_t)q * stride + i];
            }
        }
    }
    free(a_idx); free(a_val); free(b_idx); free(b_val); free(a_cnt); free(b_cnt);
    return rc;
}

/*
 * compute_sparse_entries_from_rowmajored_X_and_colmajored_M, pecos/core/utils/matrix.hpp:1049-1060
 * with the four do_dot_product overloads it reaches:
 *   sparse x sparse  matrix.hpp:836-859   (walk both ascending; ret += x*y on equal ids)
 *   dense  x sparse  matrix.hpp:870-877   (ret += x[y.idx[s]] * y.val[s], every s)
 *   dense  x dense   matrix.hpp:861-868   (ret += x[i]*y[i], i ascending)
 * x_* : CSR if x_dense == NULL else row-major dense; w_*: CSC if w_dense == NULL else col-major
 * dense with `dim` rows per column.
 */
int orc_sparse_inner_products(const uint64_t* x_indptr, const uint32_t* x_idx, const float* x_val,
                              const float* x_dense, const uint64_t* w_indptr,
                              const uint32_t* w_idx, const float* w_val, const float* w_dense,
                              uint32_t dim, uint64_t len, const uint32_t* rows,
                              const uint32_t* cols, float* out) {
    for (uint64_t i = 0; i < len; ++i) {
        const uint32_t r = rows[i], c = cols[i];
        float ret = 0.0f;
        if (x_dense && w_dense) {
            const float* x = x_dense + (size_t)r * dim; const float* w = w_dense + (size_t)c * dim;
            for (uint32_t d = 0; d < dim; ++d) { float pr = x[d] * w[d]; ret = ret + pr; }
        } else if (x_dense) {
            const float* x = x_dense + (size_t)r * dim;
            for (uint64_t s = w_indptr[c]; s < w_indptr[c + 1]; ++s) {
                float pr = x[w_idx[s]] * w_val[s]; ret = ret + pr;
            }
        } else if (w_dense) {
            const float* w = w_dense + (size_t)c * dim;
            for (uint64_t s = x_indptr[r]; s < x_indptr[r + 1]; ++s) {
                float pr = w[x_idx[s]] * x_val[s]; ret = ret + pr;
            }
        } else {
            uint64_t s = x_indptr[r], se = x_indptr[r + 1], t = w_indptr[c], te = w_indptr[c + 1];
            while (s < se && t < te) {
                if (x_idx[s] == w_idx[t]) { float pr = x_val[s] * w_val[t]; ret = ret + pr; ++s; ++t; }
                else if (x_idx[s] < w_idx[t]) ++s;
                else ++t;
            }
        }
        out[i] = ret;
    }
    return 0;
}

/*
 * HierarchicalMLModel::predict_on_selected_outputs, inference.hpp:2507-2571, with
 *   the per-layer patterns   S_{l-1} = pattern(S_l x C_l) (smat_x_smat, sorted indices)   :2527-2541
 *   prolongate_sparse_predictions                                                        :1302-1358
 *   w_ops<csc_t>::compute_sparse_predictions / vector_ops::inner_product                 :1018-1078, 1081-1149
 *     sparse X: res = 0; res += bias*w_bias (if explicit); res += dot(x, w)   (dot accumulated separately)
 *     dense  X: bias>0: res = bias*w_bias, then res += x[idx]*w over the non-bias entries in order
 *               bias<=0: dot over all entries
 * (only LAYER_TYPE_CSC supports this entry point in the reference, :2143-2147).
 * sel_*: CSR pattern of the selected outputs (N x L).  Output: out_idx/out_val in the reference's
 * traversal order, row_ptr identical to sel_indptr.  Returns 0, or -1 / -2 (a selected label has no
 * path to the root / appears twice: the reference would emit uninitialised slots).
 */
static int orc_u32_cmp(const void* a, const void* b) {
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

int orc_predict_selected(const orc_layer_t* layers, uint32_t depth, uint32_t n_rows,
                         const uint64_t* x_indptr, const uint32_t* x_idx, const float* x_val,
                         const float* x_dense, uint32_t x_cols,
                         const uint64_t* sel_indptr, const uint32_t* sel_idx,
                         int pp_kind, int pp_p, uint32_t* out_idx, float* out_val) {
    int rc = 0;
    /* child -> parent maps */
    uint32_t** parent = (uint32_t**)calloc(depth, sizeof(uint32_t*));
    for (uint32_t l = 0; l < depth; ++l) {
        const orc_csc_t* C = &layers[l].C;
        parent[l] = (uint32_t*)malloc(((size_t)C->rows + 1) * 4);
        for (uint32_t i = 0; i < C->rows; ++i) parent[l][i] = 0xFFFFFFFFu;
        for (uint32_t p = 0; p < C->cols; ++p)
            for (uint64_t c = C->col_ptr[p]; c < C->col_ptr[p + 1]; ++c) parent[l][C->row_idx[c]] = p;
    }
    uint32_t max_w_rows = 0;
    for (uint32_t l = 0; l < depth; ++l) if (layers[l].W.rows > max_w_rows) max_w_rows = layers[l].W.rows;
    float* xs = (float*)calloc((size_t)max_w_rows + 1, sizeof(float));
    uint8_t* touched = (uint8_t*)calloc((size_t)max_w_rows + 1, 1);
    for (uint32_t q = 0; q < n_rows && rc == 0; ++q) {
        const uint64_t sb = sel_indptr[q], se = sel_indptr[q + 1];
        const size_t n_sel = (size_t)(se - sb);
        /* patterns bottom-up: pat[l] = sorted unique nodes of layer l */
        uint32_t** pat = (uint32_t**)calloc(depth, sizeof(uint32_t*));
        size_t* npat = (size_t*)calloc(depth, sizeof(size_t));
        pat[depth - 1] = (uint32_t*)malloc((n_sel + 1) * 4);
        memcpy(pat[depth - 1], sel_idx + sb, n_sel * 4);
        qsort(pat[depth - 1], n_sel, 4, orc_u32_cmp);          /* membership is a set (valid_cols, :1325-1329) */
        for (size_t i = 1; i < n_sel; ++i) if (pat[depth - 1][i] == pat[depth - 1][i - 1]) rc = -2;
        npat[depth - 1] = n_sel;
        for (uint32_t l = depth - 1; l > 0; --l) {
            pat[l - 1] = (uint32_t*)malloc((npat[l] + 1) * 4);
            size_t m = 0;
            for (size_t i = 0; i < npat[l]; ++i) {
                const uint32_t pr = pat[l][i] < layers[l].C.rows ? parent[l][pat[l][i]] : 0xFFFFFFFFu;
                if (pr == 0xFFFFFFFFu) { rc = -1; break; }
                pat[l - 1][m++] = pr;
            }
            qsort(pat[l - 1], m, 4, orc_u32_cmp);
            size_t u = 0;
            for (size_t i = 0; i < m; ++i) if (i == 0 || pat[l - 1][i] != pat[l - 1][i - 1]) pat[l - 1][u++] = pat[l - 1][i];
            npat[l - 1] = u;
        }
        if (!x_dense)
            for (uint64_t t = x_indptr[q]; t < x_indptr[q + 1]; ++t)
                if (x_idx[t] <= max_w_rows) { xs[x_idx[t]] = x_val[t]; touched[x_idx[t]] = 1; }
        const float* xd = x_dense ? x_dense + (size_t)q * x_cols : NULL;
        /* top-down */
        uint32_t root = 0; float one = 1.0f;
        uint32_t* prev_node = &root; float* prev_v = &one; size_t n_prev = 1;
        int prev_owned = 0;
        for (uint32_t l = 0; l < depth && rc == 0; ++l) {
            const orc_layer_t* L = &layers[l];
            const int kind = pp_kind >= 0 ? pp_kind : L->pp_kind;
            const int pw = pp_kind >= 0 ? pp_p : L->pp_p;
            const int use_bias = L->bias > 0.0f;
            uint32_t* node = (uint32_t*)malloc((npat[l] + 1) * 4);
            float* val = (float*)malloc((npat[l] + 1) * 4);
            size_t k = 0;
            for (size_t i = 0; i < n_prev && rc == 0; ++i) {
                const uint32_t p = prev_node[i];
                if (p >= L->C.cols) { rc = -1; break; }
                for (uint64_t c = L->C.col_ptr[p]; c < L->C.col_ptr[p + 1]; ++c) {
                    const uint32_t j = L->C.row_idx[c];
                    if (!bsearch(&j, pat[l], npat[l], 4, orc_u32_cmp)) continue;   /* valid_cols.count(...) */
                    if (k >= npat[l]) { rc = -2; break; }
                    const uint64_t cb = L->W.col_ptr[j], ce = L->W.col_ptr[j + 1];
                    const int has_b = use_bias && ce > cb && L->W.row_idx[ce - 1] == L->W.rows - 1;
                    float res = 0.0f;
                    if (xd) {
                        if (use_bias) {
                            uint64_t range = ce;
                            if (has_b) { range = ce - 1; float pr = L->bias * L->W.val[ce - 1]; res = res + pr; }
                            for (uint64_t e = cb; e < range; ++e) { float pr = xd[L->W.row_idx[e]] * L->W.val[e]; res = res + pr; }
                        } else {
                            float ret = 0.0f;
                            for (uint64_t e = cb; e < ce; ++e) { float pr = xd[L->W.row_idx[e]] * L->W.val[e]; ret = ret + pr; }
                            res = ret;
                        }
                    } else {
                        if (has_b) { float pr = L->bias * L->W.val[ce - 1]; res = res + pr; }
                        float ret = 0.0f;
                        for (uint64_t e = cb; e < ce; ++e) {
                            const uint32_t f = L->W.row_idx[e];
                            if (touched[f]) { float pr = xs[f] * L->W.val[e]; ret = ret + pr; }
                        }
                        res = res + ret;
                    }
                    float v = orc_transform(kind, pw, res);
                    if (l > 0) v = orc_combine(kind, v, prev_v[i]);
                    node[k] = j; val[k] = v; ++k;
                }
            }
            if (rc == 0 && k != npat[l]) rc = -2;
            if (prev_owned) { free(prev_node); free(prev_v); }
            prev_node = node; prev_v = val; n_prev = k; prev_owned = 1;
        }
        if (rc == 0) {
            if (n_prev != n_sel) rc = -2;
            else for (size_t i = 0; i < n_sel; ++i) { out_idx[sb + i] = prev_node[i]; out_val[sb + i] = prev_v[i]; }
        }
        if (prev_owned) { free(prev_node); free(prev_v); }
        if (!x_dense)
            for (uint64_t t = x_indptr[q]; t < x_indptr[q + 1]; ++t)
                if (x_idx[t] <= max_w_rows) { xs[x_idx[t]] = 0.0f; touched[x_idx[t]] = 0; }
        for (uint32_t l = 0; l < depth; ++l) free(pat[l]);
        free(pat); free(npat);
    }
    for (uint32_t l = 0; l < depth; ++l) free(parent[l]);
    free(parent); free(xs); free(touched);
    return rc;
}
