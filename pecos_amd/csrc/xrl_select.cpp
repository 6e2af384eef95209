// predict_on_selected_outputs: score a given (query, label) sparsity pattern through the tree.
//
// Reference: HierarchicalMLModel::predict_on_selected_outputs (inference.hpp:2507-2571):
//   * per-layer patterns bottom-up, S_{l-1} = pattern(S_l x C_l) with sorted indices (:2527-2541);
//   * per layer top-down, prolongate_sparse_predictions (:1302-1358) walks the previous layer's
//     entries IN THEIR ORDER and appends, for each, the children (in C's stored order) that belong to
//     this layer's pattern -- that walk defines the order of the output row;
//   * scores come from the CSC route (vector_ops::inner_product, :1018-1078), K4 in xrl_kernels.hip.
// The pattern bookkeeping is integer work on small sets and stays on the host; the inner products,
// transform and combine run on the GPU, one launch per layer.
#include <algorithm>

#include "xrl_predict.h"

namespace xrl {

void predict_selected(Model& m, const QueriesDev& X, uint32_t s_rows, uint32_t s_cols, const uint64_t* s_ptr,
                      const uint32_t* s_idx, const char* post_processor, std::vector<uint32_t>& out_idx,
                      std::vector<float>& out_val, const SelectedInit* init) {
    const size_t T = m.layers.size();
    const Layer& last = *m.layers.back();
    const uint32_t out_cols = last.reordered ? last.c_rows : last.w_cols;
    if (s_rows != X.rows) fail("Instance dimension of query and selected output matrix do not match");
    if (s_cols != out_cols) fail("Label dimension of selected output matrix does not match");
    if (!X.dense && X.cols != m.nr_features && X.cols != m.layers[0]->w_rows) fail("Feature dimension of query matrix does not match weight matrix");
    const uint32_t N = s_rows;
    const uint64_t nnz = s_ptr[N];

    // ---- patterns, bottom-up (sorted unique per query)
    std::vector<std::vector<uint64_t>> pat_ptr(T, std::vector<uint64_t>(N + 1, 0));
    std::vector<std::vector<uint32_t>> pat(T);
    pat[T - 1].assign(s_idx, s_idx + nnz);
    for (uint32_t q = 0; q <= N; ++q) pat_ptr[T - 1][q] = s_ptr[q];
    for (uint32_t q = 0; q < N; ++q) {
        auto b = pat[T - 1].begin() + s_ptr[q], e = pat[T - 1].begin() + s_ptr[q + 1];
        std::sort(b, e);
        if (std::adjacent_find(b, e) != e) fail("selected_outputs_csr row " + std::to_string(q) + " holds a label twice");
        if (b != e && *(e - 1) >= out_cols) fail("selected_outputs_csr holds a label id out of range");
    }
    for (size_t l = T - 1; l > 0; --l) {
        const Layer& L = *m.layers[l];
        std::vector<uint32_t> tmp;
        for (uint32_t q = 0; q < N; ++q) {
            tmp.clear();
            for (uint64_t i = pat_ptr[l][q]; i < pat_ptr[l][q + 1]; ++i) {
                const uint32_t pr = L.h_parent[pat[l][i]];
                if (pr == 0xFFFFFFFFu) fail("selected label " + std::to_string(pat[l][i]) + " has no parent in layer " + std::to_string(l) + " (pruned tree)");
                tmp.push_back(pr);
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            pat[l - 1].insert(pat[l - 1].end(), tmp.begin(), tmp.end());
            pat_ptr[l - 1][q + 1] = pat[l - 1].size();
        }
    }

    // ---- traversal order, top-down (prolongate_sparse_predictions)
    std::vector<std::vector<uint32_t>> node(T), ppos(T), pair_q(T);
    std::vector<uint32_t> mark;
    // previous-layer predictions entering layer 0: the implicit root (ones(N x 1)), explicit csr_codes, or
    // ones(N x C.cols) without combine (single-layer API, libpecos.cpp:237-274)
    std::vector<uint64_t> init_ptr; std::vector<uint32_t> init_node; std::vector<float> init_val;
    const bool has_init = init != nullptr;
    if (has_init) {
        const uint32_t P0 = m.layers[0]->c_cols;
        init_ptr.assign(N + 1, 0);
        if (init->codes) {
            if (init->codes->rows != N) fail("Instance dimension of query and prev_layer_pred matrix do not match");
            if (init->codes->cols != P0) fail("Label dimension of prev_layer_pred and C matrix do not match");
            const uint64_t cn = init->codes->row_ptr[N];
            init_ptr.assign(init->codes->row_ptr, init->codes->row_ptr + N + 1);
            init_node.assign(init->codes->col_idx, init->codes->col_idx + cn);
            init_val.assign(init->codes->val, init->codes->val + cn);
        } else {
            init_node.resize((size_t)N * P0); init_val.assign((size_t)N * P0, 1.0f);
            for (uint32_t q = 0; q < N; ++q) { init_ptr[q + 1] = (uint64_t)(q + 1) * P0; for (uint32_t p = 0; p < P0; ++p) init_node[(size_t)q * P0 + p] = p; }
        }
    }
    for (size_t l = 0; l < T; ++l) {
        const Layer& L = *m.layers[l];
        mark.assign((size_t)L.c_rows + 1, 0u);
        node[l].reserve(pat[l].size()); ppos[l].reserve(pat[l].size()); pair_q[l].reserve(pat[l].size());
        for (uint32_t q = 0; q < N; ++q) {
            const uint32_t stamp = q + 1;
            for (uint64_t i = pat_ptr[l][q]; i < pat_ptr[l][q + 1]; ++i) mark[pat[l][i]] = stamp;
            const uint64_t before = node[l].size();
            const uint64_t pb = l ? pat_ptr[l - 1][q] : (has_init ? init_ptr[q] : 0), pe = l ? pat_ptr[l - 1][q + 1] : (has_init ? init_ptr[q + 1] : 1);
            for (uint64_t i = pb; i < pe; ++i) {
                const uint32_t parent = l ? node[l - 1][i] : (has_init ? init_node[i] : 0u);   // previous layer's ORDERED list
                if (parent >= L.c_cols) fail("selected-output walk left the tree");
                for (uint64_t c = L.h_c_ptr[parent]; c < L.h_c_ptr[parent + 1]; ++c) {
                    const uint32_t j = L.h_c_idx[c];
                    if (mark[j] == stamp) { node[l].push_back(j); ppos[l].push_back((uint32_t)(i - pb)); pair_q[l].push_back(q); }
                }
            }
            if (node[l].size() - before != pat_ptr[l][q + 1] - pat_ptr[l][q]) fail("selected-output pattern is inconsistent with the cluster chain");
        }
    }

    // ---- device: one K4 launch per layer
    if (!m.ws) m.ws = std::make_unique<Workspace>();
    hipStream_t stream = m.stream;
    DevBuf d_node, d_ppos, d_q, d_off[2], d_val[2];
    if (has_init) { d_off[1].upload(init_ptr); d_val[1].upload(init_val); }   // plays "layer -1" (prv of layer 0)
    for (size_t l = 0; l < T; ++l) {
        Layer& L = *m.layers[l];
        ensure_device_csc(L);
        const uint64_t np = node[l].size();
        d_node.upload(node[l]); d_ppos.upload(ppos[l]); d_q.upload(pair_q[l]);
        const int cur = (int)(l & 1), prv = cur ^ 1;
        d_off[cur].upload(pat_ptr[l]);
        d_val[cur].reserve(np * 4);
        const PostProc pp = post_processor ? parse_post_processor(post_processor) : L.pp;
        launch_k4_selected(L.d_csc_ptr.as<uint64_t>(), L.d_csc_idx.as<uint32_t>(), L.d_csc_val.as<float>(), L.w_rows, L.bias, X,
                           d_q.as<uint32_t>(), d_node.as<uint32_t>(), d_ppos.as<uint32_t>(),
                           (l || has_init) ? d_off[prv].as<uint64_t>() : nullptr, (l || has_init) ? d_val[prv].as<float>() : nullptr,
                           d_val[cur].as<float>(), np, pp, (l == 0 && (!has_init || init->no_prev_pred)) ? 1 : 0, stream);
        XRL_HIP(hipStreamSynchronize(stream));   // the upload buffers are reused by the next layer
    }
    out_idx = std::move(node[T - 1]);
    out_val.resize(out_idx.size());
    if (!out_val.empty())
        XRL_HIP(hipMemcpy(out_val.data(), d_val[(T - 1) & 1].p, out_val.size() * 4, hipMemcpyDeviceToHost));
}

}  // namespace xrl
