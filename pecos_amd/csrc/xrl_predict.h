// Host orchestrator of the beam search (HierarchicalMLModel::predict, inference.hpp:2446-2488).
#pragma once
#include "xrl_kernels.h"
#include "xrl_model.h"

namespace xrl {


struct Queries {
    int device = 0;
    DevBuf ptr, idx, val;
    QueriesDev dev{};
    uint64_t nnz = 0;
};

struct PredictOpts {
    uint32_t beam_size = 0;          // 0 = per-layer param.json value
    uint32_t only_topk = 0;          // 0 = last layer's param.json value
    const char* post_processor = nullptr;
    const BeamDev* initial = nullptr;  // non-null: explicit previous-layer predictions (csr_codes)
    uint32_t initial_max = 0;          // max entries per row in `initial`
    double* stats_out = nullptr;       // host [depth * kStatsPerLayer], see launch_stats
    bool no_prev_pred = false;         // explicit initial beam but no combine (fill_ones case, libpecos.cpp:219-222)
    uint64_t initial_cand_bound = 0;   // explicit initial beam: max over rows of the candidates it prolongates to (0 = bound from the largest chunks)
    uint32_t reserve_rows = 0;         // callers that predict row ranges of different sizes: the largest one (scratch is sized once, no realloc mid-pipeline)
    bool csc_route = false;            // the reference's CSC arithmetic (w_ops<csc_t>): K0 -> K1C -> K2 on every layer
};

uint32_t effective_topk(const Model& m, uint32_t only_topk);
void resolve_profile(Model& m);
// predict_on_selected_outputs (xrl_select.cpp): values for the pattern (s_ptr, s_idx), in the reference's walk order
struct ScipyCsrF32View { uint32_t rows, cols; const uint64_t* row_ptr; const uint32_t* col_idx; const float* val; };
struct SelectedInit { const ScipyCsrF32View* codes; bool no_prev_pred; };   // explicit predictions entering layer 0
void predict_selected(Model& m, const QueriesDev& X, uint32_t s_rows, uint32_t s_cols, const uint64_t* s_ptr,
                      const uint32_t* s_idx, const char* post_processor, std::vector<uint32_t>& out_idx,
                      std::vector<float>& out_val, const SelectedInit* init = nullptr);   // synchronise and fold pending hipEvent pairs into m.profile

// Enqueue the whole beam search on `stream`; results land in fixed-stride device buffers.
void predict_device(Model& m, const QueriesDev& X, const PredictOpts& o, uint32_t* d_out_idx, float* d_out_val,
                    uint32_t* d_out_cnt, uint32_t out_stride, hipStream_t stream, bool sync,
                    uint32_t row_begin = 0, uint32_t row_count = 0xFFFFFFFFu);

}  // namespace xrl
