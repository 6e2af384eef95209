// K2 for top-k sizes beyond what a workgroup's LDS holds (only_topk / beam_size > 20 480): the reference's sorted_csr
// (inference.hpp:1223-1298) has no cap -- it sorts every query's candidate row.  Here: one SEGMENTED RADIX SORT (rocPRIM) over the
// candidate rows of the whole batch, keys = the order-preserving score keys of xrl_device.h inverted (ascending sort = score descending),
// values = candidate positions; a radix sort is stable, so equal scores keep their position order -- the reference's tie-break
// (:1265-1273).  Then the first k of every row are mapped to child ids (reorder_prediction, :1776-1784) and written out.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "xrl_device.h"
#include "xrl_kernels.h"

namespace xrl {

namespace {

__global__ void __launch_bounds__(256)
k2big_keys(const float* __restrict__ cand, const uint32_t* __restrict__ ncand, uint32_t nrows, uint32_t stride, uint32_t* __restrict__ keys,
           uint32_t* __restrict__ vals, uint32_t* __restrict__ offs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i <= nrows) offs[i] = (uint32_t)(i * stride);
    if (i >= (uint64_t)nrows * stride) return;
    const uint32_t q = (uint32_t)(i / stride), p = (uint32_t)(i % stride);
    keys[i] = p < ncand[q] ? ~score_key(cand[i]) : 0xFFFFFFFFu;       // (score_key is never 0 for a candidate: its inverse never collides with the filler)
    vals[i] = p;
}

struct EmitArgs {
    const uint32_t* chunk_col; const uint32_t* perm_inv;
    const uint32_t* p_idx; const uint32_t* p_cnt; uint32_t p_stride;
    const uint32_t* cand_off; const uint32_t* ncand; const float* cand; const uint32_t* pos_sorted;
    uint32_t* out_idx; float* out_val; uint32_t* out_cnt;
    uint32_t nrows, beam_in, cand_stride, k, out_stride;
    int implicit_root;
};

__global__ void __launch_bounds__(256) k2big_emit(EmitArgs a) {
    const uint64_t q = blockIdx.x;
    const uint32_t n = min(a.ncand[q], a.cand_stride), kk = min(a.k, n);
    const uint32_t bcnt = a.implicit_root ? 1u : min(a.p_cnt[q], a.beam_in);
    for (uint32_t i = threadIdx.x; i < kk; i += 256u) {
        const uint32_t pos = a.pos_sorted[q * a.cand_stride + i];
        uint32_t parent = 0, off = 0;
        if (!a.implicit_root) {
            uint32_t lo = 0, hi = bcnt;                                 // last beam slot whose block starts at or before the position
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.cand_off[q * a.beam_in + mid] <= pos) lo = mid; else hi = mid; }
            off = a.cand_off[q * a.beam_in + lo];
            parent = a.p_idx[q * a.p_stride + lo];
        }
        const uint32_t child = a.chunk_col[parent] + (pos - off);
        a.out_idx[q * a.out_stride + i] = a.perm_inv ? a.perm_inv[child] : child;
        a.out_val[q * a.out_stride + i] = a.cand[q * a.cand_stride + pos];
    }
    if (threadIdx.x == 0) a.out_cnt[q] = kk;
}

}  // namespace

void launch_k2_topk_big(const LayerDev& L, const LayerPlan& P, BeamDev prev, const uint32_t* cand_off, const uint32_t* ncand, const float* cand,
                        uint32_t* out_idx, float* out_val, uint32_t* out_cnt, uint32_t out_stride, hipStream_t s) {
    if (P.nrows == 0) return;
    const uint64_t total = (uint64_t)P.nrows * P.cand_stride;
    if (total > 0xFFFFFFF0ull) fail("k2: candidate buffer exceeds 2^32 floats; lower max_batch_rows");
    // (a rare path: scratch is allocated per call and released after the stream has drained)
    uint32_t *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr, *offs = nullptr; void* tmp = nullptr;
    auto release = [&] { for (void* p : {(void*)keys, (void*)keys2, (void*)vals, (void*)vals2, (void*)offs, tmp}) if (p) (void)hipFree(p); };
    try {
        XRL_HIP(hipMalloc(&keys, total * 4)); XRL_HIP(hipMalloc(&keys2, total * 4)); XRL_HIP(hipMalloc(&vals, total * 4)); XRL_HIP(hipMalloc(&vals2, total * 4));
        XRL_HIP(hipMalloc(&offs, ((size_t)P.nrows + 1) * 4));
        hipLaunchKernelGGL(k2big_keys, dim3((uint32_t)((total + 256) / 256)), dim3(256), 0, s, cand, ncand, P.nrows, P.cand_stride, keys, vals, offs);
        XRL_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        XRL_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (unsigned int)total, P.nrows, offs, offs + 1, 0, 32, s));
        XRL_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
        XRL_HIP(rocprim::segmented_radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (unsigned int)total, P.nrows, offs, offs + 1, 0, 32, s));
        EmitArgs a;
        a.chunk_col = L.chunk_col; a.perm_inv = L.perm_inv; a.p_idx = prev.idx; a.p_cnt = prev.cnt; a.p_stride = prev.stride;
        a.cand_off = cand_off; a.ncand = ncand; a.cand = cand; a.pos_sorted = vals2;
        a.out_idx = out_idx; a.out_val = out_val; a.out_cnt = out_cnt;
        a.nrows = P.nrows; a.beam_in = P.beam_in; a.cand_stride = P.cand_stride; a.k = P.k; a.out_stride = out_stride; a.implicit_root = P.implicit_root;
        hipLaunchKernelGGL(k2big_emit, dim3(P.nrows), dim3(256), 0, s, a);
        XRL_HIP(hipGetLastError());
        XRL_HIP(hipStreamSynchronize(s));
    } catch (...) { release(); throw; }
    release();
}

}  // namespace xrl
