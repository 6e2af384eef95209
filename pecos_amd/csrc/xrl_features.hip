// Query-side featurizer kernels: what produces X stays on the device, so the beam search needs no H2D of X.
//
//   tfidf_weight_kernel   the WEIGHTING half of the reference's TF-IDF vectorizer -- BaseVectorizer::get_sorted_feature,
//                         pecos/core/utils/tfidf.hpp:798-822 (c_tfidf_predict, libpecos.cpp:427-445): term counts of a document, in
//                         ascending feature id, -> binary / sublinear tf -> x idf -> l1 / l2 normalisation.  Tokenisation and the
//                         n-gram lookup (string work, :775-793) stay the reference's host code; their output -- a CSR of term
//                         COUNTS -- is this kernel's input.
//
// Arithmetic follows the reference's float32 code operation by operation: the norm is accumulated SEQUENTIALLY in ascending
// feature order (one wavefront per document, a scalar loop over its lanes), multiply and add rounded separately (-ffp-contract=off), sqrtf / division IEEE.
// With sublinear_tf the reference calls glibc's logf, whose last bit the device's logf may not share: <= 1 ulp there.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "xrl_kernels.h"

namespace xrl {

// One wavefront per document: coalesced loads, the weighting of 64 entries at a time in parallel, and the norm summed by a SCALAR
// loop over the lanes (v_readlane -> v_add with a scalar operand), i.e. sequentially in ascending feature order like the reference.
// (count and out may be the SAME array -- no __restrict__ on them; a column id outside [0, cols) is reported through *err, as the
//  reference's idx_idf.at() would throw, instead of reading some other feature's idf.)
// Rows are SEGMENTS of row_ptr: row r is [row_ptr[r * seg_stride + seg_off], row_ptr[r * seg_stride + seg_off + 1]) -- stride 1 / offset 0
// for a plain CSR; an ensemble of nb base vectorizers (hstack, tfidf.hpp:1417-1423) weights base b's part of every document with
// stride nb, offset b and that base's parameters, then the whole rows once more with no weighting at all (normalize_csr, :1318-1354).
__global__ void __launch_bounds__(256)
tfidf_weight_kernel(const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col_idx, const float* count,
                    const float* __restrict__ idf, uint32_t rows, uint32_t cols, int binary, int sublinear_tf, int norm_p,
                    float* out, uint32_t seg_stride, uint32_t seg_off, uint32_t* __restrict__ err) {
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (r >= rows) return;
    const uint64_t b = row_ptr[(uint64_t)r * seg_stride + seg_off], e = row_ptr[(uint64_t)r * seg_stride + seg_off + 1];
    float denom = 0.0f;
    for (uint64_t c0 = b; c0 < e; c0 += 64u) {
        const uint64_t t = c0 + lane;
        const bool ok = t < e;
        float v = 0.0f;
        if (ok) {
            v = binary ? 1.0f : count[t];                                      // tfidf.hpp:800
            if (sublinear_tf) v = (float)((double)logf(v) + 1.0);              // :801  (std::log(float) + 1.0)
            if (idf) {                                                         // :802-804
                const uint32_t f = col_idx[t];
                if (f >= cols) { if (err) *err = 1u; }
                else v = __fmul_rn(v, idf[f]);
            }
            out[t] = v;
        }
        const float term = norm_p == 1 ? fabsf(v) : __fmul_rn(v, v);           // :806-809
        const uint32_t n = (uint32_t)min((uint64_t)64u, e - c0);
        for (uint32_t i = 0; i < n; ++i)
            denom = __fadd_rn(denom, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(term), (int)i)));
    }
    if (fabsf(denom) < FLT_EPSILON) denom = 1.0f;                              // :814-815
    else if (norm_p == 2) denom = sqrtf(denom);                                // :816-817
    for (uint64_t t = b + lane; t < e; t += 64u) out[t] = __fdiv_rn(out[t], denom);   // :819-821 (each lane re-reads what it wrote)
}

void launch_tfidf_weight(const uint64_t* row_ptr, const uint32_t* col_idx, const float* count, const float* idf, uint32_t rows, uint32_t cols,
                         int binary, int sublinear_tf, int norm_p, float* out, hipStream_t s, uint32_t seg_stride, uint32_t seg_off, uint32_t* err) {
    if (rows == 0) return;
    if (norm_p != 1 && norm_p != 2) fail("tfidf: invalid normalize option, norm_p: [ 1| 2]");
    hipLaunchKernelGGL(tfidf_weight_kernel, dim3((rows + 3u) / 4u), dim3(256), 0, s, row_ptr, col_idx, count, idf, rows, cols, binary,
                       sublinear_tf, norm_p, out, seg_stride, seg_off, err);
    XRL_HIP(hipGetLastError());
}

}  // namespace xrl
