// Device-side helpers shared by the HIP kernels (xrl_kernels.hip, xrl_k1q.hip): the reference's
// post-processors, and a wavefront-wide top-k with the reference's ordering.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "xrl_model.h"

namespace xrl {

__device__ __forceinline__ void wave_sync_lds() {
    // LDS operations of one wavefront execute in program order; this only stops the compiler
    // from moving LDS accesses across the point (cross-lane RAW through LDS inside a wave).
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// Guard of the exact bound pruning (xrl_predict.cpp): "a child's score is at most its parent's bound" fails only when a child's
// score is NaN, and a NaN needs a non-finite operand or an intermediate that overflows (inf - inf, 0 * inf).  A query whose
// largest |x| is finite and small enough that no partial sum of (features + bias) products can reach the fp32 range keeps
// every accumulator finite on every layer; all others are never pruned (they take the same path as prune = 0).
//   xmax_bits: max over the query's values of (bits & 0x7FFFFFFF) -- a NaN / inf compares above every finite value
//   wmax:      max over the layers of |weight| * max(1, |bias|), +inf when a weight is not finite
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool prune_guard_ok(uint32_t xmax_bits, uint32_t n_feat, float wmax) {
    if (xmax_bits >= 0x7F800000u) return false;
    const float b = fmaxf(__uint_as_float(xmax_bits), 1.0f) * wmax * (float)(n_feat + 2u);
    return b < 1.0e37f;                                             // (NaN / inf compare false)
}

// ---------------------------------------------------------------------------------------------
// post-processor (inference.hpp:192-240).  The reference lambdas take `const float&`:
//   sigmoid / log-sigmoid evaluate std::exp(float) (= expf) and continue in double;
//   l{p}-hinge keeps z in a FLOAT, then pow/exp in double.  Results are cast to float (:1369).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ref_expf(float x) { return (float)exp((double)x); }

// z^p for the integer p of l{p}-hinge.  z is a float, so z*z is EXACT in double (48-bit product) and
// z^3 = (z*z)*z, z^4 = (z*z)*(z*z) carry a single rounding: they are the correctly rounded powers,
// which is what glibc's pow returns (its error bound is < 1 ULP, correctly rounded in practice).
// Larger p fall back to pow().
__device__ __forceinline__ double hinge_pow4(float zf, int p) {   // p in [0, 4] only
    const double z = (double)zf;
    const double z2 = z * z;
    return p == 0 ? 1.0 : p == 1 ? z : p == 2 ? z2 : p == 3 ? z2 * z : z2 * z2;
}

__device__ __forceinline__ double hinge_pow(float zf, int p) {
    const double z = (double)zf;
    switch (p) {
    case 0: return 1.0;
    case 1: return z;
    case 2: return z * z;
    case 3: return (z * z) * z;
    case 4: { const double t = z * z; return t * t; }
    default: return pow(z, (double)p);
    }
}

// PPC: compile-time post-processor CLASS.  0 = light (noop, l{p}-hinge and log-l{p}-hinge with
// p <= 4: at most one fp64 exp, ~12 VGPRs), 1 = generic (adds sigmoid / log-sigmoid / pow(), ~42
// VGPRs).  Keeping the heavy libm paths out of the default kernels keeps them at 8 waves per SIMD.
template <int PPC>
__device__ __forceinline__ float pp_transform(int kind, int p, float v) {
    if (PPC == 0) {
        if (kind == PP_LP_HINGE) { const float z = (float)fmax(0.0, 1.0 - (double)v); return (float)exp(-hinge_pow4(z, p)); }
        if (kind == PP_LOG_LP_HINGE) { const float z = (float)fmax(0.0, 1.0 - (double)v); return (float)(-hinge_pow4(z, p)); }
        return v;
    }
    switch (kind) {
    case PP_SIGMOID: return (float)(1.0 / (1.0 + (double)ref_expf(-v)));
    case PP_LOG_SIGMOID: return (float)(-log(1.0 + (double)ref_expf(-v)));
    case PP_LP_HINGE: {
        const float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)exp(-hinge_pow(z, p));
    }
    case PP_LOG_LP_HINGE: {
        const float z = (float)fmax(0.0, 1.0 - (double)v);
        return (float)(-hinge_pow(z, p));
    }
    default: return v;
    }
}

__device__ __forceinline__ float pp_combine(int kind, float x, float parent) {
    switch (kind) {
    case PP_SIGMOID:
    case PP_LP_HINGE: return __fmul_rn(x, parent);        // std::multiplies<float>
    case PP_LOG_SIGMOID:
    case PP_LOG_LP_HINGE: return __fadd_rn(x, parent);    // std::plus<float>
    default: return x;
    }
}

inline int pp_class(const PostProc& pp) {
    if (pp.kind == PP_NOOP) return 0;
    if ((pp.kind == PP_LP_HINGE || pp.kind == PP_LOG_LP_HINGE) && pp.p >= 0 && pp.p <= 4) return 0;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Wavefront top-k with the comparator of sorted_csr (inference.hpp:1265-1273): value descending,
// ties -> smaller candidate POSITION first.
//
// Every lane holds NS candidates in registers; candidate (r, lane) has position r*64 + lane.  Scores are
// mapped to unsigned keys that order like the floats do (-0.0 and +0.0 share a key: the reference compares
// with operator>, for which they tie); key 0 marks "no candidate".  The k-th largest key T is found by
// bisection on the key VALUE with wavefront ballots (a count per step, scalar control flow, early exit as
// soon as a threshold separates exactly k candidates; for NS >= 4 the search interval is first narrowed to
// [k-th largest per-lane maximum, maximum]); candidates above T plus the first (k - #above) candidates equal
// to T in position order are compacted into LDS in position order and ranked by counting.
// Replaces ~28 serial shuffle-insertions per 64 candidates (round 1's K2) by ~10-20 ballot steps.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t score_key(float v) {
    uint32_t b = __float_as_uint(v);
    if (b == 0x80000000u) b = 0u;                                  // -0.0 ties with +0.0
    const uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return k ? k : 1u;                                             // 0 is reserved for "no candidate"
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {   // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <int N>
__device__ __forceinline__ uint32_t wave_count_ge(const uint32_t (&key)[N], uint32_t t) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < N; ++r) c += (uint32_t)__popcll(__ballot(key[r] >= t));
    return c;
}

// kk-th largest key, searched in [lo, hi] given count(key >= lo) >= kk and count(key > hi) < kk.
// exact: the returned threshold t satisfies count(key >= t) == kk (early exit); otherwise t is the kk-th largest
// key itself and count(key >= t) may exceed kk (ties at t).
template <int N>
__device__ __forceinline__ uint32_t wave_bisect_kth(const uint32_t (&key)[N], uint32_t kk, uint32_t lo, uint32_t hi, bool& exact) {
    exact = false;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1) + 1u;           // in (lo, hi]
        const uint32_t c = wave_count_ge<N>(key, mid);
        if (c >= kk) { lo = mid; if (c == kk) { exact = true; break; } }
        else hi = mid - 1u;
    }
    return lo;
}

// Maximum of x over the wavefront, as a scalar: four DPP steps leave every lane with its 16-lane row's maximum (max is idempotent,
// so mirrored / overlapping exchanges are fine), the four rows meet in SGPRs.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true));   // row_half_mirror
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true));   // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return max(max(a, b), max(c, d));
}

// Small k: take the best remaining candidate k times.  The bisection below costs ~15 scalar instructions per step and needs all 32
// steps when the k-th score is tied (saturated post-processor outputs are the rule on deep trees) -- about 1000 scalar instructions
// per selection, which made the query-stationary kernels scalar-issue bound (one scalar unit per CU; profiles/r03_pruning_topk.md).  One
// extraction is a wavefront maximum (DPP), a ballot per candidate register to find the FIRST holder of that maximum -- lowest
// register, then lowest lane = lowest position, the reference's tie-break -- two readlanes and three selects: ~19 vector and ~12
// scalar instructions, and the winners come out already ranked (lane i receives the i-th best).
template <int NS>
__device__ __forceinline__ uint32_t wave_topk_extract(uint32_t (&kx)[NS], const uint32_t (&sbits)[NS], const uint32_t (&payload)[NS],
                                                      uint32_t k, int lane, uint32_t& o_rank, uint32_t& o_sbits, uint32_t& o_payload) {
    uint32_t osb = 0, opl = 0, i = 0;
    for (; i < k; ++i) {
        uint32_t m = kx[0];
#pragma unroll
        for (int r = 1; r < NS; ++r) m = max(m, kx[r]);
        const uint32_t mx = wave_max_u32(m);
        if (mx == 0u) break;                                         // no candidate left
        bool found = false;
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            if (!found) {
                const unsigned long long b = __ballot(kx[r] == mx);
                if (b != 0ull) {
                    const int wl = __builtin_ctzll(b);
                    const int sb = __builtin_amdgcn_readlane((int)sbits[r], wl), pl = __builtin_amdgcn_readlane((int)payload[r], wl);
                    const bool dst = (uint32_t)lane == i;
                    osb = dst ? (uint32_t)sb : osb;
                    opl = dst ? (uint32_t)pl : opl;
                    kx[r] = lane == wl ? 0u : kx[r];
                    found = true;
                }
            }
        }
    }
    o_rank = (uint32_t)lane; o_sbits = osb; o_payload = opl;
    return i;
}
constexpr uint32_t kTopkExtractMaxK = 20;   // above: the bisection's fixed cost is the smaller one

// Returns kk = min(k, #candidates).  Lanes [0, kk) receive one selected candidate each: its final rank in
// (value desc, position asc) order, its score bits and its payload.  sc: 64 uint2 of wavefront-private LDS.  k <= 64.
// The keys are CONSUMED (the extraction clears the winners in place: no second copy in registers).
template <int NS>
__device__ __forceinline__ uint32_t wave_topk(uint32_t (&key)[NS], const uint32_t (&sbits)[NS], const uint32_t (&payload)[NS],
                                              uint32_t k, uint2* sc, int lane, uint32_t& o_rank, uint32_t& o_sbits, uint32_t& o_payload) {
    if (k <= kTopkExtractMaxK) return wave_topk_extract<NS>(key, sbits, payload, k, lane, o_rank, o_sbits, o_payload);
    o_rank = 0; o_sbits = 0; o_payload = 0;
    const uint32_t n_valid = wave_count_ge<NS>(key, 1u);
    const uint32_t kk = min(k, n_valid);
    if (kk == 0) return 0;
    uint32_t lo = 1u, hi = 0xFFFFFFFFu;
    bool exact = false;
    if (NS >= 4) {
        uint32_t m[1] = {key[0]};
#pragma unroll
        for (int r = 1; r < NS; ++r) m[0] = max(m[0], key[r]);
        uint32_t mx = m[0];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
        hi = __builtin_amdgcn_readfirstlane(mx);
        bool e1;
        lo = wave_bisect_kth<1>(m, kk, 0u, hi, e1);                // >= kk lanes hold a candidate >= lo
        if (lo == 0u) lo = 1u;
    }
    const uint32_t T = wave_bisect_kth<NS>(key, kk, lo, hi, exact);
    uint32_t need_eq = 0;
    if (!exact) need_eq = kk - (T == 0xFFFFFFFFu ? 0u : wave_count_ge<NS>(key, T + 1u));
    uint32_t base = 0, eq_seen = 0;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        const bool above = exact ? key[r] >= T : key[r] > T;
        const bool eq = !exact && key[r] == T;
        const unsigned long long meq = __ballot(eq);
        const bool sel = above || (eq && eq_seen + lanes_below(meq) < need_eq);
        const unsigned long long msel = __ballot(sel);
        if (sel) sc[base + lanes_below(msel)] = make_uint2(sbits[r], payload[r]);
        base += (uint32_t)__popcll(msel); eq_seen += (uint32_t)__popcll(meq);
    }
    wave_sync_lds();
    // compacted in position order: rank = #(larger keys) + #(equal keys earlier in the list)
    const uint2 mine = sc[(uint32_t)lane < kk ? lane : 0];
    const uint32_t mk = score_key(__uint_as_float(mine.x));
    uint32_t rank = 0;
    for (uint32_t j = 0; j < kk; ++j) {
        const uint32_t kj = score_key(__uint_as_float(sc[j].x));
        rank += (kj > mk || (kj == mk && j < (uint32_t)lane)) ? 1u : 0u;
    }
    wave_sync_lds();
    o_rank = rank; o_sbits = mine.x; o_payload = mine.y;
    return kk;
}

}  // namespace xrl
