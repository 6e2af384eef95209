// Microbenchmark for the next K1Q step (profiles/r03_pruning_topk.md section 5): what does the per-load
// 64-bit vector address add cost, and does the scalar-base load form (global_load_dword v, v_off32, s[base:base+1]) written in
// inline assembly pay?  One wavefront per "query": F pseudo-random feature rows of a [rows x ld] f32 matrix, lane = column,
// U loads in flight, acc += x * w with a separately rounded multiply and add (as in K1Q).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/k1q_load_forms scripts/k1q_load_forms.hip && /tmp/k1q_load_forms
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int U = 8;

__device__ __forceinline__ uint32_t next_feature(uint32_t& s, uint32_t rows) {   // cheap LCG: uniform over the rows
    s = s * 1664525u + 1013904223u;
    return (uint32_t)(((uint64_t)s * rows) >> 32);
}

// A: what hipcc emits for K1Q today -- the lane's 64-bit column address is loop-invariant, per load one v_lshl_add_u64 adds the row offset
__global__ void __launch_bounds__(256) form_vector_add(const uint32_t* __restrict__ wd, uint32_t ld4, uint32_t rows, int F, float* __restrict__ out, uint32_t nq) {
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (q >= nq) return;
    uint32_t s = __builtin_amdgcn_readfirstlane(q * 2654435761u + 12345u);
    const uint32_t woff = lane * 4u;
    float acc = 0.0f;
    for (int t = 0; t < F; t += U) {
        uint32_t w[U]; float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t f = next_feature(s, rows);
            x[u] = (float)(f & 255u) * 0.001f;
            w[u] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(wd) + (uint64_t)f * ld4 + woff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __fadd_rn(acc, __fmul_rn(x[u], __uint_as_float(w[u])));
    }
    out[(size_t)q * 64u + lane] = acc;
}

// B: scalar row base + 32-bit lane offset, the load written by hand; ONE s_waitcnt per batch that carries the loaded registers
// (so that the compiler cannot move a use above it)
__global__ void __launch_bounds__(256) form_scalar_base(const uint32_t* __restrict__ wd, uint32_t ld4, uint32_t rows, int F, float* __restrict__ out, uint32_t nq) {
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (q >= nq) return;
    uint32_t s = __builtin_amdgcn_readfirstlane(q * 2654435761u + 12345u);
    const uint32_t woff = lane * 4u;
    float acc = 0.0f;
    for (int t = 0; t < F; t += U) {
        uint32_t w[U]; float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t f = next_feature(s, rows);
            x[u] = (float)(f & 255u) * 0.001f;
            const uint64_t row = reinterpret_cast<uint64_t>(wd) + (uint64_t)f * ld4;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(w[u]) : "v"(woff), "s"(row) : "memory");
        }
        static_assert(U == 8, "the wait below lists eight registers");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __fadd_rn(acc, __fmul_rn(x[u], __uint_as_float(w[u])));
    }
    out[(size_t)q * 64u + lane] = acc;
}

int main() {
    const uint32_t rows = 135000, ld = 8192, nq = 490000;      // Amazon-670K level 3: 4.4 GB
    const int F = 80;
    uint32_t* wd = nullptr; float *oa = nullptr, *ob = nullptr;
    CHECK(hipMalloc(&wd, (size_t)rows * ld * 4));
    CHECK(hipMemset(wd, 0x3c, (size_t)rows * ld * 4));           // 0x3c3c3c3c = 0.0115 as a float
    CHECK(hipMalloc(&oa, (size_t)nq * 64 * 4)); CHECK(hipMalloc(&ob, (size_t)nq * 64 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid((nq + 3) / 4), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        float ma = 0, mb = 0;
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(form_vector_add, grid, block, 0, 0, wd, ld * 4u, rows, F, oa, nq); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ma, e0, e1));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(form_scalar_base, grid, block, 0, 0, wd, ld * 4u, rows, F, ob, nq); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&mb, e0, e1));
        std::printf("rep %d: vector 64-bit add %.3f ms   scalar base (inline asm) %.3f ms\n", rep, ma, mb);
    }
    std::vector<float> ha(4096), hb(4096);
    CHECK(hipMemcpy(ha.data(), oa, 4096 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), ob, 4096 * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 4096; ++i) bad += ha[i] != hb[i];
    std::printf("outputs %s\n", bad ? "DIFFER" : "identical");
    return bad != 0;
}
