// FETCH_SIZE / L2-request calibration for the access patterns of the XR-Linear kernels (MI355X_MICROARCH.md: FETCH_SIZE is
// only calibrated for wide coalesced streams -- "calibrate on a known byte count in your own access pattern").
//   stream16   every lane reads 16 B, coalesced, over 2 GiB                      (known: 2 GiB fetched, no reuse)
//   gather8    every lane reads 8 B at a hashed 8-byte slot of a 4 GiB buffer    (K1's entry / bitmap gathers; 2^27 gathers = 1 GiB useful)
//   gather8s   the same inside a 64 MiB window (Infinity-Cache resident)         (do on-die hits reach FETCH_SIZE?)
//   seg64      16 lanes read 64 contiguous bytes at a hashed 64-byte slot, 4 GiB (K1Q's one segment per (feature, chunk); 2^25 segments = 2 GiB useful)
//   seg128     32 lanes read 128 contiguous bytes at a hashed 128-byte slot      (a full line per request)
// Run under rocprofv3 --kernel-trace --pmc <counter> (one counter set per run) and compare with the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void __launch_bounds__(256) stream16(const uint4* __restrict__ p, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) gather8(const uint2* __restrict__ p, uint64_t slots, uint64_t per_thread, uint32_t* out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t k = 0; k < per_thread; k += 4) {
        uint2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[mix(t * per_thread + k + u) % slots];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int LANES>   // LANES lanes read LANES*4 contiguous bytes at a hashed aligned slot
__global__ void __launch_bounds__(256) segload(const uint32_t* __restrict__ p, uint64_t slots, uint64_t per_group, uint32_t* out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t g = t / LANES; const uint32_t l = t % LANES;
    uint32_t acc = 0;
    for (uint64_t k = 0; k < per_group; k += 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[(mix(g * per_group + k + u) % slots) * LANES + l];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t GiB = 1ull << 30;
    void* buf; uint32_t* out;
    CK(hipMalloc(&buf, 4 * GiB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, 4 * GiB));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timed = [&](const char* name, double useful, auto&& launch) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("%-9s useful_bytes=%.0f ms=%.3f useful_GBps=%.1f\n", name, useful, ms, useful / ms / 1e6);
        }
    };
    timed("stream16", 2.0 * GiB, [&] { hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, 2 * GiB / 16, out); });
    const uint64_t threads = 1ull << 22;   // 16384 blocks x 256
    timed("gather8", (double)(threads * 32 * 8), [&] { hipLaunchKernelGGL(gather8, dim3(16384), dim3(256), 0, 0, (const uint2*)buf, 4 * GiB / 8, 32, out); });
    timed("gather8s", (double)(threads * 32 * 8), [&] { hipLaunchKernelGGL(gather8, dim3(16384), dim3(256), 0, 0, (const uint2*)buf, (64ull << 20) / 8, 32, out); });
    timed("seg64", (double)(threads / 16 * 128 * 64), [&] { hipLaunchKernelGGL(segload<16>, dim3(16384), dim3(256), 0, 0, (const uint32_t*)buf, 4 * GiB / 64, 128, out); });
    timed("seg128", (double)(threads / 32 * 128 * 128), [&] { hipLaunchKernelGGL(segload<32>, dim3(16384), dim3(256), 0, 0, (const uint32_t*)buf, 4 * GiB / 128, 128, out); });
    return 0;
}
