// Does the LDS float atomic add (ds_add_f32) round exactly like v_add_f32, denormals included?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const float* a, const float* b, float* o_valu, float* o_lds, int n) {
    __shared__ float acc[256];
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
        int i = base + threadIdx.x;
        float x = i < n ? a[i] : 0.f, y = i < n ? b[i] : 0.f;
        acc[threadIdx.x] = x;
        __syncthreads();
        atomicAdd(&acc[threadIdx.x], y);
        __syncthreads();
        if (i < n) { o_lds[i] = acc[threadIdx.x]; o_valu[i] = __fadd_rn(x, y); }
        __syncthreads();
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<float> a(n), b(n), ov(n), ol(n);
    std::mt19937_64 rng(1);
    for (int i = 0; i < n; ++i) {
        uint32_t u = (uint32_t)rng(), w = (uint32_t)rng();
        int mode = i & 7;
        if (mode == 0) { u &= 0x807FFFFFu; }                      // denormal a
        if (mode == 1) { w &= 0x807FFFFFu; }                      // denormal b
        if (mode == 2) { u &= 0x807FFFFFu; w &= 0x807FFFFFu; }    // both denormal
        if (mode == 3) { u = (u & 0x807FFFFFu) | 0x00800000u; w = (w & 0x807FFFFFu) | 0x00800000u; w ^= 0x80000000u & ~u; }  // near-denormal cancellation
        if (mode >= 4) { u = (u & 0x80FFFFFFu) | 0x3F000000u; w = (w & 0x80FFFFFFu) | (0x3F000000u - ((w >> 28) << 23)); } // ordinary magnitudes
        memcpy(&a[i], &u, 4); memcpy(&b[i], &w, 4);
    }
    float *da, *db, *dv, *dl;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, da, db, dv, dl, n);
    hipMemcpy(ov.data(), dv, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ol.data(), dl, n * 4, hipMemcpyDeviceToHost);
    long mism = 0, mism_host = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t x, y; memcpy(&x, &ov[i], 4); memcpy(&y, &ol[i], 4);
        volatile float h = a[i] + b[i]; uint32_t z; float hh = h; memcpy(&z, &hh, 4);
        bool nanx = (x & 0x7FFFFFFFu) > 0x7F800000u, nany = (y & 0x7FFFFFFFu) > 0x7F800000u;
        if (x != z && !(nanx)) ++mism_host;
        if (x != y && !(nanx && nany)) { ++mism; if (shown++ < 8) printf("i=%d a=%a b=%a valu=%a lds=%a\n", i, a[i], b[i], ov[i], ol[i]); }
    }
    printf("pairs %d  ds_add_f32 != v_add_f32: %ld   v_add_f32 != host add: %ld\n", n, mism, mism_host);
    return 0;
}
