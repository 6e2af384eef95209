// Which accesses does gfx950 treat as out of range for a RAW buffer resource (stride 0, offen)?  K1Q relies on the answer in three places
// (k1q_load_w's scalar row offset, the lane offset that switches a load off, the presence-mask loads).  The "buffer" is the middle third of a
// larger allocation whose dwords hold their own index, so that an access the hardware lets through never faults and shows where it landed.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/brp scripts/buffer_range_probe.hip && /tmp/brp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Case { uint32_t voff, soff, nrec; };
__global__ void probe(const uint32_t* base, const Case* cs, int n, uint32_t* out) {
    for (int i = 0; i < n; ++i) {
        const Case c = cs[i];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, (int)c.nrec, 0x00020000);
        const uint32_t so = __builtin_amdgcn_readfirstlane(c.soff);
        out[i * 64 + threadIdx.x] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(c.voff + threadIdx.x * 4u), (int)so, 0);
    }
}
// scattered lanes: every lane its own row of a [rows x pw] array (voffset = row * pw * 4), the word selected by the scalar offset
__global__ void scattered(const uint32_t* pres, uint32_t rows, uint32_t pw, uint32_t* out, int iters) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(pres), 0, (int)(rows * pw * 4u), 0x00020000);
    uint32_t s = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 7u, acc = 0u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t row = (uint32_t)(((uint64_t)s * rows) >> 32);
        const uint32_t word = __builtin_amdgcn_readfirstlane((uint32_t)i % pw);
        const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(row * pw * 4u), (int)(word * 4u), 0);
        acc += (v == row * pw + word) ? 0u : 1u;
    }
    out[blockIdx.x * 64u + threadIdx.x] = acc;
}

int main() {
    const uint32_t N = 4u << 20;                       // dwords per third (16 MiB)
    uint32_t* all = nullptr; CHECK(hipMalloc(&all, (size_t)3 * N * 4));
    std::vector<uint32_t> h((size_t)3 * N); for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
    CHECK(hipMemcpy(all, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const uint32_t B = N * 4;                          // bytes of the middle third = the "buffer"
    const std::vector<Case> cs = {
        {64, 1u << 20, B},            // small lane offset, large scalar offset, sum in range
        {1u << 20, 64, B},            // large lane offset, small scalar offset, sum in range
        {B - 256, 0, B},              // last 256 bytes through the lane offset
        {0, B - 256, B},              // ... through the scalar offset
        {B - 256, 512, B},            // lane offset in range on its own, sum beyond the end
        {512, B - 256, B},            // scalar offset in range on its own, sum beyond the end
        {0, B, B},                    // scalar offset == num_records
        {0xFFFFFFF0u - 63 * 4, 4096, B},   // the "switched off" lane offset
        {0xFFFFFFF0u - 63 * 4, 0, B},
        {64, 4096, 1024},             // num_records smaller than the scalar offset (one-row resource, row selected by soffset)
        {64, 4096, 0},                // num_records 0
    };
    Case* dc = nullptr; uint32_t* dout = nullptr;
    CHECK(hipMalloc(&dc, cs.size() * sizeof(Case))); CHECK(hipMemcpy(dc, cs.data(), cs.size() * sizeof(Case), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dout, cs.size() * 64 * 4));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, all + N, dc, (int)cs.size(), dout);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> o(cs.size() * 64); CHECK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cs.size(); ++i) {
        // expected when let through: dword index N + (voff + soff) / 4 + lane  (mod 2^32 arithmetic shown as the hardware did it)
        std::printf("case %2zu voff=0x%08x soff=0x%08x num_records=0x%08x : lane0=%u lane63=%u  (in-buffer dword would be %llu)\n", i, cs[i].voff, cs[i].soff, cs[i].nrec,
                    o[i * 64], o[i * 64 + 63], (unsigned long long)N + ((unsigned long long)cs[i].voff + cs[i].soff) / 4);
    }
    {   // K1Q's presence-mask access pattern: 135002 rows x 16 words, lane = row
        const uint32_t rows = 135002, pw = 16;
        uint32_t* pres = nullptr; CHECK(hipMalloc(&pres, (size_t)rows * pw * 4));
        std::vector<uint32_t> hp((size_t)rows * pw); for (size_t i = 0; i < hp.size(); ++i) hp[i] = (uint32_t)i;
        CHECK(hipMemcpy(pres, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        uint32_t* bad = nullptr; CHECK(hipMalloc(&bad, 4096 * 64 * 4));
        hipLaunchKernelGGL(scattered, dim3(4096), dim3(64), 0, 0, pres, rows, pw, bad, 256);
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> hb(4096 * 64); CHECK(hipMemcpy(hb.data(), bad, hb.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long nb = 0; for (uint32_t x : hb) nb += x;
        std::printf("scattered rows: %llu wrong values of %llu loads\n", nb, 4096ull * 64 * 256);
    }
    return 0;
}
