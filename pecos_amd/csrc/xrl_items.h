// The (query, beam slot, tile) work item K0 lays out and K1 / K1G consume (device code only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace xrl {

struct alignas(16) ItemDesc {   // 32 bytes; tile == kNoTile: inactive slot
    uint32_t q, tile, out_off; float pscore;
    uint64_t x_begin; uint32_t x_len, pad;   // CSR queries: the query row's range in col_idx / val (saves K1 a dependent lookup)
};
__host__ __device__ inline ItemDesc make_item(uint32_t q, uint32_t tile, uint32_t out_off, float ps, uint64_t xb, uint32_t xl) {
    ItemDesc d; d.q = q; d.tile = tile; d.out_off = out_off; d.pscore = ps; d.x_begin = xb; d.x_len = xl; d.pad = 0u; return d;
}
constexpr uint32_t kNoTile = 0xFFFFFFFFu;

}  // namespace xrl
