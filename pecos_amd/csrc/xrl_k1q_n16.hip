// K1Q kernel instantiations, one share per translation unit (xrl_k1q_impl.h)
#include "xrl_k1q_impl.h"

namespace xrl {
void k1q_launch_n16(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v, int ppc) { if (ppc) k1q_launch_variant<16, 1, false>(a, grid, s, v); else k1q_launch_variant<16, 0, false>(a, grid, s, v); }
}  // namespace xrl
