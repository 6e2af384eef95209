// K1Q kernel instantiations, one share per translation unit (xrl_k1q_impl.h)
#include "xrl_k1q_impl.h"

namespace xrl {
void k1q_launch_n1p1(const K1QArgs& a, dim3 grid, hipStream_t s, const K1QVariant& v) { k1q_launch_variant<1, 1, true>(a, grid, s, v); }
}  // namespace xrl
