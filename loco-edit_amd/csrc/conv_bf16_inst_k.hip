// Explicit instantiations of the tap-pair 16x16x32 conv launcher (see conv_pair_kernel.h).  Compiled with the max-memory-clause
// scheduler strategy (Makefile: under max-ilp the tangent / cotangent forms spill).
#include "conv_pair_kernel.h"

namespace loco {
template void launch_pair_b<PR_BF16X3, CM_NONE>(const ConvArgs&, hipStream_t);
template void launch_pair_b<PR_BF16X3, CM_GN_SILU>(const ConvArgs&, hipStream_t);
template void launch_pair_b<PR_BF16X3, CM_TAN_SILU>(const ConvArgs&, hipStream_t);
template void launch_pair_b<PR_BF16X3, CM_COT_SILU>(const ConvArgs&, hipStream_t);
}  // namespace loco
