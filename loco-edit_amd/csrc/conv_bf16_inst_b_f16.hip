// Explicit instantiations of the low-precision conv launcher (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace loco {
template void launch_tile_b<PR_F16, 9, CM_GN_SILU>(const ConvArgs&, hipStream_t);
template void launch_tile_b<PR_F16, 9, CM_GN>(const ConvArgs&, hipStream_t);
}  // namespace loco
