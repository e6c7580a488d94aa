// Explicit instantiations of the low-precision conv launcher (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace loco {
template void launch_tile_b<PR_BF16X3, 9, CM_NONE>(const ConvArgs&, hipStream_t);
}  // namespace loco
