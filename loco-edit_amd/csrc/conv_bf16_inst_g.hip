// Explicit instantiations of the low-precision conv launcher (see conv_bf16_kernel.h): the GELU forward prologue.
#include "conv_bf16_kernel.h"

namespace loco {
template void launch_tile_b<PR_BF16X3, 9, CM_GN_GELU>(const ConvArgs&, hipStream_t);
}  // namespace loco
