// Explicit instantiations of the split-bf16 conv launcher (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace loco {
template void launch_tile_b<9, CM_TAN_SILU>(const ConvArgs&, hipStream_t);
}  // namespace loco
