// Explicit instantiations of the K-concatenated conv launcher (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace loco {
template void launch_kcat_b<PR_F16, CM_GN_SILU>(const ConvArgs&, hipStream_t);
template void launch_kcat_b<PR_F16, CM_TAN_SILU>(const ConvArgs&, hipStream_t);
template void launch_kcat_b<PR_F16, CM_GN_GELU>(const ConvArgs&, hipStream_t);
}  // namespace loco
