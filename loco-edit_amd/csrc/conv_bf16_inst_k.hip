// Explicit instantiations of the tap-pair 16x16x32 conv launcher (see conv_pair_kernel.h): diagnostics build only (-DLOCO_DIAG,
// `make diag`; LOCO_CONV_PAIR=1) -- correct, measured neutral against the 32x32x16 kernel (profiles/r06_experiments.md section 6).
#ifdef LOCO_DIAG
#include "conv_pair_kernel.h"

namespace loco {
template void launch_pair_b<PR_BF16X3, CM_NONE>(const ConvArgs&, hipStream_t);
template void launch_pair_b<PR_BF16X3, CM_GN_SILU>(const ConvArgs&, hipStream_t);
template void launch_pair_b<PR_BF16X3, CM_TAN_SILU>(const ConvArgs&, hipStream_t);
template void launch_pair_b<PR_BF16X3, CM_COT_SILU>(const ConvArgs&, hipStream_t);
}  // namespace loco
#endif
