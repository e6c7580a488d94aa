// Explicit instantiations of the dual-probe conv launcher (see conv_dual_kernel.h): diagnostics build only (-DLOCO_DIAG,
// `make diag`) -- the tile measured neutral (profiles/r05_experiments.md) and is not part of the product library.
#ifdef LOCO_DIAG
#include "conv_dual_kernel.h"

namespace loco {
template void launch_dual_b<PR_BF16X3, CM_TAN_SILU>(const ConvArgs&, hipStream_t);
template void launch_dual_b<PR_BF16X3, CM_COT_SILU>(const ConvArgs&, hipStream_t);
}  // namespace loco
#endif
