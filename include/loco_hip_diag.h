/* Diagnostics of libloco_hip -- NOT part of the drop-in boundary (include/loco_hip.h).
 *
 * These two entry points exist only in a library built with -DLOCO_DIAG (`make -C loco-edit_amd/csrc diag` ->
 * loco-edit_amd/libloco_hip_diag.so); the by-hand tuning scripts under tests/ load that build through LOCO_HIP_LIB.  The
 * shipped libloco_hip.so does not export them. */
#ifndef LOCO_HIP_DIAG_H
#define LOCO_HIP_DIAG_H
#include "loco_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Tuning hook: average ms of one convolution shape (random scratch data) over `iters` launches.
 * mode: 0 raw, 1 GN+SiLU, 2 GN, 3 tangent, 4 cotangent; tile: -1 auto or a variant id. */
int  loco_bench_conv(loco_ctx* ctx, int32_t cin, int32_t cout, int32_t H, int32_t W, int32_t B, int32_t mode,
                     int32_t taps, int32_t tile, int32_t iters, float* ms_avg, void* stream);

/* Debug / test hook: copy an internal primal activation ("down.0.block.0" ...)
 * of the last forward/primal call into dst (device), returns element count or <0. */
int64_t loco_debug_tensor(loco_ctx* ctx, const char* name, float* dst, int64_t cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOCO_HIP_DIAG_H */
