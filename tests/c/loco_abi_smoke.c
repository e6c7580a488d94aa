/* Plain-C consumer of the C ABI (include/loco_hip.h): no torch, no C++.  Built and run by
 * tests/test_gpu_parity.py::test_c_abi_from_plain_c:
 *     gcc -std=c11 -I include -I /opt/rocm/include tests/c/loco_abi_smoke.c -L loco-edit_amd -lloco_hip \
 *         -L /opt/rocm/lib -lamdhip64 -o loco_abi_smoke
 *     ./loco_abi_smoke params.bin x.bin t eps_out.bin jv_out.bin
 * params.bin: repeated records { int32 name_len; char name[]; int32 ndim; int64 shape[ndim]; float data[] } (the tiny
 * Ho-DDPM configuration of the tests); x.bin: 1*3*32*32 floats followed by 2*3072 floats of probe rows.
 * Writes eps = unet(x, t) and U = J V (x0 operator, no mask) so the caller can compare them with the ctypes path. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "loco_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? loco_last_error(ctx) : "?"); return 2; } } while (0)
#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s params.bin x.bin t eps_out.bin jv_out.bin\n", argv[0]); return 1; }
    loco_ctx* ctx = NULL;
    if (loco_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    loco_unet_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.resolution = 32; cfg.in_channels = 3; cfg.out_ch = 3; cfg.ch = 32;
    cfg.num_levels = 3; cfg.ch_mult[0] = 1; cfg.ch_mult[1] = 2; cfg.ch_mult[2] = 2;
    cfg.num_res_blocks = 2; cfg.num_attn_res = 1; cfg.attn_resolutions[0] = 16;
    cfg.gn_groups = 32; cfg.gn_eps = 1e-6f; cfg.max_batch = 4;
    cfg.arch = 0; cfg.num_head_channels = -1; cfg.learn_sigma = 0;
    CHECK(loco_create(&cfg, &ctx));
    printf("%s\n", loco_version());

    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t nlen;
    int nparams = 0;
    while (fread(&nlen, 4, 1, f) == 1) {
        char name[256];
        int32_t ndim;
        int64_t shape[8], count = 1;
        if (nlen <= 0 || nlen >= 256 || fread(name, 1, (size_t)nlen, f) != (size_t)nlen) return 1;
        name[nlen] = 0;
        if (fread(&ndim, 4, 1, f) != 1 || ndim < 1 || ndim > 8 || fread(shape, 8, (size_t)ndim, f) != (size_t)ndim) return 1;
        for (int i = 0; i < ndim; ++i) count *= shape[i];
        float* data = (float*)malloc((size_t)count * sizeof(float));
        if (fread(data, sizeof(float), (size_t)count, f) != (size_t)count) return 1;
        CHECK(loco_load_param(ctx, name, data, shape, ndim, /*is_device=*/0));
        free(data);
        ++nparams;
    }
    fclose(f);
    if (loco_params_missing(ctx) != 0) { fprintf(stderr, "missing: %s\n", loco_last_error(ctx)); return 2; }

    const int n = 3 * 32 * 32, k = 2;
    float* hx = (float*)malloc(sizeof(float) * (size_t)(n + k * n));
    f = fopen(argv[2], "rb");
    if (!f || fread(hx, sizeof(float), (size_t)(n + k * n), f) != (size_t)(n + k * n)) { fprintf(stderr, "bad x.bin\n"); return 1; }
    fclose(f);
    const float t = (float)atof(argv[3]);
    float *dx, *deps, *dV, *dU;
    HIPOK(hipMalloc((void**)&dx, sizeof(float) * n));
    HIPOK(hipMalloc((void**)&deps, sizeof(float) * n));
    HIPOK(hipMalloc((void**)&dV, sizeof(float) * k * n));
    HIPOK(hipMalloc((void**)&dU, sizeof(float) * k * n));
    HIPOK(hipMemcpy(dx, hx, sizeof(float) * n, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(dV, hx + n, sizeof(float) * k * n, hipMemcpyHostToDevice));
    hipStream_t st;
    HIPOK(hipStreamCreate(&st));
    CHECK(loco_unet_forward(ctx, dx, t, 1, deps, st));
    CHECK(loco_pmp_primal(ctx, dx, t, /*alpha_bar=*/0.5f, /*mask=*/NULL, /*use_et=*/0, st));
    CHECK(loco_pmp_jvp(ctx, dV, k, dU, st));
    HIPOK(hipStreamSynchronize(st));
    float* out = (float*)malloc(sizeof(float) * (size_t)(k * n));
    HIPOK(hipMemcpy(out, deps, sizeof(float) * n, hipMemcpyDeviceToHost));
    f = fopen(argv[4], "wb"); fwrite(out, sizeof(float), (size_t)n, f); fclose(f);
    HIPOK(hipMemcpy(out, dU, sizeof(float) * k * n, hipMemcpyDeviceToHost));
    f = fopen(argv[5], "wb"); fwrite(out, sizeof(float), (size_t)(k * n), f); fclose(f);
    printf("parameters %d, workspace %lld bytes, %.3f GFLOP per evaluation\n", nparams, (long long)loco_workspace_bytes(ctx),
           loco_unet_flops(ctx) / 1e9);
    loco_destroy(ctx);
    return 0;
}
