// Standalone timing / comparison harness for the attention tangent / cotangent kernels (csrc/attn_flash.hip), no engine context:
//   attn_bench T NH CH B Lt iters [dump-prefix]
// fills q, k, v, o, the probes' dq / dk / dv / g_o with hashed random values and P with normalised positive rows, runs the
// tangent launch and the cotangent pair `iters` times between HIP events and prints the times and checksums of every output.
// With a dump prefix the outputs are written as raw float32 files (<prefix>_{out,gq,gk,gv}.bin) for bit comparisons between
// builds.  Build: tests/diag/r05.sh attn_build (hipcc, includes the kernel source directly; -DAF_WI=<bits> what-if switches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>
#include "../../loco-edit_amd/csrc/attn_flash.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed, float scale, float bias) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = bias + scale * ((float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f);
}
// rows of P: positive, summing to one (a softmax of something)
__global__ void norm_rows(float* P, int PS) {
    float* r = P + (long)blockIdx.x * PS;
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < PS; i += 256) { float v = expf(3.0f * r[i]); r[i] = v; s += v; }
    sm[threadIdx.x] = s; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k]; __syncthreads(); }
    const float inv = 1.0f / sm[0];
    for (int i = threadIdx.x; i < PS; i += 256) r[i] *= inv;
}
static float* dalloc(long n, unsigned seed, float scale, float bias = 0.f) {
    float* p; CK(hipMalloc(&p, (size_t)n * 4));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, scale, bias);
    return p;
}
static std::vector<float> g_keep[4];
static void report(const char* name, const float* d, long n, const char* prefix, int slot = -1, bool compare = false) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (slot >= 0 && compare) {
        double md = 0, mr = 0; long nd = 0;
        for (long i = 0; i < n; ++i) { double df = fabs((double)h[i] - g_keep[slot][i]); if (df > 0) ++nd; md = df > md ? df : md; mr = fabs(g_keep[slot][i]) > mr ? fabs(g_keep[slot][i]) : mr; }
        printf("  %-4s vs the converting kernels: %ld of %ld differ, max |diff| %.3e (max |ref| %.3e)\n", name, nd, n, md, mr);
    }
    if (slot >= 0 && !compare) g_keep[slot] = h;
    double s = 0, s2 = 0; long bad = 0;
    for (long i = 0; i < n; ++i) { if (!std::isfinite(h[i])) ++bad; s += h[i]; s2 += (double)h[i] * h[i]; }
    printf("  %-4s sum %.9e  sumsq %.9e  nonfinite %ld\n", name, s, s2, bad);
    if (prefix) {
        std::string fn = std::string(prefix) + "_" + name + ".bin";
        FILE* f = fopen(fn.c_str(), "wb");
        if (f) { fwrite(h.data(), 4, n, f); fclose(f); }
    }
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: attn_bench T NH CH B Lt iters [dump-prefix]\n"); return 2; }
    const int T = atoi(argv[1]), NH = atoi(argv[2]), CH = atoi(argv[3]), B = atoi(argv[4]), Lt = atoi(argv[5]), iters = atoi(argv[6]);
    const char* prefix = argc > 7 ? argv[7] : nullptr;
    const long HS = (long)CH * T, per = (long)NH * HS, PS = Lt + T;
    if (Lt ? !loco::attn_flash_text_supported(T, CH, Lt) : !loco::attn_flash_supported(T, CH)) { fprintf(stderr, "shape not supported\n"); return 2; }
    loco::AttnFlashArgs a{};
    a.T = T; a.NH = NH; a.B = B; a.CH = CH; a.scale = 1.0f / sqrtf((float)CH);
    a.q = dalloc(per, 1, 1.f); a.k = dalloc(per, 2, 1.f); a.v = dalloc(per, 3, 1.f); a.hs = HS;
    float* P = dalloc((long)NH * T * PS, 4, 1.f);
    hipLaunchKernelGGL(norm_rows, dim3((unsigned)(NH * T)), dim3(256), 0, 0, P, (int)PS);
    a.P = P;
    a.o = dalloc(per, 5, 1.f);
    a.dq = dalloc(per * B, 6, 1.f); a.dk = dalloc(per * B, 7, 1.f); a.dv = dalloc(per * B, 8, 1.f); a.bs_d = per;
    a.out = dalloc(per * B, 9, 0.f); a.bs_out = per;
    a.go = dalloc(per * B, 10, 1.f); a.bs_go = per;
    a.gq = dalloc(per * B, 11, 0.f); a.gk = dalloc(per * B, 12, 0.f); a.gv = dalloc(per * B, 13, 0.f); a.bs_g = per;
    a.delta = dalloc((long)B * NH * T, 14, 0.f);
    a.Lt = Lt;
    if (Lt) { a.kt = dalloc((long)NH * CH * Lt, 15, 1.f); a.vt = dalloc((long)NH * CH * Lt, 16, 1.f); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    loco::launch_attn_flash_tangent(a, 0);
    loco::launch_attn_flash_cotangent(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) loco::launch_attn_flash_tangent(a, 0);
    CK(hipEventRecord(e1, 0));
    for (int i = 0; i < iters; ++i) loco::launch_attn_flash_cotangent(a, 0);
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    float t_tan = 0, t_cot = 0;
    CK(hipEventElapsedTime(&t_tan, e0, e1)); CK(hipEventElapsedTime(&t_cot, e1, e2));
    // issued MFMA flops: 2 T (Lt + T) CHpad per product, 3 MFMAs each; TAN 4 products, COT 2 + 3
    printf("T %d NH %d CH %d B %d Lt %d: tangent %.1f us, cotangent pair %.1f us\n", T, NH, CH, B, Lt, 1e3 * t_tan / iters, 1e3 * t_cot / iters);
    report("out", a.out, per * B, prefix, 0);
    report("gq", a.gq, per * B, prefix, 1);
    report("gk", a.gk, per * B, prefix, 2);
    report("gv", a.gv, per * B, prefix, 3);
    // ---- the DMA-fed kernels (operands pre-split into a workspace)
    a.ws_bytes = loco::attn_flash_ws_bytes(a);
    CK(hipMalloc(&a.ws, a.ws_bytes));
    CK(hipMemset(a.out, 0, (size_t)per * B * 4)); CK(hipMemset(a.gq, 0, (size_t)per * B * 4));
    CK(hipMemset(a.gk, 0, (size_t)per * B * 4)); CK(hipMemset(a.gv, 0, (size_t)per * B * 4));
    loco::launch_attn_flash_tangent(a, 0);
    loco::launch_attn_flash_cotangent(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) loco::launch_attn_flash_tangent(a, 0);
    CK(hipEventRecord(e1, 0));
    for (int i = 0; i < iters; ++i) loco::launch_attn_flash_cotangent(a, 0);
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    CK(hipEventElapsedTime(&t_tan, e0, e1)); CK(hipEventElapsedTime(&t_cot, e1, e2));
    printf("DMA-fed (workspace %.1f MB, split pass included): tangent %.1f us, cotangent pair %.1f us\n", a.ws_bytes / 1048576.0, 1e3 * t_tan / iters, 1e3 * t_cot / iters);
    report("out", a.out, per * B, nullptr, 0, true);
    report("gq", a.gq, per * B, nullptr, 1, true);
    report("gk", a.gk, per * B, nullptr, 2, true);
    report("gv", a.gv, per * B, nullptr, 3, true);
    return 0;
}
