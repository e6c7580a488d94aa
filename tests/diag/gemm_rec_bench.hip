// Standalone comparison of the two split-bf16 GEMMs on the latent decoder's attention shapes (4096 tokens x 512 channels, 5 probes):
// gemm_bf16x3_kernel (gemm.hip: every workgroup converts its operand panels) vs gemm_rec_bf16x3 (gemm_rec.hip: operands split once
// into records, streamed by LDS-DMA).  Outputs compared bit for bit, times between HIP events (split passes / reduce included).
//   gemm_rec_bench [T] [C] [B] [iters]
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp -I../../loco-edit_amd/csrc gemm_rec_bench.hip -o bin/gemm_rec_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../loco-edit_amd/csrc/gemm.hip"
#include "../../loco-edit_amd/csrc/gemm_rec.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = scale * ((float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f);
}
static float* dalloc(long n, unsigned seed, float scale) {
    float* p; CK(hipMalloc(&p, (size_t)n * 4));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, scale);
    return p;
}

static void run(const char* name, loco::GemmArgs g, long c_count, int iters) {
    std::vector<float> ref(c_count), got(c_count);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms0 = 0, ms1 = 0;
    const float beta = g.beta;
    // converting kernel
    CK(hipMemset(g.C, 0, (size_t)c_count * 4));
    loco::launch_gemm_bf16x3(g, 0);
    CK(hipMemcpy(ref.data(), g.C, (size_t)c_count * 4, hipMemcpyDeviceToHost));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) loco::launch_gemm_bf16x3(g, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
    // record kernel
    const size_t wsb = loco::gemm_rec_ws_bytes(g);
    unsigned char* ws; CK(hipMalloc(&ws, wsb));
    CK(hipMemset(g.C, 0, (size_t)c_count * 4));
    loco::launch_gemm_rec(g, ws, 0);
    CK(hipMemcpy(got.data(), g.C, (size_t)c_count * 4, hipMemcpyDeviceToHost));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) loco::launch_gemm_rec(g, ws, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
    long nd = 0; double md = 0, mr = 0;
    for (long i = 0; i < c_count; ++i) { double d = fabs((double)ref[i] - got[i]); if (d > 0) ++nd; md = d > md ? d : md; mr = fabs(ref[i]) > mr ? fabs(ref[i]) : mr; }
    const double gf = 2.0 * g.M * g.N * g.K * (g.A2 ? 2 : 1) * g.batch * (g.batch2 > 0 ? g.batch2 : 1) * 1e-9;
    printf("%-34s M %d N %d K %d%s beta %.0f: converting %.1f us (%.0f TF/s)  records %.1f us (%.0f TF/s, ws %.0f MB)  differ %ld of %ld, max |d| %.2e (max |ref| %.2e)\n",
           name, g.M, g.N, g.K, g.A2 ? " x2" : "", beta, 1e3 * ms0 / iters, gf / (ms0 / iters), 1e3 * ms1 / iters, gf / (ms1 / iters), wsb / 1048576.0, nd, c_count, md, mr);
    CK(hipFree(ws));
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 4096, C = argc > 2 ? atoi(argv[2]) : 512, B = argc > 3 ? atoi(argv[3]) : 5, iters = argc > 4 ? atoi(argv[4]) : 5;
    const long CT = (long)C * T, TT = (long)T * T;
    float *q = dalloc(CT, 1, 1.f), *k = dalloc(CT, 2, 1.f), *v = dalloc(CT, 3, 1.f);
    float *dq = dalloc(CT * B, 4, 1.f), *dk = dalloc(CT * B, 5, 1.f), *dv = dalloc(CT * B, 6, 1.f);
    float *SP = dalloc(TT, 7, 1.0f / T), *ST = dalloc(TT * B, 8, 1.0f / T), *S2 = dalloc(TT * B, 9, 0.f), *oT = dalloc(CT * B, 10, 0.f);
    CK(hipDeviceSynchronize());
    loco::GemmArgs g;
    // dS = dq^T k   (engine.hip sa_tangent): A = dq (m = token: unit stride, k = channel), B = k shared
    std::memset(&g, 0, sizeof(g));
    g.A = dq; g.sam = 1; g.sak = T; g.sab = CT; g.Bm = k; g.sbk = T; g.sbn = 1; g.sbb = 0;
    g.C = S2; g.scm = T; g.scn = 1; g.scb = TT; g.M = T; g.N = T; g.K = C; g.batch = B; g.alpha = 1.f; g.beta = 0.f;
    run("dS = dq^T k", g, TT * B, iters);
    // the pair dq^T k + q^T dk as one launch (two K segments)
    g.A2 = q; g.sab2 = 0; g.Bm2 = dk; g.sbb2 = CT;
    run("dS = dq^T k + q^T dk", g, TT * B, iters);
    // accumulate form
    std::memset(&g, 0, sizeof(g));
    g.A = q; g.sam = 1; g.sak = T; g.sab = 0; g.Bm = dk; g.sbk = T; g.sbn = 1; g.sbb = CT;
    g.C = S2; g.scm = T; g.scn = 1; g.scb = TT; g.M = T; g.N = T; g.K = C; g.batch = B; g.alpha = 1.f; g.beta = 1.f;
    run("dS += q^T dk", g, TT * B, iters);
    // do = dv P^T: A = dv (m = channel, k = token unit stride), B[k][n] = P[n][k] shared
    std::memset(&g, 0, sizeof(g));
    g.A = dv; g.sam = T; g.sak = 1; g.sab = CT; g.Bm = SP; g.sbk = 1; g.sbn = T; g.sbb = 0;
    g.C = oT; g.scm = T; g.scn = 1; g.scb = CT; g.M = C; g.N = T; g.K = T; g.batch = B; g.alpha = 1.f; g.beta = 0.f;
    run("do = dv P^T", g, CT * B, iters);
    // do += v dP^T: A = v shared, B = dP per probe
    std::memset(&g, 0, sizeof(g));
    g.A = v; g.sam = T; g.sak = 1; g.sab = 0; g.Bm = ST; g.sbk = 1; g.sbn = T; g.sbb = TT;
    g.C = oT; g.scm = T; g.scn = 1; g.scb = CT; g.M = C; g.N = T; g.K = T; g.batch = B; g.alpha = 1.f; g.beta = 1.f;
    run("do += v dP^T", g, CT * B, iters);
    // g_k[c][j] = sum_i q[c][i] g_S[i][j]: B[k][n] = g_S[k][n] (k = row)
    std::memset(&g, 0, sizeof(g));
    g.A = q; g.sam = T; g.sak = 1; g.sab = 0; g.Bm = ST; g.sbk = T; g.sbn = 1; g.sbb = TT;
    g.C = oT; g.scm = T; g.scn = 1; g.scb = CT; g.M = C; g.N = T; g.K = T; g.batch = B; g.alpha = 1.f; g.beta = 0.f;
    run("g_k = q g_S", g, CT * B, iters);
    return 0;
}
