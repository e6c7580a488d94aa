// Generic strided batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for the
// attention products of the AttnBlock (q^T k, v P^T and their tangent /
// cotangent forms; reference diffusion.py:948-962 torch.bmm calls).  Operands
// are addressed through (row, col, batch) strides so q/k/v stay in their
// [channel][token] conv layout and transposes are never materialised.
// 64x64 tile, 4 waves (one 32x32 block each), BK = 64 (measured 1.1 ms/step faster than 32), next tile prefetched into registers.
#include "kernels.h"

namespace loco {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 64, GBN = 64, GBK = 64;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[GBK][GBM + 1];
    __shared__ float Bs[GBK][GBN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int nb2 = g.batch2 > 0 ? g.batch2 : 1;
    const int bz = blockIdx.z / nb2, hz = blockIdx.z % nb2;
    const float* A = g.A + (long)bz * g.sab + (long)hz * g.sah;
    const float* B = g.Bm + (long)bz * g.sbb + (long)hz * g.sbh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // per-thread staging coordinates: thread order follows the unit-stride dimension of each operand
    constexpr int NE = GBM * GBK / 256;     // 8 elements of A and of B per thread and tile
    int am[NE], ak[NE], bn[NE], bk[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = tid + i * 256;
        if (g.sak == 1) { am[i] = e / GBK; ak[i] = e % GBK; } else { ak[i] = e / GBM; am[i] = e % GBM; }
        if (g.sbn == 1) { bk[i] = e / GBN; bn[i] = e % GBN; } else { bn[i] = e / GBK; bk[i] = e % GBK; }
    }
    float ra[NE], rb[NE];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            bool va = (m0 + am[i] < g.M) && (k0 + ak[i] < g.K);
            bool vb = (n0 + bn[i] < g.N) && (k0 + bk[i] < g.K);
            long oa = va ? (long)(m0 + am[i]) * g.sam + (long)(k0 + ak[i]) * g.sak : 0;
            long ob = vb ? (long)(k0 + bk[i]) * g.sbk + (long)(n0 + bn[i]) * g.sbn : 0;
            float xa = A[oa], xb = B[ob];
            ra[i] = va ? xa : 0.f;
            rb[i] = vb ? xb : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += GBK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NE; ++i) { As[ak[i]][am[i]] = ra[i]; Bs[bk[i]][bn[i]] = rb[i]; }
        __syncthreads();
        if (k0 + GBK < g.K) fetch(k0 + GBK);
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            float a = As[2 * kk + khalf][wm * 32 + l31];
            float b = Bs[2 * kk + khalf][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn * 32 + l31;
    if (n < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (m >= g.M) continue;
            long off = (long)m * g.scm + (long)n * g.scn;
            float v = g.alpha * acc[r];
            float* c = g.C + (long)bz * g.scb + (long)hz * g.sch + off;
            if (g.beta != 0.f) v += g.beta * (*c);
            if (g.bias) v += g.bias[m];
            if (g.R) v += g.R[(long)bz * g.srb + (long)hz * g.sch + off];
            *c = v;
        }
    }
}

void launch_gemm(const GemmArgs& g, hipStream_t st) {
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, g.batch * (g.batch2 > 0 ? g.batch2 : 1));
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, g);
}

}  // namespace loco
