// Generic strided batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for the
// attention products of the AttnBlock (q^T k, v P^T and their tangent /
// cotangent forms; reference diffusion.py:948-962 torch.bmm calls).  Operands
// are addressed through (row, col, batch) strides so q/k/v stay in their
// [channel][token] conv layout and transposes are never materialised.
// 64x64 tile, 4 waves (one 32x32 block each), BK = 16.
#include "kernels.h"

namespace loco {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 64, GBN = 64, GBK = 16;

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

    for (int k0 = 0; k0 < g.K; k0 += GBK) {
        // stage A tile (64 x 16) and B tile (16 x 64); thread order follows the unit-stride dim
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + i * 256;
            int m, k;
            if (g.sak == 1) { m = e / GBK; k = e % GBK; } else { k = e / GBM; m = e % GBM; }
            float v = 0.f;
            if (m0 + m < g.M && k0 + k < g.K) v = A[(long)(m0 + m) * g.sam + (long)(k0 + k) * g.sak];
            As[k][m] = v;
            int n, kb;
            if (g.sbn == 1) { kb = e / GBN; n = e % GBN; } else { n = e / GBK; kb = e % GBK; }
            float w = 0.f;
            if (n0 + n < g.N && k0 + kb < g.K) w = B[(long)(k0 + kb) * g.sbk + (long)(n0 + n) * g.sbn];
            Bs[kb][n] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            float a = As[2 * kk + khalf][wm * 32 + l31];
            float b = Bs[2 * kk + khalf][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
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
