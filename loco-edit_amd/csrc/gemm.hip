// Generic strided batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for the
// attention products of the AttnBlock (q^T k, v P^T and their tangent /
// cotangent forms; reference diffusion.py:948-962 torch.bmm calls).  Operands
// are addressed through (row, col, batch) strides so q/k/v stay in their
// [channel][token] conv layout and transposes are never materialised.
// 64x64 tile, 4 waves (one 32x32 block each), BK = 64 (measured 1.1 ms/step faster than 32), next tile prefetched into registers.
#include "kernels.h"
#include <cstdlib>

namespace loco {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 64, GBN = 64, GBK = 64;

template <int NSEG>
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
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
        if (seg) {
            A = g.A2 + (long)bz * g.sab2 + (long)hz * g.sah;
            B = g.Bm2 + (long)bz * g.sbb2 + (long)hz * g.sbh;
        }
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
                if (g.colbias) v += g.colbias[n];
            if (g.R) v += g.R[(long)bz * g.srb + (long)hz * g.sch + off];
            *c = v;
        }
    }
}

// Small products (the 256-token attention of the DDPM denoiser: M = N = 256, K = 512 / 1024, 5 probes): the 64 x 64 tiling gives
// 80 workgroups of one wave per SIMD, each a serial chain of K / 2 f32 MFMAs at 64 cycles -- 18 - 39 us per launch with two
// thirds of the chip idle (4 % of the headline step).  Here a workgroup owns a 32 x 32 tile and its four waves split every
// K-step between them (wave w: k-pairs 8 w ... 8 w + 7 of the 32 of a step), the four partial tiles are summed through LDS in
// wave order: 4 x the workgroups, a quarter of the chain.  Summation order differs from the kernel above (four interleaved
// partial chains), the arithmetic is the same exact-fp32 MFMA.
constexpr int SBM = 32, SBN = 32, SBK = 64;       // (K-steps of 128 measured worse: -0.45 % vs -0.85 % per headline step)
template <int NSEG>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(GemmArgs g) {
    __shared__ float As[SBK][SBM + 1];
    __shared__ float Bs[SBK][SBN + 1];
    __shared__ float Red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int m0 = blockIdx.y * SBM, n0 = blockIdx.x * SBN;
    const int nb2 = g.batch2 > 0 ? g.batch2 : 1;
    const int bz = blockIdx.z / nb2, hz = blockIdx.z % nb2;
    const float* A = g.A + (long)bz * g.sab + (long)hz * g.sah;
    const float* B = g.Bm + (long)bz * g.sbb + (long)hz * g.sbh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NE = SBM * SBK / 256;     // 16 elements of A and of B per thread and tile
    int am[NE], ak[NE], bn[NE], bk[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = tid + i * 256;
        if (g.sak == 1) { am[i] = e / SBK; ak[i] = e % SBK; } else { ak[i] = e / SBM; am[i] = e % SBM; }
        if (g.sbn == 1) { bk[i] = e / SBN; bn[i] = e % SBN; } else { bn[i] = e / SBK; bk[i] = e % SBK; }
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
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
        if (seg) {
            A = g.A2 + (long)bz * g.sab2 + (long)hz * g.sah;
            B = g.Bm2 + (long)bz * g.sbb2 + (long)hz * g.sbh;
        }
        fetch(0);
        for (int k0 = 0; k0 < g.K; k0 += SBK) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NE; ++i) { As[ak[i]][am[i]] = ra[i]; Bs[bk[i]][bn[i]] = rb[i]; }
            __syncthreads();
            if (k0 + SBK < g.K) fetch(k0 + SBK);
#pragma unroll
            for (int kk = 0; kk < SBK / 8; ++kk) {
                float a = As[2 * (wave * (SBK / 8) + kk) + khalf][l31];
                float b = Bs[2 * (wave * (SBK / 8) + kk) + khalf][l31];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
    }
    // waves 1 .. 3 hand their partial tile to wave 0, which sums them in wave order and writes the tile
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += Red[w][r][lane];
    const int n = n0 + l31;
    if (n < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (m >= g.M) continue;
            long off = (long)m * g.scm + (long)n * g.scn;
            float v = g.alpha * acc[r];
            float* c = g.C + (long)bz * g.scb + (long)hz * g.sch + off;
            if (g.beta != 0.f) v += g.beta * (*c);
            if (g.bias) v += g.bias[m];
            if (g.colbias) v += g.colbias[n];
            if (g.R) v += g.R[(long)bz * g.srb + (long)hz * g.sch + off];
            *c = v;
        }
    }
}

// ---------------------------------------------------------------------------
// The same strided batched GEMM on the bf16 matrix pipe with SPLIT-bf16 operands (hi + lo halves, three
// v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate: fp32-faithful to ~2^-16 like the convolutions of that mode).
// Used for the attention products that are matrix-rate bound on the f32-input MFMA (1/16 of the bf16 rate): long
// contractions over many tokens, i.e. the latent decoder's 4096-token x 512-channel mid attention (score product
// K = 512, value product K = 4096).  The short multi-head products (K = 64) and the 256-token blocks are bound by the
// materialised score matrix / by launch latency and measured the same on both kernels (profiles/r02_conv_analysis.md
// section 6), so they stay on the exact kernel above.  BM x BN tile per workgroup, 4 waves of (BM/2) x (BN/2), BK = 32
// per stage.  Staging: a thread owns 8 consecutive k of one row (column), loaded with that operand's k stride --
// coalesced across lanes when the row index is the unit-stride one, 32-byte runs when k is -- converts them to one
// 16-byte hi and one 16-byte lo piece and writes the two pieces of the row's 64-byte record
// [hi k0-7|hi k8-15|lo k0-7|lo k8-15] (piece index XORed with (row>>2)&3: conflict-free ds_read_b128 fragments).
typedef __bf16 bf16x8_g __attribute__((ext_vector_type(8)));

// hi / lo halves of eight values: ONE v_cvt_pk_bf16_f32 per pair and half (the scalar __bf16 casts compile to a conversion per
// value plus shift / or packing: 11 vector instructions per pair instead of 6; same roundings, same bits -- conv_bf16_kernel.h cvt2)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    float da = a - ha;
    asm volatile("" : "+v"(da));
    const float db = b - hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){da, db}, bf16x2_));
}
__device__ __forceinline__ void split_piece(const float* v, uint4& hi, uint4& lo) {
    split_pair(v[0], v[1], hi.x, lo.x);
    split_pair(v[2], v[3], hi.y, lo.y);
    split_pair(v[4], v[5], hi.z, lo.z);
    split_pair(v[6], v[7], hi.w, lo.w);
}
__device__ __forceinline__ int grec(int row, int piece) { return row * 64 + ((piece ^ ((row >> 2) & 3)) << 4); }

template <int BM, int BN, int NSEG>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs g) {
    constexpr int BK = 32, KS = BK / 16;            // two MFMA k-steps per stage
    constexpr int TM = BM / 64, TN = BN / 64;       // 32x32 blocks per wave and dimension
    constexpr int NA = BM * 4 / 256, NB = BN * 4 / 256;   // (row, 8-k group) items per thread
    __shared__ __attribute__((aligned(16))) unsigned char As[KS * BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[KS * BN * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nb2 = g.batch2 > 0 ? g.batch2 : 1;
    const int bz = blockIdx.z / nb2, hz = blockIdx.z % nb2;
    const float* A = g.A + (long)bz * g.sab + (long)hz * g.sah;
    const float* B = g.Bm + (long)bz * g.sbb + (long)hz * g.sbh;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[NA][8], rb[NB][8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int e = tid + it * 256, m = e % BM, kq = e / BM;
            const bool vm = m0 + m < g.M;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + kq * 8 + j;
                const bool v = vm && k < g.K;
                const float x = A[v ? (long)(m0 + m) * g.sam + (long)k * g.sak : 0];
                ra[it][j] = v ? x : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int e = tid + it * 256, n = e % BN, kq = e / BN;
            const bool vn = n0 + n < g.N;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + kq * 8 + j;
                const bool v = vn && k < g.K;
                const float x = B[v ? (long)k * g.sbk + (long)(n0 + n) * g.sbn : 0];
                rb[it][j] = v ? x : 0.f;
            }
        }
    };
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
    if (seg) {
        A = g.A2 + (long)bz * g.sab2 + (long)hz * g.sah;
        B = g.Bm2 + (long)bz * g.sbb2 + (long)hz * g.sbh;
    }
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int e = tid + it * 256, m = e % BM, kq = e / BM;      // kq 0..3: k-step kq>>1, half kq&1
            uint4 hi, lo;
            split_piece(ra[it], hi, lo);
            unsigned char* rec = As + (kq >> 1) * BM * 64;
            *reinterpret_cast<uint4*>(rec + grec(m, kq & 1)) = hi;
            *reinterpret_cast<uint4*>(rec + grec(m, 2 + (kq & 1))) = lo;
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int e = tid + it * 256, n = e % BN, kq = e / BN;
            uint4 hi, lo;
            split_piece(rb[it], hi, lo);
            unsigned char* rec = Bs + (kq >> 1) * BN * 64;
            *reinterpret_cast<uint4*>(rec + grec(n, kq & 1)) = hi;
            *reinterpret_cast<uint4*>(rec + grec(n, 2 + (kq & 1))) = lo;
        }
        __syncthreads();
        if (k0 + BK < g.K) fetch(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_g ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = (wm * TM + i) * 32 + l31;
                ah[i] = *reinterpret_cast<const bf16x8_g*>(As + ks * BM * 64 + grec(m, khalf));
                al[i] = *reinterpret_cast<const bf16x8_g*>(As + ks * BM * 64 + grec(m, 2 + khalf));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = (wn * TN + j) * 32 + l31;
                bh[j] = *reinterpret_cast<const bf16x8_g*>(Bs + ks * BN * 64 + grec(n, khalf));
                bl[j] = *reinterpret_cast<const bf16x8_g*>(Bs + ks * BN * 64 + grec(n, 2 + khalf));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (m >= g.M) continue;
                const long off = (long)m * g.scm + (long)n * g.scn;
                float v = g.alpha * acc[i][j][r];
                float* c = g.C + (long)bz * g.scb + (long)hz * g.sch + off;
                if (g.beta != 0.f) v += g.beta * (*c);
                if (g.bias) v += g.bias[m];
                if (g.colbias) v += g.colbias[n];
                if (g.R) v += g.R[(long)bz * g.srb + (long)hz * g.sch + off];
                *c = v;
            }
        }
}

bool gemm_prefers_bf16x3(const GemmArgs& g) {
    // long contractions with enough work per launch to be bound by the f32-input matrix rate
    const double macs = (double)g.M * g.N * g.K * (g.A2 ? 2 : 1) * g.batch * (g.batch2 > 0 ? g.batch2 : 1);
    return g.K >= 256 && macs >= 4e9;
}

void launch_gemm_bf16x3(const GemmArgs& g, hipStream_t st) {
    const int nb = g.batch * (g.batch2 > 0 ? g.batch2 : 1);
    const long big = (long)((g.N + 127) / 128) * ((g.M + 127) / 128) * nb;
    if (big >= 512 && g.M >= 128 && g.N >= 128) {
        dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, nb);
        if (g.A2) hipLaunchKernelGGL((gemm_bf16x3_kernel<128, 128, 2>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<128, 128, 1>), grid, dim3(256), 0, st, g);
    } else {
        dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, nb);
        if (g.A2) hipLaunchKernelGGL((gemm_bf16x3_kernel<64, 64, 2>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<64, 64, 1>), grid, dim3(256), 0, st, g);
    }
}

void launch_gemm(const GemmArgs& g, hipStream_t st) {
    const int nz = g.batch * (g.batch2 > 0 ? g.batch2 : 1);
    static int small_on = -1;      // LOCO_GEMM_SMALL=0: always the 64 x 64 tiling (A/B)
    if (small_on < 0) { const char* e = getenv("LOCO_GEMM_SMALL"); small_on = e ? (atoi(e) != 0) : 1; }
    // launches whose 64 x 64 grid is about one round of the chip or less and whose contraction is long enough to be the cost
    const long wg64 = (long)((g.N + GBN - 1) / GBN) * ((g.M + GBM - 1) / GBM) * nz;
    static int small_wg = -1;      // LOCO_GEMM_SMALL_WG: largest 64 x 64 grid that still takes the small tiling
    if (small_wg < 0) { const char* e = getenv("LOCO_GEMM_SMALL_WG"); small_wg = e ? atoi(e) : 320; }      // (r05: 128 -> 289.4 ms per headline step, 200 -> 288.6, 400 -> 288.4)
    if (small_on && wg64 < small_wg && g.K >= 128) {
        dim3 sg((g.N + SBN - 1) / SBN, (g.M + SBM - 1) / SBM, nz);
        if (g.A2) hipLaunchKernelGGL(gemm_f32_small_kernel<2>, sg, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(gemm_f32_small_kernel<1>, sg, dim3(256), 0, st, g);
        return;
    }
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, g.batch * (g.batch2 > 0 ? g.batch2 : 1));
    if (g.A2) hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, st, g);
}

}  // namespace loco
