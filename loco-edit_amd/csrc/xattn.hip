// Text cross-attention of one (sample, head) fused into a single kernel (Stable Diffusion / IF-shaped denoisers: the
// `attn2` of a SpatialTransformer block, reference call sites edit.py:619-623, 1319-1322 through diffusers'
// UNet2DConditionModel).  The keys / values are the projected prompt states K_ctx, V_ctx [C][Lp] (L = 77 real tokens,
// constant with respect to the image), so
//     forward    P = softmax_l(scale q^T K),             o   = V P^T
//     tangent    R = scale P o (dS - <P, dS>),  dS = dq^T K,    do  = V R^T
//     cotangent  R = scale P o (gP - <P, gP>),  gP = g_o^T V,   g_q = K R^T
// are all "scores of X against K1 -> row operation -> values out of K2": per (sample, head) only T x L scores, tiny
// contractions (40-160 channels, 77 keys).  The generic route (strided GEMM, row kernel, strided GEMM) writes the
// [T][Lp] score rows of every (sample, head) to HBM and reads them twice more.  Here a workgroup owns 64 tokens of one
// (sample, head): K1 and K2 live in LDS ([CH][4 LG] floats, columns >= L zero), a token's score row is spread over 4
// adjacent lanes (LG columns each, in registers), the row sums are two shuffles, and the products run on the fp32 VALU
// (exact fp32 in every arithmetic mode: the contractions are too short to feed the matrix pipe).
// Tensors keep the engine's [channel][token] layout; P is [head][T][Lp] per sample (Lp = padded row length, columns
// >= L written as zero by the forward kernel).
#include "kernels.h"

namespace loco {

namespace {

constexpr int XA_TOK = 64;      // tokens per workgroup (256 threads = 64 tokens x 4 column groups)

template <int LG, bool FWD>
__global__ __launch_bounds__(256) void xattn_fused_kernel(XAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xa_sm[];
    constexpr int LW = 4 * LG;                               // columns held (>= L)
    float* const K1s = xa_sm;                                // [CH][LW]
    float* const K2s = xa_sm + (long)a.CH * LW;
    const int tid = threadIdx.x, g = tid & 3, tl = tid >> 2;
    const int t = blockIdx.x * XA_TOK + tl, h = blockIdx.y, b = blockIdx.z;
    // ---- K1, K2 of this head -> LDS (columns >= L as zero)
    for (int e = tid; e < a.CH * LW; e += 256) {
        const int c = e / LW, l = e - c * LW;
        const bool ok = l < a.L;
        K1s[e] = ok ? a.K1[(long)(h * a.CH + c) * a.Lp + l] : 0.f;
        K2s[e] = ok ? a.K2[(long)(h * a.CH + c) * a.Lp + l] : 0.f;
    }
    __syncthreads();
    // ---- scores of this token against the lane's LG columns
    float S[LG];
#pragma unroll
    for (int j = 0; j < LG; ++j) S[j] = 0.f;
    const float* xp = a.X + (long)b * a.x_bs + (long)h * a.CH * a.T + t;
    for (int c = 0; c < a.CH; ++c) {
        const float x = xp[(long)c * a.T];
        const float4* kr = reinterpret_cast<const float4*>(K1s + c * LW + g * LG);
#pragma unroll
        for (int j = 0; j < LG / 4; ++j) {
            const float4 k = kr[j];
            S[4 * j] = fmaf(x, k.x, S[4 * j]); S[4 * j + 1] = fmaf(x, k.y, S[4 * j + 1]);
            S[4 * j + 2] = fmaf(x, k.z, S[4 * j + 2]); S[4 * j + 3] = fmaf(x, k.w, S[4 * j + 3]);
        }
    }
    // ---- row operation; the 4 lanes of a token are adjacent
    float* prow = a.P + (long)(FWD ? b : 0) * a.p_bs + ((long)h * a.T + t) * a.Lp + g * LG;
    if constexpr (FWD) {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < LG; ++j) {
            S[j] = (g * LG + j < a.L) ? a.alpha * S[j] : -INFINITY;
            mx = fmaxf(mx, S[j]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < LG; ++j) { S[j] = __expf(S[j] - mx); sum += S[j]; }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < LG; ++j) S[j] *= inv;
#pragma unroll
        for (int j = 0; j < LG / 4; ++j)
            if (g * LG + 4 * j < a.Lp)
                *reinterpret_cast<float4*>(prow + 4 * j) = make_float4(S[4 * j], S[4 * j + 1], S[4 * j + 2], S[4 * j + 3]);
        // columns [4 LG, Lp) of the row (beyond what the lanes hold) are padding: zero
        for (int l = LW + g * 4; l < a.Lp; l += 16)
            *reinterpret_cast<float4*>(a.P + (long)b * a.p_bs + ((long)h * a.T + t) * a.Lp + l) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float P[LG];
#pragma unroll
        for (int j = 0; j < LG / 4; ++j) {
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g * LG + 4 * j < a.Lp) p = *reinterpret_cast<const float4*>(prow + 4 * j);
            P[4 * j] = p.x; P[4 * j + 1] = p.y; P[4 * j + 2] = p.z; P[4 * j + 3] = p.w;
        }
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < LG; ++j) dot = fmaf(P[j], S[j], dot);
        dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2);
#pragma unroll
        for (int j = 0; j < LG; ++j) S[j] = a.alpha * P[j] * (S[j] - dot);
    }
    // ---- values: O[c][t] = sum_l K2[c][l] S[l]; lane g of the token stores the channels c = g (mod 4)
    float* op = a.O + (long)b * a.o_bs + (long)h * a.CH * a.T + t;
    for (int c0 = 0; c0 < a.CH; c0 += 4) {
        float mine = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = c0 + cc;
            float acc = 0.f;
            if (c < a.CH) {
                const float4* kr = reinterpret_cast<const float4*>(K2s + c * LW + g * LG);
#pragma unroll
                for (int j = 0; j < LG / 4; ++j) {
                    const float4 k = kr[j];
                    acc = fmaf(k.x, S[4 * j], acc); acc = fmaf(k.y, S[4 * j + 1], acc);
                    acc = fmaf(k.z, S[4 * j + 2], acc); acc = fmaf(k.w, S[4 * j + 3], acc);
                }
            }
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
            if (cc == g) mine = acc;
        }
        if (c0 + g < a.CH) op[(long)(c0 + g) * a.T] = mine;
    }
}

template <int LG>
void xattn_launch(const XAttnArgs& a, hipStream_t st) {
    const dim3 grid(a.T / XA_TOK, a.NH, a.B);
    const size_t lds = (size_t)2 * a.CH * 4 * LG * sizeof(float);
    auto kf = &xattn_fused_kernel<LG, true>;
    auto kl = &xattn_fused_kernel<LG, false>;
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (first_on_device(once)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    }
    if (a.fwd) hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(kl, grid, dim3(256), lds, st, a);
}

int xattn_lg(int L) {           // columns per lane: the smallest instantiated LG with 4 LG >= L
    const int need = (L + 3) / 4;
    for (int lg : {4, 8, 12, 20, 32})
        if (lg >= need) return lg;
    return 0;
}

}  // namespace

bool xattn_fused_supported(int T, int CH, int L, int Lp) {
    const int lg = xattn_lg(L);
    return lg > 0 && (T % XA_TOK) == 0 && (Lp % 4) == 0 && (size_t)2 * CH * 4 * lg * sizeof(float) <= 150 * 1024;
}

void launch_xattn_fused(const XAttnArgs& a, hipStream_t st) {
    switch (xattn_lg(a.L)) {
        case 4: xattn_launch<4>(a, st); break;
        case 8: xattn_launch<8>(a, st); break;
        case 12: xattn_launch<12>(a, st); break;
        case 20: xattn_launch<20>(a, st); break;
        default: xattn_launch<32>(a, st); break;
    }
}

}  // namespace loco
