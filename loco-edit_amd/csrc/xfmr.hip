// Token-wise pieces of the latent-diffusion SpatialTransformer (Stable Diffusion v1 denoiser; reference call sites
// edit.py:597, 619-623, 655-658 through diffusers' UNet2DConditionModel): LayerNorm over the channels of each token and
// the GEGLU gate of the feed-forward, each with its tangent and cotangent form.  Tensors keep the engine's
// [channel][token] layout, so a token's channels are T floats apart: threads run over tokens (coalesced row segments)
// and over channel slices.  Bandwidth-bound, 12-20 bytes per element.  Token counts are multiples of 16.
#include "kernels.h"

namespace loco {

namespace {

__device__ __forceinline__ float gelu_f(float b) { return 0.5f * b * (1.0f + erff(b * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float b) {
    return 0.5f * (1.0f + erff(b * 0.70710678118654752f)) + b * 0.39894228040143268f * __expf(-0.5f * b * b);
}

// LayerNorm kernels: a workgroup owns 16 consecutive tokens; its 1024 threads are 64 channel slices x 16 tokens, so 16
// lanes read one 64-byte run of a channel row and a wave covers 4 slices.  NPER = C / 64 channels per thread stay in
// registers between the statistics and the apply (C = 320 / 640 / 1280 of the Stable Diffusion denoiser: 5 / 10 / 20);
// NPER = 0 is the general form, which re-reads the tile from cache.  Per-token sums: two shuffles inside the wave, then
// the 16 waves through LDS.  (The first version ran 8 slices of up to 160 dependent iterations on 40 workgroups: 128 us
// for a 1.6 MB tensor.)
constexpr int LN_TOK = 16, LN_SL = 64, LN_THREADS = LN_TOK * LN_SL, LN_WAVES = LN_THREADS / 64;
__device__ __forceinline__ float ln_reduce(float v, float (*sm)[LN_TOK], int tk) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    __syncthreads();
    if ((threadIdx.x & 63) < LN_TOK) sm[threadIdx.x >> 6][tk] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < LN_WAVES; ++i) r += sm[i][tk];
    return r;
}
// y = (x - mean) * rstd * gamma + beta per token; stats[0][t] = mean, stats[1][t] = rstd
template <int NPER>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* x, long xbs, int C, int T, const float* gamma,
                                                            const float* beta, float eps, float* y, long ybs, float* stats, long sbs) {
    __shared__ float sm[LN_WAVES][LN_TOK];
    const int tk = threadIdx.x & (LN_TOK - 1), sl = threadIdx.x / LN_TOK, t = blockIdx.x * LN_TOK + tk, b = blockIdx.y;
    const float* xp = x + (long)b * xbs + t;
    float* yp = y + (long)b * ybs + t;
    float xv[NPER > 0 ? NPER : 1];
    float s = 0.f;
    if constexpr (NPER > 0) {
#pragma unroll
        for (int i = 0; i < NPER; ++i) { xv[i] = xp[(long)(sl + LN_SL * i) * T]; s += xv[i]; }
    } else {
        for (int c = sl; c < C; c += LN_SL) s += xp[(long)c * T];
    }
    const float mean = ln_reduce(s, sm, tk) / (float)C;
    float m2 = 0.f;
    if constexpr (NPER > 0) {
#pragma unroll
        for (int i = 0; i < NPER; ++i) { const float d = xv[i] - mean; m2 += d * d; }
    } else {
        for (int c = sl; c < C; c += LN_SL) { const float d = xp[(long)c * T] - mean; m2 += d * d; }
    }
    const float rstd = rsqrtf(ln_reduce(m2, sm, tk) / (float)C + eps);
    if (sl == 0) {
        stats[(long)b * sbs + t] = mean;
        stats[(long)b * sbs + T + t] = rstd;
    }
    if constexpr (NPER > 0) {
#pragma unroll
        for (int i = 0; i < NPER; ++i) { const int c = sl + LN_SL * i; yp[(long)c * T] = (xv[i] - mean) * rstd * gamma[c] + beta[c]; }
    } else {
        for (int c = sl; c < C; c += LN_SL) yp[(long)c * T] = (xp[(long)c * T] - mean) * rstd * gamma[c] + beta[c];
    }
}
// tangent (COT = false): dy = rstd * gamma * (dx - mean_c(dx) - xhat * mean_c(xhat dx)),  xhat from the primal x (B = 1)
// cotangent (COT = true): gx = base + rstd * (z - mean_c(z) - xhat * mean_c(xhat z)),  z = gamma * gy
template <int NPER, bool COT>
__global__ __launch_bounds__(LN_THREADS) void ln_lin_kernel(const float* dx, long dbs, const float* xprim, const float* sprim,
                                                            int C, int T, const float* gamma, const float* base, long base_bs,
                                                            float* dy, long ybs) {
    __shared__ float sm[LN_WAVES][LN_TOK];
    const int tk = threadIdx.x & (LN_TOK - 1), sl = threadIdx.x / LN_TOK, t = blockIdx.x * LN_TOK + tk, b = blockIdx.y;
    const float mean = sprim[t], rstd = sprim[T + t];
    const float* dp = dx + (long)b * dbs + t;
    const float* xp = xprim + t;
    float* yp = dy + (long)b * ybs + t;
    const float* bp = base ? base + (long)b * base_bs + t : nullptr;
    float zv[NPER > 0 ? NPER : 1], xh[NPER > 0 ? NPER : 1];
    float m1 = 0.f, m2 = 0.f;
    if constexpr (NPER > 0) {
#pragma unroll
        for (int i = 0; i < NPER; ++i) {
            const int c = sl + LN_SL * i;
            zv[i] = dp[(long)c * T];
            xh[i] = xp[(long)c * T];
        }
#pragma unroll
        for (int i = 0; i < NPER; ++i) {
            if (COT) zv[i] *= gamma[sl + LN_SL * i];
            xh[i] = (xh[i] - mean) * rstd;
            m1 += zv[i]; m2 += xh[i] * zv[i];
        }
    } else {
        for (int c = sl; c < C; c += LN_SL) {
            const float z = (COT ? gamma[c] : 1.f) * dp[(long)c * T], h = (xp[(long)c * T] - mean) * rstd;
            m1 += z; m2 += h * z;
        }
    }
    m1 = ln_reduce(m1, sm, tk) / (float)C;
    m2 = ln_reduce(m2, sm, tk) / (float)C;
    if constexpr (NPER > 0) {
#pragma unroll
        for (int i = 0; i < NPER; ++i) {
            const int c = sl + LN_SL * i;
            float r = rstd * (COT ? 1.f : gamma[c]) * (zv[i] - m1 - xh[i] * m2);
            if (COT && bp) r += bp[(long)c * T];
            yp[(long)c * T] = r;
        }
    } else {
        for (int c = sl; c < C; c += LN_SL) {
            const float z = (COT ? gamma[c] : 1.f) * dp[(long)c * T], h = (xp[(long)c * T] - mean) * rstd;
            float r = rstd * (COT ? 1.f : gamma[c]) * (z - m1 - h * m2);
            if (COT && bp) r += bp[(long)c * T];
            yp[(long)c * T] = r;
        }
    }
}

// GEGLU: f = [value (C4 rows) | gate (C4 rows)] x T tokens
//   kind 0: out = value * gelu(gate)            kind 1 (tangent): out = dvalue * gelu(gate) + value * gelu'(gate) * dgate
//   kind 2 (cotangent): gf = [ g * gelu(gate) | g * value * gelu'(gate) ]
template <int KIND>
__global__ __launch_bounds__(256) void geglu_kernel(const float* in, long in_bs, const float* fprim, long n4, float* out, long out_bs) {
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        if (KIND == 0) {
            const float* f = in + (long)b * in_bs;
            out[(long)b * out_bs + i] = f[i] * gelu_f(f[n4 + i]);
        } else if (KIND == 1) {
            const float* d = in + (long)b * in_bs;
            const float g = fprim[n4 + i];
            out[(long)b * out_bs + i] = d[i] * gelu_f(g) + fprim[i] * dgelu_f(g) * d[n4 + i];
        } else {
            const float gg = in[(long)b * in_bs + i], g = fprim[n4 + i];
            float* o = out + (long)b * out_bs;
            o[i] = gg * gelu_f(g);
            o[n4 + i] = gg * fprim[i] * dgelu_f(g);
        }
    }
}

}  // namespace

// NPER dispatch: the register-resident forms for C = 64 * {5, 10, 20}, the general form otherwise
#define LN_DISPATCH(C, CALL)                                   \
    do {                                                       \
        if ((C) == 5 * LN_SL) { CALL(5); }                     \
        else if ((C) == 10 * LN_SL) { CALL(10); }              \
        else if ((C) == 20 * LN_SL) { CALL(20); }              \
        else { CALL(0); }                                      \
    } while (0)

void launch_ln_fwd(const float* x, long xbs, int B, int C, int T, const float* gamma, const float* beta, float eps, float* y,
                   long ybs, float* stats, long sbs, hipStream_t st) {
#define LN_CALL(N) hipLaunchKernelGGL(ln_fwd_kernel<N>, dim3(T / LN_TOK, B), dim3(LN_THREADS), 0, st, x, xbs, C, T, gamma, beta, eps, y, ybs, stats, sbs)
    LN_DISPATCH(C, LN_CALL);
#undef LN_CALL
}
void launch_ln_tan(const float* dx, long dbs, const float* xprim, const float* sprim, int B, int C, int T, const float* gamma,
                   float* dy, long ybs, hipStream_t st) {
#define LN_CALL(N) hipLaunchKernelGGL((ln_lin_kernel<N, false>), dim3(T / LN_TOK, B), dim3(LN_THREADS), 0, st, dx, dbs, xprim, sprim, C, T, gamma, (const float*)nullptr, 0L, dy, ybs)
    LN_DISPATCH(C, LN_CALL);
#undef LN_CALL
}
void launch_ln_cot(const float* gy, long gbs, const float* xprim, const float* sprim, int B, int C, int T, const float* gamma,
                   const float* base, long base_bs, float* gx, long xbs, hipStream_t st) {
#define LN_CALL(N) hipLaunchKernelGGL((ln_lin_kernel<N, true>), dim3(T / LN_TOK, B), dim3(LN_THREADS), 0, st, gy, gbs, xprim, sprim, C, T, gamma, base, base_bs, gx, xbs)
    LN_DISPATCH(C, LN_CALL);
#undef LN_CALL
}
void launch_geglu(int kind, const float* in, long in_bs, const float* fprim, int B, long n4, float* out, long out_bs,
                  hipStream_t st) {
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    dim3 grid(blocks, B);
    if (kind == 0) hipLaunchKernelGGL(geglu_kernel<0>, grid, dim3(256), 0, st, in, in_bs, fprim, n4, out, out_bs);
    else if (kind == 1) hipLaunchKernelGGL(geglu_kernel<1>, grid, dim3(256), 0, st, in, in_bs, fprim, n4, out, out_bs);
    else hipLaunchKernelGGL(geglu_kernel<2>, grid, dim3(256), 0, st, in, in_bs, fprim, n4, out, out_bs);
}

}  // namespace loco
