// Token-wise pieces of the latent-diffusion SpatialTransformer (Stable Diffusion v1 denoiser; reference call sites
// edit.py:597, 619-623, 655-658 through diffusers' UNet2DConditionModel): LayerNorm over the channels of each token and
// the GEGLU gate of the feed-forward, each with its tangent and cotangent form.  Tensors keep the engine's
// [channel][token] layout, so a token's channels are T floats apart: threads run over tokens (coalesced row segments)
// and over channel slices.  Bandwidth-bound, 12-20 bytes per element.  Token counts are multiples of 32.
#include "kernels.h"

namespace loco {

namespace {

__device__ __forceinline__ float gelu_f(float b) { return 0.5f * b * (1.0f + erff(b * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float b) {
    return 0.5f * (1.0f + erff(b * 0.70710678118654752f)) + b * 0.39894228040143268f * __expf(-0.5f * b * b);
}

// LayerNorm kernels: a workgroup owns 32 consecutive tokens; its 256 threads are 8 channel slices x 32 tokens, so a
// half-wave reads one 128-byte run of a channel row and the per-token sums over the C channels are 8 partial sums merged
// through LDS.  The 32 x C tile (41-164 KB) is re-read from cache by the second / third pass.
constexpr int LN_TOK = 32, LN_SL = 8;
__device__ __forceinline__ float ln_reduce(float v, float (*sm)[LN_TOK], int sl, int tk) {
    __syncthreads();
    sm[sl][tk] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < LN_SL; ++i) r += sm[i][tk];
    return r;
}
// y = (x - mean) * rstd * gamma + beta per token; stats[0][t] = mean, stats[1][t] = rstd
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, long xbs, int C, int T, const float* gamma,
                                                     const float* beta, float eps, float* y, long ybs, float* stats, long sbs) {
    __shared__ float sm[LN_SL][LN_TOK];
    const int tk = threadIdx.x & 31, sl = threadIdx.x >> 5, t = blockIdx.x * LN_TOK + tk, b = blockIdx.y;
    const float* xp = x + (long)b * xbs + t;
    float s = 0.f;
    for (int c = sl; c < C; c += LN_SL) s += xp[(long)c * T];
    const float mean = ln_reduce(s, sm, sl, tk) / (float)C;
    float m2 = 0.f;
    for (int c = sl; c < C; c += LN_SL) { const float d = xp[(long)c * T] - mean; m2 += d * d; }
    const float rstd = rsqrtf(ln_reduce(m2, sm, sl, tk) / (float)C + eps);
    if (sl == 0) {
        stats[(long)b * sbs + t] = mean;
        stats[(long)b * sbs + T + t] = rstd;
    }
    float* yp = y + (long)b * ybs + t;
    for (int c = sl; c < C; c += LN_SL) yp[(long)c * T] = (xp[(long)c * T] - mean) * rstd * gamma[c] + beta[c];
}
// tangent: dy = rstd * gamma * (dx - mean_c(dx) - xhat * mean_c(xhat dx)),  xhat from the primal x (B = 1)
__global__ __launch_bounds__(256) void ln_tan_kernel(const float* dx, long dbs, const float* xprim, const float* sprim, int C,
                                                     int T, const float* gamma, float* dy, long ybs) {
    __shared__ float sm[LN_SL][LN_TOK];
    const int tk = threadIdx.x & 31, sl = threadIdx.x >> 5, t = blockIdx.x * LN_TOK + tk, b = blockIdx.y;
    const float mean = sprim[t], rstd = sprim[T + t];
    const float* dp = dx + (long)b * dbs + t;
    const float* xp = xprim + t;
    float m1 = 0.f, m2 = 0.f;
    for (int c = sl; c < C; c += LN_SL) {
        const float d = dp[(long)c * T], xh = (xp[(long)c * T] - mean) * rstd;
        m1 += d; m2 += xh * d;
    }
    m1 = ln_reduce(m1, sm, sl, tk) / (float)C;
    m2 = ln_reduce(m2, sm, sl, tk) / (float)C;
    float* yp = dy + (long)b * ybs + t;
    for (int c = sl; c < C; c += LN_SL) {
        const float xh = (xp[(long)c * T] - mean) * rstd;
        yp[(long)c * T] = rstd * gamma[c] * (dp[(long)c * T] - m1 - xh * m2);
    }
}
// cotangent: gx = base + rstd * (z - mean_c(z) - xhat * mean_c(xhat z)),  z = gamma * gy
__global__ __launch_bounds__(256) void ln_cot_kernel(const float* gy, long gbs, const float* xprim, const float* sprim, int C,
                                                     int T, const float* gamma, const float* base, long base_bs, float* gx,
                                                     long xbs) {
    __shared__ float sm[LN_SL][LN_TOK];
    const int tk = threadIdx.x & 31, sl = threadIdx.x >> 5, t = blockIdx.x * LN_TOK + tk, b = blockIdx.y;
    const float mean = sprim[t], rstd = sprim[T + t];
    const float* gp = gy + (long)b * gbs + t;
    const float* xp = xprim + t;
    float m1 = 0.f, m2 = 0.f;
    for (int c = sl; c < C; c += LN_SL) {
        const float z = gamma[c] * gp[(long)c * T], xh = (xp[(long)c * T] - mean) * rstd;
        m1 += z; m2 += xh * z;
    }
    m1 = ln_reduce(m1, sm, sl, tk) / (float)C;
    m2 = ln_reduce(m2, sm, sl, tk) / (float)C;
    float* op = gx + (long)b * xbs + t;
    for (int c = sl; c < C; c += LN_SL) {
        const float xh = (xp[(long)c * T] - mean) * rstd;
        float r = rstd * (gamma[c] * gp[(long)c * T] - m1 - xh * m2);
        if (base) r += base[(long)b * base_bs + t + (long)c * T];
        op[(long)c * T] = r;
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

void launch_ln_fwd(const float* x, long xbs, int B, int C, int T, const float* gamma, const float* beta, float eps, float* y,
                   long ybs, float* stats, long sbs, hipStream_t st) {
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(T / LN_TOK, B), dim3(256), 0, st, x, xbs, C, T, gamma, beta, eps, y, ybs, stats, sbs);
}
void launch_ln_tan(const float* dx, long dbs, const float* xprim, const float* sprim, int B, int C, int T, const float* gamma,
                   float* dy, long ybs, hipStream_t st) {
    hipLaunchKernelGGL(ln_tan_kernel, dim3(T / LN_TOK, B), dim3(256), 0, st, dx, dbs, xprim, sprim, C, T, gamma, dy, ybs);
}
void launch_ln_cot(const float* gy, long gbs, const float* xprim, const float* sprim, int B, int C, int T, const float* gamma,
                   const float* base, long base_bs, float* gx, long xbs, hipStream_t st) {
    hipLaunchKernelGGL(ln_cot_kernel, dim3(T / LN_TOK, B), dim3(256), 0, st, gy, gbs, xprim, sprim, C, T, gamma, base,
                       base_bs, gx, xbs);
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
