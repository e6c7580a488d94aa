// Bandwidth-bound kernels around the convolutions: GroupNorm statistics (wave
// shuffle + LDS reductions, emitting only per-channel scale/shift for the conv
// prologue), their tangent / cotangent forms, the attention softmax and its
// Jacobian, time embedding, DDIM algebra.  All 16-byte vectorised, grid-strided.
#include "kernels.h"

namespace loco {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
// block-wide sum of a double over 256 threads; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
    return r;
}
// activation and its derivative: act_fwd / act_der of kernels.h (ACT_SILU, ACT_GELU), a wave-uniform switch in these kernels

// ---------------------------------------------------------------------------
// GroupNorm statistics.  grid = (nsplit, G, B).  Each block reduces a 16B-aligned
// slice of the group's contiguous cpg*HW floats with a two-pass (mean, then
// centred M2) scheme; slices are merged with Chan's formula in double.
// scratch[((b*G+g)*nsplit + s)*3 + {0,1,2}] = {n, mean, M2}
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* x, long bs, long len, double* scratch) {
    __shared__ double sm[4];
    const int s = blockIdx.x, nsplit = gridDim.x, g = blockIdx.y, b = blockIdx.z, G = gridDim.y;
    const float* p = x + (long)b * bs + (long)g * len;
    long per = ((len / 4 + nsplit - 1) / nsplit) * 4;
    long beg = (long)s * per, end = beg + per < len ? beg + per : len;
    if (beg > end) beg = end;
    const long n4 = (end - beg) / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p + beg);
    double sum = 0.0;
    for (long i = threadIdx.x; i < n4; i += 256) {
        float4 v = p4[i];
        sum += (double)((v.x + v.y) + (v.z + v.w));
    }
    for (long i = beg + n4 * 4 + threadIdx.x; i < end; i += 256) sum += (double)p[i];
    double tot = block_sum(sum, sm);
    const double cnt = (double)(end - beg);
    const float mean = cnt > 0 ? (float)(tot / cnt) : 0.f;
    double m2 = 0.0;
    for (long i = threadIdx.x; i < n4; i += 256) {
        float4 v = p4[i];
        float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        m2 += (double)((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
    }
    for (long i = beg + n4 * 4 + threadIdx.x; i < end; i += 256) {
        float a0 = p[i] - mean;
        m2 += (double)(a0 * a0);
    }
    // exact correction for the float-rounded mean: sum (x-m)^2 = M2 + n*(mu-m)^2, handled by storing m itself
    double M2 = block_sum(m2, sm);
    if (threadIdx.x == 0) {
        double* o = scratch + (((long)b * G + g) * nsplit + s) * 3;
        double mu = cnt > 0 ? tot / cnt : 0.0;
        // M2 about the float mean -> about the exact slice mean
        double d = mu - (double)mean;
        o[0] = cnt;
        o[1] = mu;
        o[2] = M2 - cnt * d * d;
    }
}

__global__ void gn_finalize_kernel(const double* scratch, int nsplit, int G, int C, float eps,
                                   const float* gamma, const float* beta, float* mr, float* sc, float* sh,
                                   long sbs, const float* ss_scale, const float* ss_shift) {
    const int b = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double* s = scratch + ((long)b * G + g) * nsplit * 3;
    double n = 0, mean = 0, M2 = 0;
    for (int i = 0; i < nsplit; ++i) {
        double nb = s[3 * i], mb = s[3 * i + 1], Mb = s[3 * i + 2];
        if (nb <= 0) continue;
        double nn = n + nb, d = mb - mean;
        mean += d * nb / nn;
        M2 += Mb + d * d * n * nb / nn;
        n = nn;
    }
    double var = M2 / n;
    float meanf = (float)mean;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    mr[(long)b * sbs + g * 2] = meanf;
    mr[(long)b * sbs + g * 2 + 1] = rstd;
    const int cpg = C / G;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        float gm = gamma[c];
        float a = gm * rstd, o = beta[c] - meanf * rstd * gm;
        if (ss_scale) {   // scale-shift norm: GN(h)*(1+scale)+shift (guided_diffusion unet.py:250-254)
            float f = 1.0f + ss_scale[c];
            a *= f;
            o = o * f + ss_shift[c];
        }
        sc[(long)b * sbs + c] = a;
        sh[(long)b * sbs + c] = o;
    }
}

static int gn_nsplit(long len, int BG) {
    long s = len / 16384;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    while (s > 1 && s * BG > 8192) s >>= 1;
    return (int)s;
}

void launch_gn_stats(const float* x, long bs, int B, int C, int HW, int G, float eps, const float* gamma,
                     const float* beta, float* mr, float* sc, float* sh, long stats_bs, double* scratch,
                     hipStream_t st, const float* ss_scale, const float* ss_shift) {
    long len = (long)(C / G) * HW;
    int ns = gn_nsplit(len, B * G);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(ns, G, B), dim3(256), 0, st, x, bs, len, scratch);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((G + 63) / 64, B), dim3(64), 0, st, scratch, ns, G, C, eps,
                       gamma, beta, mr, sc, sh, stats_bs, ss_scale, ss_shift);
}

// ---------------------------------------------------------------------------
// tangent / cotangent statistics (see kernels.h).  scratch[(b*G+g)*nsplit+s][2]
template <int KIND>
__global__ __launch_bounds__(256) void gn_tstats_partial(const float* d, long d_bs, const float* x, long x_bs,
                                                         int HW, int cpg, const float* sc, const float* sh,
                                                         const float* mr, long pbs_c, long pbs_g,
                                                         double* scratch, float* tst, float* tc, long tbs, int act) {
    __shared__ double sm[4];
    const int s = blockIdx.x, nsplit = gridDim.x, g = blockIdx.y, b = blockIdx.z, G = gridDim.y;
    const long len = (long)cpg * HW;
    const float* dp = d + (long)b * d_bs + (long)g * len;
    const float* xp = x + (long)b * x_bs + (long)g * len;
    const float* scb = sc + (long)b * pbs_c + g * cpg;
    const float* shb = sh + (long)b * pbs_c + g * cpg;
    const float mean = mr[(long)b * pbs_g + 2 * g], rstd = mr[(long)b * pbs_g + 2 * g + 1];
    long per = ((len / 4 + nsplit - 1) / nsplit) * 4;
    long beg = (long)s * per, end = beg + per < len ? beg + per : len;
    if (beg > end) beg = end;
    double s1 = 0.0, s2 = 0.0;
    for (long i = beg + threadIdx.x * 4; i < end; i += 1024) {
        float4 dv = *reinterpret_cast<const float4*>(dp + i);
        float4 xv = *reinterpret_cast<const float4*>(xp + i);
        int ci = (int)(i / HW);
        float scc = scb[ci], shc = shb[ci];
        float dd[4] = {dv.x, dv.y, dv.z, dv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xh = (xx[j] - mean) * rstd;
            float z;
            if (KIND == 0) z = dd[j];
            else if (KIND == 1) z = (scc / rstd) * act_der(fmaf(scc, xx[j], shc), act) * dd[j];
            else z = (scc / rstd) * dd[j];
            a1 += z;
            a2 += xh * z;
        }
        s1 += (double)a1;
        s2 += (double)a2;
    }
    double t1 = block_sum(s1, sm);
    double t2 = block_sum(s2, sm);
    if (nsplit > 1) {
        if (threadIdx.x == 0) {
            double* o = scratch + (((long)b * G + g) * nsplit + s) * 2;
            o[0] = t1;
            o[1] = t2;
        }
        return;
    }
    // single slice: finalise here (saves the second launch for the small tensors)
    const double inv_n = 1.0 / (double)len;
    const float m1 = (float)(t1 * inv_n), m2 = (float)(t2 * inv_n);
    if (threadIdx.x == 0) {
        float* o = tst + (long)b * tbs + 2 * g;
        o[0] = m1;
        o[1] = m2;
    }
    if (tc && threadIdx.x < cpg) {
        float f = KIND == 0 ? 1.0f : rstd;
        float* t = tc + (long)b * tbs + 2 * ((long)g * cpg + threadIdx.x);
        t[0] = f * m1;
        t[1] = f * m2;
    }
}

__global__ void gn_tstats_finalize(const double* scratch, int nsplit, long BG, int G, double inv_n, float* tst,
                                   float* tc, long tbs, int cpg, const float* mr, long pbs_g, int kind) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BG) return;
    float* o = tst + (i / G) * tbs + 2 * (i % G);
    double a = 0, c = 0;
    for (int s = 0; s < nsplit; ++s) {
        a += scratch[(i * nsplit + s) * 2];
        c += scratch[(i * nsplit + s) * 2 + 1];
    }
    float m1 = (float)(a * inv_n), m2 = (float)(c * inv_n);
    o[0] = m1;
    o[1] = m2;
    if (tc) {
        // per-channel expansion for the split-bf16 conv staging; the cotangent form carries rstd
        int g = (int)(i % G);
        float f = kind == 0 ? 1.0f : mr[(i / G) * pbs_g + 2 * g + 1];
        float* t = tc + (i / G) * tbs + 2 * (long)g * cpg;
        for (int k = 0; k < cpg; ++k) { t[2 * k] = f * m1; t[2 * k + 1] = f * m2; }
    }
}

void launch_gn_tstats(const float* d, long d_bs, const float* x, long x_bs, int B, int C, int HW, int G,
                      const float* sc, const float* sh, const float* mr, long pbs_c, long pbs_g, int kind,
                      float* tst, float* tc, long tst_bs, double* scratch, hipStream_t st, int act) {
    int cpg = C / G;
    long len = (long)cpg * HW;
    int ns = gn_nsplit(len, B * G);
    dim3 grid(ns, G, B);
    if (kind == 0)
        hipLaunchKernelGGL(gn_tstats_partial<0>, grid, dim3(256), 0, st, d, d_bs, x, x_bs, HW, cpg, sc, sh, mr,
                           pbs_c, pbs_g, scratch, tst, tc, tst_bs, act);
    else if (kind == 1)
        hipLaunchKernelGGL(gn_tstats_partial<1>, grid, dim3(256), 0, st, d, d_bs, x, x_bs, HW, cpg, sc, sh, mr,
                           pbs_c, pbs_g, scratch, tst, tc, tst_bs, act);
    else
        hipLaunchKernelGGL(gn_tstats_partial<2>, grid, dim3(256), 0, st, d, d_bs, x, x_bs, HW, cpg, sc, sh, mr,
                           pbs_c, pbs_g, scratch, tst, tc, tst_bs, act);
    if (ns == 1) return;
    long BG = (long)B * G;
    hipLaunchKernelGGL(gn_tstats_finalize, dim3((unsigned)((BG + 255) / 256)), dim3(256), 0, st, scratch, ns, BG,
                       G, 1.0 / (double)len, tst, tc, tst_bs, cpg, mr, pbs_g, kind);
}

// ---------------------------------------------------------------------------
// Forward statistics taken in a conv epilogue (ConvArgs::st_part): part[b][c][tile] = {mean, M2} of equally sized row
// tiles.  One wave per (b, g) merges the cpg x ntile entries of its group in double (mean = average of the means,
// M2 = sum M2 + n_t sum (mean_t - mean)^2) and writes what gn_finalize_kernel writes.
__global__ __launch_bounds__(64) void gn_fused_finalize_kernel(const float* part, int ntile, int C, int G, double inv_n,
                                                               float eps, const float* gamma, const float* beta,
                                                               float* mr, float* sc, float* sh, long sbs,
                                                               const float* ss_scale, const float* ss_shift) {
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / G;
    const float2* p = reinterpret_cast<const float2*>(part) + ((long)b * C + (long)g * cpg) * ntile;
    const int cnt = cpg * ntile;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < cnt; i += 64) {
        float2 v = p[i];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    s1 = __shfl(wave_sum(s1), 0, 64);
    s2 = __shfl(wave_sum(s2), 0, 64);
    const double mean = s1 / (double)cnt, n_t = 1.0 / (inv_n * (double)cnt);
    double dev = 0.0;
    for (int i = threadIdx.x; i < cnt; i += 64) {
        const double d = (double)p[i].x - mean;
        dev += d * d;
    }
    dev = __shfl(wave_sum(dev), 0, 64);
    double var = (s2 + n_t * dev) * inv_n;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mr[(long)b * sbs + g * 2] = meanf;
        mr[(long)b * sbs + g * 2 + 1] = rstd;
    }
    for (int k = threadIdx.x; k < cpg; k += 64) {
        const int c = g * cpg + k;
        const float gm = gamma[c];
        float a = gm * rstd, o = beta[c] - meanf * rstd * gm;
        if (ss_scale) {
            const float f = 1.0f + ss_scale[c];
            a *= f;
            o = o * f + ss_shift[c];
        }
        sc[(long)b * sbs + c] = a;
        sh[(long)b * sbs + c] = o;
    }
}

// The same for a norm over the channel concatenation [A (C1 channels) | B (C - C1)] of two tensors whose producers each left
// their own {mean, M2} tile partials (the up-path ResBlocks' norm1 over torch.cat([h, skip]), diffusion.py:182-190): a group's
// channels come from one part or from both, the parts may have been written with different tile sizes (weights n_a, n_b).
__global__ __launch_bounds__(64) void gn_fused_finalize_cat_kernel(const float* partA, int C1, int ntA, const float* partB, int ntB,
                                                                   int C, int G, int HW, float eps, const float* gamma,
                                                                   const float* beta, float* mr, float* sc, float* sh, long sbs) {
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / G, C2 = C - C1;
    const double nA = (double)HW / ntA, nB = (double)HW / ntB;            // elements per tile partial
    const float2* pa = reinterpret_cast<const float2*>(partA) + (long)b * C1 * ntA;
    const float2* pb = reinterpret_cast<const float2*>(partB) + (long)b * C2 * ntB;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        const float2* p = c < C1 ? pa + (long)c * ntA : pb + (long)(c - C1) * ntB;
        const int nt = c < C1 ? ntA : ntB;
        const double w = c < C1 ? nA : nB;
        for (int i = threadIdx.x; i < nt; i += 64) { const float2 v = p[i]; s1 += w * (double)v.x; s2 += (double)v.y; }
    }
    s1 = __shfl(wave_sum(s1), 0, 64);
    s2 = __shfl(wave_sum(s2), 0, 64);
    const double n = (double)cpg * HW, mean = s1 / n;
    double dev = 0.0;
    for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        const float2* p = c < C1 ? pa + (long)c * ntA : pb + (long)(c - C1) * ntB;
        const int nt = c < C1 ? ntA : ntB;
        const double w = c < C1 ? nA : nB;
        for (int i = threadIdx.x; i < nt; i += 64) { const double d = (double)p[i].x - mean; dev += w * d * d; }
    }
    dev = __shfl(wave_sum(dev), 0, 64);
    double var = (s2 + dev) / n;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mr[(long)b * sbs + g * 2] = meanf;
        mr[(long)b * sbs + g * 2 + 1] = rstd;
    }
    for (int k = threadIdx.x; k < cpg; k += 64) {
        const int c = g * cpg + k;
        const float gm = gamma[c];
        sc[(long)b * sbs + c] = gm * rstd;
        sh[(long)b * sbs + c] = beta[c] - meanf * rstd * gm;
    }
}
// Tangent / cotangent group means from the row-tile partials of a conv epilogue (ConvArgs::st_part, kinds ST_TAN / ST_COT; round
// 6): one wave per (sample, group) sums the cpg x ntile entries of its group in double and writes what gn_tstats_partial /
// gn_tstats_finalize write.  Tangent partials are the raw {sum d, sum x d}: mean(xhat d) = rstd (sum x d - mean sum d) / n with the
// primal {mean, rstd} of the consuming norm; cotangent partials are {sum z, sum xhat z} already.  Channels >= C1 of a
// concatenation come from the second producer's partials (tangent kind: raw sums do not depend on the norm).
__global__ __launch_bounds__(256) void gn_lin_fused_finalize_kernel(int kind, const float* partA, int C1, int ntA,
                                                                    const float* partB, int ntB, int C, int G, double inv_n,
                                                                    const float* mr, float* tst, float* tc, long tbs) {
    __shared__ double sm[4];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / G, C2 = C - C1;
    const float2* pa = reinterpret_cast<const float2*>(partA) + (long)b * C1 * ntA;
    const float2* pb = partB ? reinterpret_cast<const float2*>(partB) + (long)b * C2 * ntB : nullptr;
    // the group's entries are contiguous inside each producer's buffer ([c][tile], channels of a group adjacent): one or two runs
    const int c0 = g * cpg, c1 = c0 + cpg;
    const int a0 = c0 < C1 ? c0 : C1, a1 = c1 < C1 ? c1 : C1;                 // channels [a0, a1) from A, [b0, b1) from B
    const int b0 = (c0 > C1 ? c0 : C1) - C1, b1 = (c1 > C1 ? c1 : C1) - C1;
    const float2* ra = pa + (long)a0 * ntA; const int na = (a1 - a0) * ntA;
    const float2* rb = pb ? pb + (long)b0 * ntB : nullptr; const int nb = pb ? (b1 - b0) * ntB : 0;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < na; i += 1024) {       // four independent loads in flight per thread
        const float2 z = {0.f, 0.f};
        const float2 v0 = ra[i], v1 = i + 256 < na ? ra[i + 256] : z, v2 = i + 512 < na ? ra[i + 512] : z, v3 = i + 768 < na ? ra[i + 768] : z;
        s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (int i = threadIdx.x; i < nb; i += 1024) {
        const float2 z = {0.f, 0.f};
        const float2 v0 = rb[i], v1 = i + 256 < nb ? rb[i + 256] : z, v2 = i + 512 < nb ? rb[i + 512] : z, v3 = i + 768 < nb ? rb[i + 768] : z;
        s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    s1 = block_sum(s1, sm);
    s2 = block_sum(s2, sm);
    const float mean = mr[2 * g], rstd = mr[2 * g + 1];
    float m1, m2;
    if (kind == ST_TAN) {
        m1 = (float)(s1 * inv_n);
        m2 = (float)((double)rstd * (s2 - (double)mean * s1) * inv_n);
    } else {       // cotangent partials are sums of S g and xhat S g: z = S g / rstd
        m1 = (float)(s1 * inv_n / (double)rstd);
        m2 = (float)(s2 * inv_n / (double)rstd);
    }
    if (threadIdx.x == 0) {
        float* o = tst + (long)b * tbs + 2 * g;
        o[0] = m1;
        o[1] = m2;
    }
    if (tc) {      // per-channel expansion for the split-bf16 conv staging; the cotangent form carries rstd
        const float f = kind == ST_TAN ? 1.0f : rstd;
        for (int k = threadIdx.x; k < cpg; k += 256) {
            float* t = tc + (long)b * tbs + 2 * ((long)g * cpg + k);
            t[0] = f * m1;
            t[1] = f * m2;
        }
    }
}
void launch_gn_lin_fused_finalize(int kind, const float* partA, int C1, int ntA, const float* partB, int ntB, int B, int C, int HW,
                                  int G, const float* mr, float* tst, float* tc, long tst_bs, hipStream_t st) {
    hipLaunchKernelGGL(gn_lin_fused_finalize_kernel, dim3(G, B), dim3(256), 0, st, kind, partA, C1, ntA, partB, ntB, C, G,
                       1.0 / ((double)(C / G) * HW), mr, tst, tc, tst_bs);
}
void launch_gn_fused_finalize_cat(const float* partA, int C1, int ntA, const float* partB, int ntB, int B, int C, int HW, int G,
                                  float eps, const float* gamma, const float* beta, float* mr, float* sc, float* sh,
                                  long stats_bs, hipStream_t st) {
    hipLaunchKernelGGL(gn_fused_finalize_cat_kernel, dim3(G, B), dim3(64), 0, st, partA, C1, ntA, partB, ntB, C, G, HW, eps,
                       gamma, beta, mr, sc, sh, stats_bs);
}

void launch_gn_fused_finalize(const float* part, int ntile, int B, int C, int HW, int G, float eps, const float* gamma,
                              const float* beta, float* mr, float* sc, float* sh, long stats_bs, const float* ss_scale,
                              const float* ss_shift, hipStream_t st) {
    const double inv_n = 1.0 / ((double)(C / G) * HW);
    hipLaunchKernelGGL(gn_fused_finalize_kernel, dim3(G, B), dim3(64), 0, st, part, ntile, C, G, inv_n, eps, gamma, beta, mr,
                       sc, sh, stats_bs, ss_scale, ss_shift);
}

// ---------------------------------------------------------------------------
// Split-K epilogue + statistics.  grid = (ns, G, B): a block sums the K-slabs of its slice of one (sample, group), applies
// bias / bias2 / residual / accumulate, writes the tensor and takes the statistics of what it wrote.  One slice (the small
// tensors split-K exists for): finalised here; several: slice sums to `scratch`, finalised by the kernels the standalone
// statistics use (gn_finalize_kernel from {n, mean, M2}, gn_tstats_finalize from {sum z, sum xhat z}).
template <int KIND>
__global__ __launch_bounds__(1024) void splitk_reduce_stats_kernel(ConvArgs a, int G, float eps, const float* gamma,
                                                                  const float* beta, float* mr, float* sc, float* sh, long sbs,
                                                                  const float* ss_scale, const float* ss_shift,
                                                                  const float* prim, const float* sc_p, const float* sh_p,
                                                                  const float* mr_p, float* tst, float* tc, long tbs,
                                                                  double* scratch) {
    __shared__ double sm[16];
    const int s = blockIdx.x, nsl = gridDim.x, g = blockIdx.y, b = blockIdx.z;
    const int HW = a.Hout * a.Wout, cpg = a.Cout / G;
    const long step = (long)blockDim.x * 4;
    const long len = (long)cpg * HW, per_b = (long)a.Cout * HW, total = per_b * a.B;
    const long base = (long)g * len;                                    // offset of the group inside one sample
    long per = ((len / 4 + nsl - 1) / nsl) * 4;
    long beg = (long)s * per, end = beg + per < len ? beg + per : len;
    if (beg > end) beg = end;
    float mean_p = 0.f, rstd_p = 1.f;
    if (KIND != ST_FWD) { mean_p = mr_p[2 * g]; rstd_p = mr_p[2 * g + 1]; }
    double t1 = 0.0, t2 = 0.0;
    for (long i = beg + threadIdx.x * 4; i < end; i += step) {
        const long e = base + i;                                        // element inside the sample
        const int c = (int)(e / HW);
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        const float* pp = a.partial + (long)b * per_b + e;
        int k = 0;
        for (; k + 4 <= a.nsplit; k += 4) {                             // four slab loads in flight per step
            const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(pp + (long)k * total);
            const f32x4_t p1 = *reinterpret_cast<const f32x4_t*>(pp + (long)(k + 1) * total);
            const f32x4_t p2 = *reinterpret_cast<const f32x4_t*>(pp + (long)(k + 2) * total);
            const f32x4_t p3 = *reinterpret_cast<const f32x4_t*>(pp + (long)(k + 3) * total);
            v += (p0 + p1) + (p2 + p3);
        }
        for (; k < a.nsplit; ++k) v += *reinterpret_cast<const f32x4_t*>(pp + (long)k * total);
        float add = 0.f;
        if (a.bias) add += a.bias[c];
        if (a.bias2) add += a.bias2[(long)b * a.bias2_bs + c];
        v += add;
        if (a.res) v += a.res_scale * *reinterpret_cast<const f32x4_t*>(a.res + (long)b * a.res_bs + e);
        float* o = a.out + (long)b * a.out_bs + e;
        if (a.accumulate) v += *reinterpret_cast<const f32x4_t*>(o);
        *reinterpret_cast<f32x4_t*>(o) = v;
        float a1, a2;
        if (KIND == ST_FWD) {
            a1 = (v[0] + v[1]) + (v[2] + v[3]);
            a2 = 0.f;                                  // M2 in a second pass about the slice mean (below)
        } else {
            const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(prim + e);
            const float scc = sc_p[c], shc = sh_p[c];
            a1 = 0.f; a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - mean_p) * rstd_p;
                float z = v[j];
                if (KIND == ST_COT) z *= (scc / rstd_p) * act_der(fmaf(scc, xv[j], shc), a.act);
                a1 += z;
                a2 += xh * z;
            }
        }
        t1 += (double)a1;
        t2 += (double)a2;
    }
    t1 = block_sum(t1, sm);
    if (KIND == ST_FWD) {
        // two-pass variance as the standalone kernel: the slice was just written by these same threads, read it back
        const double n = (double)(end - beg);
        const float mu = n > 0 ? (float)(t1 / n) : 0.f;
        double m2 = 0.0;
        for (long i = beg + threadIdx.x * 4; i < end; i += step) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(a.out + (long)b * a.out_bs + base + i);
            const float d0 = v[0] - mu, d1 = v[1] - mu, d2 = v[2] - mu, d3 = v[3] - mu;
            m2 += (double)((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        }
        const double dm = (n > 0 ? t1 / n : 0.0) - (double)mu;
        t2 = block_sum(m2, sm) - n * dm * dm;          // M2 about the exact slice mean
    } else {
        t2 = block_sum(t2, sm);
    }
    if (nsl > 1) {
        if (threadIdx.x == 0) {
            if (KIND == ST_FWD) {            // {n, mean, M2} of the slice for gn_finalize_kernel's Chan merge
                double* o = scratch + (((long)b * G + g) * nsl + s) * 3;
                const double n = (double)(end - beg), mu = n > 0 ? t1 / n : 0.0;
                o[0] = n; o[1] = mu; o[2] = t2;
            } else {
                double* o = scratch + (((long)b * G + g) * nsl + s) * 2;
                o[0] = t1; o[1] = t2;
            }
        }
        return;
    }
    const double inv_n = 1.0 / (double)len;
    if (KIND == ST_FWD) {
        const double mean = t1 * inv_n;
        double var = t2 * inv_n;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (threadIdx.x == 0) {
            mr[(long)b * sbs + g * 2] = meanf;
            mr[(long)b * sbs + g * 2 + 1] = rstd;
        }
        if (threadIdx.x < cpg) {
            const int c = g * cpg + threadIdx.x;
            const float gm = gamma[c];
            float aa = gm * rstd, oo = beta[c] - meanf * rstd * gm;
            if (ss_scale) {
                const float f = 1.0f + ss_scale[c];
                aa *= f;
                oo = oo * f + ss_shift[c];
            }
            sc[(long)b * sbs + c] = aa;
            sh[(long)b * sbs + c] = oo;
        }
    } else {
        const float m1 = (float)(t1 * inv_n), m2 = (float)(t2 * inv_n);
        if (threadIdx.x == 0) {
            tst[(long)b * tbs + 2 * g] = m1;
            tst[(long)b * tbs + 2 * g + 1] = m2;
        }
        if (tc && threadIdx.x < cpg) {
            const float f = KIND == ST_TAN ? 1.0f : rstd_p;
            float* t = tc + (long)b * tbs + 2 * ((long)g * cpg + threadIdx.x);
            t[0] = f * m1;
            t[1] = f * m2;
        }
    }
}

void launch_conv_splitk_reduce_stats(const ConvArgs& a, int kind, int G, float eps, const float* gamma, const float* beta,
                                     float* mr, float* sc, float* sh, long stats_bs, const float* ss_scale,
                                     const float* ss_shift, const float* prim, const float* sc_prim, const float* sh_prim,
                                     const float* mr_prim, float* tst, float* tc, long tst_bs, double* scratch,
                                     hipStream_t st) {
    const int HW = a.Hout * a.Wout, cpg = a.Cout / G;
    const long len = (long)cpg * HW;
    const int ns = gn_nsplit(len, a.B * G);
    dim3 grid(ns, G, a.B);
    // threads: one 16-byte piece each where the slice is small (the 8x8 .. 32x32 tensors split-K exists for), 1024 at most
    long pieces = (len / ns + 3) / 4;
    int nthr = (int)((pieces + 63) / 64) * 64;
    if (nthr < 64) nthr = 64;
    if (nthr > 1024) nthr = 1024;
    if (nthr < cpg) nthr = ((cpg + 63) / 64) * 64;
#define RS(K) hipLaunchKernelGGL(splitk_reduce_stats_kernel<K>, grid, dim3(nthr), 0, st, a, G, eps, gamma, beta, mr, sc, sh, \
                                 stats_bs, ss_scale, ss_shift, prim, sc_prim, sh_prim, mr_prim, tst, tc, tst_bs, scratch)
    if (kind == ST_FWD) RS(ST_FWD); else if (kind == ST_TAN) RS(ST_TAN); else RS(ST_COT);
#undef RS
    if (ns == 1) return;
    if (kind == ST_FWD) {
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((G + 63) / 64, a.B), dim3(64), 0, st, scratch, ns, G, a.Cout, eps, gamma,
                           beta, mr, sc, sh, stats_bs, ss_scale, ss_shift);
    } else {
        const long BG = (long)a.B * G;
        hipLaunchKernelGGL(gn_tstats_finalize, dim3((unsigned)((BG + 255) / 256)), dim3(256), 0, st, scratch, ns, BG, G,
                           1.0 / (double)len, tst, tc, tst_bs, cpg, mr_prim, 0L, kind == ST_TAN ? 0 : 1);
    }
}

// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* d, long d_bs, const float* x, long x_bs,
                                                       const float* base, long base_bs, float* out, long out_bs,
                                                       int accumulate, int C, int HW, int cpg, const float* sc,
                                                       const float* sh, const float* mr, long pbs_c, long pbs_g,
                                                       const float* tst, long tbs, int act, float base_scale) {
    const int b = blockIdx.y;
    const long per = (long)C * HW;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per; i += (long)gridDim.x * 1024) {
        int c = (int)(i / HW);
        int g = c / cpg;
        float scc = sc[(long)b * pbs_c + c], shc = sh[(long)b * pbs_c + c];
        float4 xv = *reinterpret_cast<const float4*>(x + (long)b * x_bs + i);
        float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float r[4];
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = fmaf(scc, xx[j], shc);
        } else if (KIND == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = act_fwd(fmaf(scc, xx[j], shc), act);
        } else {
            float mean = mr[(long)b * pbs_g + 2 * g], rstd = mr[(long)b * pbs_g + 2 * g + 1];
            float m1 = tst[(long)b * tbs + g * 2], m2 = tst[(long)b * tbs + g * 2 + 1];
            float4 dv = *reinterpret_cast<const float4*>(d + (long)b * d_bs + i);
            float dd[4] = {dv.x, dv.y, dv.z, dv.w};
            float bb[4] = {0.f, 0.f, 0.f, 0.f};
            if (KIND >= 2 && base) {
                float4 bv = *reinterpret_cast<const float4*>(base + (long)b * base_bs + i);
                bb[0] = base_scale * bv.x; bb[1] = base_scale * bv.y; bb[2] = base_scale * bv.z; bb[3] = base_scale * bv.w;
            }
            float gm = scc / rstd;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xh = (xx[j] - mean) * rstd;
                if (KIND == 1) r[j] = scc * (dd[j] - m1 - xh * m2);
                else if (KIND == 5) r[j] = act_der(fmaf(scc, xx[j], shc), act) * scc * (dd[j] - m1 - xh * m2);
                else if (KIND == 2) r[j] = bb[j] + rstd * (gm * act_der(fmaf(scc, xx[j], shc), act) * dd[j] - m1 - xh * m2);
                else r[j] = bb[j] + rstd * (gm * dd[j] - m1 - xh * m2);
            }
        }
        float4* o = reinterpret_cast<float4*>(out + (long)b * out_bs + i);
        if (accumulate) {
            float4 ov = *o;
            r[0] += ov.x; r[1] += ov.y; r[2] += ov.z; r[3] += ov.w;
        }
        *o = make_float4(r[0], r[1], r[2], r[3]);
    }
}

void launch_gn_apply(int kind, const float* d, long d_bs, const float* x, long x_bs, const float* base,
                     long base_bs, float* out, long out_bs, int accumulate, int B, int C, int HW, int G,
                     const float* sc, const float* sh, const float* mr, long pbs_c, long pbs_g, const float* tst,
                     long tst_bs, hipStream_t st, int act, float base_scale) {
    long per = (long)C * HW;
    int blocks = (int)((per / 4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    dim3 grid(blocks, B);
    int cpg = C / G;
#define GA(K) hipLaunchKernelGGL(gn_apply_kernel<K>, grid, dim3(256), 0, st, d, d_bs, x, x_bs, base, base_bs, out, \
                                 out_bs, accumulate, C, HW, cpg, sc, sh, mr, pbs_c, pbs_g, tst, tst_bs, act, base_scale)
    switch (kind) {
        case 0: GA(0); break;
        case 1: GA(1); break;
        case 2: GA(2); break;
        case 4: GA(4); break;
        case 5: GA(5); break;
        default: GA(3); break;
    }
#undef GA
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_cache_kernel(const float* x, int C, int HW, int cpg, const float* sc,
                                                       const float* sh, const float* mr, float2* sx, int act) {
    const long per = (long)C * HW;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per; i += (long)gridDim.x * 1024) {
        int c = (int)(i / HW);
        int g = c / cpg;
        float scc = sc[c], shc = sh[c], mean = mr[2 * g], rstd = mr[2 * g + 1];
        float4 xv = *reinterpret_cast<const float4*>(x + i);
        float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float4 o0, o1;
        float r[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[2 * j] = scc * act_der(fmaf(scc, xx[j], shc), act);
            r[2 * j + 1] = (xx[j] - mean) * rstd;
        }
        o0 = make_float4(r[0], r[1], r[2], r[3]);
        o1 = make_float4(r[4], r[5], r[6], r[7]);
        float4* o = reinterpret_cast<float4*>(sx + i);
        o[0] = o0;
        o[1] = o1;
    }
}
void launch_gn_cache(const float* x, int C, int HW, int cpg, const float* sc, const float* sh, const float* mr,
                     float2* sx, hipStream_t st, int act) {
    long per = (long)C * HW;
    int blocks = (int)((per / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gn_cache_kernel, dim3(blocks), dim3(256), 0, st, x, C, HW, cpg, sc, sh, mr, sx, act);
}

// ---------------------------------------------------------------------------
// GroupNorm of token-major encoder states (kernels.h launch_ctx_groupnorm): one workgroup per group, two passes in double
__global__ __launch_bounds__(256) void ctx_groupnorm_kernel(const float* tok, int L, int D, int cpg, float eps,
                                                            const float* gamma, const float* beta, float* out) {
    __shared__ double sm[4];
    const int g = blockIdx.x, n = L * cpg;
    double s1 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s1 += (double)tok[(long)(i / cpg) * D + g * cpg + (i % cpg)];
    const double mean = block_sum(s1, sm) / (double)n;
    __syncthreads();
    double s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double d = (double)tok[(long)(i / cpg) * D + g * cpg + (i % cpg)] - mean;
        s2 += d * d;
    }
    const double var = block_sum(s2, sm) / (double)n;
    const float mf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = g * cpg + (i % cpg);
        const long o = (long)(i / cpg) * D + c;
        out[o] = (tok[o] - mf) * rstd * gamma[c] + beta[c];
    }
}
void launch_ctx_groupnorm(const float* tok, int L, int D, int G, float eps, const float* gamma, const float* beta,
                          float* out, hipStream_t st) {
    hipLaunchKernelGGL(ctx_groupnorm_kernel, dim3(G), dim3(256), 0, st, tok, L, D, D / G, eps, gamma, beta, out);
}

// ---------------------------------------------------------------------------
// softmax over rows of length T (multiple of 64): one wave per row; rows up to 1024 stay in registers, longer ones
// (the decoder's 64x64 = 4096-token mid attention) are streamed three times through the cache
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* S, long rows, int T, long bs) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = S + (long)blockIdx.y * bs + row * T;
    float v[16];
    const int per = T / 64;
    float mx = -INFINITY;
    if (per > 16) {
        for (int i = 0; i < per; ++i) mx = fmaxf(mx, p[lane + i * 64]);
        mx = wave_maxf(mx);
        mx = __shfl(mx, 0, 64);
        float sum = 0.f;
        for (int i = 0; i < per; ++i) sum += expf(p[lane + i * 64] - mx);
        sum = wave_sumf(sum);
        sum = __shfl(sum, 0, 64);
        const float inv = 1.0f / sum;
        for (int i = 0; i < per; ++i) p[lane + i * 64] = expf(p[lane + i * 64] - mx) * inv;
        return;
    }
    for (int i = 0; i < per; ++i) {
        v[i] = p[lane + i * 64];
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_maxf(mx);
    mx = __shfl(mx, 0, 64);
    float sum = 0.f;
    for (int i = 0; i < per; ++i) {
        v[i] = expf(v[i] - mx);
        sum += v[i];
    }
    sum = wave_sumf(sum);
    sum = __shfl(sum, 0, 64);
    float inv = 1.0f / sum;
    for (int i = 0; i < per; ++i) p[lane + i * 64] = v[i] * inv;
}
void launch_softmax_rows(float* S, long rows, int T, hipStream_t st, int B, long bs) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4), B), dim3(256), 0, st, S, rows, T, bs);
}

__global__ __launch_bounds__(256) void softmax_jac_kernel(float* dS, const float* P, long rows, int T,
                                                          long p_rows, float scale, long bs) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* d = dS + (long)blockIdx.y * bs + row * T;
    const float* p = P + (row % p_rows) * T;
    const int per = T / 64;
    float dv[16], pv[16];
    float dot = 0.f;
    if (per > 16) {
        for (int i = 0; i < per; ++i) dot += d[lane + i * 64] * p[lane + i * 64];
        dot = wave_sumf(dot);
        dot = __shfl(dot, 0, 64);
        for (int i = 0; i < per; ++i) d[lane + i * 64] = scale * p[lane + i * 64] * (d[lane + i * 64] - dot);
        return;
    }
    for (int i = 0; i < per; ++i) {
        dv[i] = d[lane + i * 64];
        pv[i] = p[lane + i * 64];
        dot += dv[i] * pv[i];
    }
    dot = wave_sumf(dot);
    dot = __shfl(dot, 0, 64);
    for (int i = 0; i < per; ++i) d[lane + i * 64] = scale * pv[i] * (dv[i] - dot);
}
void launch_softmax_jac(float* dS, const float* P, long rows, int T, long p_rows, float scale, hipStream_t st,
                        int B, long bs) {
    hipLaunchKernelGGL(softmax_jac_kernel, dim3((unsigned)((rows + 3) / 4), B), dim3(256), 0, st, dS, P, rows, T,
                       p_rows, scale, bs);
}

// ---------------------------------------------------------------------------
// time embedding: [sin, cos] sinusoid (divisor half-1) -> dense0 -> swish -> dense1 -> swish
// (reference diffusion.py:783-804, 154-157, and the nonlinearity(temb) of :899)
__global__ void temb_kernel(float t, int ch, int temb_ch, const float* freq, const float* w0, const float* b0,
                            const float* w1, const float* b1, float* out, int cos_first, const float* add,
                            const float* t_ptr, int act) {
    extern __shared__ float sm[];
    if (t_ptr) t = *t_ptr;     // timestep from device memory: the launch is a node of a replayed HIP graph
    float* emb = sm;           // [ch]
    float* h = sm + ch;        // [temb_ch]
    const int half = ch / 2;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float a = t * freq[i];   // freq table built on the host exactly as torch does (engine.hip)
        // [sin, cos] (Ho-DDPM, diffusion.py:800) or [cos, sin] (guided_diffusion nn.py:118)
        emb[cos_first ? half + i : i] = sinf(a);
        emb[cos_first ? i : half + i] = cosf(a);
    }
    if ((ch & 1) && threadIdx.x == 0) emb[ch - 1] = 0.f;
    __syncthreads();
    for (int o = threadIdx.x; o < temb_ch; o += blockDim.x) {
        float acc = b0[o];
        for (int i = 0; i < ch; ++i) acc = fmaf(w0[(long)o * ch + i], emb[i], acc);
        h[o] = act_fwd(acc, act);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < temb_ch; o += blockDim.x) {
        float acc = b1[o];
        for (int i = 0; i < temb_ch; ++i) acc = fmaf(w1[(long)o * temb_ch + i], h[i], acc);
        if (add) acc += add[o];          // conditioning embedding (class / pooled text), added to emb before the SiLU
        out[o] = act_fwd(acc, act);
    }
}
void launch_temb(float t, int ch, int temb_ch, const float* freq, const float* w0, const float* b0, const float* w1,
                 const float* b1, float* scratch, hipStream_t st, int cos_first, const float* add,
                 const float* t_ptr, int act) {
    hipLaunchKernelGGL(temb_kernel, dim3(1), dim3(512), (ch + temb_ch) * sizeof(float), st, t, ch, temb_ch, freq,
                       w0, b0, w1, b1, scratch, cos_first, add, t_ptr, act);
}
__global__ void set_scalar_kernel(float* p, float v) { *p = v; }
// {shader-clock counter, 100 MHz wall counter} of the CU this one-lane launch lands on: two of these around a timed
// region give the average shader clock the chip held over it (MI355X guide, DVFS give-back item 6), so a kernel time
// taken on one box can be compared with another box's at the clock each of them ran at
__global__ void clock_stamp_kernel(unsigned long long* out) {
    out[0] = __builtin_amdgcn_s_memtime();
    out[1] = __builtin_amdgcn_s_memrealtime();
}
void launch_clock_stamp(unsigned long long* out2, hipStream_t st) {
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, st, out2);
}
void launch_set_scalar(float* p, float v, hipStream_t st) {
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, p, v);
}
// out[o] = b[o] + sum_i w[o][i]*tact[i]; one wave per output row
__global__ __launch_bounds__(256) void temb_proj_kernel(const float* tact, int temb_ch, const float* w,
                                                        const float* b, int cout, float* out) {
    int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= cout) return;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int i = lane; i < temb_ch; i += 64) acc = fmaf(w[(long)o * temb_ch + i], tact[i], acc);
    acc = wave_sumf(acc);
    if (lane == 0) out[o] = acc + b[o];
}
void launch_temb_proj(const float* tact, int temb_ch, const float* w, const float* b, int cout, float* out,
                      hipStream_t st) {
    hipLaunchKernelGGL(temb_proj_kernel, dim3((cout + 3) / 4), dim3(256), 0, st, tact, temb_ch, w, b, cout, out);
}

// ---------------------------------------------------------------------------
__global__ void pool2x2_kernel(const float* in, long in_bs, float* out, long out_bs, int accumulate, int C,
                               int Ho, int Wo, float scale) {
    const int b = blockIdx.y;
    const long per = (long)C * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        int x = (int)(i % Wo);
        long r = i / Wo;
        int y = (int)(r % Ho);
        long c = r / Ho;
        const float* p = in + (long)b * in_bs + (c * 2 * Ho + 2 * y) * (2 * Wo) + 2 * x;
        float2 a = *reinterpret_cast<const float2*>(p);
        float2 d = *reinterpret_cast<const float2*>(p + 2 * Wo);
        float v = ((a.x + a.y) + (d.x + d.y)) * scale;
        float* o = out + (long)b * out_bs + i;
        if (accumulate) v += *o;
        *o = v;
    }
}
void launch_pool2x2_sum(const float* in, long in_bs, float* out, long out_bs, int accumulate, int B, int C,
                        int Hout, int Wout, hipStream_t st, float scale) {
    long per = (long)C * Hout * Wout;
    int blocks = (int)((per + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pool2x2_kernel, dim3(blocks, B), dim3(256), 0, st, in, in_bs, out, out_bs, accumulate, C,
                       Hout, Wout, scale);
}

// out[b][c][y][x] (+)= scale * in[b][c][y/2][x/2]   (nearest x2; adjoint of the 2x2 average with scale 0.25)
__global__ void upsample2x_kernel(const float* in, long in_bs, float* out, long out_bs, int accumulate, int C,
                                  int Hi, int Wi, float scale) {
    const int b = blockIdx.y;
    const int Wo = 2 * Wi, Ho = 2 * Hi;
    const long per = (long)C * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        int x = (int)(i % Wo);
        long r = i / Wo;
        int y = (int)(r % Ho);
        long c = r / Ho;
        float v = scale * in[(long)b * in_bs + (c * Hi + (y >> 1)) * Wi + (x >> 1)];
        float* o = out + (long)b * out_bs + i;
        if (accumulate) v += *o;
        *o = v;
    }
}
void launch_upsample2x(const float* in, long in_bs, float* out, long out_bs, int accumulate, float scale, int B,
                       int C, int Hin, int Win, hipStream_t st) {
    long per = (long)C * Hin * Win * 4;
    int blocks = (int)((per + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks, B), dim3(256), 0, st, in, in_bs, out, out_bs, accumulate, C,
                       Hin, Win, scale);
}

__global__ void copy_kernel(const float* in, long in_bs, float* out, long out_bs, int accumulate, long per) {
    const int b = blockIdx.y;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < per; i += (long)gridDim.x * blockDim.x * 4) {
        float4 v = *reinterpret_cast<const float4*>(in + (long)b * in_bs + i);
        float4* o = reinterpret_cast<float4*>(out + (long)b * out_bs + i);
        if (accumulate) {
            float4 w = *o;
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *o = v;
    }
}
void launch_copy(const float* in, long in_bs, float* out, long out_bs, int accumulate, int B, long per_sample,
                 hipStream_t st) {
    int blocks = (int)((per_sample / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(copy_kernel, dim3(blocks, B), dim3(256), 0, st, in, in_bs, out, out_bs, accumulate,
                       per_sample);
}

// ---------------------------------------------------------------------------
// DDIM update, op order of reference utils.py:362-374:
//   P = (x - e*c_x0_e)/c_x0_x ; next = c_next_x0*P + c_next_e*e (+ c_noise*noise)
__global__ void ddim_step_kernel(const float* x, const float* eps, const float* noise, float* out, float* x0_out,
                                 long count, float sq1mat, float sqat, float sqatn, float ce, float cn) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        float e = eps[i];
        float p = (x[i] - e * sq1mat) / sqat;
        float v = sqatn * p + ce * e;
        if (noise) v += cn * noise[i];
        if (x0_out) x0_out[i] = p;
        out[i] = v;
    }
}
void launch_ddim_step(const float* x, const float* eps, const float* noise, float* out, float* x0_out, long count,
                      float c_x0_x, float c_x0_e, float c_next_x0, float c_next_e, float c_noise, hipStream_t st) {
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(blocks), dim3(256), 0, st, x, eps, noise, out, x0_out, count, c_x0_e, c_x0_x,
                       c_next_x0, c_next_e, c_noise);
}

__global__ void masked_axpby_kernel(const float* V, const float* dE, const uint8_t* mask, float cv, float ce,
                                    float* U, long n, long total, const uint8_t* mask2, long split) {
    // rows >= split use mask2 (two subspace solves with different masks sharing one probe batch)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long row = i / n, j = i - row * n;
        const uint8_t* mk = (mask2 && row >= split) ? mask2 : mask;
        // a select, not a product with 0: a NaN / Inf outside the mask must not reach the Gram / eig of the whole block
        const bool pass = !mk || mk[j];
        U[i] = pass ? ((V ? cv * V[i] : 0.f) + ce * dE[i]) : 0.f;   // V == nullptr: output and input sizes differ (raw network Jacobian)
    }
}
void launch_masked_axpby(const float* V, const float* dE, const uint8_t* mask, float cv, float ce, float* U, int k,
                         long n, hipStream_t st, const uint8_t* mask2, long split) {
    long total = (long)k * n;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(masked_axpby_kernel, dim3(blocks), dim3(256), 0, st, V, dE, mask, cv, ce, U, n, total, mask2, split);
}

// z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) over moments [B][2 Z][HW] (mean | logvar), noise / z [B][Z][HW]:
// DiagonalGaussianDistribution.sample() behind `vae.encode(x0).latent_dist.sample() * 0.18215` (reference edit.py:594-597)
__global__ void latent_sample_kernel(const float* mom, const float* noise, float scale, float* z, long zhw, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / zhw, j = i - b * zhw;
        const float mean = mom[b * 2 * zhw + j];
        const float lv = fminf(fmaxf(mom[b * 2 * zhw + zhw + j], -30.f), 20.f);
        z[i] = scale * (mean + expf(0.5f * lv) * (noise ? noise[i] : 0.f));
    }
}
void launch_latent_sample(const float* mom, const float* noise, float scale, float* z, int B, long zhw, hipStream_t st) {
    const long total = (long)B * zhw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(latent_sample_kernel, dim3(blocks), dim3(256), 0, st, mom, noise, scale, z, zhw, total);
}

__global__ void cot_seed_kernel(const float* U, const uint8_t* mask, float cv, float ce, float* gE, float* gX0,
                                long n, long total, const uint8_t* mask2, long split) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long row = i / n, j = i - row * n;
        const uint8_t* mk = (mask2 && row >= split) ? mask2 : mask;
        float u = (!mk || mk[j]) ? U[i] : 0.f;
        gE[i] = ce * u;
        if (gX0) gX0[i] = cv * u;
    }
}
void launch_cot_seed(const float* U, const uint8_t* mask, float cv, float ce, float* gE, float* gX0, int k, long n,
                     hipStream_t st, const uint8_t* mask2, long split) {
    long total = (long)k * n;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cot_seed_kernel, dim3(blocks), dim3(256), 0, st, U, mask, cv, ce, gE, gX0, n, total, mask2, split);
}

__global__ void fill_random_kernel(float* p, long count, unsigned seed, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        p[i] = scale * ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f);
    }
}
void launch_fill_random(float* p, long count, unsigned seed, float scale, hipStream_t st) {
    hipLaunchKernelGGL(fill_random_kernel, dim3(2048), dim3(256), 0, st, p, count, seed, scale);
}

// out = sum_i coef[i] * src[i] (n <= 4 terms; src[i] may alias out)
struct LinComb { const float* src[4]; float coef[4]; int n; };
__global__ void lincomb_kernel(LinComb lc, float* out, long count) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += (long)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= count) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < lc.n; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(lc.src[k] + i);
                acc.x = fmaf(lc.coef[k], v.x, acc.x); acc.y = fmaf(lc.coef[k], v.y, acc.y);
                acc.z = fmaf(lc.coef[k], v.z, acc.z); acc.w = fmaf(lc.coef[k], v.w, acc.w);
            }
            *reinterpret_cast<float4*>(out + i) = acc;
        } else {
            for (long j = i; j < count; ++j) {
                float acc = 0.f;
                for (int k = 0; k < lc.n; ++k) acc = fmaf(lc.coef[k], lc.src[k][j], acc);
                out[j] = acc;
            }
        }
    }
}
void launch_lincomb(const float* const* src, const float* coef, int n, float* out, long count, hipStream_t st) {
    LinComb lc;
    lc.n = n;
    for (int k = 0; k < 4; ++k) { lc.src[k] = k < n ? src[k] : nullptr; lc.coef[k] = k < n ? coef[k] : 0.f; }
    long blocks = (count / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)blocks), dim3(256), 0, st, lc, out, count);
}

__global__ void add_kernel(const float* a, const float* b, float* out, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}
void launch_add(const float* a, const float* b, float* out, long count, hipStream_t st) {
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, st, a, b, out, count);
}

}  // namespace loco
