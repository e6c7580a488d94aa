// Fused self-attention of the AttnBlock (reference models/ddpm/diffusion.py:948-962 torch.bmm + softmax;
// guided_diffusion/unet.py:330-356 QKVAttentionLegacy) and its tangent / cotangent forms for the PMP-Jacobian passes.
//
// One workgroup owns ROWS = 16 query tokens of one (image / probe, head) and runs, without leaving the CU,
//   phase 1   S-type product(s) on v_mfma_f32_16x16x4_f32:   S[i][j] = sum_c X[c][i] Y[c][j]        (K = head channels)
//   phase 2   the row operation on the score tile held in LDS: softmax, or the softmax Jacobian product
//   phase 3   O-type product(s):                               O[c][i] = sum_j Z[c][j] W[i][j]        (K = tokens)
// so the [T x T] score / probability tangents never travel to HBM (only the primal P, which the tangent and cotangent
// passes re-read, and the cotangent g_S, which the column-block kernel below needs, are stored).  q, k, v stay in their
// [channel][token] conv layout; no transposes are materialised.  Arithmetic is exact fp32 (f32-input MFMA): attention
// is 0.14 % of the FLOPs, what it cost before was launches (5-6 generic GEMM / softmax kernels per block and pass).
//
//   forward    S = scale q^T k;  P = softmax_j S (stored);  o = v P^T
//   tangent    dS = dq^T k + q^T dk;  dP = scale P (dS - rowsum(P dS));  do = dv P^T + v dP^T
//   cotangent  g_P = g_o^T v;  g_S = scale P (g_P - rowsum(P g_P)) (stored);  g_q = k g_S^T      [row blocks]
//              g_k = q g_S,  g_v = g_o P                                                          [column blocks]
#include "kernels.h"

namespace loco {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ROWS = 16;      // query rows (phase 1/2) or key columns (column-block kernel) per workgroup
constexpr int KC = 32;        // head channels per phase-1 staging chunk
constexpr int CG = 64;        // output channels per phase-3 group (one 16-channel tile per wave)

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// LDS layout (floats): xs [KC][16] | big: phase 1 ys [KC][T + 16], phase 3 zs [CG][T + 2] | ps [16][T + 2] | ds [16][T + 2].
// Ys rows are padded to T + 16 (k rows 0,1 of a ds_read_b32 half-wave land in disjoint bank halves), the score tiles and
// the Z chunk to T + 2 (16 rows x 2 k-columns hit 32 distinct banks).
template <int FORM>   // 0 forward, 1 tangent, 2 cotangent (row-block part)
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, CH = a.CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.x * ROWS;
    const int h = blockIdx.y, b = blockIdx.z;
    const int SY = T + 16, SP = T + 2;
    float* xs = smem;                              // [KC][16]
    float* big = smem + KC * 16;                   // phase 1: ys [KC][SY]; phase 3: zs [CG][SP]
    const int bigsz = (KC * SY > CG * SP) ? KC * SY : CG * SP;
    float* ps = big + bigsz;                       // [16][SP]
    float* ds = ps + ROWS * SP;                    // [16][SP]
    const int ntile = T / 16;                      // score column tiles; wave w owns tiles w, w+4, ...
    const int myt = (ntile + 3 - wave) / 4;        // tiles of this wave (<= 4 for T <= 256)

    // ---------------- phase 1: score tile ----------------
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nprod = FORM == 1 ? 2 : 1;
    for (int p = 0; p < nprod; ++p) {
        const float* X = (p == 0 ? a.X1 + (long)b * a.x1_bs + (long)h * a.x1_hs : a.X2 + (long)b * a.x2_bs + (long)h * a.x2_hs);
        const float* Y = (p == 0 ? a.Y1 + (long)b * a.y1_bs + (long)h * a.y1_hs : a.Y2 + (long)b * a.y2_bs + (long)h * a.y2_hs);
        for (int c0 = 0; c0 < CH; c0 += KC) {
            __syncthreads();
            // stage X[c0..c0+KC)[i0..i0+16) and Y[c0..c0+KC)[0..T): 16-byte loads along the token axis
            for (int e = tid; e < KC * 4; e += 256) {
                int c = e >> 2, q4 = e & 3;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c0 + c < CH) v = *reinterpret_cast<const f32x4*>(X + (long)(c0 + c) * T + i0 + q4 * 4);
                *reinterpret_cast<f32x4*>(xs + c * 16 + q4 * 4) = v;
            }
            const int t4 = T / 4;
            for (int e = tid; e < KC * t4; e += 256) {
                int c = e / t4, q4 = e - c * t4;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c0 + c < CH) v = *reinterpret_cast<const f32x4*>(Y + (long)(c0 + c) * T + q4 * 4);
                *reinterpret_cast<f32x4*>(big + c * SY + q4 * 4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < KC / 4; ++kk) {
                const float av = xs[(kk * 4 + kq) * 16 + l16];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < myt) {
                        const float bv = big[(kk * 4 + kq) * SY + (wave + 4 * t) * 16 + l16];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[i = 4*kq + r][j = tile*16 + l16] -> ds
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < myt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[(4 * kq + r) * SP + (wave + 4 * t) * 16 + l16] = acc[t][r];
        }
    }
    __syncthreads();

    // ---------------- phase 2: row operation, 4 rows per wave ----------------
    const int per = T / 64;
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
        float* d = ds + row * SP;
        float* pr = ps + row * SP;
        if (FORM == 0) {
            float v[4], mx = -INFINITY;
            for (int i = 0; i < per; ++i) { v[i] = a.scale * d[lane + 64 * i]; mx = fmaxf(mx, v[i]); }
            mx = wmax(mx);
            float sum = 0.f;
            for (int i = 0; i < per; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
            sum = wsum(sum);
            const float inv = 1.0f / sum;
            float* pg = a.Sout + (long)b * a.s_bs + (long)h * a.s_hs + (long)(i0 + row) * T;
            for (int i = 0; i < per; ++i) {
                const float pv = v[i] * inv;
                pr[lane + 64 * i] = pv;
                pg[lane + 64 * i] = pv;              // the primal P the Jacobian passes re-read
            }
        } else {
            const float* pg = a.P + (long)b * a.p_bs + (long)h * a.p_hs + (long)(i0 + row) * T;
            float dv[4], pv[4], dot = 0.f;
            for (int i = 0; i < per; ++i) {
                dv[i] = d[lane + 64 * i];
                pv[i] = pg[lane + 64 * i];
                dot += dv[i] * pv[i];
            }
            dot = wsum(dot);
            float* sg = FORM == 2 ? a.Sout + (long)b * a.s_bs + (long)h * a.s_hs + (long)(i0 + row) * T : nullptr;
            for (int i = 0; i < per; ++i) {
                const float o = a.scale * pv[i] * (dv[i] - dot);
                d[lane + 64 * i] = o;
                pr[lane + 64 * i] = pv[i];
                if (FORM == 2) sg[lane + 64 * i] = o;   // g_S for the column-block kernel
            }
        }
    }

    // ---------------- phase 3: O[c][i0 + i] = sum_j Z1[c][j] W1[i][j] (+ Z2 W2) ----------------
    // forward: (v, P); tangent: (dv, P) + (v, dP); cotangent: (k, g_S)
    const int nprod3 = FORM == 1 ? 2 : 1;
    float* O = a.O + (long)b * a.o_bs + (long)h * a.o_hs;
    for (int cg = 0; cg < CH; cg += CG) {
        f32x4 oc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < nprod3; ++p) {
            const float* Z = (p == 0 ? a.Z1 + (long)b * a.z1_bs + (long)h * a.z1_hs : a.Z2 + (long)b * a.z2_bs + (long)h * a.z2_hs);
            const float* W = (FORM == 0) ? ps : (FORM == 2) ? ds : (p == 0 ? ps : ds);
            __syncthreads();      // previous users of `big` (and, first time, the phase-2 writers of ps / ds) are done
            const int t4 = T / 4;
            for (int e = tid; e < CG * t4; e += 256) {
                int c = e / t4, q4 = e - c * t4;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cg + c < CH) v = *reinterpret_cast<const f32x4*>(Z + (long)(cg + c) * T + q4 * 4);
                float* dst = big + c * SP + q4 * 4;          // SP = T + 2: rows are 8-byte aligned only
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
            __syncthreads();
            const float* zrow = big + (wave * 16 + l16) * SP;
            const float* wrow = W + l16 * SP;
            for (int kk = 0; kk < T / 4; ++kk) {
                const float av = zrow[kk * 4 + kq];
                const float bv = wrow[kk * 4 + kq];
                oc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, oc, 0, 0, 0);
            }
        }
        // D[c = 4*kq + r][i = l16]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = cg + wave * 16 + 4 * kq + r;
            if (c < CH) O[(long)c * T + i0 + l16] = oc[r];
        }
    }
}

// cotangent, column blocks: g_k[c][j] = sum_i q[c][i] g_S[i][j],  g_v[c][j] = sum_i g_o[c][i] P[i][j]
// (X1 = q, W1 = g_S -> O;  X2 = g_o, W2 = P -> O2).  One workgroup owns 16 key columns of one (probe, head).
__global__ __launch_bounds__(256) void attn_cols_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, CH = a.CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * ROWS;
    const int h = blockIdx.y, b = blockIdx.z;
    const int SP = T + 2;
    float* ws = smem;                  // [T][16 + 1]  W[i][j0 + j]   (k-major: row i)
    float* zs = smem + T * 17;         // [CG][SP]
    for (int pr = 0; pr < 2; ++pr) {
        const float* Wg = pr == 0 ? a.Sout + (long)b * a.s_bs + (long)h * a.s_hs     // g_S written by the row kernel
                                  : a.P + (long)b * a.p_bs + (long)h * a.p_hs;
        const float* Z = pr == 0 ? a.X1 + (long)b * a.x1_bs + (long)h * a.x1_hs : a.X2 + (long)b * a.x2_bs + (long)h * a.x2_hs;
        float* O = pr == 0 ? a.O + (long)b * a.o_bs + (long)h * a.o_hs : a.O2 + (long)b * a.o2_bs + (long)h * a.o2_hs;
        __syncthreads();
        for (int e = tid; e < T * 4; e += 256) {
            int i = e >> 2, q4 = e & 3;
            f32x4 v = *reinterpret_cast<const f32x4*>(Wg + (long)i * T + j0 + q4 * 4);
            float* dst = ws + i * 17 + q4 * 4;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        for (int cg = 0; cg < CH; cg += CG) {
            __syncthreads();
            const int t4 = T / 4;
            for (int e = tid; e < CG * t4; e += 256) {
                int c = e / t4, q4 = e - c * t4;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cg + c < CH) v = *reinterpret_cast<const f32x4*>(Z + (long)(cg + c) * T + q4 * 4);
                float* dst = zs + c * SP + q4 * 4;
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
            __syncthreads();
            f32x4 oc = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* zrow = zs + (wave * 16 + l16) * SP;
            for (int kk = 0; kk < T / 4; ++kk) {
                const float av = zrow[kk * 4 + kq];                 // A[m = c][k = i]
                const float bv = ws[(kk * 4 + kq) * 17 + l16];      // B[k = i][n = j]
                oc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, oc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cg + wave * 16 + 4 * kq + r;
                if (c < CH) O[(long)c * T + j0 + l16] = oc[r];
            }
        }
    }
}

size_t rows_lds(int T) {
    const int SY = T + 16, SP = T + 2;
    const int bigsz = (KC * SY > CG * SP) ? KC * SY : CG * SP;
    return (size_t)(KC * 16 + bigsz + 2 * ROWS * SP) * sizeof(float);
}

template <int FORM>
void launch_rows(const AttnArgs& a, hipStream_t st) {
    static bool big_lds = false;
    const size_t lds = rows_lds(a.T);
    if (lds > 64 * 1024 && !big_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_rows_kernel<FORM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        big_lds = true;
    }
    hipLaunchKernelGGL(attn_rows_kernel<FORM>, dim3(a.T / ROWS, a.NH, a.B), dim3(256), lds, st, a);
}

}  // namespace

// TODO(perf): measured 41 ms/step SLOWER than the five-GEMM path (latency-bound inner loops); routed off until the
// contiguous-k / ds_read_b128 rewrite lands
bool attn_supported(int T, int CH) { (void)T; (void)CH; return false; }

void launch_attn_rows(int form, const AttnArgs& a, hipStream_t st) {
    if (form == 0) launch_rows<0>(a, st);
    else if (form == 1) launch_rows<1>(a, st);
    else launch_rows<2>(a, st);
}

void launch_attn_cols(const AttnArgs& a, hipStream_t st) {
    static bool big_lds = false;
    const size_t lds = (size_t)(a.T * 17 + CG * (a.T + 2)) * sizeof(float);
    if (lds > 64 * 1024 && !big_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        big_lds = true;
    }
    hipLaunchKernelGGL(attn_cols_kernel, dim3(a.T / ROWS, a.NH, a.B), dim3(256), lds, st, a);
}

}  // namespace loco
