// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 157 TF/s peak).
//
// One kernel family serves the three passes of the PMP-Jacobian operator:
//   forward   : GroupNorm-apply + SiLU fused into the operand staging
//   tangent   : d(GroupNorm+SiLU) fused into the staging (reads primal + tangent)
//   cotangent : data-gradient conv = the same kernel on flipped/transposed weights,
//               with the GroupNorm+SiLU cotangent of the *next* layer fused in
// (replaces the cuDNN conv / ATen GroupNorm / SiLU / functorch dual-number
//  launches listed in SURVEY.md section 2.3).
//
// GEMM view: D[cout][pixel] = sum_{tap, cin} W[cin][tap][cout] * halo[cin][pixel + tap].
// A workgroup (4 waves) owns MT couts x NT output pixels (a TH x TW patch of one
// image); it walks Cin in chunks of BK=8: the input halo patch of the chunk is
// staged ONCE in LDS as [cin][halo pixel] (prologue applied while staging) and
// all 9 taps are shifted LDS reads of it; weights are staged as [tap][cin][cout].
// Both MFMA operands are then bank-conflict-free ds_read_b32 (32 consecutive
// pixels / couts per half wave).  The next chunk's global loads are issued
// before the MFMA block of the current one (register prefetch).
#include "kernels.h"
#include <cstdio>

namespace loco {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 8;

// `act`: ConvArgs::act (kernels.h ActKind), wave-uniform; this exact-fp32 kernel takes the activation at run time (CM_GN_GELU
// is dispatched as CM_GN_SILU with act = ACT_GELU)
template <int MODE>
__device__ __forceinline__ float prologue(float d, float x, float sc, float sh, float gamma, float mean,
                                          float rstd, float m1, float m2, int act) {
    if constexpr (MODE == CM_NONE) {
        return d;
    } else if constexpr (MODE == CM_GN_SILU) {
        return act_fwd(fmaf(sc, d, sh), act);
    } else if constexpr (MODE == CM_GN) {
        return fmaf(sc, d, sh);
    } else {
        float xh = (x - mean) * rstd;
        float ds = act_der(fmaf(sc, x, sh), act);
        if constexpr (MODE == CM_TAN_SILU) {
            return ds * sc * (d - m1 - xh * m2);
        } else {  // CM_COT_SILU
            return rstd * (gamma * ds * d - m1 - xh * m2);
        }
    }
}

template <int TAPS, int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_f32(ConvArgs a) {
    constexpr int MT = WM * TM * 32;
    constexpr int NT = WN * TN * 32;
    constexpr int KS = (TAPS == 9) ? 3 : 1;
    constexpr int NPOS = (TAPS == 9) ? 3 : 1;
    constexpr int WTOT = TAPS * BK * MT / 4;          // float4 elements of one weight chunk
    constexpr int NWV = (WTOT + 255) / 256;
    constexpr bool NEEDP = (MODE == CM_TAN_SILU || MODE == CM_COT_SILU);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                       // [TAPS][BK][MT]
    float* Hs = smem + TAPS * BK * MT;      // [BK][halo_sz]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int S = a.stride;
    const int TW = a.Wout < 32 ? a.Wout : 32;
    const int TH = NT / TW;
    const int tiles_x = a.Wout / TW;
    const int oy0 = (blockIdx.x / tiles_x) * TH;
    const int ox0 = (blockIdx.x % tiles_x) * TW;
    const int co0 = blockIdx.y * MT;
    const int b = blockIdx.z / a.nsplit;
    const int split = blockIdx.z % a.nsplit;

    const int halo_w = (TW - 1) * S + KS;
    const int halo_h = (TH - 1) * S + KS;
    const int halo_sz = halo_h * halo_w;

    // logical input extent seen by the conv (after upsample / zero insertion)
    const int LH = (a.upsample || a.zins) ? a.Hin * 2 : a.Hin;
    const int LW = (a.upsample || a.zins) ? a.Win * 2 : a.Win;
    const long in_plane = (long)a.Hin * a.Win;

    // per-thread halo positions (independent of the channel chunk)
    int poff[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        int pos = tid + i * 256;
        int off = -1;
        if (pos < halo_sz) {
            int hy = pos / halo_w, hx = pos - hy * halo_w;
            int Y = oy0 * S - a.pad + hy, X = ox0 * S - a.pad + hx;
            if (Y >= 0 && Y < LH && X >= 0 && X < LW) {
                if (a.upsample) off = (Y >> 1) * a.Win + (X >> 1);
                else if (a.zins) off = ((Y | X) & 1) ? -1 : (Y >> 1) * a.Win + (X >> 1);
                else off = Y * a.Win + X;
            }
        }
        poff[i] = off;
    }

    // B-operand (pixel) offsets of this lane inside the halo image
    int hoff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int p = (wn * TN + j) * 32 + l31;
        int ty = p / TW, tx = p - ty * TW;
        hoff[j] = ty * S * halo_w + tx * S;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunks = (a.Cin + BK - 1) / BK;
    const int cps = (nchunks + a.nsplit - 1) / a.nsplit;
    const int cbeg = split * cps;
    const int cend = (cbeg + cps < nchunks) ? cbeg + cps : nchunks;

    const float* inb = a.in + (long)b * a.in_bs;
    const float* prb = NEEDP ? a.prim + (long)b * a.prim_bs : nullptr;
    // primal statistics: broadcast over the probe batch when their stride is 0
    const float* scb = (MODE != CM_NONE) ? a.sc + (long)b * a.scsh_bs : nullptr;
    const float* shb = (MODE != CM_NONE) ? a.sh + (long)b * a.scsh_bs : nullptr;
    const float* mrb = NEEDP ? a.mr + (long)b * a.mr_bs : nullptr;
    const float* tsb = NEEDP ? a.tst + (long)b * a.tst_bs : nullptr;
    const int wpitch = (a.Cout + 31) & ~31;

    float hv[NPOS][BK];
    float pv[NEEDP ? NPOS : 1][BK];
    float4 wv[NWV];

    auto prefetch = [&](int chunk) {
        const int c0 = chunk * BK;
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
#pragma unroll
            for (int k = 0; k < BK; ++k) {
                float v = 0.0f, pvv = 0.0f;
                if (poff[i] >= 0 && c0 + k < a.Cin) {
                    v = inb[(long)(c0 + k) * in_plane + poff[i]];
                    if constexpr (NEEDP) pvv = prb[(long)(c0 + k) * in_plane + poff[i]];
                }
                hv[i][k] = v;
                if constexpr (NEEDP) pv[i][k] = pvv;
            }
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            int e = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < WTOT) {
                int row = e / (MT / 4), m4 = e - row * (MT / 4);   // row = k*TAPS + tap
                int k = row / TAPS;
                int co = co0 + m4 * 4;
                if (c0 + k < a.Cin && co < wpitch)
                    v = *reinterpret_cast<const float4*>(a.w + ((long)c0 * TAPS + row) * wpitch + co);
            }
            wv[i] = v;
        }
    };

    auto stage = [&](int chunk) {
        const int c0 = chunk * BK;
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            float sc = 0.f, sh = 0.f, gm = 0.f, mean = 0.f, rstd = 0.f, m1 = 0.f, m2 = 0.f;
            int c = c0 + k;
            if (c >= a.Cin) c = a.Cin - 1;
            if constexpr (MODE != CM_NONE) {
                sc = scb[c];
                sh = shb[c];
            }
            if constexpr (NEEDP) {
                int g = c / a.cpg;
                mean = mrb[2 * g];
                rstd = mrb[2 * g + 1];
                m1 = tsb[2 * g];
                m2 = tsb[2 * g + 1];
                gm = sc / rstd;   // effective gain (includes the scale-shift factor of the ADM blocks)
            }
#pragma unroll
            for (int i = 0; i < NPOS; ++i) {
                int pos = tid + i * 256;
                if (pos < halo_sz) {
                    float v = 0.0f;
                    if (poff[i] >= 0 && c0 + k < a.Cin)
                        v = prologue<MODE>(hv[i][k], NEEDP ? pv[i][k] : 0.f, sc, sh, gm, mean, rstd, m1, m2, a.act);
                    Hs[k * halo_sz + pos] = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            int e = tid + i * 256;
            if (e < WTOT) {
                int row = e / (MT / 4), m4 = e - row * (MT / 4);
                int k = row / TAPS, tap = row - k * TAPS;
                *reinterpret_cast<float4*>(Ws + (tap * BK + k) * MT + m4 * 4) = wv[i];
            }
        }
    };

    if (cbeg < cend) prefetch(cbeg);
    for (int chunk = cbeg; chunk < cend; ++chunk) {
        __syncthreads();
        stage(chunk);
        __syncthreads();
        if (chunk + 1 < cend) prefetch(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int tapoff = (TAPS == 9) ? (tap / 3) * halo_w + (tap % 3) : 0;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int k = 2 * kk + khalf;
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = Ws[(tap * BK + k) * MT + (wm * TM + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = Hs[k * halo_sz + hoff[j] + tapoff];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // epilogue: D[row = cout][col = pixel]; col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long out_plane = (long)a.Hout * a.Wout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int p = (wn * TN + j) * 32 + l31;
        int ty = p / TW, tx = p - ty * TW;
        int oy = oy0 + ty, ox = ox0 + tx;
        if (oy >= a.Hout || ox >= a.Wout) continue;
        long pix = (long)oy * a.Wout + ox;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co >= a.Cout) continue;
                float v = acc[i][j][r];
                if (a.nsplit > 1) {
                    a.partial[(((long)split * a.B + b) * a.Cout + co) * out_plane + pix] = v;
                } else {
                    if (a.bias) v += a.bias[co];
                    if (a.bias2) v += a.bias2[(long)b * a.bias2_bs + co];
                    if (a.res) v += a.res_scale * a.res[(long)b * a.res_bs + co * out_plane + pix];
                    float* o = a.out + (long)b * a.out_bs + co * out_plane + pix;
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
        }
    }
}

__global__ void conv_splitk_reduce(ConvArgs a, long total) {
    const long out_plane = (long)a.Hout * a.Wout;
    const long per_b = (long)a.Cout * out_plane;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / per_b);
        long rem = i - (long)b * per_b;
        int co = (int)(rem / out_plane);
        float v = 0.f;
        for (int s = 0; s < a.nsplit; ++s) v += a.partial[(long)s * total + i];
        if (a.bias) v += a.bias[co];
        if (a.bias2) v += a.bias2[(long)b * a.bias2_bs + co];
        if (a.res) v += a.res_scale * a.res[(long)b * a.res_bs + rem];
        float* o = a.out + (long)b * a.out_bs + rem;
        if (a.accumulate) v += *o;
        *o = v;
    }
}

// ---------------------------------------------------------------------------
struct TileCfg { int MT, NT; };

static inline int pick_tile(int Cout, int HW) {
    // 0: 128x128, 1: 128x64, 2: 32x128, 3: 64x64
    if (HW >= 128) {
        if (Cout > 64) return 0;
        if (Cout > 32) return (HW >= 128) ? 0 : 3;
        return 2;
    }
    // HW == 64
    if (Cout > 64) return 1;
    return 3;
}
int conv_pick_tile(int Cout, int HW) { return pick_tile(Cout, HW); }
static const TileCfg kTiles[4] = {{128, 128}, {128, 64}, {32, 128}, {64, 64}};

int conv_pick_nsplit(int Cin, int Cout, int Hout, int Wout, int B, int taps) {
    int t = pick_tile(Cout, Hout * Wout);
    long blocks = (long)((Hout * Wout) / kTiles[t].NT) * ((Cout + kTiles[t].MT - 1) / kTiles[t].MT) * B;
    int nchunks = (Cin + BK - 1) / BK;
    if (blocks >= 384 || nchunks < 8) return 1;
    int want = (int)((512 + blocks - 1) / blocks);
    int maxs = nchunks / 4;            // at least 4 chunks (32 channels) per split
    if (want > maxs) want = maxs;
    if (want > 32) want = 32;
    if (want < 1) want = 1;
    return want;
}

size_t conv_partial_floats(const ConvArgs& a) {
    if (a.nsplit <= 1) return 0;
    return (size_t)a.nsplit * a.B * a.Cout * a.Hout * a.Wout;
}

template <int TAPS, int WM, int WN, int TM, int TN, int MODE>
static void launch_one(const ConvArgs& a, hipStream_t st) {
    constexpr int MT = WM * TM * 32, NT = WN * TN * 32;
    constexpr int KS = (TAPS == 9) ? 3 : 1;
    int TW = a.Wout < 32 ? a.Wout : 32;
    int TH = NT / TW;
    int halo_w = (TW - 1) * a.stride + KS, halo_h = (TH - 1) * a.stride + KS;
    size_t lds = ((size_t)TAPS * BK * MT + (size_t)BK * halo_w * halo_h) * sizeof(float);
    dim3 grid((a.Hout * a.Wout) / NT, (a.Cout + MT - 1) / MT, a.B * a.nsplit);
    hipLaunchKernelGGL((conv_mfma_f32<TAPS, WM, WN, TM, TN, MODE>), grid, dim3(256), lds, st, a);
}

template <int TAPS, int MODE>
static void launch_tile(const ConvArgs& a, hipStream_t st) {
    switch (pick_tile(a.Cout, a.Hout * a.Wout)) {
        case 0: launch_one<TAPS, 2, 2, 2, 2, MODE>(a, st); break;
        case 1: launch_one<TAPS, 4, 1, 1, 2, MODE>(a, st); break;
        case 2: launch_one<TAPS, 1, 4, 1, 1, MODE>(a, st); break;
        default: launch_one<TAPS, 2, 2, 1, 1, MODE>(a, st); break;
    }
}

int conv_bf16_pick_tile(int Cout, int HW, int Bsplit);   // conv_bf16.hip
int bf16_tile_of(const ConvArgs& a);                       // conv_bf16.hip
static inline int bf16_tile_of_(const ConvArgs& a) { return bf16_tile_of(a); }

const char* conv_variant_name(const ConvArgs& a, int taps, int prec) {
    static const char* tiles[6] = {"2,2,2,2", "4,1,1,2", "1,4,1,1", "2,2,1,1", "2,2,2,4", "2,4,2,2"};
    static char names[3][2][6][6][56];
    int t = prec ? conv_bf16_pick_tile(a.Cout, a.Hout * a.Wout, a.B) : pick_tile(a.Cout, a.Hout * a.Wout);
    if (prec) { ConvArgs q = a; q.taps = taps; if (bf16_tile_of_(q) == 6) t = 0; }     // the two-per-CU variant is a 128 x 128 tile too
    if (prec && t == 4) t = 5;
    if (prec && a.stride == 2 && t == 5) t = 0;
    if (t < 0 || t > 5) t = 0;
    if (prec && taps == 9 && a.Cin2 > 0) {
        static char kn2[3][6][40];
        const int mm = (a.mode < 0 || a.mode > 5) ? 2 : a.mode;
        if (!kn2[prec][mm][0]) snprintf(kn2[prec][mm], 40, "%s<2,4,2,2,%d>", prec == 1 ? "conv_kcat_bf16x3" : "conv_kcat_f16", mm);
        return kn2[prec][mm];
    }
    if (prec == 1 && taps == 1 && a.gemm) {
        static char gn[2][40];
        char* n = gn[a.gemm_tm == 4];
        if (!n[0]) snprintf(n, 40, "conv_gemm_bf16x3<%d>", a.gemm_tm);
        return n;
    }
    if (prec == 1 && taps == 9 && a.pers_groups > 0 && !a.dual) {
        static char pn[6][40];
        const int mm = (a.mode < 0 || a.mode > 5) ? 2 : a.mode;
        if (!pn[mm][0]) snprintf(pn[mm], 40, "conv_pers_bf16x3<9,2,4,2,2,%d>", mm);
        return pn[mm];
    }
    {
        ConvArgs q = a; q.taps = taps;
        if (prec == 1 && taps == 9 && conv_pair_ok(q)) {      // the 16x16x32 tap-pair kernel (conv_pair_kernel.h)
            static char pn[5][40];
            const int mm = (a.mode < 0 || a.mode > 4) ? 0 : a.mode;
            if (!pn[mm][0]) snprintf(pn[mm], 40, "conv_pair_bf16x3<%d>", mm);
            return pn[mm];
        }
    }
    if (prec == 1 && taps == 9 && a.dual) {
        static char dn[5][40];
        const int mm = (a.mode < 0 || a.mode > 4) ? 2 : a.mode;
        if (!dn[mm][0]) snprintf(dn[mm], 40, "conv_dual_bf16x3<%d>", mm);
        return dn[mm];
    }
    int ti = taps == 9 ? 0 : 1;
    int m = a.mode;
    if (taps != 9 && m != CM_NONE) m = CM_GN;
    if (!prec && m == CM_GN_GELU) m = CM_GN_SILU;      // the exact-fp32 kernel takes the activation at run time
    if (m < 0 || m > 5) m = 2;
    char* n = names[prec][ti][t][m];
    static const char* kn[3] = {"conv_mfma_f32", "conv_mfma_bf16x3", "conv_mfma_f16"};
    if (!n[0]) snprintf(n, 56, "%s<%d,%s,%d>", kn[prec], taps, tiles[t], m);
    return n;
}

void launch_conv(const ConvArgs& a, int taps, hipStream_t st) {
    if (taps == 9) {
        switch (a.mode) {
            case CM_NONE: launch_tile<9, CM_NONE>(a, st); break;
            case CM_GN_SILU: case CM_GN_GELU: launch_tile<9, CM_GN_SILU>(a, st); break;
            case CM_TAN_SILU: launch_tile<9, CM_TAN_SILU>(a, st); break;
            case CM_COT_SILU: launch_tile<9, CM_COT_SILU>(a, st); break;
            default: launch_tile<9, CM_GN>(a, st); break;
        }
    } else {
        switch (a.mode) {
            case CM_NONE: launch_tile<1, CM_NONE>(a, st); break;
            default: launch_tile<1, CM_GN>(a, st); break;
        }
    }
}

// the same with 16 bytes per thread and four slab loads in flight (planes are multiples of 4 floats from 8x8 on)
typedef float f32x4_r __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv_splitk_reduce4(ConvArgs a, long total) {
    const long out_plane = (long)a.Hout * a.Wout;
    const long per_b = (long)a.Cout * out_plane;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (long)gridDim.x * blockDim.x * 4) {
        const int b = (int)(i / per_b);
        const long rem = i - (long)b * per_b;
        const int co = (int)(rem / out_plane);
        f32x4_r v = {0.f, 0.f, 0.f, 0.f};
        const float* pp = a.partial + i;
        int s = 0;
        for (; s + 4 <= a.nsplit; s += 4) {
            const f32x4_r p0 = *reinterpret_cast<const f32x4_r*>(pp + (long)s * total);
            const f32x4_r p1 = *reinterpret_cast<const f32x4_r*>(pp + (long)(s + 1) * total);
            const f32x4_r p2 = *reinterpret_cast<const f32x4_r*>(pp + (long)(s + 2) * total);
            const f32x4_r p3 = *reinterpret_cast<const f32x4_r*>(pp + (long)(s + 3) * total);
            v += (p0 + p1) + (p2 + p3);
        }
        for (; s < a.nsplit; ++s) v += *reinterpret_cast<const f32x4_r*>(pp + (long)s * total);
        float add = 0.f;
        if (a.bias) add += a.bias[co];
        if (a.bias2) add += a.bias2[(long)b * a.bias2_bs + co];
        v += add;
        if (a.res) v += a.res_scale * *reinterpret_cast<const f32x4_r*>(a.res + (long)b * a.res_bs + rem);
        float* o = a.out + (long)b * a.out_bs + rem;
        if (a.accumulate) v += *reinterpret_cast<const f32x4_r*>(o);
        *reinterpret_cast<f32x4_r*>(o) = v;
    }
}

void launch_conv_splitk_reduce(const ConvArgs& a, hipStream_t st) {
    if (a.nsplit > 1) {
        long total = (long)a.B * a.Cout * a.Hout * a.Wout;
        const bool vec = ((a.Hout * a.Wout) % 4) == 0 && (a.out_bs % 4) == 0 && (a.res_bs % 4) == 0 &&
                         (reinterpret_cast<uintptr_t>(a.out) % 16) == 0 && (reinterpret_cast<uintptr_t>(a.res) % 16) == 0;
        if (vec) {
            int blocks = (int)((total / 4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(conv_splitk_reduce4, dim3(blocks), dim3(256), 0, st, a, total);
            return;
        }
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_splitk_reduce, dim3(blocks), dim3(256), 0, st, a, total);
    }
}

}  // namespace loco
