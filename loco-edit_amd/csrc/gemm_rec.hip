// Strided batched GEMM on split-bf16 operands, BOTH operands by LDS-DMA from records split once per launch (round 5).
//
// The attention products of a single wide head (the latent decoder's mid attention: 4096 tokens x 512 channels) run as plain GEMMs
// with per-probe [T x T] matrices; gemm_bf16x3_kernel (gemm.hip) serves them at 60 TFLOP/s algorithmic (1.42 ms per launch of 5
// probes: 4.3 % of a config-4 solve): every 128 x 128 workgroup loads its operand panels with 4-byte loads, converts them to hi / lo
// halves (each element by 32 workgroups) and syncs twice per 32 k.  Here, as in conv_gemm_kernel.h and attn_flash.hip:
//   * `rec_split_kernel` turns each operand into records [z][K / 16][row][64 B] = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] of one
//     row's 16 consecutive k, pieces XOR-swizzled by (row >> 2) & 3 (one pass: 4 B read + 4 B written per element; an operand
//     shared by the batch is split once);
//   * `gemm_rec_bf16x3<TM>`: a workgroup of 8 waves owns 64 TM rows x 256 columns, a ring of four K-step slots filled by
//     `global_load_lds_dwordx4` three steps ahead, one raw barrier per step, fragments of the next step read under this step's
//     MFMAs (the stage loop of conv_gemm_bf16x3); two K segments (the tangent forms dq^T k + q^T dk as one launch);
//   * launches whose tile grid leaves the chip idle split K over workgroups into partial tiles + a deterministic reduce.
// Same accumulation order per output element as gemm_bf16x3_kernel without split-K (k-step by k-step, lo*hi, hi*lo, hi*hi).
// Reference call sites: the AttnBlock of the latent decoder under jvp / vjp (edit.py:655-658 `self.pipe.vae.decode`).
#include "conv_bf16_kernel.h"
#include <cstdint>
#include <cstdlib>

namespace loco {

namespace {

constexpr int GR_NSLOT = 4, GR_NT = 256;

struct RecSplitArgs { const float* X; long sr, sk, sb, sh; int R, K, nb2; unsigned char* rec; };

// grid (ceil(R / 256), ceil(K / 16), z): thread = one row of one 16-k chunk
template <bool KCONTIG>
__global__ __launch_bounds__(256) void rec_split_kernel(RecSplitArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, z = blockIdx.z, nchunk = gridDim.y;
    if (r >= a.R) return;
    const float* X = a.X + (long)(z / a.nb2) * a.sb + (long)(z % a.nb2) * a.sh + (long)r * a.sr + (long)c * 16 * a.sk;
    float v[16];
    const int kleft = a.K - c * 16;
    if (KCONTIG && kleft >= 16 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const float4* p = reinterpret_cast<const float4*>(X);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 x = p[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = k < kleft ? X[(long)k * a.sk] : 0.f;
    }
    uint4 h0, l0, h1, l1;
    split8<PR_BF16X3>(v, h0, l0);
    split8<PR_BF16X3>(v + 8, h1, l1);
    unsigned char* rp = a.rec + (((long)z * nchunk + c) * a.R) * 64;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(r, 0)) = h0;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(r, 1)) = h1;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(r, 2)) = l0;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(r, 3)) = l1;
}

struct GemmRecArgs {
    const unsigned char *a1, *b1, *a2, *b2;        // records of the K segments (a2 == nullptr: one segment)
    long a1_bb, a1_hb, b1_bb, b1_hb;               // byte strides per batch sample (0: shared) and per head
    long a2_bb, a2_hb, b2_bb, b2_hb;
    int M, N, nst1, nst2, nb2, Z;                  // K steps (of 16) per segment; z = b * nb2 + h
    float* C; long scm, scn, scb, sch; float alpha, beta;
    int ksplit; float* part;                        // ksplit > 1: partial tiles [split][z][M][N] (row-major) instead of C
};

template <int TM>
__global__ __launch_bounds__(512, 1) void gemm_rec_bf16x3(GemmRecArgs g) {
    constexpr int PR = PR_BF16X3, RB = 64, TN = 2;
    constexpr int MT = 64 * TM;
    constexpr int ABYTES = MT * RB, BBYTES = GR_NT * RB, SLOTB = ABYTES + BBYTES;
    constexpr int NA = ABYTES / (512 * 16), NB = BBYTES / (512 * 16), NDMA = NA + NB;
    static_assert(NA >= 1 && NB == 2 && GR_NSLOT * SLOTB <= 160 * 1024, "ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // tile order: the batch samples (and K splits) of one tile adjacent in dispatch order on one XCD (blocks L, L + 8, ...): an
    // operand shared by the batch is fetched from HBM once
    const int ntn = (g.N + GR_NT - 1) / GR_NT, ntm = (g.M + MT - 1) / MT, ZS = g.Z * g.ksplit;
    int t, zs;
    {
        const int L = blockIdx.x;
        if (((ntn * ntm) & 7) == 0) { const int q = L >> 3; zs = q % ZS; t = (q / ZS) * 8 + (L & 7); }
        else { zs = L % ZS; t = L / ZS; }
        t = __builtin_amdgcn_readfirstlane(t);
        zs = __builtin_amdgcn_readfirstlane(zs);
    }
    const int z = zs / g.ksplit, split = zs % g.ksplit;
    const int bz = z / g.nb2, hz = z % g.nb2;
    const int m0 = (t / ntn) * MT, n0 = (t % ntn) * GR_NT;
    const int nst_all = g.nst1 + g.nst2;
    const int sps = (nst_all + g.ksplit - 1) / g.ksplit;
    const int sbeg = split * sps, send = sbeg + sps < nst_all ? sbeg + sps : nst_all;
    const int nst = send - sbeg;

    unsigned aoffg[NA], boffg[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = tid + i * 512;
        int r = m0 + (e >> 2);
        if (r >= g.M) r = g.M - 1;                                 // (rows past M are never stored)
        aoffg[i] = (unsigned)r * 64u + (unsigned)(e & 3) * 16u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = tid + i * 512;
        int p = n0 + (e >> 2);
        if (p >= g.N) p = g.N - 1;
        boffg[i] = (unsigned)p * 64u + (unsigned)(e & 3) * 16u;
    }
    const unsigned char* const a1 = g.a1 + bz * g.a1_bb + hz * g.a1_hb;
    const unsigned char* const b1 = g.b1 + bz * g.b1_bb + hz * g.b1_hb;
    const unsigned char* const a2 = g.a2 ? g.a2 + bz * g.a2_bb + hz * g.a2_hb : nullptr;
    const unsigned char* const b2 = g.a2 ? g.b2 + bz * g.b2_bb + hz * g.b2_hb : nullptr;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef const __attribute__((address_space(1))) unsigned char glb_u8;
    auto dma = [&](int st, int slot) {                          // K-step st of this workgroup -> ring slot
        const bool live = st < nst;
        const int s = __builtin_amdgcn_readfirstlane(sbeg + (live ? st : nst - 1));
        const bool seg2 = s >= g.nst1;
        const int chunk = seg2 ? s - g.nst1 : s;
        const unsigned char* wa = (seg2 ? a2 : a1) + (long)chunk * ((long)g.M * 64);
        const unsigned char* ba = (seg2 ? b2 : b1) + (long)chunk * ((long)g.N * 64);
        unsigned char* const S = smem_b + slot * SLOTB;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((glb_u8*)(wa + (live ? aoffg[i] : 0u)), (lds_u8*)(S + i * 8192 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((glb_u8*)(ba + (live ? boffg[i] : 0u)), (lds_u8*)(S + ABYTES + i * 8192 + wave * 1024), 16, 0, 0);
    };

    const unsigned aoff_hi = (unsigned)rec_off<PR>(wm * (TM * 32) + l31, khalf), aoff_lo = (unsigned)rec_off<PR>(wm * (TM * 32) + l31, 2 + khalf);
    const unsigned boff_hi = (unsigned)ABYTES + (unsigned)rec_off<PR>(wn * 64 + l31, khalf);
    const unsigned boff_lo = (unsigned)ABYTES + (unsigned)rec_off<PR>(wn * 64 + l31, 2 + khalf);
    struct FragA { s16x8 h, l; };
    struct FragB { s16x8 h[TN], l[TN]; };
    auto load_A = [&](FragA& f, int slot, int i) {
        const unsigned char* S = smem_b + slot * SLOTB + i * (32 * RB);
        f.h = *reinterpret_cast<const s16x8*>(S + aoff_hi);
        f.l = *reinterpret_cast<const s16x8*>(S + aoff_lo);
    };
    auto load_B = [&](FragB& f, int slot) {
        const unsigned char* S = smem_b + slot * SLOTB;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.h[j] = *reinterpret_cast<const s16x8*>(S + j * (32 * RB) + boff_hi);
            f.l[j] = *reinterpret_cast<const s16x8*>(S + j * (32 * RB) + boff_lo);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto mma_row = [&](const FragA& fa, const FragB& fb, int i) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, fa.h), al = __builtin_bit_cast(bf16x8, fa.l);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, fb.h[j]), bl = __builtin_bit_cast(bf16x8, fb.l[j]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
        }
    };

    dma(0, 0); dma(1, 1); dma(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
    FragA fa[2];
    FragB fb[2];
    load_A(fa[0], 0, 0);
    load_B(fb[0], 0);
    auto step = [&](auto ktag, const int s) {
        constexpr int K4 = decltype(ktag)::value;                  // s % 4
        dma(s + 3, (K4 + 3) & 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            mma_row(fa[i & 1], fb[K4 & 1], i);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < TM) load_A(fa[(i + 1) & 1], K4, i + 1);
            else load_A(fa[0], (K4 + 1) & 3, 0);
            if (i == 0) load_B(fb[(K4 + 1) & 1], (K4 + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int s = 0; s < nst; s += 4) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nst) step(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 < nst) step(std::integral_constant<int, 2>{}, s + 2);
        if (s + 3 < nst) step(std::integral_constant<int, 3>{}, s + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- write-out: the lane's column n of D, rows by register
    if (g.ksplit > 1) {
        float* P = g.part + ((long)split * g.Z + z) * ((long)g.M * g.N);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (m < g.M) P[(long)m * g.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    float* C = g.C + (long)bz * g.scb + (long)hz * g.sch;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (m >= g.M) continue;
                float* c = C + (long)m * g.scm + (long)n * g.scn;
                float v = g.alpha * acc[i][j][r];
                if (g.beta != 0.f) v += g.beta * (*c);
                *c = v;
            }
        }
}

// C = alpha * sum over splits + beta * C, the splits in order (deterministic)
__global__ __launch_bounds__(256) void gemm_rec_reduce(const float* part, int ksplit, int Z, int nb2, int M, int N, float* C,
                                                       long scm, long scn, long scb, long sch, float alpha, float beta) {
    const long MN = (long)M * N, i = (long)blockIdx.x * 256 + threadIdx.x;
    const int z = blockIdx.y;
    if (i >= MN) return;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[((long)k * Z + z) * MN + i];
    const int m = (int)(i / N), n = (int)(i % N);
    float* c = C + (long)(z / nb2) * scb + (long)(z % nb2) * sch + (long)m * scm + (long)n * scn;
    float v = alpha * s;
    if (beta != 0.f) v += beta * (*c);
    *c = v;
}

struct RecPlan { size_t a1, b1, a2, b2, part, total; int za, zb, za2, zb2, nck, ksplit, tm; };

RecPlan rec_plan(const GemmArgs& g) {
    RecPlan p{};
    const int nb2 = g.batch2 > 0 ? g.batch2 : 1;
    p.nck = (g.K + 15) / 16;
    p.za = (g.sab ? g.batch : 1) * nb2;  p.zb = (g.sbb ? g.batch : 1) * nb2;
    p.a1 = (size_t)p.za * p.nck * g.M * 64;  p.b1 = (size_t)p.zb * p.nck * g.N * 64;
    if (g.A2) {
        p.za2 = (g.sab2 ? g.batch : 1) * nb2;  p.zb2 = (g.sbb2 ? g.batch : 1) * nb2;
        p.a2 = (size_t)p.za2 * p.nck * g.M * 64;  p.b2 = (size_t)p.zb2 * p.nck * g.N * 64;
    }
    p.tm = (g.M % 256 == 0 || g.M >= 1024) ? 4 : 2;
    const int MT = 64 * p.tm;
    const long tiles = (long)((g.M + MT - 1) / MT) * ((g.N + GR_NT - 1) / GR_NT) * g.batch * nb2;
    const int nst = p.nck * (g.A2 ? 2 : 1);
    p.ksplit = 1;
    static int ks_env = -1;      // LOCO_GEMM_REC_KSPLIT: 0 = the rule below, n = force
    if (ks_env < 0) { const char* e = getenv("LOCO_GEMM_REC_KSPLIT"); ks_env = e ? atoi(e) : 0; }
    if (ks_env > 0) p.ksplit = ks_env;
    else if (tiles < 224 && nst >= 64) {                     // less than one workgroup per CU: the fewest splits that fill the last round
        int best = 1; double beff = (double)tiles / (256.0 * ((tiles + 255) / 256));
        for (int k = 2; k <= 8; ++k) {
            if (nst / k < 16) break;
            const long w = tiles * k;
            const double eff = (double)w / (256.0 * ((w + 255) / 256));
            if (eff > beff + 0.05) { best = k; beff = eff; }
        }
        p.ksplit = best;
    }
    if (p.ksplit > nst) p.ksplit = nst;
    if (p.ksplit > 1) p.part = (size_t)p.ksplit * g.batch * nb2 * g.M * g.N * 4;
    p.total = p.a1 + p.b1 + p.a2 + p.b2 + p.part;
    return p;
}

void split_operand(const float* X, long sr, long sk, long sb, long sh, int R, int K, int zb, int nb2, unsigned char* rec, hipStream_t st) {
    RecSplitArgs a{X, sr, sk, sb, sh, R, K, nb2, rec};
    dim3 grid((R + 255) / 256, (K + 15) / 16, zb);
    if (sk == 1) hipLaunchKernelGGL(rec_split_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(rec_split_kernel<false>, grid, dim3(256), 0, st, a);
}

template <int TM>
void launch_rec_tm(const GemmRecArgs& r, hipStream_t st) {
    constexpr int MT = 64 * TM, SLOTB = MT * 64 + GR_NT * 64;
    auto kern = &gemm_rec_bf16x3<TM>;
    static DeviceOnce once;
    if (first_on_device(once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long tiles = (long)((r.M + MT - 1) / MT) * ((r.N + GR_NT - 1) / GR_NT);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * r.Z * r.ksplit)), dim3(512), (size_t)GR_NSLOT * SLOTB, st, r);
}

}  // namespace

bool gemm_rec_eligible(const GemmArgs& g) {
    const char* e = getenv("LOCO_GEMM_REC");      // LOCO_GEMM_REC=0: the converting 128 x 128 kernel (A/B switch, read per launch)
    if ((e && atoi(e) == 0) || g.bias || g.colbias || g.R) return false;
    const double macs = (double)g.M * g.N * g.K * (g.A2 ? 2 : 1) * g.batch * (g.batch2 > 0 ? g.batch2 : 1);
    return g.K >= 256 && g.M >= 128 && g.N >= 256 && macs >= 4e9;
}
size_t gemm_rec_ws_bytes(const GemmArgs& g) { return rec_plan(g).total; }

void launch_gemm_rec(const GemmArgs& g, unsigned char* ws, hipStream_t st) {
    const RecPlan p = rec_plan(g);
    const int nb2 = g.batch2 > 0 ? g.batch2 : 1;
    unsigned char *ra = ws, *rb = ra + p.a1, *ra2 = rb + p.b1, *rb2 = ra2 + p.a2;
    float* part = reinterpret_cast<float*>(rb2 + p.b2);
    split_operand(g.A, g.sam, g.sak, g.sab, g.sah, g.M, g.K, p.za, nb2, ra, st);
    split_operand(g.Bm, g.sbn, g.sbk, g.sbb, g.sbh, g.N, g.K, p.zb, nb2, rb, st);
    if (g.A2) {
        split_operand(g.A2, g.sam, g.sak, g.sab2, g.sah, g.M, g.K, p.za2, nb2, ra2, st);
        split_operand(g.Bm2, g.sbn, g.sbk, g.sbb2, g.sbh, g.N, g.K, p.zb2, nb2, rb2, st);
    }
    GemmRecArgs r{};
    const long hbA = (long)p.nck * g.M * 64, hbB = (long)p.nck * g.N * 64;      // bytes of one z of records
    r.a1 = ra; r.a1_hb = hbA; r.a1_bb = g.sab ? hbA * nb2 : 0;
    r.b1 = rb; r.b1_hb = hbB; r.b1_bb = g.sbb ? hbB * nb2 : 0;
    if (g.A2) {
        r.a2 = ra2; r.a2_hb = hbA; r.a2_bb = g.sab2 ? hbA * nb2 : 0;
        r.b2 = rb2; r.b2_hb = hbB; r.b2_bb = g.sbb2 ? hbB * nb2 : 0;
    }
    r.M = g.M; r.N = g.N; r.nst1 = p.nck; r.nst2 = g.A2 ? p.nck : 0; r.nb2 = nb2; r.Z = g.batch * nb2;
    r.C = g.C; r.scm = g.scm; r.scn = g.scn; r.scb = g.scb; r.sch = g.sch; r.alpha = g.alpha; r.beta = g.beta;
    r.ksplit = p.ksplit; r.part = part;
    if (p.tm == 4) launch_rec_tm<4>(r, st); else launch_rec_tm<2>(r, st);
    if (p.ksplit > 1) {
        const long MN = (long)g.M * g.N;
        hipLaunchKernelGGL(gemm_rec_reduce, dim3((unsigned)((MN + 255) / 256), r.Z), dim3(256), 0, st, part, p.ksplit, r.Z, nb2, g.M, g.N,
                           g.C, g.scm, g.scn, g.scb, g.sch, g.alpha, g.beta);
    }
}

}  // namespace loco
