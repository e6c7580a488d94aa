// Implicit-GEMM 3x3 / 1x1 convolution on v_mfma_f32_32x32x16_bf16 with SPLIT-bf16
// operands ("bf16x3"): every fp32 operand is carried as hi + lo bf16 halves and each
// product is three MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate).  The dropped
// lo*lo term and the rounding of lo are <= 2^-16 relative per product, i.e. the
// result is fp32-faithful to ~1.5e-5 while running on the 2.5 PF/s matrix pipe
// (effective peak 2500/3 = 833 TF/s vs 157 TF/s for the f32-input MFMA).
//
// Structure (same GEMM view as conv.hip: D[cout][pixel]): a workgroup owns MT couts
// x NT pixels; Cin is walked in chunks of 16 channels (one MFMA k-step).  The halo
// patch of a chunk is converted ONCE into LDS pixel records [hi k0-7|hi k8-15|lo k0-7|
// lo k8-15] (80-byte pitch: conflict-free ds_read_b128 with affine offsets) with the
// GroupNorm/SiLU forward, tangent or cotangent map applied in fp32 BEFORE the split;
// weights are staged one kernel row (3 taps) at a time from a pre-split, pre-swizzled
// global layout by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write); their
// 64-byte records are XOR-swizzled (chunk ^= (index>>2)&3) for the A-operand reads.
// One raw s_barrier per stage; the stage body is branch-free and fenced into
// scheduling regions so LDS reads, halo conversion and re-loads sit under the MFMAs
// (see the pipeline comment in the kernel and profiles/r01_conv_whatif.md).
//
// This header holds the kernel template and its per-(TAPS, MODE) launcher; conv_bf16_inst_*.hip instantiate
// disjoint subsets so the variants compile in parallel, conv_bf16.hip holds the tile heuristics and the dispatch.
#pragma once
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace loco {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));             // 16-byte global load from a 4-byte aligned address
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 f32x2_u __attribute__((aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));      // 16-byte operand fragment, reinterpreted per arithmetic

// Operand arithmetic (template parameter PR):
//   PR_BF16X3: every fp32 operand as hi + lo bf16, 3 MFMAs per product (fp32-faithful, 64-byte records
//              [hi k0-7|hi k8-15|lo k0-7|lo k8-15]);
//   PR_F16:    operands rounded to f16 (11 significant bits), ONE v_mfma_f32_32x32x16_f16 per product, 32-byte
//              records [k0-7|k8-15]: a third of the matrix work, half the LDS and weight traffic.
enum : int { PR_BF16X3 = 0, PR_F16 = 1 };

constexpr int BKC = 16;
constexpr bool LIVE_MASK = true;   // loads issued for chunks past the end of the range read one address instead of the last chunk again
constexpr bool EARLY_W = true;   // 3x3 stride-1 kernels: first two weight stages issued before the index setup (see conv_lowp_body)
constexpr int NDUMMY = 8;   // spare halo records per buffer: lanes without a halo item store there instead of branching

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence: ~10 VALU instructions fewer per element of the halo
// conversion of the forward (GroupNorm + SiLU) convs, 2.6 % per denoiser evaluation
__device__ __forceinline__ float sigmoidf2_(float y) { return __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

// byte offset of logical 16-byte chunk q inside the weight record of index p: 64-byte records (bf16x3) XOR the chunk
// with (p>>2)&3, 32-byte records (f16) with (p>>3)&1 -- either way the 16 lanes of a ds_read_b128 service group hit 16
// disjoint 4-dword bank groups
// Quotient of small non-negative integers (n < 2^22, d >= 1) by a float reciprocal and one correction step: 8 vector
// instructions instead of the ~35 of the generic 32-bit division (a tile's index setup has eight run-time divisions)
__device__ __forceinline__ int idiv_small(int n, int d) {
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    q += (r >= d) ? 1 : ((r < 0) ? -1 : 0);
    return q;
}
__device__ __forceinline__ void idivmod_small(int n, int d, int& q, int& r) { q = idiv_small(n, d); r = n - q * d; }

template <int PR>
__device__ __forceinline__ int rec_off(int p, int q) {
    if constexpr (PR == PR_F16) return p * 32 + ((q ^ ((p >> 3) & 1)) << 4);
    else return p * 64 + ((q ^ ((p >> 2) & 3)) << 4);
}
// Halo records use a PADDED pitch instead of the XOR swizzle: 80 bytes = 20 dwords (48 bytes = 12 dwords for the
// 32-byte f16 records), so 16 consecutive records hit 16 disjoint 4-dword bank groups (ds_read_b128 conflict-free) AND
// the offset stays affine in the record index -- a tap shift is then a wave-uniform addend, where the swizzle needed one
// precomputed VGPR offset per (tap, operand).
template <int PR> constexpr int halo_pitch() { return PR == PR_F16 ? 48 : 80; }
template <int PR> constexpr int rec_bytes() { return PR == PR_F16 ? 32 : 64; }
template <int PR>
__device__ __forceinline__ int hrec_off(int p, int q) { return p * halo_pitch<PR>() + (q << 4); }

// two fp32 values -> one dword of packed 16-bit operands (hi part) and, for bf16x3, the dword of their residuals
template <int PR>
__device__ __forceinline__ void cvt2(float a, float b, unsigned& hi, unsigned& lo) {
    if constexpr (PR == PR_F16) {
        _Float16 ha = (_Float16)a, hb = (_Float16)b;
        hi = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
        lo = 0;
    } else {
        // both halves of a dword by ONE v_cvt_pk_bf16_f32 each (scalar __bf16 casts compile to one conversion per value
        // plus shift / or packing: 11 vector instructions per pair instead of 6); the hi values go back to fp32 as the
        // dword's two 16-bit fields -- same roundings, same bits
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
        const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
        // the residuals as two scalar subtractions: left to itself the compiler packs them into one v_pk_add_f32, which
        // costs more beside MFMAs than the two v_sub_f32 it replaces (measured: 304.5 vs 302.8 ms per step)
        float da = a - ha;
        asm volatile("" : "+v"(da));
        const float db = b - hb;
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){da, db}, bf16x2_));
    }
}

template <int PR>
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
    cvt2<PR>(v[0], v[1], hi.x, lo.x);
    cvt2<PR>(v[2], v[3], hi.y, lo.y);
    cvt2<PR>(v[4], v[5], hi.z, lo.z);
    cvt2<PR>(v[6], v[7], hi.w, lo.w);
}

constexpr int max_halo(int NT, int taps, bool s2) {
    if (taps == 1) return NT;
    // TW = 32 (or the image width when smaller); worst case over the supported widths
    if (NT >= 128) {
        int th = NT / 32;
        return s2 ? (2 * th + 1) * 65 : (th + 2) * 34;
    }
    // NT = 64: an 8x8 output tile (8x8 images) or two rows of 32 (wider images: 4 x 34 halo, stride 2: 5 x 65)
    return s2 ? 5 * 65 : 4 * 34;
}

// Sums of N values per lane over groups of LW consecutive lanes by recursive halving (a reduce-scatter, then a plain butterfly for
// what is left): step M pairs lane l with l ^ M, each keeps one half of its values and adds the partner's -- N - 1 + log2(LW / N)
// shuffles instead of N log2(LW) (the epilogue statistics: 32 sums of a 128 x 256 tile over 64 lanes in 32 shuffles, not 192).
// Afterwards lane l holds, in av[0 .. rs_left), the group totals of the values i + sum_b bit_b(l) (N >> (b + 1)), b < rs_steps.
constexpr int rs_steps(int n, int lw) { int s = 0; while ((1 << s) < lw && (n >> s) > 1) ++s; return s; }
template <int N, int M, int LW>
__device__ __forceinline__ void rs_reduce(float* av, const int lane) {
    if constexpr (M < LW && N > 1) {
        constexpr int H = N / 2;
        const bool up = (lane & M) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float keep = up ? av[i + H] : av[i];
            const float send = up ? av[i] : av[i + H];
            av[i] = keep + __shfl_xor(send, M, 64);
        }
        rs_reduce<H, M * 2, LW>(av, lane);
    } else if constexpr (M < LW) {
#pragma unroll
        for (int i = 0; i < N; ++i) av[i] += __shfl_xor(av[i], M, 64);
        rs_reduce<N, M * 2, LW>(av, lane);
    }
}

// Epilogue of the low-precision conv kernel: the WM x WN waves (threads 0 .. WM*WN*64-1) write their accumulator tiles.
// D[row = cout][col = pixel]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// `stage_round(h, S)`: the wave's accumulator values of write-out round h (cout rows (wm * TM + h) * 32 ... + 31 of the tile) into
// the staging image S[WM * 32][NT] -- the one place that knows the accumulator layout (conv_pair_kernel.h passes its own)
// EPI: what of the epilogue's optional work is compiled in (its registers are live across the staging round trip either way).
// EPI_COT1: the norm-cotangent term of a 1x1 operator (ConvArgs::cot_d) -- the per-pixel 1x1 kernels only: the engine puts it
// into ResBlock shortcut operators; EPI_STATS: the statistics sinks (ConvArgs::st_part) -- not the DMA-fed GEMM, whose launches
// never carry one (conv_lowp_can_fuse_stats)
constexpr int EPI_COT1 = 1, EPI_STATS = 2;
template <int WM, int WN, int TM, int TN, int EPI = EPI_STATS, class StageRound>
__device__ __forceinline__ void conv_lowp_epilogue_staged(const ConvArgs& a, StageRound&& stage_round, unsigned char* smem_b, const int co0,
                                                          const int oy0, const int ox0, const int TW, const int tile_id, const int b,
                                                          const int split);
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_lowp_epilogue_direct(const ConvArgs& a, f32x16 (&acc)[TM][TN], const int co0, const int oy0,
                                                          const int ox0, const int TW, const int b, const int split);
template <int WM, int WN, int TM, int TN, int EPI = EPI_STATS>
__device__ __forceinline__ void conv_lowp_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], unsigned char* smem_b, const int co0,
                                                   const int oy0, const int ox0, const int TW, const int tile_id, const int b,
                                                   const int split) {
    constexpr int NT = WN * TN * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, khalf = lane >> 5, wm = wave / WN, wn = wave % WN;
    if (co0 + WM * TM * 32 <= a.Cout) {
        conv_lowp_epilogue_staged<WM, WN, TM, TN, EPI>(a, [&](const int h, float* S) {      // (h is a constant after unrolling)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    S[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * NT + (wn * TN + j) * 32 + l31] = acc[h][j][r];
        }, smem_b, co0, oy0, ox0, TW, tile_id, b, split);
        return;
    }
    conv_lowp_epilogue_direct<WM, WN, TM, TN>(a, acc, co0, oy0, ox0, TW, b, split);
}
template <int WM, int WN, int TM, int TN, int EPI, class StageRound>
__device__ __forceinline__ void conv_lowp_epilogue_staged(const ConvArgs& a, StageRound&& stage_round, unsigned char* smem_b, const int co0,
                                                          const int oy0, const int ox0, const int TW, const int tile_id, const int b,
                                                          const int split) {
    constexpr int NTHR = WM * WN * 64;
    constexpr bool COT1 = (EPI & EPI_COT1) != 0, STATS = (EPI & EPI_STATS) != 0;
    constexpr int MT = WM * TM * 32;
    constexpr int NT = WN * TN * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const long out_plane = (long)a.Hout * a.Wout;
    {
        // Whole cout tiles leave through LDS: in the accumulator layout a lane owns ONE pixel of 16 couts, i.e. 64 dword
        // stores per lane and tile, and the tile's write-out is bound by store ISSUE, not by bandwidth (512 wave-stores of
        // 256 bytes per workgroup).  Staged through the (now idle) operand buffers as S[cout][pixel], a lane reads back 4
        // consecutive pixels of one cout and the tile leaves in 16-byte stores: a quarter of the vector-memory
        // instructions for the same bytes, residual / accumulate reads likewise as 16-byte loads.  One round per
        // accumulator row block h (WM x 32 couts x NT pixels <= 64 KB).
        constexpr int SROWS = WM * 32, NQ = NT / 4, NTASK = (SROWS * NQ) / NTHR;
        static_assert((SROWS * NQ) % NTHR == 0 && (NQ & (NQ - 1)) == 0, "epilogue task split");
        float* const S = reinterpret_cast<float*>(smem_b);
        const bool part = a.nsplit > 1;
        float* const ob = part ? a.partial + (((long)split * a.B + b) * a.Cout) * out_plane : a.out + (long)b * a.out_bs;
        const float* const rb = (!part && a.res) ? a.res + (long)b * a.res_bs : nullptr;
        const float* const b2 = (!part && a.bias2) ? a.bias2 + (long)b * a.bias2_bs : nullptr;
        const float* const b1 = part ? nullptr : a.bias;
        const bool accu = !part && a.accumulate;
        const int twsh = 31 - __builtin_clz((unsigned)TW);      // (a power of two)
        // Tangent / cotangent group means of the finished tile (round 6, below): what they need of the primal under the round's rows
        // -- tangent: x of the output tensor (4 bytes per element); cotangent: the consuming norm's cached {S, xhat} records (8
        // bytes: no transcendental per element here) -- is requested at the top of each round, ahead of its LDS round trip, residual
        // loads and stores (shared by the probes of the tile on one XCD: served by its L2), and used after the round's stores
        const bool lin_st = STATS && (a.st_kind == ST_TAN || a.st_kind == ST_COT) && !part;
        const bool lin_cot = a.st_kind == ST_COT;
        // Norm-cotangent term (ConvArgs::cot_d; 1x1 operators): its operands -- the cotangent behind the norm (4 bytes per element),
        // the norm's {S, xhat} records (8, shared by the tile's probes) and the channel's {rstd m1, rstd m2} -- are requested at the
        // top of the round as well, into the registers the statistics' records use (a launch has one of the two)
        const bool cot_ep = COT1 && a.cot_d && !part && !lin_st;
        const float* const cot_db = cot_ep ? a.cot_d + (long)b * a.cot_d_bs : nullptr;
        const f32x2* const cot_tb = cot_ep ? reinterpret_cast<const f32x2*>(a.cot_tc + (long)b * a.cot_tc_bs) : nullptr;
        // Task q of a round = staging row q * RSTEP + row0, pixels 4 * quad0 ... + 3: the lane's (row0, quad0) are fixed, the task
        // adds a compile-time row count -- cout and tensor offset of a task are one per-lane base plus a wave-uniform term (scalar
        // arithmetic), not per-task registers held across the round
        constexpr int RSTEP = NTHR / NQ;
        static_assert(NTHR % NQ == 0 && 32 % RSTEP == 0, "epilogue task rows");
        const int row0 = tid / NQ, quad0 = tid & (NQ - 1);
        const unsigned pix0 = (unsigned)((oy0 + ((quad0 * 4) >> twsh)) * a.Wout + ox0 + ((quad0 * 4) & (TW - 1)));
        auto co_of = [&](int h, int q) -> int { return co0 + row0 + (((q * RSTEP) >> 5) * TM + h) * 32 + ((q * RSTEP) & 31); };
        auto off_of = [&](int h, int q) -> unsigned {
            const unsigned crow = (unsigned)((((q * RSTEP) >> 5) * TM + h) * 32 + ((q * RSTEP) & 31));      // compile-time after unrolling
            return (unsigned)(co0 + row0) * (unsigned)out_plane + pix0 + crow * (unsigned)out_plane;
        };
        constexpr int NV = 2 * TM * NTASK;                 // row sums per lane: {s1, s2} per (round, task)
        float av[STATS ? NV : 1];
#pragma unroll
        for (int h = 0; h < TM; ++h) {
            if (h > 0) {                                   // the previous round's read-back is done in every wave
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            stage_round(h, S);
            // (requested here: the round's accumulator registers are free, the loads fly under the LDS round trip)
            f32x4 xs[(STATS || COT1) ? NTASK : 1][2];      // tangent: [q][0] = x of 4 pixels; cotangent: {S0,x0,S1,x1}, {S2,x2,S3,x3}
            f32x4 cd[COT1 ? NTASK : 1];
            if constexpr (COT1) {
                if (cot_ep) {
#pragma unroll
                    for (int q = 0; q < NTASK; ++q) {
                        const unsigned o = off_of(h, q);
                        cd[q] = *reinterpret_cast<const f32x4*>(cot_db + o);
                        const f32x4* sp4 = reinterpret_cast<const f32x4*>(a.cot_sx + o);
                        xs[q][0] = sp4[0];
                        xs[q][1] = sp4[1];
                    }
                }
            }
            if constexpr (STATS) if (lin_st) {
#pragma unroll
                for (int q = 0; q < NTASK; ++q) {
                    const unsigned o = off_of(h, q);
                    if (lin_cot) {
                        const f32x4* sp4 = reinterpret_cast<const f32x4*>(a.st_sx + o);
                        xs[q][0] = sp4[0];
                        xs[q][1] = sp4[1];
                    } else {
                        xs[q][0] = *reinterpret_cast<const f32x4*>(a.st_x + o);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f32x4 v[NTASK], rv[NTASK];
#pragma unroll
            for (int q = 0; q < NTASK; ++q) v[q] = *reinterpret_cast<const f32x4*>(&S[(q * RSTEP + row0) * NT + quad0 * 4]);
            if (rb || accu) {
#pragma unroll
                for (int q = 0; q < NTASK; ++q) {
                    f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
                    if (rb) t4 = a.res_scale * *reinterpret_cast<const f32x4*>(rb + off_of(h, q));
                    if (accu) t4 += *reinterpret_cast<const f32x4*>(ob + off_of(h, q));
                    rv[q] = t4;
                }
#pragma unroll
                for (int q = 0; q < NTASK; ++q) v[q] += rv[q];
            }
            if constexpr (COT1) {
                if (cot_ep) {      // out += S d - (rstd m1 + xhat rstd m2): a task row is one channel
#pragma unroll
                    for (int q = 0; q < NTASK; ++q) {
                        const f32x2 tc2 = cot_tb[co_of(h, q)];      // (one per channel: L2, not worth eight register pairs across the round trip)
                        const float c1 = tc2[0], c2 = tc2[1];
                        v[q][0] += fmaf(xs[q][0][0], cd[q][0], -fmaf(xs[q][0][1], c2, c1));
                        v[q][1] += fmaf(xs[q][0][2], cd[q][1], -fmaf(xs[q][0][3], c2, c1));
                        v[q][2] += fmaf(xs[q][1][0], cd[q][2], -fmaf(xs[q][1][1], c2, c1));
                        v[q][3] += fmaf(xs[q][1][2], cd[q][3], -fmaf(xs[q][1][3], c2, c1));
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NTASK; ++q) {
                float add = 0.f;
                if (b1) add += b1[co_of(h, q)];
                if (b2) add += b2[co_of(h, q)];
                v[q] += add;
                __builtin_nontemporal_store(v[q], reinterpret_cast<f32x4*>(ob + off_of(h, q)));   // streamed once: keep L2 for the shared primal cache / weights
            }
            if constexpr (STATS) if (lin_st) {      // the lane's share of the round's row sums (see behind the round loop)
#pragma unroll
                for (int q = 0; q < NTASK; ++q) {
                    const f32x4 vv = v[q];
                    float s1, s2;
                    if (!lin_cot) {
                        const f32x4 xv = xs[q][0];
                        s1 = (vv[0] + vv[1]) + (vv[2] + vv[3]);
                        s2 = fmaf(xv[0], vv[0], xv[1] * vv[1]) + fmaf(xv[2], vv[2], xv[3] * vv[3]);
                    } else {       // z = S g (the 1 / rstd of the group is applied by gn_lin_fused_finalize)
                        const float z0 = xs[q][0][0] * vv[0], z1 = xs[q][0][2] * vv[1], z2 = xs[q][1][0] * vv[2], z3 = xs[q][1][2] * vv[3];
                        s1 = (z0 + z1) + (z2 + z3);
                        s2 = fmaf(xs[q][0][1], z0, xs[q][0][3] * z1) + fmaf(xs[q][1][1], z2, xs[q][1][3] * z3);
                    }
                    av[(h * NTASK + q) * 2] = s1;
                    av[(h * NTASK + q) * 2 + 1] = s2;
                }
            }
            // Forward GroupNorm statistics of the finished tile for the norm that consumes this tensor (kernels.h
            // ConvArgs::st_part): the NQ lanes of a task row hold one cout over the tile's NT pixels; sums about a pivot
            // inside the row's data (its first value in this tile: no cancellation of sum x^2 - n mean^2 when |mean| >> std),
            // butterfly over the row's lanes, one lane writes {mean, M2} of the row tile.  (The tangent / cotangent means:
            // behind the round loop.  Round 4 tried them here from the {S, xhat} records of the output tile -- 8 more bytes per
            // element read in the latency-exposed round, 9-26 us per launch against the 16-21 us of the standalone pass.)
            if (STATS && a.st_kind == ST_FWD && !part) {
                const int ntile = (a.Hout * a.Wout) / NT;
                float* const sp = a.st_part + (long)b * a.Cout * ntile * 2;
#pragma unroll
                for (int q = 0; q < NTASK; ++q) {
                    const float pivot = __shfl(v[q][0], lane & ~(NQ - 1) & 63, 64);
                    const float d0 = v[q][0] - pivot, d1 = v[q][1] - pivot, d2 = v[q][2] - pivot, d3 = v[q][3] - pivot;
                    float s1 = (d0 + d1) + (d2 + d3);
                    float s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
                    for (int m = 1; m < (NQ < 64 ? NQ : 64); m <<= 1) {
                        s1 += __shfl_xor(s1, m, 64);
                        s2 += __shfl_xor(s2, m, 64);
                    }
                    const float m = s1 * (1.0f / NT);
                    const int t = q * NTHR + tid;
                    if ((t & (NQ - 1)) == 0)
                        *reinterpret_cast<f32x2*>(sp + ((long)co_of(h, q) * ntile + tile_id) * 2) = f32x2{pivot + m, s2 - s1 * m};
                }
            }
        }
        // Tangent / cotangent group means of the finished tile for the norm that consumes this tensor (round 6; kernels.h
        // ConvArgs::st_part).  Tangent: raw {sum d, sum x d} per cout row (norm-independent: gn_lin_fused_finalize forms
        // mean(xhat d) = rstd (sum x d - mean sum d) / n in double); cotangent: {sum S g, sum xhat S g} from the consuming norm's
        // primal cache (gn_lin_fused_finalize divides by rstd: z = (sc / rstd) act'(y) g = S g / rstd, what gn_tstats_partial<1> sums).  Replaces that kernel's pass over the tensor (reference op: the
        // GroupNorm inside jvp / vjp, models/ddpm/diffusion.py:810-811 under edit.py:2455,2479).  Every round left the lane's share
        // of its row sums in av[]; all 2 x TM x NTASK of them go through ONE recursive-halving reduction over the row's NQ lanes.
        if constexpr (STATS) if (lin_st) {
            constexpr int LW = NQ < 64 ? NQ : 64, STEPS = rs_steps(NV, LW), LEFT = NV >> STEPS;
            static_assert((NV & (NV - 1)) == 0, "row sums per lane");
            rs_reduce<NV, 1, LW>(av, lane);
            const int gl = lane & (LW - 1);
            if (gl < (1 << STEPS)) {
                int base = 0;
#pragma unroll
                for (int sb = 0; sb < STEPS; ++sb) base += ((gl >> sb) & 1) * (NV >> (sb + 1));
                const int ntile = (a.Hout * a.Wout) / NT;
                float* const sp = a.st_part + (long)b * a.Cout * ntile * 2;
#pragma unroll
                for (int i = 0; i < LEFT; ++i) {
                    const int idx = base + i, hq = idx >> 1;
                    const int h = hq / NTASK, q = hq - h * NTASK;
                    const int row = (q * NTHR + tid) / NQ;
                    const int co = co0 + ((row >> 5) * TM + h) * 32 + (row & 31);
                    sp[((long)co * ntile + tile_id) * 2 + (idx & 1)] = av[i];
                }
            }
        }
    }
}
// partial cout tiles (Cout not a multiple of the tile): straight from the accumulator layout, one dword per store
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_lowp_epilogue_direct(const ConvArgs& a, f32x16 (&acc)[TM][TN], const int co0, const int oy0,
                                                          const int ox0, const int TW, const int b, const int split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, khalf = lane >> 5, wm = wave / WN, wn = wave % WN;
    const long out_plane = (long)a.Hout * a.Wout;
    const bool full_co = false;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int p = (wn * TN + j) * 32 + l31;
        int ty = p / TW, tx = p - ty * TW;
        int oy = oy0 + ty, ox = ox0 + tx;
        const unsigned pix = (unsigned)(oy * a.Wout + ox);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cob = co0 + (wm * TM + i) * 32 + 4 * khalf;
            if (a.nsplit > 1) {
                float* pb = a.partial + (((long)split * a.B + b) * a.Cout) * out_plane;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = cob + (r & 3) + 8 * (r >> 2);
                    if (full_co || co < a.Cout) pb[(long)co * out_plane + pix] = acc[i][j][r];
                }
            } else {
                float* ob = a.out + (long)b * a.out_bs;
                const float* rb = a.res ? a.res + (long)b * a.res_bs : nullptr;
                const float* b2 = a.bias2 ? a.bias2 + (long)b * a.bias2_bs : nullptr;
                if (rb || a.accumulate) {
                    // The residual may alias the output (nin shortcut written in place), so the compiler cannot
                    // move a residual load above the previous store: gather the 16 residual / accumulate values of
                    // the tile first (each thread only touches its own elements), then add and store.
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int co = cob + (r & 3) + 8 * (r >> 2);
                        const bool ok = full_co || co < a.Cout;
                        const long off = (long)(ok ? co : a.Cout - 1) * out_plane + pix;
                        float t = 0.f;
                        if (rb) t = a.res_scale * rb[off];
                        if (a.accumulate) t += ob[off];
                        rv[r] = t;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int co = cob + (r & 3) + 8 * (r >> 2);
                        if (!(full_co || co < a.Cout)) continue;
                        float v = acc[i][j][r] + rv[r];
                        if (a.bias) v += a.bias[co];
                        if (b2) v += b2[co];
                        __builtin_nontemporal_store(v, &ob[(long)co * out_plane + pix]);   // streamed once: keep L2 for the shared primal cache / weights (2 ms per step)
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int co = cob + (r & 3) + 8 * (r >> 2);
                        if (!(full_co || co < a.Cout)) continue;
                        float v = acc[i][j][r];
                        if (a.bias) v += a.bias[co];
                        if (b2) v += b2[co];
                        __builtin_nontemporal_store(v, &ob[(long)co * out_plane + pix]);   // streamed once: keep L2 for the shared primal cache / weights (2 ms per step)
                    }
                }
            }
        }
    }
}

// The same write-out (whole cout tiles, no split-K) in rounds of WM x RPW cout rows through a staging area S that the operand
// buffers do not overlap: the persistent kernel (PHASE 3 of conv_lowp_body) keeps the next probe's first operands in LDS while
// the finished tile leaves.  RPW = 16: 2 x 16 x 256 x 4 B = 32 KB per round.
template <int WM, int WN, int TM, int TN, int RPW>
__device__ __forceinline__ void conv_lowp_epilogue_rounds(const ConvArgs& a, f32x16 (&acc)[TM][TN], float* const S, const int co0,
                                                          const int oy0, const int ox0, const int TW, const int tile_id, const int b) {
    constexpr int NTHR = WM * WN * 64, NT = WN * TN * 32, NQ = NT / 4;
    constexpr int SROWS = WM * RPW, NTASK = (SROWS * NQ) / NTHR, NRB = RPW / 8;
    static_assert((RPW == 8 || RPW == 16 || RPW == 32) && (SROWS * NQ) % NTHR == 0 && NQ == 64, "round geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const unsigned out_plane = (unsigned)(a.Hout * a.Wout);
    float* const ob = a.out + (long)b * a.out_bs;
    const float* const rb = a.res ? a.res + (long)b * a.res_bs : nullptr;
    const float* const b2 = a.bias2 ? a.bias2 + (long)b * a.bias2_bs : nullptr;
    const bool accu = a.accumulate != 0;
    const int twsh = TW == 32 ? 5 : (TW == 16 ? 4 : 3);
    const int ntile = (a.Hout * a.Wout) / NT;
    float* const sp = (a.st_kind == ST_FWD) ? a.st_part + (long)b * a.Cout * ntile * 2 : nullptr;
#pragma unroll
    for (int h = 0; h < TM; ++h) {
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += NRB) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < NRB; ++g)
#pragma unroll
                    for (int r3 = 0; r3 < 4; ++r3)
                        S[(wm * RPW + g * 8 + 4 * khalf + r3) * NT + (wn * TN + j) * 32 + l31] = acc[h][j][(g0 + g) * 4 + r3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f32x4 v[NTASK];
            unsigned off[NTASK];
            int cos_[NTASK];
#pragma unroll
            for (int q = 0; q < NTASK; ++q) {
                const int t = q * NTHR + tid;
                const int srow = t / NQ, quad = t & (NQ - 1);
                v[q] = *reinterpret_cast<const f32x4*>(&S[srow * NT + quad * 4]);
                const int co = co0 + ((srow / RPW) * TM + h) * 32 + g0 * 8 + (srow % RPW);
                const int p = quad * 4;
                cos_[q] = co;
                off[q] = (unsigned)co * out_plane + (unsigned)((oy0 + (p >> twsh)) * a.Wout + ox0 + (p & (TW - 1)));
            }
            if (rb || accu) {
                f32x4 rv[NTASK];
#pragma unroll
                for (int q = 0; q < NTASK; ++q) {
                    f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
                    if (rb) t4 = a.res_scale * *reinterpret_cast<const f32x4*>(rb + off[q]);
                    if (accu) t4 += *reinterpret_cast<const f32x4*>(ob + off[q]);
                    rv[q] = t4;
                }
#pragma unroll
                for (int q = 0; q < NTASK; ++q) v[q] += rv[q];
            }
#pragma unroll
            for (int q = 0; q < NTASK; ++q) {
                float add = 0.f;
                if (a.bias) add += a.bias[cos_[q]];
                if (b2) add += b2[cos_[q]];
                v[q] += add;
                __builtin_nontemporal_store(v[q], reinterpret_cast<f32x4*>(ob + off[q]));
            }
            if (sp) {      // forward GroupNorm statistics of the finished tile: see conv_lowp_epilogue
#pragma unroll
                for (int q = 0; q < NTASK; ++q) {
                    const float pivot = __shfl(v[q][0], 0, 64);
                    const float d0 = v[q][0] - pivot, d1 = v[q][1] - pivot, d2 = v[q][2] - pivot, d3 = v[q][3] - pivot;
                    float s1 = (d0 + d1) + (d2 + d3);
                    float s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) {
                        s1 += __shfl_xor(s1, m, 64);
                        s2 += __shfl_xor(s2, m, 64);
                    }
                    const float m = s1 * (1.0f / NT);
                    if (lane == 0)
                        *reinterpret_cast<f32x2*>(sp + ((long)cos_[q] * ntile + tile_id) * 2) = f32x2{pivot + m, s2 - s1 * m};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the round's read-back is done in every wave
            __builtin_amdgcn_s_barrier();
        }
    }
}

// PHASE 0: the whole kernel.  K-concatenated pair (ResBlock conv2 + its 1x1 shortcut on the block input, one accumulator
// tile, one write-out): PHASE 1 = the first operator's stage loop only (accumulators zeroed, no epilogue), PHASE 2 = the
// second operator's stage loop on top of the same accumulators, then the epilogue.
// PHASE 3 (round 5): PERSISTENT over the probes of one (pixel tile, cout tile).  The workgroup walks probes b, b + G, b + 2G, ...
// (G = a.pers_groups workgroups share a tile) as ONE continuous stream of chunks: the halo parts and weight stages of the next
// probe's first chunk are loaded / converted under the current probe's last chunk exactly like the next chunk of the same probe
// (only the wave-uniform base pointers change between probes of one tile: index setup once per workgroup, the shared primal-cache
// patch stays in this CU's caches), and the finished tile leaves through a staging area of its own (32 KB rounds behind the
// operand buffers) so the prefetched operands survive the write-out.  r05 stamps of the 128 -> 128 tangent conv: index setup
// 2.4 k + prologue 9.0 k of a tile's 94.9 k cycles, every tile.  Needs an even number of chunks (buffer parities restart per probe),
// whole cout tiles and no split-K.
template <int PR, int TAPS, int WM, int WN, int TM, int TN, int MODE, int STG, int PHASE = 0>
__device__ __forceinline__ void conv_lowp_body(const ConvArgs& a, f32x16 (&acc)[TM][TN]) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int HP = halo_pitch<PR>();
    constexpr int RB = rec_bytes<PR>();               // bytes of one operand record (16 k-values of one pixel / cout)
    constexpr int NPC = RB / 16;                      // 16-byte pieces per record
    constexpr int MT = WM * TM * 32;
    constexpr int NT = WN * TN * 32;
    constexpr int KS = (TAPS == 9) ? 3 : 1;
    constexpr int NTS = KS;                          // taps per weight stage (one kernel row)
    constexpr int NROW = (TAPS == 9) ? 3 : 1;        // weight stages per channel chunk
    constexpr bool NEEDP = (MODE == CM_TAN_SILU || MODE == CM_COT_SILU);
    constexpr bool PERS = (PHASE == 3);
    // GEN: general per-pixel staging (stride 2, upsample, zero insertion, partial channel chunks, caller-owned
    // tensors); !GEN: 16-byte loads of 4 consecutive pixels from padded arena tensors (stride-1 convs)
    // STG 3: vector staging in the compact LDS layout of the 128 x 128 tile that fits TWO workgroups per CU (<= 81 920 B: two
    //        weight buffers, i.e. no cross-stage operand prefetch, and one 64-byte dump record per halo buffer);
    // STG 0: vector staging; 1: per-pixel staging, stride 1 (also upsample / zero-insert / partial chunks /
    // caller-owned tensors); 2: per-pixel staging sized for the stride-2 halo
    constexpr bool GEN = (STG == 1 || STG == 2);
    constexpr bool COMPACT = (STG == 3);
    constexpr int NITEM = GEN ? (2 * max_halo(NT, TAPS, STG == 2) + NTHR - 1) / NTHR : 1;
    constexpr int WTOT = NTS * MT * NPC;             // 16-byte pieces of one weight stage
    constexpr int NWV = (WTOT + NTHR - 1) / NTHR;
    // cross-stage operand prefetch (see the stage loop): 3x3 stride-1 variants; the stride-2 halo leaves no LDS for a
    // third weight buffer, and the 1x1 convs are HBM-bound
    constexpr bool XPF = (TAPS == 9 && STG != 2 && STG != 3);
    constexpr int NWB = XPF ? 3 : 2;                 // weight stage buffers
    // 1x1 operators on the per-pixel path: a stage is only TM*TN MFMA groups long, far shorter than a memory latency, so
    // the operands of DEEP_D chunks are kept in flight in a register ring (pixels AND weights by plain loads: the
    // compiler then counts vmcnt per ring slot by itself, which it cannot do past an LDS-DMA) -- see the stage loop
    constexpr bool DEEP1 = (TAPS == 1 && STG == 1 && !NEEDP);
    constexpr int DEEP_D = 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
#ifdef LOCO_DUAL_STAMP
    // diagnostics build only (tests/diag/lowp_stamps.py): s_memtime at the phase boundaries of every workgroup, wave 0 lane 0
#define LP_STAMP(i) do { if (threadIdx.x == 0 && PHASE == 0) reinterpret_cast<unsigned long long*>(a.partial)[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LP_STAMP(i) do { } while (0)
#endif
    LP_STAMP(0);
    // double-buffered: weight stage s lives in W buffer s&1, the halo of chunk c in H buffer c&1
    constexpr int WBYTES = NTS * MT * RB;
    unsigned char* const Wsb = smem_b;               // 2 x [NTS][MT] records
    unsigned char* Hsb;                              // 2 x [halo_sz] records (set below, needs halo_sz)
    unsigned char* Ws = smem_b;                      // W / H buffer the operand reads and the halo conversion address
    unsigned char* Hs = smem_b;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int S = a.stride;
    // Pixel tile: TW x NT / TW pixels, TW = 32 columns (or the image width when smaller) -- except for a pointwise operator at
    // stride 1, which has no halo to keep compact: its tile is NT CONSECUTIVE pixels of the plane (whole image rows, or a run
    // inside one), so every channel of the tile is one contiguous run for the staging loads, the write-out and the epilogue's
    // operand streams instead of NT / 32 segments of 128 bytes (same bits; these launches are bound by memory: -3 ... -4 %)
    // (image widths that are a power of two: the run then divides the tile or the row)
    const int TW = (TAPS == 1 && PHASE == 0 && a.stride == 1 && !a.upsample && !a.zins && (a.Wout & (a.Wout - 1)) == 0)
                       ? (a.Wout < NT ? a.Wout : NT) : (a.Wout < 32 ? a.Wout : 32);
    const int TH = NT / TW;
    const int tiles_x = a.Wout / TW;
    // Block order: the probes (and K-splits) of one (pixel tile, cout tile) are adjacent in dispatch order
    // and land on the same XCD, so the shared primal {S, xhat} cache and the weights of the tile are
    // served from that XCD's L2 instead of HBM once per probe (blocks L and L+8 share an XCD).
    int tile_id, cot_id, zid;
    {
        const int ntile = (a.Hout * a.Wout) / NT, ncot = (a.Cout + MT - 1) / MT, Z = PERS ? a.pers_groups : a.B * a.nsplit;
        const int NTC = ntile * ncot, L = blockIdx.x;
        int T = 0;
        if ((ntile & 7) == 0) {
            // same XCD (L mod 8), adjacent in dispatch order: first the cout tiles of one (pixel tile, probe) -- they
            // read the same input patch -- then the other probes of that pixel tile (same primal cache and weights)
            int q = L >> 3, q2, t8;
            idivmod_small(q, ncot, q2, cot_id);
            idivmod_small(q2, Z, t8, zid);
            tile_id = t8 * 8 + (L & 7);
        } else {
            if ((NTC & 7) == 0) { int q = L >> 3, t8; idivmod_small(q, Z, t8, zid); T = t8 * 8 + (L & 7); }
            else idivmod_small(L, Z, T, zid);
            idivmod_small(T, ntile, cot_id, tile_id);
        }
        // integer division runs on the VALU: pin the (wave-uniform) results back into SGPRs so everything derived
        // from them (chunk range, tile origin, batch bases) is scalar arithmetic and saddr-form addressing
        tile_id = __builtin_amdgcn_readfirstlane(tile_id);
        cot_id = __builtin_amdgcn_readfirstlane(cot_id);
        zid = __builtin_amdgcn_readfirstlane(zid);
    }
    int ty0_, tx0_;
    idivmod_small(tile_id, tiles_x, ty0_, tx0_);
    const int oy0 = __builtin_amdgcn_readfirstlane(ty0_ * TH);
    const int ox0 = __builtin_amdgcn_readfirstlane(tx0_ * TW);
    const int co0 = cot_id * MT;
    // PERS: zid = the workgroup's group = its first probe; it walks np probes, pstep apart
    int zb_, zs_;
    idivmod_small(zid, a.nsplit, zb_, zs_);
    const int b = PERS ? zid : __builtin_amdgcn_readfirstlane(zb_);
    const int split = PERS ? 0 : __builtin_amdgcn_readfirstlane(zs_);
    const int pstep = PERS ? a.pers_groups : 1;
    const int np = PERS ? (a.B - b + pstep - 1) / pstep : 1;
    int pq = 0;                                       // probe of the walk the stage loop is multiplying (PERS)

    if constexpr (XPF && PHASE != 2 && EARLY_W) {
        // The first two weight stages leave BEFORE the per-thread index setup below (a dozen run-time integer divisions,
        // ~1 us): their L2 -> LDS latency then runs under it instead of after it.  Same addresses as dma_w() further down.
        const int nchunks_ = (a.Cin + BKC - 1) / BKC;
        const int cps_ = __builtin_amdgcn_readfirstlane((nchunks_ + a.nsplit - 1) / a.nsplit);
        const int cbeg_ = split * cps_;
        const int cend_ = (cbeg_ + cps_ < nchunks_) ? cbeg_ + cps_ : nchunks_;
        if (cend_ > cbeg_) {
            typedef __attribute__((address_space(3))) unsigned char lds_u8_;
            typedef const __attribute__((address_space(1))) unsigned char glb_u8_;
            const int wpitch_ = (a.Cout + 31) & ~31;
            const unsigned char* wbase0 = reinterpret_cast<const unsigned char*>(a.wb) +
                                          (unsigned)(cbeg_ * TAPS) * ((unsigned)wpitch_ * (unsigned)RB);
#pragma unroll
            for (int st_ = 0; st_ < 2; ++st_)              // kernel rows 0 and 1 of the first chunk
#pragma unroll
                for (int i = 0; i < NWV; ++i) {
                    int e = tid + i * NTHR;
                    if (e >= WTOT) e = WTOT - 1;
                    const int tap = e / (MT * NPC), rem = e - tap * (MT * NPC);
                    int rec = co0 + rem / NPC;
                    if (rec >= wpitch_) rec = wpitch_ - 1;
                    const unsigned rel = (unsigned)(((st_ * NTS + tap) * wpitch_ + rec) * NPC + (rem % NPC)) * 16u;
                    const int e0 = wave * 64 + i * NTHR;
                    if ((WTOT % NTHR) == 0 || e0 < WTOT)
                        __builtin_amdgcn_global_load_lds((glb_u8_*)(wbase0 + rel), (lds_u8_*)(smem_b + st_ * WBYTES + e0 * 16), 16, 0, 0);
                }
        }
    }
    const int halo_w = (TW - 1) * S + KS;
    const int halo_h = (TH - 1) * S + KS;
    const int halo_sz = halo_h * halo_w;
    // + dump records for the lanes without a halo item (compact layout: ONE, cut to the 64 bytes a record's data takes)
    const int HBYTES = COMPACT ? halo_sz * HP + RB : (halo_sz + NDUMMY) * HP;
    Hsb = smem_b + NWB * WBYTES;

    const int LH = (a.upsample || a.zins) ? a.Hin * 2 : a.Hin;
    const int LW = (a.upsample || a.zins) ? a.Win * 2 : a.Win;
    const long in_plane = (long)a.Hin * a.Win;

    // staging items of this thread: (octet of 8 channels, halo position).  Loads are issued
    // UNCONDITIONALLY from clamped addresses (uniform base + 32-bit per-lane offset) and masked
    // afterwards: per-element branches around loads serialise them (one vmcnt(0) per element).
    int ipos[NITEM], ioct[NITEM];
    unsigned ivoff[NITEM];
    bool ival[NITEM];
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
        int it = tid + i * NTHR;
        int oct, pos;
        idivmod_small(it, halo_sz, oct, pos);
        int off = -1;
        if (it < 2 * halo_sz) {
            int hy, hx;
            idivmod_small(pos, halo_w, hy, hx);
            int Y = oy0 * S - a.pad + hy, X = ox0 * S - a.pad + hx;
            if (Y >= 0 && Y < LH && X >= 0 && X < LW) {
                if (a.upsample) off = (Y >> 1) * a.Win + (X >> 1);
                else if (a.zins) off = ((Y | X) & 1) ? -1 : (Y >> 1) * a.Win + (X >> 1);
                else off = Y * a.Win + X;
            }
        } else {
            pos = -1; oct = 0;
        }
        ipos[i] = pos; ioct[i] = oct;
        ival[i] = off >= 0;
        ivoff[i] = (unsigned)(oct * 8 * (int)in_plane + (off >= 0 ? off : 0));
    }

    // ---- vector staging item of this thread (!GEN): 4 consecutive halo pixels x 4 channels (quarter chunk q4)
    const int nseg = (halo_w + 3) >> 2;
    int v_q4 = 0, v_pos0 = -1, v_voff = 0, v_cnt = 0;
    unsigned v_pm = 0;                       // per-pixel validity bits
    int v_rec[4] = {0, 0, 0, 0};             // LDS record of each of the 4 pixels (a dump record when not staged)
    unsigned v_goff = 0;                     // byte offset of the item's first pixel / channel from (chunk base - 16 floats)
    if constexpr (!GEN) {
        const int per_q = halo_h * nseg;
        // Lane order of the staging items.  The LDS stores of one pixel index put a dword or two into each record; the
        // record pitch (80 / 48 bytes) sends them to bank 8*row + 16*seg + 4*octet + 2*half (mod 32), so only the low bits
        // of (row, seg, octet, half) can spread a half-wave over banks.  Default order below: lane bits = {seg & 3, row & 1,
        // half, octet}: 16 distinct bank pairs per half-wave (2-way conflict) while 4 consecutive segments (64 bytes of
        // one image row) stay adjacent lanes for the global loads.  The plain order (segments fastest, then rows, then
        // channel quarters) is 8-way conflicted; it remains the fallback when the padded index space does not fit the
        // workgroup.
        const int nsg4 = (nseg + 3) >> 2, nhy2 = (halo_h + 1) >> 1;
        bool have = false;
        int hy = 0, sg = 0;
        if (32 * nsg4 * nhy2 <= NTHR) {
            const int hi = tid >> 5;                       // (seg >> 2, row >> 1)
            v_q4 = (tid >> 3) & 3;
            int hq_, hr_;
            idivmod_small(hi, nsg4, hq_, hr_);
            sg = hr_ * 4 + (tid & 3);
            hy = hq_ * 2 + ((tid >> 2) & 1);
            have = hi < nsg4 * nhy2 && sg < nseg && hy < halo_h;
            if (!have) v_q4 = 0;
        } else if (tid < 4 * per_q) {
            int rem;
            idivmod_small(tid, per_q, v_q4, rem);
            idivmod_small(rem, nseg, hy, sg);
            have = true;
        }
        if (have) {
            int Y = oy0 - a.pad + hy, X0 = ox0 - a.pad + 4 * sg;
            bool rowok = (Y >= 0 && Y < a.Hin);
            v_pos0 = hy * halo_w + 4 * sg;
            v_cnt = halo_w - 4 * sg < 4 ? halo_w - 4 * sg : 4;
            v_voff = rowok ? Y * a.Win + X0 : 0;          // may be -1 / run 3 floats past the plane: arenas are padded
#pragma unroll
            for (int pxi = 0; pxi < 4; ++pxi)
                if (rowok && X0 + pxi >= 0 && X0 + pxi < a.Win && pxi < v_cnt) v_pm |= 1u << pxi;
        }
#pragma unroll
        for (int pxi = 0; pxi < 4; ++pxi)
            v_rec[pxi] = (v_pos0 >= 0 && pxi < v_cnt) ? v_pos0 + pxi : (COMPACT ? halo_sz : halo_sz + (tid & (NDUMMY - 1)));
        // the row start may sit one float before the plane (left border): bias by 16 floats so the offset stays >= 0
        v_goff = (unsigned)(((long)v_q4 * 4 * in_plane + v_voff + 16) * 4);
    }

    int hoff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int p = (wn * TN + j) * 32 + l31;
        int ty, tx;
        idivmod_small(p, TW, ty, tx);
        hoff[j] = ty * S * halo_w + tx * S;
    }
    int hbyte[TN];                           // byte offset of the lane's hi operand chunk inside a halo buffer
#pragma unroll
    for (int j = 0; j < TN; ++j) hbyte[j] = hrec_off<PR>(hoff[j], khalf);
    // A-operand (weight) record offsets inside one tap block: cout-local index fixed per lane
    int aoff_hi[TM], aoff_lo[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int p = (wm * TM + i) * 32 + l31;
        aoff_hi[i] = rec_off<PR>(p, khalf);
        aoff_lo[i] = PR == PR_F16 ? 0 : rec_off<PR>(p, 2 + khalf);
    }

    if constexpr (PHASE != 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }

    const int nchunks = (a.Cin + BKC - 1) / BKC;
    const int cps = __builtin_amdgcn_readfirstlane((nchunks + a.nsplit - 1) / a.nsplit);
    const int cbeg = split * cps;
    const int cend = (cbeg + cps < nchunks) ? cbeg + cps : nchunks;

    const float* inb = a.in + (long)b * a.in_bs;
    const float2* sxb = NEEDP ? a.sx : nullptr;                     // primal (S, xhat) cache, B = 1
    const float* scb = (MODE == CM_GN_SILU || MODE == CM_GN || MODE == CM_GN_GELU) ? a.sc + (long)b * a.scsh_bs : nullptr;
    const float* shb = (MODE == CM_GN_SILU || MODE == CM_GN || MODE == CM_GN_GELU) ? a.sh + (long)b * a.scsh_bs : nullptr;
    const int wpitch = (a.Cout + 31) & ~31;                         // records per tap in the global layout
    const uint4* wg = reinterpret_cast<const uint4*>(a.wb);

    float hv[NITEM][8];
    float2 pv[NEEDP ? NITEM : 1][8];
    // per-channel constants of the chunk in flight: wave-uniform (scalar loads), selected per lane by octet
    float cA[(MODE == CM_NONE) ? 1 : 16], cB[(MODE == CM_NONE) ? 1 : 16];
    const float* tcb = NEEDP ? a.tc + (long)b * a.tc_bs : nullptr;   // {m1,m2} (tangent) or {rstd*m1, rstd*m2} (cotangent)

    // !GEN: the item (4 channels x 4 pixels) is staged in NPART parts of KP channels, one part per weight stage, so
    // only KP channels are live in registers at a time (3x3: 2 parts of 2 channels; 1x1: the whole item).  Parts
    // split the CHANNELS, not the pixels: every load stays a full 16-byte run of 4 pixels.
    constexpr int NPART = (!GEN && NROW == 3) ? 2 : 1;
    constexpr int KP = 4 / NPART;
    // The staged values stay whole 16-byte vectors from the load to their last use (components are sub-register
    // reads): split into scalars, the register allocator parks the load results in freed fragment registers and copies
    // single dwords out behind a vmcnt wait, which exposes the memory latency in every stage.
    struct HaloRegs {
        f32x4 dq[GEN ? 1 : KP];                       // !GEN: [channel of the part] x 4 pixels
        f32x4 sq[(!GEN && NEEDP) ? KP : 1][2];        // {S, xhat} of those pixels: {S0,x0,S1,x1}, {S2,x2,S3,x3}
        f32x4 cq[(!GEN && MODE != CM_NONE) ? (KP == 4 ? 2 : 1) : 1];   // per-channel constants, see cq_a / cq_b
    };
    // constants of channel kk of the part: tangent / cotangent modes load interleaved {a, b} pairs; the forward modes load
    // the a's (scale) and the b's (shift) as two runs -- KP == 2: one vector {a0,a1,b0,b1}; KP == 4: {a0..a3}, {b0..b3}
    auto cq_a = [&](const HaloRegs& R, int kk) -> float {
        if constexpr (GEN || MODE == CM_NONE) return 0.f;
        else if constexpr (NEEDP) { if constexpr (KP == 4) return R.cq[kk >> 1][(kk & 1) * 2]; else return R.cq[0][kk * 2]; }
        else return R.cq[0][kk];
    };
    auto cq_b = [&](const HaloRegs& R, int kk) -> float {
        if constexpr (GEN || MODE == CM_NONE) return 0.f;
        else if constexpr (NEEDP) { if constexpr (KP == 4) return R.cq[kk >> 1][(kk & 1) * 2 + 1]; else return R.cq[0][kk * 2 + 1]; }
        else if constexpr (KP == 4) return R.cq[1][kk];
        else return R.cq[0][2 + kk];
    };
    HaloRegs hr;                                      // the part in flight inside the stage loop
    // Past the end of the chunk range the stage body keeps issuing its loads (it is branch-free: one scheduling region); they
    // used to re-read the LAST chunk -- one chunk of halo data and two weight stages per tile that nobody consumes (12 % of a
    // 128-channel conv's halo bytes through the CU's vector-memory path).  `pf_live == false` (wave-uniform) collapses such
    // a load onto one address: every lane reads the same 16 bytes, one cache line per instruction.
    bool pf_live = true, dma_live = true;
    int pf_dq = 0;                 // PERS: 1 while a halo part of the NEXT probe of the walk is being loaded
#ifdef LOCO_DUAL_STAMP
    const bool wi_halo = (a.no_deep & 2) != 0, wi_dma = (a.no_deep & 4) != 0, wi_novalu = (a.no_deep & 8) != 0;
#endif
    auto prefetch_hv = [&](HaloRegs& R, int chunk, int part) {
        // wave-uniform chunk base (SGPR pair) + 32-bit per-lane byte offset: global_load saddr form, no 64-bit VALU
        // (32-bit scalar offset arithmetic: one sample's tensor is far below 4 GB)
        const unsigned cb = (unsigned)chunk * ((unsigned)(BKC * 4) * (unsigned)in_plane);
        const long pofs = PERS ? (long)(pq + pf_dq) * pstep : 0;      // probes past the workgroup's first one (wave-uniform)
        const char* pk = reinterpret_cast<const char*>(inb + pofs * a.in_bs - 16) + cb;
        const char* sk = reinterpret_cast<const char*>(reinterpret_cast<const float*>(sxb) - 32) + 2u * cb;
        const unsigned pl = (unsigned)in_plane * 4u;
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) {
#ifdef LOCO_DUAL_STAMP
            const unsigned po = (pf_live && !wi_halo) ? v_goff + (unsigned)(part * KP + kk) * pl : 64u;
#else
            const unsigned po = pf_live ? v_goff + (unsigned)(part * KP + kk) * pl : 64u;     // 64 = the first element itself
#endif
            // 4-byte aligned 16-byte loads (global memory tolerates dword alignment)
            R.dq[kk] = *reinterpret_cast<const f32x4_u*>(pk + po);
            if constexpr (NEEDP) {
                R.sq[kk][0] = *reinterpret_cast<const f32x4_u*>(sk + 2u * po);
                R.sq[kk][1] = *reinterpret_cast<const f32x4_u*>(sk + 2u * po + 16);
            }
        }
        const int c0 = chunk * BKC + v_q4 * 4 + part * KP;    // first channel of this part
        if constexpr (MODE != CM_NONE) {
            if constexpr (NEEDP) {      // {m1,m2} (tangent) or {rstd*m1, rstd*m2} (cotangent) per channel
                const float* tq = tcb + pofs * a.tc_bs;
                R.cq[0] = *reinterpret_cast<const f32x4_u*>(tq + 2 * c0);
                if constexpr (KP == 4) R.cq[1] = *reinterpret_cast<const f32x4_u*>(tq + 2 * c0 + 4);
            } else if constexpr (KP == 4) {
                R.cq[0] = *reinterpret_cast<const f32x4_u*>(scb + pofs * a.scsh_bs + c0);
                R.cq[1] = *reinterpret_cast<const f32x4_u*>(shb + pofs * a.scsh_bs + c0);
            } else {
                const float* sq_ = scb + pofs * a.scsh_bs;
                const float* hq_ = shb + pofs * a.scsh_bs;
                const f32x2 sa = *reinterpret_cast<const f32x2_u*>(sq_ + c0), sb = *reinterpret_cast<const f32x2_u*>(hq_ + c0);
                R.cq[0] = f32x4{sa[0], sa[1], sb[0], sb[1]};
            }
        }
    };
    auto stage_hv = [&](const HaloRegs& R, int part) {
#ifdef LOCO_DUAL_STAMP
        if (wi_novalu) return;
#endif
        const int oct = v_q4 >> 1, half = v_q4 & 1;
#pragma unroll
        for (int pxi = 0; pxi < 4; ++pxi) {
            float r[KP];
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                float d = R.dq[kk][pxi];
                float v = d;
                if constexpr (MODE == CM_GN_SILU) {
                    float y = fmaf(cq_a(R, kk), d, cq_b(R, kk));
                    v = y * sigmoidf2_(y);
                } else if constexpr (MODE == CM_GN_GELU) {
                    v = act_fwd(fmaf(cq_a(R, kk), d, cq_b(R, kk)), ACT_GELU);
                } else if constexpr (MODE == CM_GN) {
                    v = fmaf(cq_a(R, kk), d, cq_b(R, kk));
                } else if constexpr (NEEDP) {
                    const float Sv = R.sq[kk][pxi >> 1][(pxi & 1) * 2], xh = R.sq[kk][pxi >> 1][(pxi & 1) * 2 + 1];
                    // (explicit fused forms: the dual-probe tile of conv_dual_kernel.h computes the same bits)
                    if constexpr (MODE == CM_TAN_SILU) v = Sv * fmaf(-xh, cq_b(R, kk), d - cq_a(R, kk));
                    else v = fmaf(-xh, cq_b(R, kk), fmaf(Sv, d, -cq_a(R, kk)));
                }
                r[kk] = ((v_pm >> pxi) & 1u) ? v : 0.0f;
            }
            unsigned char* dst = Hs + hrec_off<PR>(v_rec[pxi], oct) + half * 8 + part * (KP * 2);
            if constexpr (KP == 4) {
                uint2 h, l;
                cvt2<PR>(r[0], r[1], h.x, l.x);
                cvt2<PR>(r[2], r[3], h.y, l.y);
                *reinterpret_cast<uint2*>(dst) = h;
                if constexpr (PR == PR_BF16X3) *reinterpret_cast<uint2*>(dst + 32) = l;
            } else {
                unsigned h, l;
                cvt2<PR>(r[0], r[1], h, l);
                *reinterpret_cast<unsigned*>(dst) = h;
                if constexpr (PR == PR_BF16X3) *reinterpret_cast<unsigned*>(dst + 32) = l;
            }
        }
    };
    auto load_consts = [&](int chunk) {
        const int c0 = chunk * BKC;
        if constexpr (MODE != CM_NONE) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                int c = c0 + kk < a.Cin ? c0 + kk : a.Cin - 1;
                if constexpr (NEEDP) { cA[kk] = tcb[2 * c]; cB[kk] = tcb[2 * c + 1]; }
                else { cA[kk] = scb[c]; cB[kk] = shb[c]; }
            }
        }
    };
    auto load_regs = [&](float (&hvd)[NITEM][8], int chunk) {
        const int c0 = chunk * BKC;
        if (c0 + BKC <= a.Cin) {
            // fast path: all 16 channels exist -> wave-uniform plane base + per-lane 32-bit offset
            if constexpr (TAPS == 9 && !NEEDP) {
                // 3x3 operators on the per-pixel path that read one value per element (stride 2, folded upsampling, caller-owned
                // inputs; raw and forward forms): one 64-bit product per chunk, one add per plane, 32-bit byte offsets masked by the
                // wave-uniform `pf_live` (see the register-ring loop of the 1x1 operators).  Not in the 1x1 instances (the shared
                // subexpressions flipped that loop's address form) and not in the tangent / cotangent forms (config 5 +2.5 %)
                const unsigned lm = pf_live ? 0xffffffffu : 0u;
                const long pb4 = in_plane * 4;
                const unsigned char* pk = reinterpret_cast<const unsigned char*>(inb) + (long)c0 * pb4;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int i = 0; i < NITEM; ++i) hvd[i][k] = *reinterpret_cast<const float*>(pk + ((ivoff[i] * 4u) & lm));
                    pk += pb4;
                }
            } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* pk = inb + (long)(c0 + k) * in_plane;
                const float2* sk = NEEDP ? sxb + (long)(c0 + k) * in_plane : nullptr;
#pragma unroll
                for (int i = 0; i < NITEM; ++i) {
                    hvd[i][k] = pk[pf_live ? ivoff[i] : 0u];
                    if constexpr (NEEDP) pv[i][k] = sk[pf_live ? ivoff[i] : 0u];
                }
            }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NITEM; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int c = c0 + ioct[i] * 8 + k;
                    int cc = c < a.Cin ? c : a.Cin - 1;
                    unsigned vo = ivoff[i] - (unsigned)(ioct[i] * 8 * (int)in_plane);
                    float v = inb[(long)cc * in_plane + vo];
                    hvd[i][k] = c < a.Cin ? v : 0.0f;
                    if constexpr (NEEDP) {
                        float2 pp = sxb[(long)cc * in_plane + vo];
                        pv[i][k] = c < a.Cin ? pp : make_float2(0.f, 0.f);
                    }
                }
            }
        }
    };
    auto prefetch_h = [&](int chunk, int part) {
        if constexpr (!GEN) { prefetch_hv(hr, chunk, part); return; }
        load_consts(chunk);
        load_regs(hv, chunk);
    };
    // Weight pieces of this thread: byte offset from the (chunk, row) base.  Records past the padded cout range
    // are CLAMPED to the last one instead of zeroed: a D row only depends on its own A row, and rows >= Cout are
    // never stored, so the duplicate is harmless and the loads stay unconditional.
    // The pre-split, pre-swizzled weight records go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x
    // 16 bytes land contiguously at a wave-uniform LDS base), so a weight stage costs no VGPRs and no ds_write.
    unsigned wrel[NWV];
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
        int e = tid + i * NTHR;
        if (e >= WTOT) e = WTOT - 1;
        int tap = e / (MT * NPC), rem = e - tap * (MT * NPC);
        int rec = co0 + rem / NPC;
        if (rec >= wpitch) rec = wpitch - 1;
        wrel[i] = (unsigned)((tap * wpitch + rec) * NPC + (rem % NPC)) * 16u;
    }
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef const __attribute__((address_space(1))) unsigned char glb_u8;
    auto dma_w = [&](int chunk, int row, unsigned char* Wdst) {
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>(wg) +
                                     (unsigned)(chunk * TAPS + row * NTS) * ((unsigned)wpitch * (unsigned)RB);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e0 = (wave * 64 + i * NTHR);                 // first piece of this wave's 1 KiB slab
            if ((WTOT % NTHR) == 0 || e0 < WTOT)                   // wave-uniform
#ifdef LOCO_DUAL_STAMP
                __builtin_amdgcn_global_load_lds((glb_u8*)(wbase + ((dma_live && !wi_dma) ? wrel[i] : 0u)), (lds_u8*)(Wdst + e0 * 16), 16, 0, 0);
#else
                __builtin_amdgcn_global_load_lds((glb_u8*)(wbase + (dma_live ? wrel[i] : 0u)), (lds_u8*)(Wdst + e0 * 16), 16, 0, 0);
#endif
        }
    };
    // whole_tag: the caller has checked that the operator has whole 16-channel chunks (no per-value channel bound)
    auto stage_regs = [&](const float (&hvs)[NITEM][8], int chunk, unsigned char* Hd, auto whole_tag) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
        const int c0 = chunk * BKC;
#pragma unroll
        for (int i = 0; i < NITEM; ++i) {
            if (ipos[i] < 0) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int c = c0 + ioct[i] * 8 + k;
                float r = 0.0f;
                if (ival[i] && (WHOLE || c < a.Cin)) {
                    float d = hvs[i][k];
                    if constexpr (MODE == CM_NONE) {
                        r = d;
                    } else {
                        const float ca = ioct[i] ? cA[8 + k] : cA[k];
                        const float cb = ioct[i] ? cB[8 + k] : cB[k];
                        if constexpr (MODE == CM_GN_SILU) {
                            float y = fmaf(ca, d, cb);
                            r = y * sigmoidf2_(y);
                        } else if constexpr (MODE == CM_GN_GELU) {
                            r = act_fwd(fmaf(ca, d, cb), ACT_GELU);
                        } else if constexpr (MODE == CM_GN) {
                            r = fmaf(ca, d, cb);
                        } else {
                            float Sv = pv[i][k].x, xh = pv[i][k].y;
                            if constexpr (MODE == CM_TAN_SILU) r = Sv * fmaf(-xh, cb, d - ca);
                            else r = fmaf(-xh, cb, fmaf(Sv, d, -ca));
                        }
                    }
                }
                v[k] = r;
            }
            uint4 hi, lo;
            split8<PR>(v, hi, lo);
            *reinterpret_cast<uint4*>(Hd + hrec_off<PR>(ipos[i], ioct[i])) = hi;
            if constexpr (PR == PR_BF16X3) *reinterpret_cast<uint4*>(Hd + hrec_off<PR>(ipos[i], 2 + ioct[i])) = lo;
        }
    };
    auto stage_h = [&](int chunk, int part) {
        if constexpr (!GEN) { stage_hv(hr, part); return; }
        stage_regs(hv, chunk, Hs, std::false_type{});
    };

    // MFMA operand fragments of one tap; two sets alternate so the ds_reads of tap t+1 are in flight under the
    // MFMAs of tap t (one exposed LDS latency per stage instead of one per operand)
    constexpr int NLO = PR == PR_BF16X3 ? 1 : 0;     // lo halves exist only in the split arithmetic
    struct Frag { s16x8 ah[TM], al[NLO ? TM : 1], bh[TN], bl[NLO ? TN : 1]; };
    auto load_frag = [&](Frag& f, int row, int tp) {
        const int tapoff = (TAPS == 9) ? row * halo_w + tp : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.ah[i] = *reinterpret_cast<const s16x8*>(Ws + tp * MT * RB + aoff_hi[i]);
#if defined(LOCO_WI_NOLO)
            if constexpr (NLO) f.al[i] = f.ah[i];      // what-if build (WRONG results): half of the operand fragment reads
#else
            if constexpr (NLO) f.al[i] = *reinterpret_cast<const s16x8*>(Ws + tp * MT * RB + aoff_lo[i]);
#endif
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned char* hp = Hs + tapoff * HP + hbyte[j];     // uniform tap shift + per-lane record base
            f.bh[j] = *reinterpret_cast<const s16x8*>(hp);
#if defined(LOCO_WI_NOLO)
            if constexpr (NLO) f.bl[j] = f.bh[j];
#else
            if constexpr (NLO) f.bl[j] = *reinterpret_cast<const s16x8*>(hp + 32);
#endif
        }
    };
    auto mma_frag = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (PR == PR_F16) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.ah[i]),
                                                                       __builtin_bit_cast(f16x8, f.bh[j]), acc[i][j], 0, 0, 0);
                } else {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, f.ah[i]), al = __builtin_bit_cast(bf16x8, f.al[i]);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, f.bh[j]), bl = __builtin_bit_cast(bf16x8, f.bl[j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                }
            }
    };

    // the same products split in two: head = the (0,0) sub-tile, tail = the rest (cross-stage prefetch pipeline)
    auto mma_one = [&](const Frag& f, int i, int j) {
        if constexpr (PR == PR_F16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.ah[i]),
                                                               __builtin_bit_cast(f16x8, f.bh[j]), acc[i][j], 0, 0, 0);
        } else {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, f.ah[i]), al = __builtin_bit_cast(bf16x8, f.al[i]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, f.bh[j]), bl = __builtin_bit_cast(bf16x8, f.bl[j]);
#if defined(LOCO_WI_MFMA16)
            // what-if build (profiles/r06_experiments.md: WRONG results, timing only): every 32x32x16 MFMA as two 16x16x32 MFMAs of
            // the same operand registers (same matrix-pipe cycles, same register traffic) -- what the MFMA SHAPE does to the
            // clock the chip holds under this kernel (MI355X_MICROARCH.md, DVFS give-back item 7)
            {
                f32x4 q0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, q1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                f32x4 q2 = {acc[i][j][8], acc[i][j][9], acc[i][j][10], acc[i][j][11]}, q3 = {acc[i][j][12], acc[i][j][13], acc[i][j][14], acc[i][j][15]};
                q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, q1, 0, 0, 0);
                q2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, q2, 0, 0, 0);
                q3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, q3, 0, 0, 0);
                q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, q1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[i][j][r] = q0[r]; acc[i][j][4 + r] = q1[r]; acc[i][j][8 + r] = q2[r]; acc[i][j][12 + r] = q3[r]; }
                return;
            }
#elif defined(LOCO_WI_2MFMA)
            // what-if build (WRONG results): two of the three MFMAs per product -- the slope of the launch time in the MFMA count
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
            (void)al;
            return;
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
        }
    };
    auto mma_frag_head = [&](const Frag& f) { mma_one(f, 0, 0); };
    auto mma_frag_tail = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (i + j > 0) mma_one(f, i, j);
    };

    // Software pipeline with ONE barrier per stage (s = NROW*chunk + row).  Stage s multiplies out of W[s&1] /
    // H[chunk&1].  It opens by launching the LDS-DMA of stage s+1's weights into W[(s+1)&1]; between its taps it
    // converts one part of the NEXT chunk's halo (loaded into registers at the end of an earlier stage) into
    // H[(chunk+1)&1]; it closes with vmcnt(0) (the DMA, issued a whole stage ago, has landed), the register loads of
    // the next halo part, and a RAW barrier so those loads stay in flight across it.  Buffers written in stage s
    // were last read in stage s-1, which every wave has left; everything read in stage s was complete before the
    // barrier that opened it.  The stage body is BRANCH-FREE (one basic block): past the end the loads are clamped
    // to the last chunk and the stores land in buffers nobody reads, so the scheduler can sink the staging VALU /
    // LDS-write work into the shadow of the MFMAs.
    //   halo schedule, 3x3 vector path (2 parts A,B):  row0: store A | load B   row1: store B   row2: load A'
    //   3x3 per-pixel path (1 part):                   row0: load           row2: store
    //   1x1:                                           every stage: store | load
    const int nch = cend - cbeg;
    const int clast = cend - 1;
    auto cclamp = [&](int c) { return __builtin_amdgcn_readfirstlane(c < clast ? c : clast); };   // keep it scalar
    // chunk x of the current probe's range, possibly past its end: resolves to the chunk to load and sets pf_live / pf_dq
    auto vres = [&](int x) -> int {
        if (x <= clast) { pf_live = true; pf_dq = 0; return __builtin_amdgcn_readfirstlane(x); }
        if (PERS && pq + 1 < np) { pf_live = true; pf_dq = 1; return __builtin_amdgcn_readfirstlane(x - nch); }      // (PERS: cbeg = 0)
        pf_live = !LIVE_MASK; pf_dq = 0;
        return clast;
    };
    auto stage_end = [&]() {
        // all LDS writes of this wave retired, LDS-DMA landed; then a bare s_barrier (no vmcnt drain of newer loads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if constexpr (XPF) {
        // ---- 3x3, stride 1: CROSS-STAGE operand prefetch ------------------------------------------------------------
        // The operand fragments of a tap are read from LDS one whole tap ahead, ACROSS the stage barrier: while the
        // MFMAs of tap t run, the ds_reads of tap t+1 are in flight -- also when tap t+1 opens the next stage -- so no
        // stage starts with an exposed LDS burst (8 waves x 16 ds_read_b128 right behind the barrier, ~600 cycles of
        // idle matrix pipe per stage before).  Two fragment sets alternate per tap; a chunk has 9 taps, so the roles
        // flip every chunk and the loop is unrolled over two chunks (compile-time parity P).  Reading stage s+1's
        // weights inside stage s needs them complete one barrier earlier: THREE weight buffers, the LDS-DMA runs two
        // stages ahead (buffer index = kernel row: 3 rows, 3 buffers).  The halo of chunk c+1 is complete before row 2
        // of chunk c starts (vector path: parts in rows 0 and 1; per-pixel path: converted in row 1).
        Frag fr[2];
        auto stage_of = [&](int ci_, int row_, int& c_out, int& r_out) {   // (chunk, row) of the stage `row_` may overflow into
            c_out = cclamp(cbeg + ci_ + row_ / NROW);
            r_out = row_ % NROW;
        };
        LP_STAMP(1);
        if (nch > 0) {
            if constexpr (!(PHASE != 2 && EARLY_W)) {      // (otherwise issued at the top of the kernel)
                int c1, r1;
                dma_w(cbeg, 0, Wsb);
                stage_of(0, 1, c1, r1);
                dma_w(c1, r1, Wsb + WBYTES);
            }
            Hs = Hsb;
            if constexpr (!GEN && NPART == 2) {
                // both parts of the first chunk are loaded together (a second register set that only lives here) so
                // the workgroup pays one memory latency, not two, before its first stage
                HaloRegs hr2;
                prefetch_hv(hr, cbeg, 0);
                prefetch_hv(hr2, cbeg, 1);
                stage_hv(hr, 0);
                stage_hv(hr2, 1);
            } else {
                prefetch_h(cbeg, 0);
                stage_h(cbeg, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prefetch_h(cclamp(cbeg + 1), 0);          // pending part / chunk expected by the first chunk's conversion
            stage_end();
            Ws = Wsb; Hs = Hsb;
            load_frag(fr[0], 0, 0);                   // the only exposed operand read of the tile
        }
        LP_STAMP(2);
        auto chunk_body = [&](auto ptag, const int ci) {
            constexpr int P = decltype(ptag)::value;
            const int chunk = cbeg + ci;
            unsigned char* const Hcur = Hsb + P * HBYTES;
            unsigned char* const Hnxt = Hsb + (1 - P) * HBYTES;
#pragma unroll
            for (int row = 0; row < NROW; ++row) {
                unsigned char* const Wcur = Wsb + row * WBYTES;
                unsigned char* const Wnx1 = Wsb + ((row + 1) % 3) * WBYTES;
                unsigned char* const Wnx2 = Wsb + ((row + 2) % 3) * WBYTES;
                // Halo schedule of a chunk (one register set hr; a part is loaded >= one whole stage before it is converted):
                //   vector path, parts A, B:  row 0: tap 1 converts A(c+1), tap 2 loads B(c+1);  row 1: tap 2 converts B(c+1);
                //                             row 2: tap 0 loads A(c+2)
                //   per-pixel path, one part: row 1: tap 1 converts (c+1);  row 2: loads (c+2) at the end of the stage
                const int cv1 = NPART == 2 ? (row == 0 ? 0 : -1) : (row == 1 ? 0 : -1);    // part converted under tap 1
                const int cv2 = (NPART == 2 && row == 1) ? 1 : -1;                         // part converted under tap 2
                const int ld0 = (NPART == 2 && row == 2) ? 0 : -1;                         // part loaded under tap 0
                const int ld2 = (NPART == 2 && row == 0) ? 1 : -1;                         // part loaded under tap 2
                const int lde = (NPART == 1 && row == 2) ? 0 : -1;                         // part loaded at the stage end
                constexpr int NLD = KP + (NEEDP ? 2 * KP + 1 : (MODE != CM_NONE ? 2 : 0));
                {
                    // weights of the stage after next; past the probe's last chunk: the first stages of the walk's next probe
                    // (PERS: the same cout tile, chunk 0 again) or, at the very end, a collapsed dead load
                    const int x = chunk + (row + 2) / NROW;
                    int c2;
                    if (x <= clast) { c2 = x; dma_live = true; }
                    else if (PERS && pq + 1 < np) { c2 = x - nch; dma_live = true; }
                    else { c2 = clast; dma_live = !LIVE_MASK; }
                    dma_w(__builtin_amdgcn_readfirstlane(c2), (row + 2) % NROW, Wnx2);
                    dma_live = true;
                }
                Frag& fa = fr[(P + row) & 1];          // taps 0 and 2 of this stage
                Frag& fb = fr[(P + row + 1) & 1];      // tap 1, then tap 0 of the next stage
                // One tap = [first MFMAs of the tap | ds_reads of the NEXT tap | remaining MFMAs (+ staging work)]: the
                // reads are issued once the matrix pipe has work queued and have the rest of the tap to land, and the
                // wait in front of a tap's first MFMA only covers reads issued a whole tap earlier.
                Ws = Wcur; Hs = Hcur;
                mma_frag_head(fa);
                __builtin_amdgcn_sched_barrier(0);
                load_frag(fb, row, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ld0 >= 0) { const int cx = vres(chunk + 2); prefetch_h(cx, ld0); pf_live = true; pf_dq = 0; }
                mma_frag_tail(fa);
                __builtin_amdgcn_sched_barrier(0);
                mma_frag_head(fb);
                __builtin_amdgcn_sched_barrier(0);
                load_frag(fa, row, 2);
                __builtin_amdgcn_sched_barrier(0);
                if (cv1 >= 0) { Hs = Hnxt; stage_h(cclamp(chunk + 1), cv1); }
                mma_frag_tail(fb);
                __builtin_amdgcn_sched_barrier(0);
                mma_frag_head(fa);
                __builtin_amdgcn_sched_barrier(0);
                // operands of the NEXT stage's first tap (same chunk: next kernel row; last row: next chunk's halo)
                Ws = Wnx1;
                if (row + 1 < NROW) { Hs = Hcur; load_frag(fb, row + 1, 0); }
                else { Hs = Hnxt; load_frag(fb, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
                // conversion and re-load never share a scheduling region: the loads then land directly in the registers
                // the conversion has finished reading (no copies behind a vmcnt wait)
                if (cv2 >= 0) { Hs = Hnxt; stage_h(cclamp(chunk + 1), cv2); }
                if (ld2 >= 0) { const int cx = vres(chunk + 1); prefetch_h(cx, ld2); pf_live = true; pf_dq = 0; }
                mma_frag_tail(fa);
                __builtin_amdgcn_sched_barrier(0);
                // the LDS-DMA of this stage (older than the part loads issued in it) must have landed before the barrier
                if (ld0 >= 0 || ld2 >= 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lde >= 0) { const int cx = vres(chunk + 2); prefetch_h(cx, lde); pf_live = true; pf_dq = 0; }
                }
                stage_end();
            }
        };
        if constexpr (PERS) {
            // one continuous stream of chunks over the probes of the walk (nch is even: the buffer parities restart with each
            // probe); a finished tile leaves through its own staging area while the next probe's operands wait in the buffers
            float* const Sst = reinterpret_cast<float*>(Hsb + 2 * HBYTES);
            for (pq = 0; pq < np; ++pq) {
                for (int ci = 0; ci < nch; ci += 2) {
                    chunk_body(std::integral_constant<int, 0>{}, ci);
                    chunk_body(std::integral_constant<int, 1>{}, ci + 1);
                }
                conv_lowp_epilogue_rounds<WM, WN, TM, TN, 16>(a, acc, Sst, co0, oy0, ox0, TW, tile_id, b + pq * pstep);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the walk's last (dead) loads
            return;
        } else {
        for (int ci = 0; ci < nch; ci += 2) {
            chunk_body(std::integral_constant<int, 0>{}, ci);
            if (ci + 1 < nch) chunk_body(std::integral_constant<int, 1>{}, ci + 1);
        }
        }
    } else {
    bool deep_done = false;
    if constexpr (DEEP1) {
      if ((a.Cin % BKC) == 0 && !a.no_deep) {     // whole chunks only (uniform); partial last chunks take the two-buffer loop below
        deep_done = true;
        // ---- 1x1, per-pixel path: register ring, DEEP_D chunks in flight -----------------------------------------------
        // Stage ci multiplies chunk ci out of W / H buffer ci & 1; meanwhile it commits chunk ci + 1 (converted pixels and
        // weight pieces, loaded DEEP_D - 1 stages ago into ring slot (ci + 1) % D) to the other buffer and re-issues that
        // slot's loads for chunk ci + 1 + D.  No LDS-DMA and no hand-written vmcnt here: every wait is the compiler's own
        // count for the slot it is about to read (vmcnt(27) = three chunks of 8 + 1 loads), so D - 1 chunks stay in flight
        // across the stage barrier.  (A third W / H buffer with the operand fragments read one stage ahead, as in the 3x3
        // loop, measured no better: profiles/r03_experiments.md #10.)
        constexpr int D = DEEP_D;
        const long plane_bytes = in_plane * 4;
        unsigned ivoffb[NITEM];
#pragma unroll
        for (int i = 0; i < NITEM; ++i) ivoffb[i] = ivoff[i] * 4u;
        float hvr[D][NITEM][8];
        f32x4 wr[D][NWV];
        const unsigned char* const wgb = reinterpret_cast<const unsigned char*>(wg);
        // `chunk` may lie past the range (the ring keeps D chunks in flight to the last stage): such loads are collapsed onto
        // one address per instruction instead of re-reading the last chunk (`LIVE_MASK`; with D = 4 that was half a 128-channel
        // operator's input again)
        auto issue = [&](auto stag, int chunk_raw) {
            constexpr int sl = decltype(stag)::value;
            const bool live = LIVE_MASK ? (chunk_raw <= clast) : true;
            const int chunk = cclamp(chunk_raw);
            const int c0 = chunk * BKC;
            // Scalar base (one 64-bit product per chunk, then one add per channel plane) + a 32-bit per-lane BYTE offset masked by the
            // wave-uniform `live`: the stage loop of this kernel is bound by instruction issue, not by the matrix pipe (r05 stamps:
            // 2.2 k cycles per stage of 12 MFMAs; ~45 of a stage's ~230 instructions were 64-bit address products, ~27 more the
            // per-load 64-bit vector adds and compare / select pairs of `live ? off : 0`)
            const unsigned lm = live ? 0xffffffffu : 0u;
            const unsigned char* pk = reinterpret_cast<const unsigned char*>(inb) + (long)c0 * plane_bytes;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int i = 0; i < NITEM; ++i) hvr[sl][i][k] = *reinterpret_cast<const float*>(pk + (ivoffb[i] & lm));
                pk += plane_bytes;
            }
            const unsigned char* wbase = wgb + (unsigned)(chunk * TAPS) * ((unsigned)wpitch * (unsigned)RB);
#pragma unroll
            for (int i = 0; i < NWV; ++i) wr[sl][i] = *reinterpret_cast<const f32x4*>(wbase + (wrel[i] & lm));
        };
        auto commit = [&](auto stag, int chunk, unsigned char* Hd, unsigned char* Wd) {
            constexpr int sl = decltype(stag)::value;
            if constexpr (MODE != CM_NONE) load_consts(chunk);
            stage_regs(hvr[sl], chunk, Hd, std::true_type{});
#pragma unroll
            for (int i = 0; i < NWV; ++i)
                if ((WTOT % NTHR) == 0 || tid + i * NTHR < WTOT)
                    *reinterpret_cast<f32x4*>(Wd + (tid + i * NTHR) * 16) = wr[sl][i];
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
        static_assert(D == 4, "ring slots are spelled out below");
        if (nch > 0) {
            issue(S0{}, cbeg);
            issue(S1{}, cbeg + 1);
            issue(S2{}, cbeg + 2);
            issue(S3{}, cbeg + 3);
            commit(S0{}, cbeg, Hsb, Wsb);
            issue(S0{}, cbeg + 4);
            stage_end();
        }
        auto body = [&](auto stag, const int ci) {        // stag = ring slot of chunk ci + 1
            const int chunk = cbeg + ci;
            unsigned char* const Wcur = Wsb + (ci & 1) * WBYTES;
            unsigned char* const Wnxt = Wsb + ((ci + 1) & 1) * WBYTES;
            unsigned char* const Hcur = Hsb + (ci & 1) * HBYTES;
            unsigned char* const Hnxt = Hsb + ((ci + 1) & 1) * HBYTES;
            Ws = Wcur; Hs = Hcur;
            Frag f0;
            load_frag(f0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_frag(f0);
            commit(stag, cclamp(chunk + 1), Hnxt, Wnxt);
            issue(stag, chunk + 1 + D);
            stage_end();
        };
        int ci = 0;
        for (; ci + D <= nch; ci += D) {                  // whole ring turns: one basic block, exact waits
            body(S1{}, ci); body(S2{}, ci + 1); body(S3{}, ci + 2); body(S0{}, ci + 3);
        }
        if (ci < nch) body(S1{}, ci);
        if (ci + 1 < nch) body(S2{}, ci + 1);
        if (ci + 2 < nch) body(S3{}, ci + 2);
      }
    }
    if (!deep_done) {
    if (nch > 0) {
        dma_w(cbeg, 0, Wsb);
        Hs = Hsb;
        if constexpr (!GEN && NPART == 2) {
            // both parts of the first chunk are loaded together (a second register set that only lives here) so the
            // workgroup pays one memory latency, not two, before its first stage
            HaloRegs hr2;
            prefetch_hv(hr, cbeg, 0);
            prefetch_hv(hr2, cbeg, 1);
            stage_hv(hr, 0);
            stage_hv(hr2, 1);
        } else {
#pragma unroll
            for (int part = 0; part < NPART; ++part) {
                prefetch_h(cbeg, part);
                stage_h(cbeg, part);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NROW == 1 || NPART == 2) prefetch_h(cclamp(cbeg + 1), 0);      // pending part expected by the first stage
        stage_end();
    }
    for (int ci = 0; ci < nch; ++ci) {
        const int chunk = cbeg + ci;
#pragma unroll
        for (int row = 0; row < NROW; ++row) {
            const int s_ = ci * NROW + row;
            unsigned char* const Wcur = Wsb + (s_ & 1) * WBYTES;
            unsigned char* const Wnxt = Wsb + ((s_ + 1) & 1) * WBYTES;
            unsigned char* const Hcur = Hsb + (ci & 1) * HBYTES;
            unsigned char* const Hnxt = Hsb + ((ci + 1) & 1) * HBYTES;
            // which halo part of the next chunk this row converts (-1: none) and which it loads at its end
            constexpr bool one = (NROW == 1);
            const int st_part = one ? 0 : (NPART == 2 ? (row == 0 ? 0 : row == 1 ? 1 : -1) : (row == 2 ? 0 : -1));
            // vector path: a part's registers are re-loaded right after they are converted (mid-stage), so the
            // loads have one to two whole stages to land; per-pixel path: loads at the end of row 0
            const int ld_part = one ? 0 : (NPART == 2 ? (row == 0 ? 1 : row == 1 ? 0 : -1) : (row == 0 ? 0 : -1));
            const int ld_chunk = one ? chunk + 2 : ((NPART == 2 && row == 1) ? chunk + 2 : chunk + 1);
            constexpr bool MIDLOAD = !GEN;          // issue the loads inside the stage, behind the conversion
            // vector-memory instructions of one part's loads (for the counted wait that leaves them in flight)
            constexpr int NLD = KP + (NEEDP ? 2 * KP + 1 : (MODE != CM_NONE ? 2 : 0));
            // Regions fenced by sched_barrier(0): the machine scheduler would otherwise sink every ds_read to just
            // before its first use and expose one LDS latency per operand.
            {
                const int r1 = (row + 1) % NROW, dc = (row + 1) / NROW;
                dma_live = LIVE_MASK ? (chunk + dc <= clast) : true;
                dma_w(cclamp(chunk + dc), r1, Wnxt);
                dma_live = true;
            }
            Ws = Wcur; Hs = Hcur;
            Frag f0, f1;
            load_frag(f0, row, 0);
            if (NTS > 1) load_frag(f1, row, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_frag(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (NTS > 2) load_frag(f0, row, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (NTS > 1) mma_frag(f1);
            // halo conversion, then the re-load of its registers, share one scheduling region with the MFMAs of taps
            // 1 and 2 (a fence between them measured 2.7 ms/step slower; conversion under tap 0 13 ms slower)
            if (st_part >= 0) { Hs = Hnxt; stage_h(cclamp(chunk + 1), st_part); Hs = Hcur; }
            if (MIDLOAD && ld_part >= 0) { pf_live = LIVE_MASK ? (ld_chunk <= clast) : true; prefetch_h(cclamp(ld_chunk), ld_part); pf_live = true; }
            if (NTS > 2) mma_frag(f0);
            __builtin_amdgcn_sched_barrier(0);
            // the LDS-DMA of this stage (older than the part loads just issued) must have landed before the barrier
            if (MIDLOAD && ld_part >= 0) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!MIDLOAD && ld_part >= 0) { pf_live = LIVE_MASK ? (ld_chunk <= clast) : true; prefetch_h(cclamp(ld_chunk), ld_part); pf_live = true; }
            }
            stage_end();
        }
    }
    }

    }

    if constexpr (PHASE == 1) {
        // the second operator's prologue overwrites the operand buffers: every wave has left the last stage (its closing
        // barrier); the fragment reads issued ahead for a stage that does not exist are retired here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }
    LP_STAMP(3);
    conv_lowp_epilogue<WM, WN, TM, TN, (TAPS == 1 && PHASE == 0) ? (EPI_COT1 | EPI_STATS) : EPI_STATS>(a, acc, smem_b, co0, oy0, ox0, TW, tile_id, b, split);
    LP_STAMP(4);
#undef LP_STAMP
}

// ---------------------------------------------------------------------------
// Kernel entry points: one name per arithmetic so profiles tell them apart.
// Waves per SIMD the register allocation has to leave room for.  The four-wave (256-thread) tiles of the 1x1 operators count on TWO
// workgroups per CU -- one's write-out under the other's stage loop; these launches are bound by memory latency, not by the matrix
// pipe -- and a four-wave workgroup alone only asks for one wave per SIMD: without the bound the allocator took 304 registers
// (240 + 64 accumulation registers) once the epilogue carried the tangent / cotangent sums (round 6), i.e. ONE workgroup per CU
// (128 -> 256 @256^2 with the norm-cotangent term: 375 -> 500 us in the flow).  STG 3: the compact 3x3 tile built for two per CU.
constexpr int lowp_min_waves(int taps, int nthr, int stg) { return (stg == 3 || (taps == 1 && nthr <= 256)) ? 2 : 1; }
template <int TAPS, int WM, int WN, int TM, int TN, int MODE, int STG>
__global__ __launch_bounds__(WM * WN * 64, lowp_min_waves(TAPS, WM * WN * 64, STG)) void conv_mfma_bf16x3(ConvArgs a) {
    f32x16 acc[TM][TN];
    conv_lowp_body<PR_BF16X3, TAPS, WM, WN, TM, TN, MODE, STG>(a, acc);
}
template <int TAPS, int WM, int WN, int TM, int TN, int MODE, int STG>
__global__ __launch_bounds__(WM * WN * 64, lowp_min_waves(TAPS, WM * WN * 64, STG)) void conv_mfma_f16(ConvArgs a) {
    f32x16 acc[TM][TN];
    conv_lowp_body<PR_F16, TAPS, WM, WN, TM, TN, MODE, STG>(a, acc);
}

#ifdef LOCO_DIAG
// persistent over the probes of a tile (PHASE 3 of conv_lowp_body): diagnostics build only (measured neutral in the flow)
template <int TAPS, int WM, int WN, int TM, int TN, int MODE, int STG>
__global__ __launch_bounds__(WM * WN * 64, 1) void conv_pers_bf16x3(ConvArgs a) {
    f32x16 acc[TM][TN];
    conv_lowp_body<PR_BF16X3, TAPS, WM, WN, TM, TN, MODE, STG, 3>(a, acc);
}
#endif

// K-concatenated ResBlock tail: out = conv3x3(map(in)) + conv1x1(in2) + biases (+ residual): the 3x3 operator of `a` (vector
// staging, MODE) and, on the same accumulator tile, the 1x1 operator {in2, Cin2, wb2} on the RAW block input (per-pixel
// staging with the register ring) -- reference models/ddpm/diffusion.py:887-912 (`x = nin_shortcut(x); return x + h`): one
// write-out instead of the shortcut's store and conv2's read-modify-write of the block output, one launch instead of two.
template <int PR, int WM, int WN, int TM, int TN, int MODE>
__device__ __forceinline__ void conv_lowp_kcat(const ConvArgs& a) {
    f32x16 acc[TM][TN];
    conv_lowp_body<PR, 9, WM, WN, TM, TN, MODE, 0, 1>(a, acc);
    ConvArgs a2 = a;
    a2.in = a.in2; a2.in_bs = a.in2_bs; a2.Cin = a.Cin2; a2.wb = a.wb2; a2.mode = CM_NONE; a2.pad = 0; a2.in_padded = a.in2_padded;
    conv_lowp_body<PR, 1, WM, WN, TM, TN, CM_NONE, 1, 2>(a2, acc);
}
template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void conv_kcat_bf16x3(ConvArgs a) { conv_lowp_kcat<PR_BF16X3, WM, WN, TM, TN, MODE>(a); }
template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void conv_kcat_f16(ConvArgs a) { conv_lowp_kcat<PR_F16, WM, WN, TM, TN, MODE>(a); }

template <int PR, int TAPS, int WM, int WN, int TM, int TN, int MODE, int STG>
static void launch_one_b2(const ConvArgs& a, hipStream_t st) {
    constexpr int MT = WM * TM * 32, NT = WN * TN * 32;
    constexpr int KS = (TAPS == 9) ? 3 : 1;
    int TW = a.Wout < 32 ? a.Wout : 32;
    int TH = NT / TW;
    int halo_w = (TW - 1) * a.stride + KS, halo_h = (TH - 1) * a.stride + KS;
    const int nwb = (TAPS == 9 && STG != 2 && STG != 3) ? 3 : 2;
    size_t lds = (size_t)nwb * KS * MT * rec_bytes<PR>() + 2 * ((size_t)halo_w * halo_h + NDUMMY) * halo_pitch<PR>();
    if (STG == 3) lds = (size_t)nwb * KS * MT * rec_bytes<PR>() + 2 * ((size_t)halo_w * halo_h * halo_pitch<PR>() + rec_bytes<PR>());
    const size_t stage_bytes = (size_t)WM * 32 * NT * 4;      // epilogue staging tile S[WM*32 couts][NT pixels]
    if (lds < stage_bytes) lds = stage_bytes;
    dim3 grid(((a.Hout * a.Wout) / NT) * ((a.Cout + MT - 1) / MT) * a.B * a.nsplit);
#ifdef LOCO_DIAG
    if constexpr (PR == PR_BF16X3 && TAPS == 9 && STG == 0 && WM == 2 && WN == 4 && TM == 2 && TN == 2) {
        if (a.pers_groups > 0) {      // conv_pers_plan: one workgroup walks the probes b, b + G, ... of its tile
            auto pk = &conv_pers_bf16x3<TAPS, WM, WN, TM, TN, MODE, STG>;
            static DeviceOnce ponce;
            if (first_on_device(ponce))
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const size_t plds = (size_t)nwb * KS * MT * rec_bytes<PR>() + 2 * ((size_t)halo_w * halo_h + NDUMMY) * halo_pitch<PR>() +
                                (size_t)WM * 16 * NT * 4;      // operand buffers + the write-out's own 32 KB staging area
            dim3 pgrid(((a.Hout * a.Wout) / NT) * ((a.Cout + MT - 1) / MT) * a.pers_groups);
            hipLaunchKernelGGL(pk, pgrid, dim3(WM * WN * 64), plds, st, a);
            return;
        }
    }
#endif
    // (if constexpr: a ?: of the two addresses instantiates BOTH arithmetics' kernels in every translation unit)
    void (*kern)(ConvArgs);
    if constexpr (PR == PR_F16) kern = &conv_mfma_f16<TAPS, WM, WN, TM, TN, MODE, STG>;
    else kern = &conv_mfma_bf16x3<TAPS, WM, WN, TM, TN, MODE, STG>;
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (first_on_device(once)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, a);
}

template <int PR, int TAPS, int WM, int WN, int TM, int TN, int MODE>
static void launch_one_b(const ConvArgs& a, hipStream_t st) {
    const bool general = a.stride != 1 || a.upsample || a.zins || (a.Cin % BKC) != 0 || !a.in_padded;
    if constexpr (MODE == CM_NONE) {
        // raw inputs: 3x3 convs take the vector path when the layout allows (measured 66.1 -> 62.8 ms/step);
        // 1x1 convs stay on the per-pixel path (38.1 vs 39.2 ms/step)
        if (a.stride == 2) launch_one_b2<PR, TAPS, WM, WN, TM, TN, MODE, 2>(a, st);
        else if (TAPS == 9 && !general) launch_one_b2<PR, TAPS, WM, WN, TM, TN, MODE, 0>(a, st);
        else launch_one_b2<PR, TAPS, WM, WN, TM, TN, MODE, 1>(a, st);
        return;
    }
    if constexpr (MODE == CM_GN_SILU || MODE == CM_TAN_SILU || MODE == CM_GN_GELU) {
        if (general) { launch_one_b2<PR, TAPS, WM, WN, TM, TN, MODE, 1>(a, st); return; }
    }
    launch_one_b2<PR, TAPS, WM, WN, TM, TN, MODE, 0>(a, st);
}


template <int PR, int MODE>
void launch_kcat_b(const ConvArgs& a, hipStream_t st) {
    constexpr int WM = 2, WN = 4, TM = 2, TN = 2, MT = 128, NT = 256;
    const int TW = a.Wout < 32 ? a.Wout : 32, TH = NT / TW;
    // LDS of the 3x3 phase (three weight stages + two halo buffers); the 1x1 phase and the epilogue tile fit inside it
    size_t lds = (size_t)3 * 3 * MT * rec_bytes<PR>() + 2 * ((size_t)(TW + 2) * (TH + 2) + NDUMMY) * halo_pitch<PR>();
    dim3 grid(((a.Hout * a.Wout) / NT) * ((a.Cout + MT - 1) / MT) * a.B * a.nsplit);
    void (*kern)(ConvArgs);
    if constexpr (PR == PR_F16) kern = &conv_kcat_f16<WM, WN, TM, TN, MODE>;
    else kern = &conv_kcat_bf16x3<WM, WN, TM, TN, MODE>;
    static DeviceOnce once;
    if (first_on_device(once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, a);
}

int bf16_tile_of(const ConvArgs& a);     // conv_bf16.hip

template <int PR, int TAPS, int MODE>
void launch_tile_b(const ConvArgs& a, hipStream_t st) {
    const int tile = bf16_tile_of(a);
    switch (tile) {
        case 5:                                                           // 128 x 256, 8 compute waves (64 x 64 each)
            launch_one_b<PR, TAPS, 2, 4, 2, 2, MODE>(a, st);
            break;
#ifdef LOCO_DIAG
        case 6:                                                           // 128 x 128, 4 waves, compact LDS: two workgroups per CU (LOCO_CONV_2WG=1; diagnostics build)
            if constexpr (TAPS == 9) { launch_one_b2<PR, TAPS, 2, 2, 2, 2, MODE, 3>(a, st); break; }
            launch_one_b<PR, TAPS, 2, 2, 2, 2, MODE>(a, st);
            break;
#endif
        case 0: launch_one_b<PR, TAPS, 2, 2, 2, 2, MODE>(a, st); break;
        case 1: launch_one_b<PR, TAPS, 4, 1, 1, 2, MODE>(a, st); break;
        case 2: launch_one_b<PR, TAPS, 1, 4, 1, 1, MODE>(a, st); break;
        default: launch_one_b<PR, TAPS, 2, 2, 1, 1, MODE>(a, st); break;
    }
}


}  // namespace loco
