// Dual-probe 3x3 convolution tile (round 5): one workgroup = 128 couts x TWO probes' 256-pixel tiles (the same 8 x 32 pixel
// patch of two samples of the batch) on v_mfma_f32_32x32x16_bf16 with split-bf16 operands, PERSISTENT over its work units.
//
// Why (profiles/r05_conv3x3_tan_pmc_mem_*.csv): on the 128 x 256 tile every workgroup pulls 72 KB of weight
// records + 65 KB of halo data per 16-channel chunk through its CU's vector-memory path (L2 serves ~2.3 GB per launch of the
// 128 -> 128 tangent conv, the texture-data unit waits for the cache a third of the launch), and ~20 % of a tile's life is
// per-tile fixed cost (index setup, first loads, write-out: 1 700 of 3 000 vector instructions per wave sit outside the stage
// loop).  Here
//   * each weight tap staged in LDS feeds TWO probes' pixel tiles (weight bytes per FLOP halved), the two probes share the loads
//     of the primal {S, xhat} cache in the tangent / cotangent forms (halo bytes per FLOP -33 %): -42 % bytes per FLOP;
//   * a wave holds 64 couts x (2 x 64) pixels = 128 accumulator registers; the weight fragment of a tap is read from LDS ONCE and
//     used for both probes' halves (24 MFMA groups per tap and wave, 72 x 3 MFMAs per kernel row);
//   * weights travel by LDS-DMA in a ring of SIX one-tap slots (8 KB each), issued five taps ahead; one raw s_barrier per tap;
//     operand fragments are read one half-tap ahead across the barrier (nothing exposed but the first tap of a unit);
//   * the halo of the next chunk is loaded one part (2 of 4 channels x 4 pixels x both probes) at a time into ONE register set,
//     two to four half-taps before its conversion (an L2 hit returns in ~300 cycles, a half-tap is ~1 000);
//   * the per-thread index setup happens once per workgroup, the workgroup then walks its units (grid = min(units, CUs)).
// LDS: 4 halo buffers [probe][chunk parity] x 348 records x 80 B = 111 360 B + 6 x 8 192 B = 160 512 B; the epilogue stages the
// accumulators through the same memory (rounds of 2 x RPW cout rows x 512 pixels) and writes 16-byte nontemporal stores.
//
// Eligibility (conv_bf16.hip conv_dual_ok): 3x3, stride 1, pad 1, whole 16-channel chunks, input in a padded engine arena, images
// at least 32 wide, whole 128-cout tiles, no split-K, modes raw / GroupNorm+SiLU forward / tangent / cotangent.  An odd last probe
// runs on the 128 x 256 kernel of conv_bf16_kernel.h.  Reference: the convs of models/ddpm/diffusion.py:855-912 under
// torch.autograd.functional.jvp / vjp (edit.py:2406-2504).
#pragma once
#include "conv_bf16_kernel.h"

namespace loco {

constexpr int DU_HW = 34, DU_HH = 10, DU_HALO = DU_HW * DU_HH, DU_NSLOT = 6;


template <int PR>
struct DualLds {
    static constexpr int HP = halo_pitch<PR>();
    static constexpr int RB = rec_bytes<PR>();
    static constexpr int HBYTES = (DU_HALO + NDUMMY) * HP;       // one halo buffer
    static constexpr int WSLOT = 128 * RB;                       // one tap of 128 couts
    static constexpr int H_OFF = 0;                              // H[probe][parity] at (parity * 2 + probe) * HBYTES
    static constexpr int W_OFF = 4 * HBYTES;
    static constexpr int TOTAL = W_OFF + DU_NSLOT * WSLOT;
};

// Epilogue of the dual tile: rounds of 2 (wave rows) x RPW cout rows x 512 pixels through LDS, then 16-byte stores.
// acc[i][j]: i = 32-cout block of the wave, j = {probe 0: 0, 1; probe 1: 2, 3} 32-pixel blocks.
template <int RPW>
__device__ __forceinline__ void conv_dual_epilogue(const ConvArgs& a, f32x16 (&acc)[2][4], float* const S, const int co0,
                                                   const int oy0, const int ox0, const int tile_id, const int b0) {
    constexpr int NTHR = 512, NT = 512, NQ = NT / 4;
    constexpr int SROWS = 2 * RPW, NTASK = (SROWS * NQ) / NTHR, NRB = RPW / 8;      // NRB: 8-row register groups per round
    static_assert(RPW == 8 || RPW == 16 || RPW == 32, "rows per wave and round");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned out_plane = (unsigned)(a.Hout * a.Wout);
    const int probe = wave & 1;                    // task rows: 128 quads = 2 waves of one row; odd waves hold probe 1's half
    const int b = b0 + probe;
    float* const ob = a.out + (long)b * a.out_bs;
    const float* const rb = a.res ? a.res + (long)b * a.res_bs : nullptr;
    const float* const b2 = a.bias2 ? a.bias2 + (long)b * a.bias2_bs : nullptr;
    const bool accu = a.accumulate != 0;
    const int ntile = (a.Hout * a.Wout) / 256;
    float* const sp = (a.st_kind == ST_FWD) ? a.st_part + (long)b * a.Cout * ntile * 2 : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += NRB) {
            // accumulators -> S[wm * RPW + row][pixel column]
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < NRB; ++g)
#pragma unroll
                    for (int r3 = 0; r3 < 4; ++r3)
                        S[(wm * RPW + g * 8 + 4 * khalf + r3) * NT + (j >> 1) * 256 + (wn * 2 + (j & 1)) * 32 + l31] =
                            acc[i][j][(g0 + g) * 4 + r3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q0 = 0; q0 < NTASK; q0 += 8) {
                constexpr int NB = NTASK < 8 ? NTASK : 8;
                f32x4 v[NB];
                unsigned off[NB];
                int cos_[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int t = (q0 + q) * NTHR + tid;
                    const int srow = t >> 7, quad = t & 127;
                    v[q] = *reinterpret_cast<const f32x4*>(&S[srow * NT + quad * 4]);
                    const int co = co0 + ((srow / RPW) * 2 + i) * 32 + g0 * 8 + (srow % RPW);
                    const int p = (quad & 63) * 4;
                    cos_[q] = co;
                    off[q] = (unsigned)co * out_plane + (unsigned)((oy0 + (p >> 5)) * a.Wout + ox0 + (p & 31));
                }
                if (rb || accu) {
                    f32x4 rv[NB];
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
                        if (rb) t4 = a.res_scale * *reinterpret_cast<const f32x4*>(rb + off[q]);
                        if (accu) t4 += *reinterpret_cast<const f32x4*>(ob + off[q]);
                        rv[q] = t4;
                    }
#pragma unroll
                    for (int q = 0; q < NB; ++q) v[q] += rv[q];
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    float add = 0.f;
                    if (a.bias) add += a.bias[cos_[q]];
                    if (b2) add += b2[cos_[q]];
                    v[q] += add;
                    __builtin_nontemporal_store(v[q], reinterpret_cast<f32x4*>(ob + off[q]));
                }
                if (sp) {
                    // forward GroupNorm statistics of the finished tile (see conv_lowp_epilogue): the 64 lanes of a wave hold one
                    // cout row of one probe's 256 pixels
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        const float pivot = __shfl(v[q][0], 0, 64);
                        const float d0 = v[q][0] - pivot, d1 = v[q][1] - pivot, d2 = v[q][2] - pivot, d3 = v[q][3] - pivot;
                        float s1 = (d0 + d1) + (d2 + d3);
                        float s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
                        for (int m = 1; m < 64; m <<= 1) {
                            s1 += __shfl_xor(s1, m, 64);
                            s2 += __shfl_xor(s2, m, 64);
                        }
                        const float m = s1 * (1.0f / 256.0f);
                        if (lane == 0)
                            *reinterpret_cast<f32x2*>(sp + ((long)cos_[q] * ntile + tile_id) * 2) = f32x2{pivot + m, s2 - s1 * m};
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the round's read-back is done in every wave
            __builtin_amdgcn_s_barrier();
        }
    }
}

template <int PR, int MODE>
__global__ __launch_bounds__(512, 1) void conv_dual_bf16x3(ConvArgs a, int nunits) {
    using L = DualLds<PR>;
    constexpr int HP = L::HP, RB = L::RB, HBYTES = L::HBYTES, WSLOT = L::WSLOT;
    constexpr bool NEEDP = (MODE == CM_TAN_SILU || MODE == CM_COT_SILU);
    constexpr bool HASC = (MODE != CM_NONE);
    // vector-memory instructions of one halo part (2 channels x 4 pixels x 2 probes [+ the shared {S, xhat} pairs] + constants)
    constexpr int NL = 4 + (NEEDP ? 4 + 2 : (HASC ? 4 : 0));
    static_assert(PR == PR_BF16X3, "one LDS-DMA per wave and tap: 64-byte records");
    static_assert(L::TOTAL <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* const Hb = smem_b + L::H_OFF;
    unsigned char* const Wb = smem_b + L::W_OFF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    const int W_ = a.Wout, H_ = a.Hout;
    const int tiles_x = W_ >> 5;
    const int ntile = (H_ * W_) >> 8, ncot = a.Cout >> 7, npair = a.B >> 1;
    const unsigned in_plane = (unsigned)(a.Hin * a.Win);
    const int nch = a.Cin / BKC;
    const int clast = nch - 1;
    const int wpitch = (a.Cout + 31) & ~31;

    // ---- per-thread constants of the workgroup (tile-invariant) ---------------------------------------------------------
    // staging item: 4 consecutive halo pixels x 4 channels (quarter chunk v_q4); lane order as in conv_lowp_body
    const int hi_ = tid >> 5;
    const int v_q4 = (tid >> 3) & 3;
    const int sg = (hi_ % 3) * 4 + (tid & 3);
    const int hy = (hi_ / 3) * 2 + ((tid >> 2) & 1);
    const bool have = hi_ < 15 && sg < 9 && hy < DU_HH;
    const int v_cnt = have ? (DU_HW - 4 * sg < 4 ? DU_HW - 4 * sg : 4) : 0;
    // byte offset of the item's pixel records inside a halo buffer (+ octet / half): record pxi of an item sits pxi * HP behind
    // the first.  The last segment of a row holds 2 pixels: its other two, and all four of a lane without an item, go to dump
    // records (dump index tid & 3, + pxi <= 6 < NDUMMY).  Two registers: the base of pixels 0, 1 and the base of pixels 2, 3.
    unsigned vb01, vb23;
    {
        const unsigned inrec = (unsigned)(v_q4 >> 1) * 16u + (unsigned)(v_q4 & 1) * 8u;
        const unsigned vdst0 = (unsigned)(hy * DU_HW + 4 * sg) * HP + inrec;
        const unsigned vdump = (unsigned)(DU_HALO + (tid & 3)) * HP + inrec;
        vb01 = v_cnt >= 2 ? vdst0 : vdump;
        vb23 = v_cnt == 4 ? vdst0 : vdump;
    }
    // operand fragment offsets.  B (pixel records): p = (wn * 2 + jj) * 32 + l31 -> patch row wn * 2 + jj, column l31: jj = 1 sits
    // DU_HW records behind jj = 0.  A (weight records): cout block i = 1 sits 32 records behind i = 0 (same swizzle phase)
    const unsigned hbyte0 = (unsigned)hrec_off<PR>((wn * 2) * DU_HW + l31, khalf);
    const unsigned aoff_hi0 = (unsigned)L::W_OFF + (unsigned)rec_off<PR>(wm * 64 + l31, khalf);
    const unsigned aoff_lo0 = (unsigned)L::W_OFF + (unsigned)rec_off<PR>(wm * 64 + l31, 2 + khalf);
    const unsigned wlane = (unsigned)tid * 16u;          // this lane's 16 bytes of a tap's 8 KB of weight records

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef const __attribute__((address_space(1))) unsigned char glb_u8;

    struct HaloRegs {
        f32x4 d0[2], d1[2];                   // [channel of the part] x 4 pixels, probe 0 / probe 1
        f32x4 sq[NEEDP ? 2 : 1][2];           // {S, xhat} of those pixels (shared by the probes)
        f32x4 cq[2];                          // tangent / cotangent constants per probe: {m1_0, m2_0, m1_1, m2_1}
        f32x2 ca[2], cs[2];                   // forward constants per probe: scale {a0, a1}, shift {b0, b1}
    };
    struct Frag { s16x8 h[2], l[2]; };       // [i] (weights: 32-cout block) or [jj] (pixels: 32-pixel block)

#ifdef LOCO_DUAL_STAMP
    // diagnostics build only (tests/diag/dual_stamps.py): s_memtime at the phase boundaries of every unit, wave 0 lane 0 -> workspace
    unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>(a.partial);
#define DU_STAMP(i) do { if (tid == 0) stamp_base[((u / gridDim.x) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DU_STAMP(i) do { } while (0)
#endif
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        DU_STAMP(0);
        // ---- unit -> (pixel tile, cout tile, probe pair); blocks u and u + 8 share an XCD: first the cout tiles of one
        // (pixel tile, pair) -- same input patch -- then the other pairs of that pixel tile (same primal cache and weights)
        int tile_id, cot_id, pair;
        if ((ntile & 7) == 0) {
            int q = u >> 3;
            cot_id = q % ncot; q /= ncot;
            pair = q % npair;
            tile_id = (q / npair) * 8 + (u & 7);
        } else {
            pair = u % npair;
            const int T = u / npair;
            tile_id = T % ntile; cot_id = T / ntile;
        }
        tile_id = __builtin_amdgcn_readfirstlane(tile_id);
        cot_id = __builtin_amdgcn_readfirstlane(cot_id);
        pair = __builtin_amdgcn_readfirstlane(pair);
        const int oy0 = (tile_id / tiles_x) * 8, ox0 = (tile_id % tiles_x) * 32;
        const int co0 = cot_id * 128;
        const int b0 = pair * 2;

        // tile-dependent part of the staging item
        unsigned v_pm = 0, v_goff;
        {
            const int Y = oy0 - 1 + hy, X0 = ox0 - 1 + 4 * sg;
            const bool rowok = have && Y >= 0 && Y < a.Hin;
#pragma unroll
            for (int pxi = 0; pxi < 4; ++pxi)
                if (rowok && X0 + pxi >= 0 && X0 + pxi < a.Win && pxi < v_cnt) v_pm |= 1u << pxi;
            const int voff = rowok ? Y * a.Win + X0 : 0;
            v_goff = (unsigned)(((long)v_q4 * 4 * in_plane + voff + 16) * 4);
        }
        const float* const inb0 = a.in + (long)b0 * a.in_bs;
        const float* const inb1 = inb0 + a.in_bs;
        const float2* const sxb = NEEDP ? a.sx : nullptr;
        const float* cb0 = nullptr; const float* cb1 = nullptr;      // tangent / cotangent {m1, m2} pairs per channel
        const float *sc0 = nullptr, *sc1 = nullptr, *sh0 = nullptr, *sh1 = nullptr;
        if constexpr (NEEDP) { cb0 = a.tc + (long)b0 * a.tc_bs; cb1 = cb0 + a.tc_bs; }
        else if constexpr (HASC) {
            sc0 = a.sc + (long)b0 * a.scsh_bs; sc1 = sc0 + a.scsh_bs;
            sh0 = a.sh + (long)b0 * a.scsh_bs; sh1 = sh0 + a.scsh_bs;
        }
        const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(a.wb) + (unsigned)co0 * (unsigned)RB;

        auto cclamp = [&](int c) { return __builtin_amdgcn_readfirstlane(c < clast ? c : clast); };
        // The part loads are issued by inline asm and waited for by hr_wait<N>() (an s_waitcnt tied to the registers): left to the
        // compiler, the first use of a loaded register gets `s_waitcnt vmcnt(0)` (its scoreboard does not count across the opaque
        // waits and the loop edge), which also waits for every weight DMA in flight -- the ring's five-tap lead collapsed to zero
        // twice per chunk (r05: 16 k of a unit's 190 k cycles).  vmcnt retires in order, so N = the vector-memory instructions
        // issued after the part's loads: the DMAs of the taps since.
        auto gload4 = [&](f32x4& dst, const char* base, unsigned off) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(off), "s"(base) : "memory");
        };
        auto gload2 = [&](f32x2& dst, const char* base, unsigned off) {
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(dst) : "v"(off), "s"(base) : "memory");
        };
        auto prefetch = [&](HaloRegs& R, int chunk_raw, int part) {
#ifdef LOCO_DUAL_STAMP
            const bool live = chunk_raw <= clast && !(a.no_deep & 2);      // what-if: halo loads collapsed onto one address
#else
            const bool live = chunk_raw <= clast;
#endif
            const int chunk = cclamp(chunk_raw);
            const unsigned cb = (unsigned)chunk * ((unsigned)(BKC * 4) * in_plane);
            const char* pk0 = reinterpret_cast<const char*>(inb0 - 16) + cb;
            const char* pk1 = reinterpret_cast<const char*>(inb1 - 16) + cb;
            const char* sk = reinterpret_cast<const char*>(reinterpret_cast<const float*>(sxb) - 32) + 2u * cb;
            const unsigned pl = in_plane * 4u;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned po = live ? v_goff + (unsigned)(part * 2 + kk) * pl : 64u;
                gload4(R.d0[kk], pk0, po);
                gload4(R.d1[kk], pk1, po);
                if constexpr (NEEDP) {
                    gload4(R.sq[kk][0], sk, 2u * po);
                    gload4(R.sq[kk][1], sk, 2u * po + 16u);
                }
            }
            const unsigned c0 = (unsigned)(chunk * BKC + v_q4 * 4 + part * 2);
            if constexpr (NEEDP) {
                gload4(R.cq[0], reinterpret_cast<const char*>(cb0), 8u * c0);
                gload4(R.cq[1], reinterpret_cast<const char*>(cb1), 8u * c0);
            } else if constexpr (HASC) {
                // {a0, a1} and {b0, b1} of each probe: cq[probe] = {a0, a1, b0, b1} (see convert)
                gload2(R.ca[0], reinterpret_cast<const char*>(sc0), 4u * c0);
                gload2(R.cs[0], reinterpret_cast<const char*>(sh0), 4u * c0);
                gload2(R.ca[1], reinterpret_cast<const char*>(sc1), 4u * c0);
                gload2(R.cs[1], reinterpret_cast<const char*>(sh1), 4u * c0);
            }
        };
        auto hr_wait = [&](HaloRegs& R, auto ntag) {
            constexpr int N = decltype(ntag)::value;
            if constexpr (NEEDP)
                asm volatile("s_waitcnt vmcnt(%10)" : "+v"(R.d0[0]), "+v"(R.d0[1]), "+v"(R.d1[0]), "+v"(R.d1[1]), "+v"(R.sq[0][0]),
                             "+v"(R.sq[0][1]), "+v"(R.sq[1][0]), "+v"(R.sq[1][1]), "+v"(R.cq[0]), "+v"(R.cq[1]) : "n"(N) : "memory");
            else if constexpr (HASC)
                asm volatile("s_waitcnt vmcnt(%8)" : "+v"(R.d0[0]), "+v"(R.d0[1]), "+v"(R.d1[0]), "+v"(R.d1[1]), "+v"(R.ca[0]),
                             "+v"(R.cs[0]), "+v"(R.ca[1]), "+v"(R.cs[1]) : "n"(N) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R.d0[0]), "+v"(R.d0[1]), "+v"(R.d1[0]), "+v"(R.d1[1]) : "n"(N) : "memory");
        };
        // one (part, probe) of a chunk's halo: map, split, store into halo buffer Hd
        auto convert = [&](const HaloRegs& R, int part, int probe, unsigned char* Hd, int px0 = 0, int px1 = 4) {
#pragma unroll
            for (int pxi = px0; pxi < px1; ++pxi) {
                float r[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const float d = probe ? R.d1[kk][pxi] : R.d0[kk][pxi];
                    const float ca = NEEDP ? R.cq[probe][kk * 2] : R.ca[probe][kk], cb = NEEDP ? R.cq[probe][kk * 2 + 1] : R.cs[probe][kk];
                    float v = d;
                    if constexpr (MODE == CM_GN_SILU) {
                        const float y = fmaf(ca, d, cb);
                        v = y * sigmoidf2_(y);
                    } else if constexpr (MODE == CM_GN_GELU) {
                        v = act_fwd(fmaf(ca, d, cb), ACT_GELU);
                    } else if constexpr (NEEDP) {
                        const float Sv = R.sq[kk][pxi >> 1][(pxi & 1) * 2], xh = R.sq[kk][pxi >> 1][(pxi & 1) * 2 + 1];
                        if constexpr (MODE == CM_TAN_SILU) v = Sv * fmaf(-xh, cb, d - ca);
                        else v = fmaf(-xh, cb, fmaf(Sv, d, -ca));
                    }
                    r[kk] = ((v_pm >> pxi) & 1u) ? v : 0.0f;
                }
                unsigned h, l;
                cvt2<PR>(r[0], r[1], h, l);
                unsigned char* dst = Hd + (pxi < 2 ? vb01 : vb23) + pxi * HP + part * 4;
                *reinterpret_cast<unsigned*>(dst) = h;
                *reinterpret_cast<unsigned*>(dst + 32) = l;
            }
        };
        // weights of tap `tap` of chunk `chunk` -> ring slot: 8 KB contiguous in the global layout, 1 KB per wave
        auto dma_w = [&](int chunk_raw, int tap, int slot) {
#ifdef LOCO_DUAL_STAMP
            const bool live = chunk_raw <= clast && !(a.no_deep & 4);      // what-if: weight DMAs collapsed onto one address
#else
            const bool live = chunk_raw <= clast;
#endif
            const int chunk = cclamp(chunk_raw);
            const unsigned char* src = wbase + (unsigned)(chunk * 9 + tap) * ((unsigned)wpitch * (unsigned)RB);
            __builtin_amdgcn_global_load_lds((glb_u8*)(src + (live ? wlane : 0u)), (lds_u8*)(Wb + slot * WSLOT + wave * 1024), 16, 0, 0);
        };

        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        // fA: weight fragments of a tap, two sets alternate per tap (the next tap's is read under this tap's second half).
        // fB: ONE set of pixel fragments: its 32-pixel block jj is re-loaded (other probe / next tap) as soon as the two MFMA groups
        // that read it have issued, two groups (>= 192 cycles) before its next use.
        Frag fA[2], fB;
        // Fragment reads and their waits are inline asm as well: the compiler answers "needs the reads issued before the last two"
        // with `s_waitcnt lgkmcnt(0)`, which exposes the LDS latency of the reads just issued twice per tap (r05: 17 k cycles of
        // a unit's loop with no memory traffic at all).  LDS operations retire in order: frag_wait<N> = all but the N youngest.
        // Issue order per tap: [p0: B1.jj0 (2) | B1.jj1 (2) | p1: A' (4) + B0'.jj0 (2) | B0'.jj1 (2)]; the LDS stores of a
        // conversion in between only make a wait stricter.
        const unsigned hb_lo = hbyte0, hb_hi = hbyte0 + 2u * HBYTES;      // bases of halo buffers 0, 1 / 2, 3 (16-bit offsets)
        auto lds_read = [&](s16x8& dst, unsigned addr, auto offtag) {
            constexpr int OFF = decltype(offtag)::value;
            static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
        };
        auto load_A = [&](Frag& f, auto slottag) {
            constexpr int SL = decltype(slottag)::value;
            lds_read(f.h[0], aoff_hi0, std::integral_constant<int, SL * WSLOT>{});
            lds_read(f.l[0], aoff_lo0, std::integral_constant<int, SL * WSLOT>{});
            lds_read(f.h[1], aoff_hi0, std::integral_constant<int, SL * WSLOT + 32 * RB>{});
            lds_read(f.l[1], aoff_lo0, std::integral_constant<int, SL * WSLOT + 32 * RB>{});
        };
        // pixel block jj of halo buffer BUF (= parity * 2 + probe) at byte offset TO (tap shift)
        auto load_B1 = [&](Frag& f, auto jjtag, auto buftag, auto totag) {
            constexpr int JJ = decltype(jjtag)::value, BUF = decltype(buftag)::value, TO = decltype(totag)::value;
            constexpr int OFF = (BUF & 1) * HBYTES + TO + JJ * (DU_HW * HP);
            lds_read(f.h[JJ], BUF < 2 ? hb_lo : hb_hi, std::integral_constant<int, OFF>{});
            lds_read(f.l[JJ], BUF < 2 ? hb_lo : hb_hi, std::integral_constant<int, OFF + 32>{});
        };
        // all but the N youngest LDS operations have retired: the weight fragment A and pixel block jj of B may be used
        auto frag_wait = [&](Frag& A, Frag& B, auto jjtag, auto ntag) {
            constexpr int JJ = decltype(jjtag)::value, N = decltype(ntag)::value;
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(A.h[0]), "+v"(A.l[0]), "+v"(A.h[1]), "+v"(A.l[1]), "+v"(B.h[JJ]), "+v"(B.l[JJ])
                         : "n"(N) : "memory");
        };
        auto mma_one = [&](const Frag& A, const Frag& B, int probe, int i, int jj) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, A.h[i]), al = __builtin_bit_cast(bf16x8, A.l[i]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, B.h[jj]), bl = __builtin_bit_cast(bf16x8, B.l[jj]);
            f32x16& c = acc[i][probe * 2 + jj];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
        };
        auto mma_pair = [&](const Frag& A, const Frag& B, int probe, int jj) {      // both cout blocks against pixel block jj
            mma_one(A, B, probe, 0, jj);
            mma_one(A, B, probe, 1, jj);
        };

        // ---- prologue of the unit --------------------------------------------------------------------------------------
        HaloRegs hr;
        DU_STAMP(1);
#pragma unroll
        for (int g = 0; g < 5; ++g) dma_w(0, g, g);                    // taps 0 .. 4 of chunk 0 -> slots 0 .. 4
        {
            HaloRegs hr2;
            prefetch(hr, 0, 0);
            prefetch(hr2, 0, 1);
            hr_wait(hr, std::integral_constant<int, NL>{});
            hr_wait(hr2, std::integral_constant<int, 0>{});
            convert(hr, 0, 0, Hb);
            convert(hr, 0, 1, Hb + HBYTES);
            convert(hr2, 1, 0, Hb);
            convert(hr2, 1, 1, Hb + HBYTES);
        }
        prefetch(hr, 1, 0);                                             // part A of chunk 1, pending at the first chunk's tap 3
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL) : "memory");    // everything older than it (the five DMAs) has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        DU_STAMP(2);
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I6 = std::integral_constant<int, 6>;
        load_A(fA[0], I0{});
        load_B1(fB, I0{}, I0{}, I0{});                                  // the only exposed operand reads of the unit
        load_B1(fB, I1{}, I0{}, I0{});

        // ---- chunk loop: 9 taps, one barrier each; P = chunk parity (halo buffers), ring slot of tap t = (3 P + t) % 6 ----------
        // Halo schedule of a chunk in half-taps h = 2 t + half (one register set hr, 18 half-taps):
        //   part A of chunk c+1: probe 0 converted at h 4, 5, probe 1 at h 6, 7 (ONE pixel of the item per MFMA pair: ~17 vector
        //   instructions beside 6 MFMAs.  Both waves of a SIMD run this stream in lock step, and a wave blocked on the matrix pipe
        //   cannot issue the vector work behind it: more than ~3 vector instructions per MFMA do not hide -- r05 what-if: with a
        //   (part, probe) converted inside one half-tap the conversions cost 9 % of the launch); part B loads at h 8, is converted
        //   at h 13, 14 (probe 0: read by every wave at the end of tap 8) and h 15, 16 (probe 1: read in the next tap 0); part A of
        //   chunk c+2 loads at h 17.  Five half-taps (~2.4 us) in flight each.
        //   (Tried and rejected: the two wave groups' schedules one tap apart, +5 % on the loop.)
#ifdef LOCO_DUAL_STAMP
        const bool wi_novalu = (a.no_deep & 8) != 0, wi_nobar = (a.no_deep & 16) != 0;      // what-if switches (timing only)
#endif
        auto tap = [&](auto ptag, auto ttag, const int ci) {
            constexpr int P = decltype(ptag)::value, t = decltype(ttag)::value;
            constexpr int slot = (3 * P + t) % DU_NSLOT, slotn = (slot + 1) % DU_NSLOT, slotd = (slot + 5) % DU_NSLOT;
            constexpr int tapoff = ((t / 3) * DU_HW + (t % 3)) * HP;
            constexpr int tapoffn = (((t + 1) / 3) * DU_HW + ((t + 1) % 3)) * HP;
            constexpr int AS = (P + t) & 1;
            unsigned char* const Hn0 = Hb + ((1 - P) * 2) * HBYTES;       // halo of the next chunk, probe 0 / 1
            unsigned char* const Hn1 = Hb + ((1 - P) * 2 + 1) * HBYTES;
            // vector work of MFMA pair q (0 .. 3) of this tap: half-tap h = 2 t + q / 2, pixel 2 (h & 1) + (q & 1) of a conversion
            auto work = [&](auto qtag) {
                constexpr int q = decltype(qtag)::value, h = 2 * t + q / 2, px = 2 * (h & 1) + (q & 1);
#ifdef LOCO_DUAL_STAMP
                if (wi_novalu) return;
#endif
                // first use of a part: its loads have landed (younger: the DMAs of taps 0, 1, 2 / of taps 5, 6)
                if constexpr (h == 4 && q == 0) hr_wait(hr, std::integral_constant<int, 3>{});
                if constexpr (h == 13 && q == 2) hr_wait(hr, std::integral_constant<int, 2>{});
                if constexpr (h == 4 || h == 5) convert(hr, 0, 0, Hn0, px, px + 1);
                else if constexpr (h == 6 || h == 7) convert(hr, 0, 1, Hn1, px, px + 1);
                else if constexpr (h == 13 || h == 14) convert(hr, 1, 0, Hn0, 2 * ((h - 13) & 1) + (q & 1), 2 * ((h - 13) & 1) + (q & 1) + 1);
                else if constexpr (h == 15 || h == 16) convert(hr, 1, 1, Hn1, 2 * ((h - 15) & 1) + (q & 1), 2 * ((h - 15) & 1) + (q & 1) + 1);
                else if constexpr (h == 8 && (q & 1) == 0) prefetch(hr, ci + 1, 1);
                else if constexpr (h == 17 && (q & 1) == 1) prefetch(hr, ci + 2, 0);
            };
            { constexpr int t5 = t + 5; dma_w(ci + (t5 >= 9 ? 1 : 0), t5 % 9, slotd); }     // weights five taps ahead
            __builtin_amdgcn_sched_barrier(0);
            // probe 0's half of the tap; probe 1's pixel fragments replace probe 0's block by block
            using BC0 = std::integral_constant<int, P * 2>;            // halo buffers: this chunk probe 0 / 1, next chunk probe 0
            using BC1 = std::integral_constant<int, P * 2 + 1>;
            using BN0 = std::integral_constant<int, (1 - P) * 2>;
            using TO = std::integral_constant<int, tapoff>;
            using TON = std::integral_constant<int, tapoffn>;
            work(std::integral_constant<int, 0>{});
            frag_wait(fA[AS], fB, I0{}, I2{});
            mma_pair(fA[AS], fB, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_B1(fB, I0{}, BC1{}, TO{});
            __builtin_amdgcn_sched_barrier(0);
            work(std::integral_constant<int, 1>{});
            frag_wait(fA[AS], fB, I1{}, I2{});
            mma_pair(fA[AS], fB, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_B1(fB, I1{}, BC1{}, TO{});
            __builtin_amdgcn_sched_barrier(0);
            // probe 1's half; the next tap's weight fragment and probe 0's pixel fragments are read under it
            work(std::integral_constant<int, 2>{});
            frag_wait(fA[AS], fB, I0{}, I2{});
            mma_pair(fA[AS], fB, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_A(fA[1 - AS], std::integral_constant<int, slotn>{});
            if constexpr (t < 8) load_B1(fB, I0{}, BC0{}, TON{});
            else load_B1(fB, I0{}, BN0{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            work(std::integral_constant<int, 3>{});
            frag_wait(fA[AS], fB, I1{}, I6{});
            mma_pair(fA[AS], fB, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t < 8) load_B1(fB, I1{}, BC0{}, TON{});
            else load_B1(fB, I1{}, BN0{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            // the weights of tap t + 2 (issued at the start of tap t - 3) have landed: younger are the DMAs of taps t - 2 .. t
            // and the part loads issued since (taps 4 and 8: one part in every window of four taps except the one that ends in tap 3)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(t == 3 ? 3 : 3 + NL) : "memory");
            // LDS stores of the conversions are read by other waves behind the barriers that close taps 7 (probe 0's next halo,
            // first read at the end of tap 8) and 8 (probe 1's, first read in the next tap 0); LDS operations retire in order, so
            // these two waits cover the earlier stores too.  Elsewhere the fragment reads just issued stay in flight across the
            // barrier.
            if constexpr (t == 7 || t == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef LOCO_DUAL_STAMP
            if (!wi_nobar)
#endif
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto chunk_body = [&](auto ptag, const int ci) {
            tap(ptag, std::integral_constant<int, 0>{}, ci); tap(ptag, std::integral_constant<int, 1>{}, ci);
            tap(ptag, std::integral_constant<int, 2>{}, ci); tap(ptag, std::integral_constant<int, 3>{}, ci);
            tap(ptag, std::integral_constant<int, 4>{}, ci); tap(ptag, std::integral_constant<int, 5>{}, ci);
            tap(ptag, std::integral_constant<int, 6>{}, ci); tap(ptag, std::integral_constant<int, 7>{}, ci);
            tap(ptag, std::integral_constant<int, 8>{}, ci);
        };
        for (int ci = 0; ci < nch; ci += 2) {
            chunk_body(std::integral_constant<int, 0>{}, ci);
            if (ci + 1 < nch) chunk_body(std::integral_constant<int, 1>{}, ci + 1);
        }
        DU_STAMP(3);
        // the ring's last (dead) DMAs and the fragment reads issued ahead retire before the accumulators take over the LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        DU_STAMP(4);
        conv_dual_epilogue<32>(a, acc, reinterpret_cast<float*>(smem_b), co0, oy0, ox0, tile_id, b0);
        DU_STAMP(5);
    }
#undef DU_STAMP
}

template <int PR, int MODE>
void launch_dual_b(const ConvArgs& a, hipStream_t st) {
    using L = DualLds<PR>;
    const int ntile = (a.Hout * a.Wout) / 256, ncot = a.Cout / 128, npair = a.B / 2;
    const int nunits = ntile * ncot * npair;
    size_t lds = L::TOTAL;
    const size_t stage_bytes = (size_t)64 * 512 * 4;
    if (lds < stage_bytes) lds = stage_bytes;
    auto kern = &conv_dual_bf16x3<PR, MODE>;
    static DeviceOnce once;
    static int ncu = 256;
    if (first_on_device(once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
    }
    // persistent: one workgroup per CU walks units u, u + grid, ... (a multiple of 8 keeps a workgroup's units on one XCD)
    int grid = nunits < ncu ? nunits : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, nunits);
}

}  // namespace loco
