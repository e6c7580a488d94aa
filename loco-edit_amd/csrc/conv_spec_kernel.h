// Role-split 3x3 convolution kernel (stride 1, padded arena tensors, 128 couts x 256 pixels per workgroup): the same
// implicit GEMM, operand records, LDS layout and epilogue as conv_lowp_body<.., 9, 2, 4, 2, 2, MODE, 0> -- but the work of
// a stage is divided between waves by ROLE instead of every wave running the whole program in lock step:
//
//   waves 0..7  (compute): ds_read of the operand fragments + MFMA, nothing else.  No global memory instruction, no halo
//                conversion, no vmcnt wait: 128 accumulator + fragment registers instead of 256 with spills.
//   waves 8..11 (staging): one per SIMD, beside that SIMD's two compute waves.  They issue the LDS-DMA of the weight stage
//                after next, convert the next chunk's halo (GroupNorm / SiLU / tangent / cotangent map in fp32, hi + lo
//                split, ds_write) and re-load its registers, and they alone wait for memory.
//
// Why: in the lock-step kernel the measured launch is the SUM of its ingredients (profiles/r02_conv_analysis.md section 2:
// halo work 55 us + MFMA 145 us + epilogue + the rest of a 340 us launch; removing the halo work alone returned 16 %), because
// both waves of a SIMD convert, wait and multiply at the same time.  Here a SIMD's matrix pipe is fed by two waves that never
// leave the MFMA stream while the third wave's VALU / memory work fills the issue slots they leave (MFMA and VALU are separate
// pipes, MI355X_MICROARCH.md "Wave scheduling").  Twelve waves per CU = three per SIMD = 168 registers per lane, which the
// compute role fits once the staging state is gone.
//
// Pipeline (stage s = 3 * chunk + kernel row, ONE s_barrier per stage, all 12 waves):
//   compute, stage s:  taps 0..2 out of W[s % 3] / H[chunk & 1]; the fragments of the NEXT stage's first tap are read at
//                      the end of stage s (no exposed LDS latency behind the barrier), so W[(s+1) % 3] -- and, in row 2,
//                      H[(chunk+1) & 1] -- must be complete at the barrier that OPENS stage s;
//   staging, stage s:  LDS-DMA of W(s+2) into W[(s+2) % 3] (last read in stage s-1), issued first so it has the whole
//                      stage to land; row 0: convert part A of chunk+1 into H[(chunk+1) & 1] (last read in row 2 of
//                      chunk-1), re-load A's registers for chunk+2; row 1: the same for part B; row 2: nothing to convert;
//                      then the counted vmcnt that retires the DMA but leaves the re-loads in flight, lgkmcnt(0), barrier.
//   A part's registers are re-loaded right after their conversion and converted a whole chunk (three stages) later.
// The LDS-DMA is issued from inline assembly (cdna_hip_programming.md 5.7): behind the builtin, hipcc drains vmcnt(0) in
// front of the next ds_write, which would serialise the staging wave's DMA with its own conversion.
#pragma once

namespace loco {

// DMAC 1: the compute waves issue the weight LDS-DMA of the stage after next themselves (three pieces per wave at the top of
// the stage, retired by their own vmcnt(0) in front of the stage barrier: a whole stage later); 0: the staging waves do.
template <int PR, int MODE, int DMAC>
__device__ __forceinline__ void conv_spec_body(const ConvArgs& a) {
    constexpr int WM = 2, WN = 4, TM = 2, TN = 2, TAPS = 9;
    constexpr int NCOMP = WM * WN * 64;                // 512 compute threads (waves 0..7)
    constexpr int NSTG = 256;                          // staging threads (waves 8..11)
    constexpr int NSLOT = 2;                           // staging items per thread: the item space of the 512-thread kernel
    constexpr int HP = halo_pitch<PR>();
    constexpr int RB = rec_bytes<PR>();
    constexpr int NPC = RB / 16;
    constexpr int MT = WM * TM * 32;                   // 128
    constexpr int NT = WN * TN * 32;                   // 256
    constexpr int NTS = 3, NROW = 3;
    constexpr bool NEEDP = (MODE == CM_TAN_SILU || MODE == CM_COT_SILU);
    constexpr int WTOT = NTS * MT * NPC;               // 16-byte pieces of one weight stage
    constexpr int NDMA = DMAC ? NCOMP : NSTG;          // threads that issue the weight LDS-DMA
    constexpr int NWV = (WTOT + NDMA - 1) / NDMA;      // pieces per issuing thread (bf16x3: 6 staging / 3 compute)
    constexpr int WBYTES = NTS * MT * RB;
    constexpr int KP = 2;                              // channels per part (two parts A, B of a 4-channel item)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* const Wsb = smem_b;                 // 3 x [3 taps][MT] records
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool staging = wave >= WM * WN;

    const int TW = a.Wout < 32 ? a.Wout : 32;
    const int TH = NT / TW;
    const int tiles_x = a.Wout / TW;
    int tile_id, cot_id, zid;
    {   // same block order as the lock-step kernel (probes / cout tiles of one pixel tile adjacent and on one XCD)
        const int ntile = (a.Hout * a.Wout) / NT, ncot = (a.Cout + MT - 1) / MT, Z = a.B * a.nsplit;
        const int NTC = ntile * ncot, L = blockIdx.x;
        int T = 0;
        if ((ntile & 7) == 0) {
            int q = L >> 3;
            cot_id = q % ncot; q /= ncot;
            zid = q % Z;
            tile_id = (q / Z) * 8 + (L & 7);
        } else {
            if ((NTC & 7) == 0) { int q = L >> 3; zid = q % Z; T = (q / Z) * 8 + (L & 7); }
            else { zid = L % Z; T = L / Z; }
            tile_id = T % ntile; cot_id = T / ntile;
        }
        tile_id = __builtin_amdgcn_readfirstlane(tile_id);
        cot_id = __builtin_amdgcn_readfirstlane(cot_id);
        zid = __builtin_amdgcn_readfirstlane(zid);
    }
    const int oy0 = __builtin_amdgcn_readfirstlane((tile_id / tiles_x) * TH);
    const int ox0 = __builtin_amdgcn_readfirstlane((tile_id % tiles_x) * TW);
    const int co0 = cot_id * MT;
    const int b = __builtin_amdgcn_readfirstlane(zid / a.nsplit);
    const int split = __builtin_amdgcn_readfirstlane(zid % a.nsplit);
    const int halo_w = TW + 2, halo_h = TH + 2;
    const int halo_sz = halo_h * halo_w;
    const int HBYTES = (halo_sz + NDUMMY) * HP;
    unsigned char* const Hsb = smem_b + 3 * WBYTES;

    const int nchunks = (a.Cin + BKC - 1) / BKC;
    const int cps = __builtin_amdgcn_readfirstlane((nchunks + a.nsplit - 1) / a.nsplit);
    const int cbeg = split * cps;
    const int cend = (cbeg + cps < nchunks) ? cbeg + cps : nchunks;
    const int nch = cend - cbeg;
    const int clast = cend - 1;
    auto cclamp = [&](int c) { return __builtin_amdgcn_readfirstlane(c < clast ? c : clast); };
    // (chunk, row) of the stage `row_` rows past the start of chunk ci_
    auto stage_of = [&](int ci_, int row_, int& c_out, int& r_out) {
        c_out = cclamp(cbeg + ci_ + row_ / NROW);
        r_out = row_ % NROW;
    };

    // ---- weight stage LDS-DMA (issued by the staging waves, or by the compute waves when DMAC): pieces of this thread as
    // byte offsets from the (chunk, row) base; records past the padded cout range are clamped (their rows are never stored)
    const int wpitch = (a.Cout + 31) & ~31;
    const unsigned char* const wgb = reinterpret_cast<const unsigned char*>(a.wb);
    const int dtid = DMAC ? tid : tid - NCOMP;        // index among the issuing threads
    const int dwave = DMAC ? wave : wave - WM * WN;
    unsigned wrel[NWV];
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
        int e = dtid + i * NDMA;
        if (e < 0) e = 0;
        if (e >= WTOT) e = WTOT - 1;
        const int tap = e / (MT * NPC), rem = e - tap * (MT * NPC);
        int rec = co0 + rem / NPC;
        if (rec >= wpitch) rec = wpitch - 1;
        wrel[i] = (unsigned)((tap * wpitch + rec) * NPC + (rem % NPC)) * 16u;
    }
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    auto dma_w = [&](int chunk, int row, unsigned char* Wdst) {
        const unsigned char* wbase = wgb + (unsigned)(chunk * TAPS + row * NTS) * ((unsigned)wpitch * (unsigned)RB);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            // 64 lanes x 16 bytes land contiguously at the wave-uniform LDS address M0 (+ 16 * lane)
            if ((WTOT % NDMA) != 0 && dwave * 64 + i * NDMA >= WTOT) continue;      // wave-uniform
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                (unsigned)(size_t)(lds_u8*)(Wdst + (dwave * 64 + i * NDMA) * 16));
            const unsigned char* gsrc = wbase + wrel[i];
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
    };

    if (staging) {
        // =========================================== staging role =====================================================
        const int stid = tid - NCOMP;                  // 0..255
        const long in_plane = (long)a.Hin * a.Win;
        const float* inb = a.in + (long)b * a.in_bs;
        const float2* sxb = NEEDP ? a.sx : nullptr;
        const float* scb = (MODE == CM_GN_SILU || MODE == CM_GN) ? a.sc + (long)b * a.scsh_bs : nullptr;
        const float* shb = (MODE == CM_GN_SILU || MODE == CM_GN) ? a.sh + (long)b * a.scsh_bs : nullptr;
        const float* tcb = NEEDP ? a.tc + (long)b * a.tc_bs : nullptr;

        // ---- items: 4 consecutive halo pixels x 4 channels (quarter chunk q4); slot i is item stid + 256 i of the same
        // lane order as the 512-thread kernel (lane bits {seg & 3, row & 1, half, octet}: 2-way LDS store conflicts, 64-byte
        // global runs), the plain order when the padded index space does not fit
        const int nseg = (halo_w + 3) >> 2;
        int v_q4[NSLOT], v_rec[NSLOT][4];
        unsigned v_pm[NSLOT], v_goff[NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int vt = stid + sl * NSTG;
            const int per_q = halo_h * nseg;
            const int nsg4 = (nseg + 3) >> 2, nhy2 = (halo_h + 1) >> 1;
            bool have = false;
            int hy = 0, sg = 0, q4 = 0;
            if (32 * nsg4 * nhy2 <= NSLOT * NSTG) {
                const int hi = vt >> 5;
                q4 = (vt >> 3) & 3;
                sg = (hi % nsg4) * 4 + (vt & 3);
                hy = (hi / nsg4) * 2 + ((vt >> 2) & 1);
                have = hi < nsg4 * nhy2 && sg < nseg && hy < halo_h;
                if (!have) q4 = 0;
            } else if (vt < 4 * per_q) {
                q4 = vt / per_q;
                int rem = vt - q4 * per_q;
                hy = rem / nseg; sg = rem - hy * nseg;
                have = true;
            }
            int pos0 = -1, voff = 0, cnt = 0;
            unsigned pm = 0;
            if (have) {
                const int Y = oy0 - a.pad + hy, X0 = ox0 - a.pad + 4 * sg;
                const bool rowok = (Y >= 0 && Y < a.Hin);
                pos0 = hy * halo_w + 4 * sg;
                cnt = halo_w - 4 * sg < 4 ? halo_w - 4 * sg : 4;
                voff = rowok ? Y * a.Win + X0 : 0;
#pragma unroll
                for (int pxi = 0; pxi < 4; ++pxi)
                    if (rowok && X0 + pxi >= 0 && X0 + pxi < a.Win && pxi < cnt) pm |= 1u << pxi;
            }
#pragma unroll
            for (int pxi = 0; pxi < 4; ++pxi)
                v_rec[sl][pxi] = (pos0 >= 0 && pxi < cnt) ? pos0 + pxi : halo_sz + (stid & (NDUMMY - 1));
            v_q4[sl] = q4; v_pm[sl] = pm;
            v_goff[sl] = (unsigned)(((long)q4 * 4 * in_plane + voff + 16) * 4);
        }

        struct HaloRegs {
            f32x4 dq[KP];
            f32x4 sq[NEEDP ? KP : 1][2];
            f32x4 cq[1];
        };
        auto cq_a = [&](const HaloRegs& R, int kk) -> float {
            if constexpr (MODE == CM_NONE) return 0.f;
            else if constexpr (NEEDP) return R.cq[0][kk * 2];
            else return R.cq[0][kk];
        };
        auto cq_b = [&](const HaloRegs& R, int kk) -> float {
            if constexpr (MODE == CM_NONE) return 0.f;
            else if constexpr (NEEDP) return R.cq[0][kk * 2 + 1];
            else return R.cq[0][2 + kk];
        };
        auto prefetch = [&](HaloRegs& R, int chunk, int part, int sl) {
            const unsigned cb = (unsigned)chunk * ((unsigned)(BKC * 4) * (unsigned)in_plane);
            const char* pk = reinterpret_cast<const char*>(inb - 16) + cb;
            const char* sk = reinterpret_cast<const char*>(reinterpret_cast<const float*>(sxb) - 32) + 2u * cb;
            const unsigned pl = (unsigned)in_plane * 4u;
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                const unsigned po = v_goff[sl] + (unsigned)(part * KP + kk) * pl;
                R.dq[kk] = *reinterpret_cast<const f32x4_u*>(pk + po);
                if constexpr (NEEDP) {
                    R.sq[kk][0] = *reinterpret_cast<const f32x4_u*>(sk + 2u * po);
                    R.sq[kk][1] = *reinterpret_cast<const f32x4_u*>(sk + 2u * po + 16);
                }
            }
            const int c0 = chunk * BKC + v_q4[sl] * 4 + part * KP;
            if constexpr (MODE != CM_NONE) {
                if constexpr (NEEDP) {
                    R.cq[0] = *reinterpret_cast<const f32x4_u*>(tcb + 2 * c0);
                } else {
                    const f32x2 sa = *reinterpret_cast<const f32x2_u*>(scb + c0), sb = *reinterpret_cast<const f32x2_u*>(shb + c0);
                    R.cq[0] = f32x4{sa[0], sa[1], sb[0], sb[1]};
                }
            }
        };
        // vector-memory instructions of one prefetch (for the counted wait that leaves the re-loads in flight)
        constexpr int NLD = KP + (NEEDP ? 2 * KP + 1 : (MODE != CM_NONE ? 2 : 0));
        auto convert = [&](const HaloRegs& R, int part, int sl, unsigned char* Hd) {
            const int oct = v_q4[sl] >> 1, half = v_q4[sl] & 1;
#pragma unroll
            for (int pxi = 0; pxi < 4; ++pxi) {
                float r[KP];
#pragma unroll
                for (int kk = 0; kk < KP; ++kk) {
                    const float d = R.dq[kk][pxi];
                    float v = d;
                    if constexpr (MODE == CM_GN_SILU) {
                        const float y = fmaf(cq_a(R, kk), d, cq_b(R, kk));
                        v = y * sigmoidf2_(y);
                    } else if constexpr (MODE == CM_GN) {
                        v = fmaf(cq_a(R, kk), d, cq_b(R, kk));
                    } else if constexpr (NEEDP) {
                        const float Sv = R.sq[kk][pxi >> 1][(pxi & 1) * 2], xh = R.sq[kk][pxi >> 1][(pxi & 1) * 2 + 1];
                        if constexpr (MODE == CM_TAN_SILU) v = Sv * (d - cq_a(R, kk) - xh * cq_b(R, kk));
                        else v = Sv * d - cq_a(R, kk) - xh * cq_b(R, kk);
                    }
                    r[kk] = ((v_pm[sl] >> pxi) & 1u) ? v : 0.0f;
                }
                unsigned char* dst = Hd + hrec_off<PR>(v_rec[sl][pxi], oct) + half * 8 + part * (KP * 2);
                unsigned h, l;
                cvt2<PR>(r[0], r[1], h, l);
                *reinterpret_cast<unsigned*>(dst) = h;
                if constexpr (PR == PR_BF16X3) *reinterpret_cast<unsigned*>(dst + 32) = l;
            }
        };
        auto stage_end = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };

        // ONE register set per slot, alternating between the two parts of a chunk: a part is loaded one stage before it is
        // converted (row 2 loads A of the next-but-one chunk ... see the schedule below).  Two sets (every part a whole
        // chunk in flight) do not fit the 168 registers of a 12-wave workgroup, and a spill here is worse than a stall: the
        // scratch re-load's vmcnt wait also waits for every older vector-memory operation of the wave, i.e. for the weight
        // DMA and the halo loads issued just before it.
        //   row 0: convert A(c+1) -> [DMA] -> load B(c+1)      row 1: convert B(c+1) -> [DMA]      row 2: [DMA] -> load A(c+2)
        // The DMA comes AFTER the conversion: hipcc does not see the asm DMA, believes the part's loads are the only
        // outstanding operations and waits vmcnt(0) in front of the conversion -- with the DMA in flight that wait would
        // cover it too.
        HaloRegs hr[NSLOT];
        if (nch > 0) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) prefetch(hr[sl], cbeg, part, sl);
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) convert(hr[sl], part, sl, Hsb);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!DMAC) {
                int c1, r1;
                dma_w(cbeg, 0, Wsb);
                stage_of(0, 1, c1, r1);
                dma_w(c1, r1, Wsb + WBYTES);
            }
            const int cn = cclamp(cbeg + 1);
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) prefetch(hr[sl], cn, 0, sl);
            // the two weight stages (older than the part loads) have landed; the part loads stay in flight
            if constexpr (!DMAC) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSLOT * NLD) : "memory");
            stage_end();
        }
        for (int ci = 0; ci < nch; ++ci) {
            const int chunk = cbeg + ci;
            unsigned char* const Hnxt = Hsb + ((ci + 1) & 1) * HBYTES;
#pragma unroll
            for (int row = 0; row < NROW; ++row) {
                __builtin_amdgcn_sched_barrier(0);
                if (row < 2) {
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) convert(hr[sl], row, sl, Hnxt);      // part A in row 0, part B in row 1
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!DMAC) {
                    int c2, r2;
                    stage_of(ci, row + 2, c2, r2);
                    dma_w(c2, r2, Wsb + ((row + 2) % 3) * WBYTES);       // weights of the stage after next
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (row == 0) {
                    const int cn = cclamp(chunk + 1);
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) prefetch(hr[sl], cn, 1, sl);
                    if constexpr (!DMAC) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSLOT * NLD) : "memory");
                } else if (row == 1) {
                    if constexpr (!DMAC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    const int cn = cclamp(chunk + 2);
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) prefetch(hr[sl], cn, 0, sl);
                    if constexpr (!DMAC) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSLOT * NLD) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                stage_end();
            }
        }
        return;                                                 // the epilogue belongs to the compute waves
    }

    // ============================================= compute role =======================================================
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int hbyte[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = (wn * TN + j) * 32 + l31;
        const int ty = p / TW, tx = p - ty * TW;
        hbyte[j] = hrec_off<PR>(ty * halo_w + tx, khalf);
    }
    int aoff_hi[TM], aoff_lo[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (wm * TM + i) * 32 + l31;
        aoff_hi[i] = rec_off<PR>(p, khalf);
        aoff_lo[i] = PR == PR_F16 ? 0 : rec_off<PR>(p, 2 + khalf);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int NLO = PR == PR_BF16X3 ? 1 : 0;
    struct Frag { s16x8 ah[TM], al[NLO ? TM : 1], bh[TN], bl[NLO ? TN : 1]; };
    auto load_frag = [&](Frag& f, const unsigned char* Ws, const unsigned char* Hs, int row, int tp) {
        const int tapoff = row * halo_w + tp;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.ah[i] = *reinterpret_cast<const s16x8*>(Ws + tp * MT * RB + aoff_hi[i]);
            if constexpr (NLO) f.al[i] = *reinterpret_cast<const s16x8*>(Ws + tp * MT * RB + aoff_lo[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned char* hp = Hs + tapoff * HP + hbyte[j];
            f.bh[j] = *reinterpret_cast<const s16x8*>(hp);
            if constexpr (NLO) f.bl[j] = *reinterpret_cast<const s16x8*>(hp + 32);
        }
    };
    auto mma_one = [&](const Frag& f, int i, int j) {
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
    };
    auto mma_head = [&](const Frag& f) { mma_one(f, 0, 0); };
    auto mma_tail = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (i + j > 0) mma_one(f, i, j);
    };
    // the barrier of the compute role: its ds_reads of the buffers the staging waves overwrite next have been consumed by
    // MFMAs already, the fragments read for the next stage's first tap stay in flight across it
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    Frag fr[2];
    if (nch > 0) {
        if constexpr (DMAC) {
            int c1, r1;
            dma_w(cbeg, 0, Wsb);
            stage_of(0, 1, c1, r1);
            dma_w(c1, r1, Wsb + WBYTES);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();                                        // prologue: first halo chunk and two weight stages are in LDS
        load_frag(fr[0], Wsb, Hsb, 0, 0);             // the only exposed operand read of the tile
    }
    auto chunk_body = [&](auto ptag, const int ci) {
        constexpr int P = decltype(ptag)::value;
        unsigned char* const Hcur = Hsb + P * HBYTES;
        unsigned char* const Hnxt = Hsb + (1 - P) * HBYTES;
#pragma unroll
        for (int row = 0; row < NROW; ++row) {
            unsigned char* const Wcur = Wsb + row * WBYTES;
            unsigned char* const Wnx1 = Wsb + ((row + 1) % 3) * WBYTES;
            Frag& fa = fr[(P + row) & 1];              // taps 0 and 2 of this stage
            Frag& fb = fr[(P + row + 1) & 1];          // tap 1, then tap 0 of the next stage
            // one tap = [first MFMAs of the tap | ds_reads of the NEXT tap | remaining MFMAs]
            mma_head(fa);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMAC) {                      // weights of the stage after next, behind the first MFMAs of the stage
                int c2, r2;
                stage_of(ci, row + 2, c2, r2);
                dma_w(c2, r2, Wsb + ((row + 2) % 3) * WBYTES);
                __builtin_amdgcn_sched_barrier(0);
            }
            load_frag(fb, Wcur, Hcur, row, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_tail(fa);
            __builtin_amdgcn_sched_barrier(0);
            mma_head(fb);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(fa, Wcur, Hcur, row, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma_tail(fb);
            __builtin_amdgcn_sched_barrier(0);
            mma_head(fa);
            __builtin_amdgcn_sched_barrier(0);
            if (row + 1 < NROW) load_frag(fb, Wnx1, Hcur, row + 1, 0);
            else load_frag(fb, Wnx1, Hnxt, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_tail(fa);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMAC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this stage's DMA, issued a stage ago
            bar();
        }
    };
    for (int ci = 0; ci < nch; ci += 2) {
        chunk_body(std::integral_constant<int, 0>{}, ci);
        if (ci + 1 < nch) chunk_body(std::integral_constant<int, 1>{}, ci + 1);
    }
    // the staging waves have ended (a barrier only counts live waves); the compute waves retire their last (unused)
    // fragment reads before the operand buffers become the epilogue's staging tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    conv_lowp_epilogue<WM, WN, TM, TN>(a, acc, smem_b, co0, oy0, ox0, TW, tile_id, b, split);
}

template <int MODE, int DMAC>
__global__ __launch_bounds__(768) void conv_spec_bf16x3(ConvArgs a) { conv_spec_body<PR_BF16X3, MODE, DMAC>(a); }
template <int MODE, int DMAC>
__global__ __launch_bounds__(768) void conv_spec_f16(ConvArgs a) { conv_spec_body<PR_F16, MODE, DMAC>(a); }

// can this launch take the role-split kernel?  (what launch_one_b sends to <9,2,4,2,2,MODE,0>)
inline bool conv_spec_ok(const ConvArgs& a) {
    return a.stride == 1 && !a.upsample && !a.zins && (a.Cin % BKC) == 0 && a.in_padded && a.pad == 1;
}

int conv_spec_dma_by_compute();       // conv_bf16.hip (LOCO_SPEC_DMA: 1 = the compute waves issue the weight LDS-DMA)

template <int PR, int MODE, int DMAC>
static void launch_conv_spec_v(const ConvArgs& a, hipStream_t st) {
    constexpr int MT = 128, NT = 256;
    const int TW = a.Wout < 32 ? a.Wout : 32, TH = NT / TW;
    const int halo_w = TW + 2, halo_h = TH + 2;
    size_t lds = (size_t)3 * 3 * MT * rec_bytes<PR>() + 2 * ((size_t)halo_w * halo_h + NDUMMY) * halo_pitch<PR>();
    const size_t stage_bytes = (size_t)2 * 32 * NT * 4;
    if (lds < stage_bytes) lds = stage_bytes;
    dim3 grid(((a.Hout * a.Wout) / NT) * ((a.Cout + MT - 1) / MT) * a.B * a.nsplit);
    auto kern = PR == PR_F16 ? &conv_spec_f16<MODE, DMAC> : &conv_spec_bf16x3<MODE, DMAC>;
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (first_on_device(once))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(kern, grid, dim3(768), lds, st, a);
}
template <int PR, int MODE>
static void launch_conv_spec(const ConvArgs& a, hipStream_t st) {
    if (conv_spec_dma_by_compute()) launch_conv_spec_v<PR, MODE, 1>(a, st);
    else launch_conv_spec_v<PR, MODE, 0>(a, st);
}

}  // namespace loco
