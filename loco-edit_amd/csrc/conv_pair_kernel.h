// 3x3 stride-1 convolution on v_mfma_f32_16x16x32_bf16 with split-bf16 operands ("bf16x3"), K = TWO TAPS x 16 channels per MFMA.
//
// Why (profiles/r06_experiments.md section 1): the 128 x 256 tile of conv_bf16_kernel.h is bound by the clock the chip holds under
// random-data matrix work, not by its cycle count (same binary on all-zero operands: -17 ... -21 %), and the MFMA SHAPE alone is
// worth -6 ... -8 % of a launch (every 32x32x16 MFMA issued as two 16x16x32 MFMAs of the same registers).  A 16x16x32 MFMA
// contracts over 32 k-values: here the 16 channels of a chunk for tap t (lanes 0 - 31: k-groups 0, 1) and for tap t + 1 (lanes
// 32 - 63: k-groups 2, 3) -- a convolution is a sum over taps as much as over channels -- so the chunk stays 16 channels, the operand
// records, the weight layout in global memory and the LDS traffic per product stay what they are, and a product is still three
// MFMAs (lo*hi, hi*lo, hi*hi).  Same workgroup tile as the kernel it replaces (128 couts x 8 x 32 pixels, eight waves of 64 x 64),
// same epilogue (conv_lowp_epilogue_staged: LDS-staged write-out with residual / bias / norm-cotangent term / statistics).
//
// What differs:
//  * one stage = one TAP PAIR (48 MFMAs per wave), nine stages per two channel chunks (18 taps); needs an even number of chunks.
//  * LDS images are PIECE PLANES, not records: a fragment read of the 16x16x32 operand takes lane (row r = lane & 15, k-group
//    q = lane >> 4); the four 16-lane service groups of a ds_read_b128 each hold every r once, so any layout whose 16-byte slot
//    index is r modulo 16 is conflict-free.  Weights: [tap][cout block of 16][piece hi0|hi1|lo0|lo1][16 rows] -- the LDS-DMA
//    un-swizzles the global records on the way (each lane fetches the piece its LDS slot wants); halo: [piece][position], plane
//    size a multiple of 256 bytes.
//  * fragments are single-buffered and refilled IN PLACE: the A fragments of row block i are re-read for the next pair as soon as
//    row block i's MFMAs are issued, the B fragments during the last row block (and its tail at the top of the next stage) --
//    64 fragment registers like the kernel it replaces, not the 128 a double-buffered tap pair would need.
//  * four weight slots, LDS-DMA three pairs ahead; halo conversion spread over stages 0, 2, 5, 7 of the nine.
// Results differ from the 32x32x16 kernel by summation order only (tests compare at 1e-6).  Opt-in / default: conv_pair_ok().
#pragma once
#include "conv_bf16_kernel.h"

namespace loco {

constexpr int PK_NTHR = 512, PK_MT = 128, PK_NT = 256;
constexpr int PK_HW = 34, PK_HSZ = 340;            // halo of the 8 x 32 pixel tile
constexpr int PK_HPOS = 352;                       // positions per plane: 340 + 12 dump positions (lanes without a staging item)
constexpr int PK_PLANE = PK_HPOS * 16;             // 5632 B = 22 x 256
constexpr int PK_HBYTES = 4 * PK_PLANE;            // one halo buffer
constexpr int PK_TAPB = 8192;                      // one tap: 8 cout blocks x 4 pieces x 16 rows x 16 B
constexpr int PK_WSLOT = 2 * PK_TAPB;              // a tap pair
constexpr int PK_NSLOT = 4;
constexpr int PK_LDS = PK_NSLOT * PK_WSLOT + 2 * PK_HBYTES;      // 65536 + 45056 = 110592 B

template <int MODE>
__global__ __launch_bounds__(PK_NTHR, 1) void conv_pair_bf16x3(ConvArgs a) {
    constexpr int PR = PR_BF16X3;
    constexpr bool NEEDP = (MODE == CM_TAN_SILU || MODE == CM_COT_SILU);
    constexpr int KP = 2;                             // channels of a staging part
    constexpr int NLD = KP + (NEEDP ? 2 * KP + 1 : (MODE != CM_NONE ? 2 : 0));      // vector-memory instructions of one part's loads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* const Wsb = smem_b;
    unsigned char* const Hsb = smem_b + PK_NSLOT * PK_WSLOT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4, sel = kq >> 1;
    const int wm = wave >> 2, wn = wave & 3;

    // block -> (pixel tile, cout tile, probe): the XCD-aware order of conv_lowp_body
    int tile_id, cot_id, zid;
    {
        const int ntile = (a.Hout * a.Wout) / PK_NT, ncot = a.Cout / PK_MT, Z = a.B;
        const int NTC = ntile * ncot, L = blockIdx.x;
        int T = 0;
        if ((ntile & 7) == 0) {
            int q = L >> 3, q2, t8;
            idivmod_small(q, ncot, q2, cot_id);
            idivmod_small(q2, Z, t8, zid);
            tile_id = t8 * 8 + (L & 7);
        } else {
            if ((NTC & 7) == 0) { int q = L >> 3, t8; idivmod_small(q, Z, t8, zid); T = t8 * 8 + (L & 7); }
            else idivmod_small(L, Z, T, zid);
            idivmod_small(T, ntile, cot_id, tile_id);
        }
        tile_id = __builtin_amdgcn_readfirstlane(tile_id);
        cot_id = __builtin_amdgcn_readfirstlane(cot_id);
        zid = __builtin_amdgcn_readfirstlane(zid);
    }
    const int tiles_x = a.Wout >> 5;
    int ty0_, tx0_;
    idivmod_small(tile_id, tiles_x, ty0_, tx0_);
    const int oy0 = __builtin_amdgcn_readfirstlane(ty0_ * 8);
    const int ox0 = __builtin_amdgcn_readfirstlane(tx0_ * 32);
    const int co0 = cot_id * PK_MT;
    const int b = zid;
    const int nch = a.Cin >> 4;                        // channel chunks (even)
    const int clast = nch - 1;
    const long in_plane = (long)a.Hin * a.Win;

    // ---- weights: LDS piece e = i * 512 + tid of a pair slot = tap i, cout block `wave`, piece lane >> 4, row lane & 15
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef const __attribute__((address_space(1))) unsigned char glb_u8;
    const int wpitch = (a.Cout + 31) & ~31;
    const unsigned wtap = (unsigned)wpitch * 64u;                  // bytes of one tap's records in global memory
    const int cl = wave * 16 + r16;                                // cout inside the tile
    const unsigned wrel = (unsigned)(co0 + cl) * 64u + (unsigned)((kq ^ ((cl >> 2) & 3)) << 4);      // the global records are XOR-swizzled
    const unsigned char* const wgb = reinterpret_cast<const unsigned char*>(a.wb);
    // taps tg, tg + 1 counted from the tile's first chunk (tg = 9 * chunk + tap); past the end: one collapsed dead load per wave
    auto dma_pair = [&](const int tg, const int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool live = (tg + i) < 9 * nch;
            const unsigned char* src = wgb + (live ? (unsigned)(tg + i) * wtap + wrel : 0u);
            __builtin_amdgcn_global_load_lds((glb_u8*)src, (lds_u8*)(Wsb + slot * PK_WSLOT + i * PK_TAPB + wave * 1024), 16, 0, 0);
        }
    };

    // the first three pairs leave BEFORE the per-thread index setup below: their L2 -> LDS latency runs under it
    dma_pair(0, 0);
    dma_pair(2, 1);
    dma_pair(4, 2);

    // ---- halo staging item of this thread: 4 consecutive halo pixels x 4 channels (quarter chunk v_q4), as conv_lowp_body
    int v_q4 = (tid >> 3) & 3, v_cnt = 0, v_pos0 = -1;
    unsigned v_pm = 0;
    unsigned v_goff;
    {
        const int hi_ = tid >> 5;
        int hq_, hr_;
        idivmod_small(hi_, 3, hq_, hr_);
        const int sg = hr_ * 4 + (tid & 3), hy = hq_ * 2 + ((tid >> 2) & 1);
        const bool have = hi_ < 15 && sg < 9 && hy < 10;
        if (!have) v_q4 = 0;
        int v_voff = 0;
        if (have) {
            const int Y = oy0 - 1 + hy, X0 = ox0 - 1 + 4 * sg;
            const bool rowok = (Y >= 0 && Y < a.Hin);
            v_pos0 = hy * PK_HW + 4 * sg;
            v_cnt = PK_HW - 4 * sg < 4 ? PK_HW - 4 * sg : 4;
            v_voff = rowok ? Y * a.Win + X0 : 0;
#pragma unroll
            for (int pxi = 0; pxi < 4; ++pxi)
                if (rowok && X0 + pxi >= 0 && X0 + pxi < a.Win && pxi < v_cnt) v_pm |= 1u << pxi;
        }
        if (v_pos0 < 0) { v_pos0 = PK_HSZ + (tid & 7); v_cnt = 1; }      // no item: every pixel goes to one dump position
        v_goff = (unsigned)(((long)v_q4 * 4 * in_plane + v_voff + 16) * 4);
    }
    const float* const inb = a.in + (long)b * a.in_bs;
    const float2* const sxb = NEEDP ? a.sx : nullptr;
    const float* const scb = (MODE == CM_GN_SILU) ? a.sc + (long)b * a.scsh_bs : nullptr;
    const float* const shb = (MODE == CM_GN_SILU) ? a.sh + (long)b * a.scsh_bs : nullptr;
    const float* const tcb = NEEDP ? a.tc + (long)b * a.tc_bs : nullptr;
    struct HaloRegs {
        f32x4 dq[KP];
        f32x4 sq[NEEDP ? KP : 1][2];
        f32x4 cq;
    };
    HaloRegs hr;
    auto prefetch_hv = [&](HaloRegs& hr, const int chunk_raw, const int part) {
        const bool live = chunk_raw <= clast;
        const int chunk = __builtin_amdgcn_readfirstlane(live ? chunk_raw : clast);
        const unsigned cb = (unsigned)chunk * (64u * (unsigned)in_plane);
        const char* pk = reinterpret_cast<const char*>(inb - 16) + cb;
        const char* sk = reinterpret_cast<const char*>(reinterpret_cast<const float*>(sxb) - 32) + 2u * cb;
        const unsigned pl = (unsigned)in_plane * 4u;
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) {
            const unsigned po = live ? v_goff + (unsigned)(part * KP + kk) * pl : 64u;
            hr.dq[kk] = *reinterpret_cast<const f32x4_u*>(pk + po);
            if constexpr (NEEDP) {
                hr.sq[kk][0] = *reinterpret_cast<const f32x4_u*>(sk + 2u * po);
                hr.sq[kk][1] = *reinterpret_cast<const f32x4_u*>(sk + 2u * po + 16);
            }
        }
        const int c0 = chunk * 16 + v_q4 * 4 + part * KP;
        if constexpr (NEEDP) {
            hr.cq = *reinterpret_cast<const f32x4_u*>(tcb + 2 * c0);            // {a0, b0, a1, b1}
        } else if constexpr (MODE == CM_GN_SILU) {
            const f32x2 sa = *reinterpret_cast<const f32x2_u*>(scb + c0), sb = *reinterpret_cast<const f32x2_u*>(shb + c0);
            hr.cq = f32x4{sa[0], sb[0], sa[1], sb[1]};
        }
    };
    // converts the part in `hr` into the piece planes of halo buffer Hd: channels c = v_q4 * 4 + part * 2 (+ 1) of 4 pixels
    auto stage_hv = [&](const HaloRegs& hr, unsigned char* const Hd, const int part, const int p0, const int p1) {
        const int c0 = v_q4 * 4 + part * KP;
        unsigned char* const pl0 = Hd + (c0 >> 3) * PK_PLANE + (c0 & 7) * 2;
#pragma unroll
        for (int pxi = p0; pxi < p1; ++pxi) {
            float r[KP];
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                const float d = hr.dq[kk][pxi];
                const float ca = hr.cq[kk * 2], cb = hr.cq[kk * 2 + 1];
                float v = d;
                if constexpr (MODE == CM_GN_SILU) {
                    const float y = fmaf(ca, d, cb);
                    v = y * sigmoidf2_(y);
                } else if constexpr (NEEDP) {
                    const float Sv = hr.sq[kk][pxi >> 1][(pxi & 1) * 2], xh = hr.sq[kk][pxi >> 1][(pxi & 1) * 2 + 1];
                    if constexpr (MODE == CM_TAN_SILU) v = Sv * fmaf(-xh, cb, d - ca);
                    else v = fmaf(-xh, cb, fmaf(Sv, d, -ca));
                }
                r[kk] = ((v_pm >> pxi) & 1u) ? v : 0.0f;
            }
            unsigned h, l;
            cvt2<PR>(r[0], r[1], h, l);
            // pixels past the item's count (the 2-pixel last segment of a halo row; lanes without an item) go to a dump position
            unsigned char* dst = pl0 + (pxi < v_cnt ? v_pos0 + pxi : PK_HSZ + 8 + (tid & 3)) * 16;
            *reinterpret_cast<unsigned*>(dst) = h;
            *reinterpret_cast<unsigned*>(dst + 2 * PK_PLANE) = l;
        }
    };

    // ---- fragments (single set, refilled in place)
    s16x8 Ah[4], Al[4], Bh[4], Bl[4];
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int aoff = sel * PK_TAPB + wm * 4096 + (kq & 1) * 256 + r16 * 16;      // + slot, + row block i * 1024, lo: + 512
    const int hbl = ((wn * 2) * PK_HW + r16) * 16 + (kq & 1) * PK_PLANE;           // + per-pair tap origin, + col block j term, lo: + 2 planes
    auto load_A = [&](const int i, const int slot) {
        const unsigned char* w = Wsb + slot * PK_WSLOT + aoff + i * 1024;
        Ah[i] = *reinterpret_cast<const s16x8*>(w);
        Al[i] = *reinterpret_cast<const s16x8*>(w + 512);
    };
    // B fragment j of the pair whose first tap is group tap t0 (0 .. 18, even): k-groups 0, 1 read tap t0, k-groups 2, 3 tap t0 + 1.
    // The second tap's halo origin lies 16 bytes further (next column), 512 (next kernel row: t0 % 9 = 2 or 5) or one buffer less 70
    // positions (t0 = 8: first tap of the next chunk) -- three per-lane bases hbl + sel * that distance; everything else (buffer,
    // tap shift of the first tap, column block, lo plane) is a compile-time immediate of the ds_read (r06 counters: per-stage selects
    // and address adds made the kernel issue 27 % more vector instructions than the 32x32x16 kernel).
    const unsigned char* const hb16 = Hsb + hbl + sel * 16;
    const unsigned char* const hb512 = Hsb + hbl + sel * 512;
    const unsigned char* const hbW = Hsb + hbl + sel * (PK_HBYTES - (2 * PK_HW + 2) * 16);
    auto load_B = [&](const int j, const int t0) {
        const int tp = t0 % 9;
        const unsigned char* const base = tp == 8 ? hbW : ((tp % 3) == 2 ? hb512 : hb16);
        const unsigned char* h = base + ((t0 / 9) & 1) * PK_HBYTES + ((tp / 3) * PK_HW + tp % 3) * 16 + ((j >> 1) * PK_HW + (j & 1) * 16) * 16;
        Bh[j] = *reinterpret_cast<const s16x8*>(h);
        Bl[j] = *reinterpret_cast<const s16x8*>(h + 2 * PK_PLANE);
    };
    auto mma = [&](const int i, const int j) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, Ah[i]), al = __builtin_bit_cast(bf16x8, Al[i]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, Bh[j]), bl = __builtin_bit_cast(bf16x8, Bl[j]);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i][j], 0, 0, 0);
    };
    auto stage_end = [&](const int left_in_flight) {
        // everything issued before this stage has landed (the LDS-DMA of the previous stage; the part loads a later stage converts)
#if defined(PKW_NODMA) || defined(PKW_NOLOAD)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (left_in_flight == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + NLD) : "memory");
#endif
#ifndef PKW_NOBAR      // what-if builds (timing only, results wrong): no stage barrier / no conversion / no LDS-DMA / no part loads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#ifndef PKW_NOTIE
        // Every fragment read issued so far has returned (the wait above).  The compiler cannot see an inline-asm wait: by its own
        // count the reads of the previous stage are still pending, and it drains lgkmcnt to 0 in front of the next stage's first
        // MFMA -- i.e. behind that stage's fresh tail reads: one exposed LDS latency per stage.  Passing the LIVE fragment registers
        // (A 0 - 2, B 0 - 1: the others are dead here) through an empty asm makes them plain values again.
        // (one statement: as five separate ones the tangent form spills seven registers)
        asm volatile("" : "+v"(Ah[0]), "+v"(Al[0]), "+v"(Ah[1]), "+v"(Al[1]), "+v"(Ah[2]), "+v"(Al[2]), "+v"(Bh[0]), "+v"(Bl[0]), "+v"(Bh[1]), "+v"(Bl[1]));
#endif
    };

    // ---- prologue: (pairs 0, 1, 2 of the weights left above) the whole first chunk's halo, part A of the second chunk in flight
    {
        // both parts of the first chunk are loaded together (a second register set that only lives here): one memory latency, not two
        HaloRegs hr2;
        prefetch_hv(hr, 0, 0);
        prefetch_hv(hr2, 0, 1);
        stage_hv(hr, Hsb, 0, 0, 4);
        stage_hv(hr2, Hsb, 1, 0, 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prefetch_hv(hr, 1, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) load_A(i, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) load_B(j, 0);

    // ---- nine stages per two chunks
    auto stage = [&](auto stag, const int g) {
        constexpr int S = decltype(stag)::value;
        constexpr int SLOT = S & 3, NSLOT_ = (S + 1) & 3;           // this pair's weight slot / the next pair's (9 stages: see the group loop)
        const int slot = (g * 9 + S) & 3, nslot = (slot + 1) & 3;
        (void)SLOT; (void)NSLOT_;
        // weights three pairs ahead, into the slot of the pair that ran in the previous stage
        constexpr int CV = (S == 0) ? 0 : (S == 2) ? 1 : (S == 5) ? 2 : (S == 7) ? 3 : -1;      // conversion step of the group
        // In a conversion stage the LDS-DMA is issued BEHIND the conversion: the compiler cannot count vmcnt past an LDS-DMA and
        // drains it to 0 in front of the first use of the part's registers -- with the DMA at the top of the stage that is one full
        // L2 -> LDS latency exposed in four of nine stages (what-if timing, r06_experiments.md section 6: 54 of 245 us)
#ifndef PKW_NODMA
        if (CV < 0) dma_pair((g * 9 + S + 3) * 2, (slot + 3) & 3);
#endif
        constexpr int hcur = 2 * S, hnxt = 2 * S + 2;      // first group tap of this pair (the tail of its B fragments is read here) / of the next
        load_B(2, hcur);
        load_B(3, hcur);
        load_A(3, slot);
        __builtin_amdgcn_sched_barrier(0);
        const int c0 = 2 * g;
        unsigned char* const Hconv = Hsb + (CV < 2 ? PK_HBYTES : 0);      // parts of chunk c0 + 1 go to halo buffer 1, of chunk c0 + 2 to buffer 0
        // Tile order: the pair's B fragments 2, 3 (and A fragment 3) were requested at the top of this stage -- their registers were
        // busy to the end of the previous one -- so the stage opens on columns 0, 1 of row blocks 0 and 1 (12 MFMAs) before it
        // touches them.  A fragment i is re-read for the next pair as soon as row block i is issued, B fragments 0, 1 inside the
        // last row block; halo conversion (two pixels per region) and the next part's loads sit between.
        mma(0, 0); mma(0, 1); mma(1, 0); mma(1, 1);
#ifndef PKW_NOCONV
        if (CV >= 0) stage_hv(hr, Hconv, CV & 1, 0, 2);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 2); mma(0, 3);
        __builtin_amdgcn_sched_barrier(0);
        load_A(0, nslot);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 2); mma(1, 3);
#ifndef PKW_NOCONV
        if (CV >= 0) stage_hv(hr, Hconv, CV & 1, 2, 4);
#endif
#ifndef PKW_NODMA
        if (CV >= 0) { __builtin_amdgcn_sched_barrier(0); dma_pair((g * 9 + S + 3) * 2, (slot + 3) & 3); }
#endif
        __builtin_amdgcn_sched_barrier(0);
        load_A(1, nslot);
        __builtin_amdgcn_sched_barrier(0);
        mma(2, 0); mma(2, 1); mma(2, 2); mma(2, 3);
        // then the registers take the next part: B of the same chunk, or A of the chunk after
#ifndef PKW_NOLOAD
        if (CV >= 0) prefetch_hv(hr, c0 + 1 + (CV + 1) / 2, (CV + 1) & 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        load_A(2, nslot);
        __builtin_amdgcn_sched_barrier(0);
        mma(3, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_B(0, hnxt);
        __builtin_amdgcn_sched_barrier(0);
        mma(3, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_B(1, hnxt);
        __builtin_amdgcn_sched_barrier(0);
        mma(3, 2); mma(3, 3);
        stage_end(CV >= 0 ? 2 + NLD : 2);
    };
    for (int g = 0; g < (nch >> 1); ++g) {
        stage(std::integral_constant<int, 0>{}, g); stage(std::integral_constant<int, 1>{}, g); stage(std::integral_constant<int, 2>{}, g);
        stage(std::integral_constant<int, 3>{}, g); stage(std::integral_constant<int, 4>{}, g); stage(std::integral_constant<int, 5>{}, g);
        stage(std::integral_constant<int, 6>{}, g); stage(std::integral_constant<int, 7>{}, g); stage(std::integral_constant<int, 8>{}, g);
    }
    // the dead loads / LDS-DMAs past the end have landed before the staging image of the write-out takes the buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    conv_lowp_epilogue_staged<2, 4, 2, 2>(a, [&](const int h, float* S) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    S[(wm * 32 + ii * 16 + 4 * kq + r) * PK_NT + wn * 64 + j * 16 + r16] = acc[2 * h + ii][j][r];
    }, smem_b, co0, oy0, ox0, 32, tile_id, b, 0);
}

template <int PR, int MODE>
void launch_pair_b(const ConvArgs& a, hipStream_t st) {
    static_assert(PR == PR_BF16X3, "the tap-pair kernel exists for the split-bf16 arithmetic");
    auto kern = &conv_pair_bf16x3<MODE>;
    static DeviceOnce once;
    if (first_on_device(once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    dim3 grid(((a.Hout * a.Wout) / PK_NT) * (a.Cout / PK_MT) * a.B);
    hipLaunchKernelGGL(kern, grid, dim3(PK_NTHR), PK_LDS, st, a);
}

}  // namespace loco
