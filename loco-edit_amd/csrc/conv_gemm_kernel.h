// Compute-shaped 1x1 operators (the SpatialTransformer's to_q/k/v, to_out, GEGLU 8C / 4C maps, proj_in / proj_out of the latent-
// diffusion denoiser: K = 320 ... 5120 input channels, 140 - 570 FLOP per byte) as a split-bf16 GEMM D[cout][pixel] = W x X whose
// operands BOTH arrive by LDS-DMA (round 5).
//
// Why: the per-pixel 1x1 kernel of conv_bf16_kernel.h converts its 16-channel pixel records in the multiplying workgroup -- for a
// 3x3 conv that conversion is shared by 9 taps, for a 1x1 it is paid per chunk and per cout tile (20 cout tiles of a 320 -> 2560
// map re-convert the same activations 20 times: ~11 vector instructions per MFMA), and a stage is one chunk = 12 MFMAs per wave
// between barriers.  Measured 64 - 214 TFLOP/s on these shapes (r04_experiments.md 18).  Here
//   * the activations are split ONCE per launch into the weights' own record format [Cin/16][pixel][hi k0-7|hi k8-15|lo k0-7|lo
//     k8-15] (act_split_kernel: one pass, 4 B read + 4 B written per element, the GroupNorm affine of the attention qkv maps
//     applied on the way) into the split-K workspace;
//   * the GEMM kernel is DMA + ds_read + MFMA only: a ring of four K-step slots (weights MT x 64 B + pixels 256 x 64 B each),
//     issued three steps ahead, one raw barrier per step, operand fragments of the next step read under this step's MFMAs;
//   * a workgroup owns 64 TM couts x 256 pixels (TM = 4: 256 couts, 24 MFMA groups x 3 per wave and step, 21 B per cycle and CU of
//     operand traffic; TM = 2 where 256-cout tiles would waste more than they save), split-K over workgroups where the tile
//     grid alone leaves the chip idle (K-heavy maps at 16 x 16: 5120 -> 1280).
// Same accumulation order per output element as the per-pixel kernel within one K range (chunk by chunk, lo*hi, hi*lo, hi*hi).
// Reference call sites: the `self.pipe.unet(...)` of edit.py:655-658 (diffusers BasicTransformerBlock linear layers).
#pragma once
#include "conv_bf16_kernel.h"

namespace loco {

constexpr int GM_NSLOT = 4, GM_NT = 256;

// fp32 [C][HW] (batch stride in_bs) -> split records [B][C/16][HW][64 B], 16-byte pieces XOR-swizzled by (pixel >> 2) & 3 like
// the weight records (rec_off): grid (HW / 256, C / 16, B), one thread per pixel, two octets of channels each
template <int MODE>
__global__ __launch_bounds__(256) void act_split_kernel(const float* in, long in_bs, int HW, const float* sc, const float* sh,
                                                        long scsh_bs, unsigned char* rec) {
    const int p = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y, b = blockIdx.z, nchunk = gridDim.y;
    const float* ip = in + (long)b * in_bs + (long)chunk * BKC * HW + p;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = ip[(long)k * HW];
    if constexpr (MODE == CM_GN) {
        const float* scb = sc + (long)b * scsh_bs + chunk * BKC;
        const float* shb = sh + (long)b * scsh_bs + chunk * BKC;
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(scb[k], v[k], shb[k]);
    }
    uint4 h0, l0, h1, l1;
    split8<PR_BF16X3>(v, h0, l0);
    split8<PR_BF16X3>(v + 8, h1, l1);
    unsigned char* rp = rec + (((long)b * nchunk + chunk) * HW) * 64;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(p, 0)) = h0;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(p, 1)) = h1;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(p, 2)) = l0;
    *reinterpret_cast<uint4*>(rp + rec_off<PR_BF16X3>(p, 3)) = l1;
}

template <int TM>
__global__ __launch_bounds__(512, 1) void conv_gemm_bf16x3(ConvArgs a, const unsigned char* brec) {
    constexpr int PR = PR_BF16X3, RB = 64;
    constexpr int WM = 2, WN = 4, TN = 2;
    constexpr int MT = WM * TM * 32;
    constexpr int ABYTES = MT * RB, BBYTES = GM_NT * RB, SLOTB = ABYTES + BBYTES;
    constexpr int NA = ABYTES / (512 * 16), NB = BBYTES / (512 * 16), NDMA = NA + NB;      // 16-byte pieces per thread and step
    static_assert(NA >= 1 && NB == 2 && GM_NSLOT * SLOTB <= 160 * 1024, "ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    const int HW = a.Hout * a.Wout;
    const int TW = a.Wout < 32 ? a.Wout : 32;
    const int tiles_x = a.Wout / TW;
    // block order as conv_lowp_body: probes (and K-splits) of one (pixel tile, cout tile) adjacent, on one XCD
    int tile_id, cot_id, zid;
    {
        const int ntile = HW / GM_NT, ncot = (a.Cout + MT - 1) / MT, Z = a.B * a.nsplit;
        const int L = blockIdx.x;
        if ((ntile & 7) == 0) {
            int q = L >> 3;
            cot_id = q % ncot; q /= ncot;
            zid = q % Z;
            tile_id = (q / Z) * 8 + (L & 7);
        } else {
            zid = L % Z;
            const int T = L / Z;
            tile_id = T % ntile; cot_id = T / ntile;
        }
        tile_id = __builtin_amdgcn_readfirstlane(tile_id);
        cot_id = __builtin_amdgcn_readfirstlane(cot_id);
        zid = __builtin_amdgcn_readfirstlane(zid);
    }
    const int oy0 = (tile_id / tiles_x) * (GM_NT / TW), ox0 = (tile_id % tiles_x) * TW;
    const int co0 = cot_id * MT;
    const int b = zid / a.nsplit, split = zid % a.nsplit;
    const int nchunks = a.Cin / BKC;
    const int cps = (nchunks + a.nsplit - 1) / a.nsplit;
    const int cbeg = split * cps;
    const int cend = cbeg + cps < nchunks ? cbeg + cps : nchunks;
    const int nst = cend - cbeg;                                  // K-steps of this workgroup
    const int wpitch = (a.Cout + 31) & ~31;

    // DMA sources of this thread: weight pieces (records past the padded cout range clamped to the last one: rows >= Cout are
    // never stored) and pixel pieces (the tile's rows are contiguous runs of TW records in the image-ordered record array)
    unsigned aoffg[NA], boffg[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = tid + i * 512;
        int r = co0 + (e >> 2);
        if (r >= wpitch) r = wpitch - 1;
        aoffg[i] = (unsigned)r * 64u + (unsigned)(e & 3) * 16u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = tid + i * 512, p = e >> 2;
        const int gp = (oy0 + p / TW) * a.Wout + ox0 + (p % TW);
        boffg[i] = (unsigned)gp * 64u + (unsigned)(e & 3) * 16u;
    }
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(a.wb);
    const unsigned char* const bbase = brec + (long)b * nchunks * HW * 64;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef const __attribute__((address_space(1))) unsigned char glb_u8;
    auto dma = [&](int st, int slot) {                          // K-step st of this workgroup -> ring slot
        const bool live = st < nst;
        const int chunk = __builtin_amdgcn_readfirstlane(cbeg + (live ? st : nst - 1));
        const unsigned char* wa = wbase + (unsigned)chunk * ((unsigned)wpitch * 64u);
        const unsigned char* ba = bbase + (long)chunk * HW * 64;
        unsigned char* const S = smem_b + slot * SLOTB;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((glb_u8*)(wa + (live ? aoffg[i] : 0u)), (lds_u8*)(S + i * 8192 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((glb_u8*)(ba + (live ? boffg[i] : 0u)), (lds_u8*)(S + ABYTES + i * 8192 + wave * 1024), 16, 0, 0);
    };

    // operand fragment offsets inside a slot (records XOR-swizzled: rec_off)
    const unsigned aoff_hi = (unsigned)rec_off<PR>(wm * (TM * 32) + l31, khalf), aoff_lo = (unsigned)rec_off<PR>(wm * (TM * 32) + l31, 2 + khalf);
    const unsigned boff_hi = (unsigned)ABYTES + (unsigned)rec_off<PR>(wn * 64 + l31, khalf);
    const unsigned boff_lo = (unsigned)ABYTES + (unsigned)rec_off<PR>(wn * 64 + l31, 2 + khalf);
    // Operand fragments: the pixel fragment of a step (both 32-pixel blocks) is used by every cout row block of the step and is
    // double-buffered across steps; the weight fragments are used once per step and row block and travel through a ring of two
    // row blocks, each read from LDS while the previous row block's six MFMA groups run.  Every wait therefore covers reads that
    // were issued a whole row block earlier.
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

    // ---- prologue: steps 0, 1, 2 in flight; steps 0 and 1 landed before the first barrier -----------------------------------
    dma(0, 0); dma(1, 1); dma(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
    FragA fa[2];
    FragB fb[2];
    load_A(fa[0], 0, 0);                                          // the only exposed operand reads of the tile
    load_B(fb[0], 0);
    // ---- K loop: step s multiplies out of slot s % 4; it opens by sending step s + 3 into the slot step s - 1 has left, reads the
    // fragments of step s + 1 (landed and fenced one barrier ago) under its own MFMAs, and closes once step s + 2 has landed
    auto step = [&](auto ktag, const int s) {
        constexpr int K4 = decltype(ktag)::value;                  // s % 4
        dma(s + 3, (K4 + 3) & 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            mma_row(fa[i & 1], fb[K4 & 1], i);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < TM) load_A(fa[(i + 1) & 1], K4, i + 1);
            else load_A(fa[0], (K4 + 1) & 3, 0);                  // (TM is even: row block 0 of the next step sits in fa[0] again)
            if (i == 0) load_B(fb[(K4 + 1) & 1], (K4 + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");      // all but step s + 3's pieces
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int s = 0; s < nst; s += 4) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nst) step(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 < nst) step(std::integral_constant<int, 2>{}, s + 2);
        if (s + 3 < nst) step(std::integral_constant<int, 3>{}, s + 3);
    }
    // the ring's last (dead) DMAs and the fragment reads issued ahead retire before the accumulators take over the LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    conv_lowp_epilogue<WM, WN, TM, TN, 0>(a, acc, smem_b, co0, oy0, ox0, TW, tile_id, b, split);      // (no statistics sink, no norm-cotangent term: conv_gemm_plan)
}

void launch_conv_gemm(const ConvArgs& a, hipStream_t st);      // conv_bf16_inst_j.hip

}  // namespace loco
