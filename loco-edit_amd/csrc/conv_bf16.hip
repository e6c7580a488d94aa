// Split-bf16 ("bf16x3") convolution: tile heuristics and dispatch.  The kernel template lives in
// conv_bf16_kernel.h and is instantiated per (TAPS, MODE) in conv_bf16_inst_*.hip.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace loco {

constexpr int BKC = 16;
int conv_pick_tile(int Cout, int HW);   // conv.hip
template <int PR, int TAPS, int MODE> void launch_tile_b(const ConvArgs& a, hipStream_t st);   // conv_bf16_inst_*.hip
enum : int { PR_BF16X3 = 0, PR_F16 = 1 };

int g_bf16_tile_override = -1;   // debug / tuning: force a tile variant for the big-image case

// B = samples of the launch (NOT multiplied by a split-K factor: the choice must not depend on the factor it determines)
int conv_bf16_pick_tile(int Cout, int HW, int B) {
    static int b1 = -1;
    if (b1 < 0) { const char* e = getenv("LOCO_B1_TILE"); b1 = e ? atoi(e) : 1024; }
    // single-sample passes (the B = 1 inversion / to-t chains) at 64x64 and below: 64 x 64 tiles fill the chip with a quarter
    // of the split-K of the 128 x 256 tile (LOCO_B1_TILE: largest H*W it applies to, default 32 x 32: 4.29 vs 4.51 ms per evaluation; 0 = off)
    static int b1_maxb = -1;      // LOCO_B1_TILE_MAXB: largest batch the rule applies to (probe groups of a two-stream pass are 2 - 3 samples)
    if (b1_maxb < 0) { const char* e = getenv("LOCO_B1_TILE_MAXB"); b1_maxb = e ? atoi(e) : 1; }
    if (b1 > 0 && B <= b1_maxb && HW <= b1 && HW >= 64 && (Cout % 64) == 0) return 3;
    int t = conv_pick_tile(Cout, HW);
    if (t == 0 && HW >= 256) {
        if (g_bf16_tile_override >= 0) return g_bf16_tile_override;
        return 5;      // 128 x 256 tile, 8 waves: fastest measured variant down to 160 workgroups
    }
    return t;
}

// split-K factor for the split-bf16 kernels: splitting costs a partial round trip + a reduce launch, so only
// split when the un-split grid would leave more than half of the CUs idle
static int conv_1x1_tile0_maxhw();
int conv_bf16_pick_nsplit(int Cin, int Cout, int Hout, int Wout, int B, int chip_share, int taps, int lanes) {
    const int HW = Hout * Wout;
    int t = conv_bf16_pick_tile(Cout, HW, B);
    if (t == 4) t = 5;
    static int taps_aware = -1, use_ceil = -1, minch = -1;      // A/B switches of the round-6 rule below
    if (taps_aware < 0) { const char* e = getenv("LOCO_SPLITK_TAPS"); taps_aware = e ? atoi(e) : 1; }
    if (use_ceil < 0) { const char* e = getenv("LOCO_SPLITK_CEIL"); use_ceil = e ? atoi(e) : 0; }
    if (minch < 0) { const char* e = getenv("LOCO_SPLITK_MINCH"); minch = e ? atoi(e) : 4; if (minch < 1) minch = 1; }
    // (the 1x1 operators' 128 x 128 tile: bf16_tile_of -- twice the workgroups of the 128 x 256 tile this count assumed until round 6)
    if (taps_aware && taps == 1 && t == 5 && B >= 2 && HW <= conv_1x1_tile0_maxhw()) t = 0;
    static const int MTs[7] = {128, 128, 32, 64, 128, 128, 128}, NTs[7] = {128, 64, 128, 64, 256, 256, 128};
    long blocks = (long)(HW / NTs[t]) * ((Cout + MTs[t] - 1) / MTs[t]) * B;
    int nchunks = (Cin + BKC - 1) / BKC;
    // Workgroups a launch should reach: one per CU, or one per CU of the launch's SHARE of the chip -- two guidance branches
    // enqueue their passes side by side on two streams (tloco.BranchStreams -> loco_set_chip_share(ctx, 2)): each launch then
    // has about half of the chip, and half the splits mean half the partial traffic and reduce work (config 5: 315 -> 296 ms
    // per solve).  LOCO_SPLITK_TARGET overrides (A/B switch).
    static int target_env = -1;
    if (target_env < 0) { const char* e = getenv("LOCO_SPLITK_TARGET"); target_env = e ? atoi(e) : 0; }
    // (lanes == 2: the launch belongs to one of the two probe groups a pass runs side by side on two streams (run_lanes): the other
    //  group's launches want CUs too, and a factor that fills all 256 serialises the two -- 240 measured best, 251.2 - 251.6 against
    //  253.5 - 254.0 ms per step at 256 and 251.5 - 252.5 at 224; on ONE stream 256 is: 259.2 - 259.8 against 260.4 - 261.2 at 224)
    int target = target_env > 0 ? target_env : (chip_share > 1 ? 256 / chip_share : (lanes == 2 ? 240 : 256));
    if (target < 32) target = 32;
    // (aiming the two-per-CU 1x1 tile at two workgroups per CU measured +0.5 %: more partial traffic than overlap)
    if (blocks >= target / 2 || nchunks < 8) return 1;
    // Round 6: the LARGEST factor whose workgroups still fit the target in one round (floor).  The rounded-up quotient put 20 tiles x
    // 13 splits = 260 workgroups on 256 CUs (1024 -> 512 @16^2, 5 probes): four of them ran as a second round of a launch whose
    // workgroups hold one CU each, and 86 - 127 tiles got 3 splits = two rounds of thirds instead of one round of halves.
    int want = use_ceil ? (int)((target + blocks - 1) / blocks) : (int)(target / blocks);
    int maxs = nchunks / minch;
    if (want > maxs) want = maxs;
    if (want > 32) want = 32;
    return want < 1 ? 1 : want;
}

// A/B switch (default off): the big-image 3x3 convs on 128 x 128 tiles of four waves in the compact LDS layout (STG 3 of
// conv_bf16_kernel.h, 81 920 B), two workgroups per CU, instead of one 128 x 256 workgroup of eight waves
static int conv_two_per_cu() {
#ifdef LOCO_DIAG
    static int v = -1;
    if (v < 0) { const char* e = getenv("LOCO_CONV_2WG"); v = e ? (atoi(e) != 0) : 0; }
    return v;
#else
    return 0;      // the compact two-workgroups-per-CU tile is compiled into the diagnostics build only (`make diag`)
#endif
}
// 1x1 operators up to this many pixels run the 128 x 128 tile of four waves (two workgroups per CU: one's write-out under the
// other's K loop) instead of 128 x 256: r05, 5 probes: 256 -> 256 @128^2 74.7 -> 63.5 us, 512 -> 512 @64^2 73.2 -> 64.7, 128 -> 128 @256^2
// 87.2 -> 81.5; the wide maps at 256^2 are equal or slower (128 -> 256: 159.9 vs 160.6, 256 -> 128: 143.9 vs 150.3).  LOCO_1X1_TILE0_MAXHW
static int conv_1x1_tile0_maxhw() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LOCO_1X1_TILE0_MAXHW"); v = e ? atoi(e) : 65536; }      // headline step 290.8 -> 289.2 ms (-0.55 %) up to 128^2; with the stage loop freed of its address arithmetic also at 256^2 (-0.3 %); 0 = off
    return v;
}
int bf16_tile_of(const ConvArgs& a) {
    int tile = conv_bf16_pick_tile(a.Cout, a.Hout * a.Wout, a.B);
    if (tile == 4) tile = 5;
    if (tile == 5 && a.taps == 1 && !a.gemm && a.B >= 2 && a.Hout * a.Wout <= conv_1x1_tile0_maxhw()) return 0;
    if (tile == 5 && conv_two_per_cu() && a.taps == 9 && a.stride == 1 && !a.upsample && !a.zins && (a.Cin % BKC) == 0 &&
        a.in_padded && a.pad == 1 && a.Wout >= 32) return 6;
    if (a.stride == 2 && tile == 5) tile = 0;   // the double-buffered stride-2 halo of a 256-pixel tile exceeds LDS
    return tile;
}
int conv_bf16_tile_pixels(const ConvArgs& a) {
    if (a.gemm) return 256;
    static const int NTs[7] = {128, 64, 128, 64, 256, 256, 128};
    return NTs[bf16_tile_of(a)];
}

int conv_bf16_tile_couts(const ConvArgs& a) {
    if (a.gemm) return 64 * a.gemm_tm;
    static const int MTs[7] = {128, 128, 32, 64, 128, 128, 128};
    return MTs[bf16_tile_of(a)];
}
bool conv_lowp_can_kcat(const ConvArgs& a) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("LOCO_KCAT"); on = e ? (atoi(e) != 0) : 1; }
    return on && a.Cin2 > 0 && a.nsplit == 1 && bf16_tile_of(a) == 5 && a.stride == 1 && !a.upsample && !a.zins && a.pad == 1 &&
           (a.Cin % BKC) == 0 && (a.Cin2 % BKC) == 0 && a.in_padded &&
           (a.mode == CM_GN_SILU || a.mode == CM_GN_GELU || a.mode == CM_TAN_SILU);
}
// the epilogue statistics exist on the LDS-staged path of whole cout tiles (conv_bf16_kernel.h) and need the finished
// sums, i.e. no split-K
bool conv_lowp_can_fuse_stats(const ConvArgs& a) {
    return a.nsplit <= 1 && !a.gemm && (a.Cout % conv_bf16_tile_couts(a)) == 0;      // (the DMA-fed GEMM's epilogue takes none)
}

template <int PR, int MODE> void launch_kcat_b(const ConvArgs& a, hipStream_t st);   // conv_bf16_inst_*.hip
template <int PR, int MODE> void launch_pair_b(const ConvArgs& a, hipStream_t st);   // conv_bf16_inst_k.hip (conv_pair_kernel.h)

// The 16x16x32 tap-pair kernel (conv_pair_kernel.h, round 6) takes the launches the 128 x 256 tile of the lock-step kernel would
// take when the shape allows: 3x3, stride 1, padded arena input, whole 16-channel chunks in an even number, whole 128-cout tiles,
// image rows a multiple of the 32-pixel tile width, no split-K, no K-concatenated shortcut.  LOCO_CONV_PAIR=0: the 32x32x16 kernel.
bool conv_pair_ok(const ConvArgs& a) {
    static int on = -1, minhw = -1;
    if (on < 0) { const char* e = getenv("LOCO_CONV_PAIR"); on = e ? (atoi(e) != 0) : 1; }
    // r06 (tests/diag/pair_check.py, 5 probes, 300 launches per number): -5.7 ... -6.2 % raw, -1.3 ... -5.6 % forward, -1.4 ... -4.0 %
    // tangent / cotangent at 256^2 and 128^2; +-1 % at 64^2 (fewer, shorter tiles: its two-latency prologue shows) -> from 128^2 up
    if (minhw < 0) { const char* e = getenv("LOCO_PAIR_MINHW"); minhw = e ? atoi(e) : 16384; }
    if (!on || a.Hout * a.Wout < minhw || a.taps != 9 || a.Cin2 > 0 || a.nsplit != 1 || a.stride != 1 || a.upsample || a.zins || a.pad != 1 || !a.in_padded ||
        (a.Cin % (2 * BKC)) != 0 || (a.Cout % 128) != 0 || (a.Wout % 32) != 0 || (a.Hout % 8) != 0 || a.pers_groups || a.dual)
        return false;
    if (!(a.mode == CM_NONE || a.mode == CM_GN_SILU || a.mode == CM_TAN_SILU || a.mode == CM_COT_SILU)) return false;
    return bf16_tile_of(a) == 5;
}
template <int PR, int MODE> void launch_dual_b(const ConvArgs& a, hipStream_t st);   // conv_bf16_inst_h / _i.hip (conv_dual_kernel.h)

// Dual-probe tile policy.  Opt-in (LOCO_CONV_DUAL=1): measured neutral on the headline (+1.2 %) and on config 3 (-0.2 %) in round 5
// (profiles/r05_experiments.md: the tile halves the weight bytes per FLOP, but weight traffic is not what a launch waits for).
// LOCO_DUAL_MIN_UNITS: fewest dual units (pixel tiles x cout tiles x probe pairs) a launch must have -- below ~3/4 of the CUs
// the wider tile leaves too much of the chip idle (the 128 x 128 level of the headline: 64 tiles x 2 pairs).
static int conv_dual_min_units() {
#ifndef LOCO_DIAG
    return -1;      // the dual-probe tile is compiled into the diagnostics build only (`make diag`)
#endif
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("LOCO_CONV_DUAL");
        if (!e || atoi(e) == 0) v = -1;
        else { const char* m = getenv("LOCO_DUAL_MIN_UNITS"); v = m ? atoi(m) : 192; }
    }
    return v;
}
bool conv_dual_ok(const ConvArgs& a) {
    const int minu = conv_dual_min_units();
    if (minu < 0) return false;
    if (a.taps != 9 || a.Cin2 > 0 || a.nsplit != 1 || a.stride != 1 || a.upsample || a.zins || a.pad != 1 || (a.Cin % BKC) != 0 ||
        !a.in_padded || a.Wout < 32 || (a.Wout % 32) != 0 || (a.Hout % 8) != 0 || (a.Cout % 128) != 0 || a.B < 2 || a.cot_d)
        return false;
    if (!(a.mode == CM_NONE || a.mode == CM_GN_SILU || a.mode == CM_TAN_SILU || a.mode == CM_COT_SILU)) return false;
    if (bf16_tile_of(a) != 5) return false;
    const long units = (long)((a.Hout * a.Wout) / 256) * (a.Cout / 128) * (a.B / 2);
    return units >= minu;
}
// `a` restricted to its samples nb .. B - 1
static ConvArgs conv_shift_batch(const ConvArgs& a, int nb) {
    ConvArgs t = a;
    t.B = a.B - nb;
    t.in += (long)nb * a.in_bs; t.out += (long)nb * a.out_bs;
    if (t.prim) t.prim += (long)nb * a.prim_bs;
    if (t.bias2) t.bias2 += (long)nb * a.bias2_bs;
    if (t.res) t.res += (long)nb * a.res_bs;
    if (t.sc) { t.sc += (long)nb * a.scsh_bs; t.sh += (long)nb * a.scsh_bs; }
    if (t.mr) t.mr += (long)nb * a.mr_bs;
    if (t.tst) t.tst += (long)nb * a.tst_bs;
    if (t.tc) t.tc += (long)nb * a.tc_bs;
    if (t.in2) t.in2 += (long)nb * a.in2_bs;
    if (t.cot_d) { t.cot_d += (long)nb * a.cot_d_bs; t.cot_tc += (long)nb * a.cot_tc_bs; }
    if (t.st_part) t.st_part += (long)nb * a.Cout * ((a.Hout * a.Wout) / conv_bf16_tile_pixels(a)) * 2;
    return t;
}
int conv_lowp_plan(const ConvArgs& a, int taps, int prec, ConvArgs parts[2]) {
    parts[0] = a;
    parts[0].dual = 0;
    ConvArgs q = a; q.taps = taps;
    if (prec != 1 || taps != 9 || !conv_dual_ok(q)) return 1;
    parts[0].B = a.B & ~1;
    parts[0].dual = 1;
    if (!(a.B & 1)) return 1;
    parts[1] = conv_shift_batch(a, a.B - 1);
    parts[1].dual = 0;
    parts[1].pers_groups = 0;
    return 2;
}

void launch_conv_gemm(const ConvArgs& a, hipStream_t st);      // conv_bf16_inst_j.hip (conv_gemm_kernel.h)

bool conv_pers_plan(ConvArgs& a) {
    // Opt-in (LOCO_CONV_PERS=1: raw / forward forms, 2: every form): measured neutral to slightly negative in the flow (r05:
    // headline 299.4 vs 300.2 ms, 25-frame decode step 44.1 -> 45.0 ms) although the isolated raw / forward launches gain 5 %
    a.pers_groups = 0;
#ifndef LOCO_DIAG
    return false;      // the persistent form is compiled into the diagnostics build only (`make diag`)
#endif
    static int on = -1;
    if (on < 0) { const char* e = getenv("LOCO_CONV_PERS"); on = e ? (atoi(e) != 0) : 0; }
    if (!on || a.taps != 9 || a.Cin2 > 0 || a.nsplit != 1 || a.stride != 1 || a.upsample || a.zins || a.pad != 1 || !a.in_padded ||
        (a.Cin % (2 * BKC)) != 0 || (a.Cout % 128) != 0 || a.B < 2 || a.cot_d || bf16_tile_of(a) != 5)
        return false;
    // The raw-input and forward forms only (r05, 128 -> 128 @256^2, 5 probes: 246 -> 232 us raw, 261 -> 248 us GroupNorm + SiLU).  The
    // tangent / cotangent forms LOSE under a walk over probes (274 -> 276, 128 -> 256: 523 -> 552 us): with one workgroup per
    // (tile, probe) the five probes of a pixel tile run at the same time on one XCD and share one fetch of the tile's primal
    // {S, xhat} cache (8 of their 12 bytes per element); walked one after the other by one CU, the cache is fetched five times.
    // LOCO_CONV_PERS=2 walks them too (A/B).
    static int all_modes = -1;
    if (all_modes < 0) { const char* e = getenv("LOCO_CONV_PERS"); all_modes = (e && atoi(e) == 2) ? 1 : 0; }
    if (!(a.mode == CM_NONE || a.mode == CM_GN_SILU || a.mode == CM_GN_GELU ||
          (all_modes && (a.mode == CM_TAN_SILU || a.mode == CM_COT_SILU)))) return false;
    // G workgroups share a tile, each walks ceil(B / G) probes: as many groups as keep the grid within one round of the chip
    const long wg0 = (long)((a.Hout * a.Wout) / 256) * (a.Cout / 128);
    int G = wg0 >= 256 ? 1 : (int)(256 / wg0);
    if (G > a.B) G = a.B;
    const int per = (a.B + G - 1) / G;
    // worth it when a workgroup gets at least two probes, the walks are balanced and the grid fills most of the chip
    if (per < 2 || a.B * 10 < G * per * 8 || wg0 * G < 192) return false;
    a.pers_groups = G;
    return true;
}

bool conv_gemm_plan(ConvArgs& a) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("LOCO_CONV_GEMM"); on = e ? (atoi(e) != 0) : 1; }
    a.gemm = 0;
    const long HW = (long)a.Hout * a.Wout;
    if (!on || a.stride != 1 || a.upsample || a.zins || (a.Cin % BKC) != 0 || a.Cin < 320 || (HW % 256) != 0 || (a.Wout % 16) != 0 ||
        a.Cout < 128 || !a.partial || a.Cin2 > 0 || !(a.mode == CM_NONE || a.mode == CM_GN) || a.cot_d)
        return false;      // (cot_d: the norm-cotangent term lives in the per-pixel 1x1 kernels' epilogue; this kernel's takes neither it nor statistics)
    {   // the kernel's 256-pixel tile is TW = min(Wout, 32) columns x 256 / TW rows: both must divide the map (a 48 x 48 map passes
        // the tests above and would be walked out of bounds: ADVICE r05)
        const int TW = a.Wout < 32 ? a.Wout : 32;
        if ((256 % TW) != 0 || (a.Wout % TW) != 0 || (a.Hout % (256 / TW)) != 0) return false;
    }
    // Where it pays (tests/diag/gemm_check.py, 5 probes, us per launch old -> new incl. the split pass): the split pass costs 8 bytes
    // per input element, so the map must have many cout tiles to spread it over (320 -> 2560 @64^2 190 -> 150, 640 -> 5120 @32^2
    // 180 -> 121, 1280 -> 10240 @16^2 164 -> 104) or be one of the K-heavy, pixel-poor maps the per-pixel kernel runs at 64 - 118
    // TFLOP/s behind split-K (2560 -> 640 @32^2 144 -> 125, 5120 -> 1280 @16^2 272 -> 82).  It does not pay for the q/k/v maps
    // (3 C couts: 84 -> 105), the C -> C maps and the wide-input, narrow-output maps at 64^2 (2560 -> 320: 160 -> 218).
    static int all_shapes = -1;      // LOCO_GEMM_ALL=1: every eligible shape (A/B)
    if (all_shapes < 0) { const char* e = getenv("LOCO_GEMM_ALL"); all_shapes = e ? atoi(e) : 0; }
    if (!all_shapes && !(a.Cout >= 640 && (a.Cout >= 4 * a.Cin || a.Cin >= 2560))) return false;
    // cout tile: 256 where the padding to whole tiles costs < 10 %, else 128
    const int pad256 = (a.Cout + 255) / 256 * 256;
    static int force_tm = -1, no_split = -1;      // bring-up knobs: LOCO_GEMM_TM=2|4, LOCO_GEMM_NOSPLIT=1
    if (force_tm < 0) { const char* e = getenv("LOCO_GEMM_TM"); force_tm = e ? atoi(e) : 0; }
    if (no_split < 0) { const char* e = getenv("LOCO_GEMM_NOSPLIT"); no_split = e ? atoi(e) : 0; }
    int tm = force_tm ? force_tm : ((pad256 * 10 <= a.Cout * 11) ? 4 : 2);
    const int mt = 64 * tm;
    const long tiles = (HW / 256) * ((a.Cout + mt - 1) / mt) * a.B;
    const size_t rec_floats = (size_t)a.B * a.Cin * HW;
    int ns = 1;
    if (tiles < 192 && !no_split) {
        const int nchunks = a.Cin / BKC;
        ns = (int)((256 + tiles - 1) / tiles);
        if (ns > 8) ns = 8;
        while (ns > 1 && nchunks / ns < 8) --ns;
        while (ns > 1 && (size_t)ns * a.B * a.Cout * HW + rec_floats > a.partial_floats) --ns;
    }
    if ((ns > 1 ? (size_t)ns * a.B * a.Cout * HW : 0) + rec_floats > a.partial_floats) return false;
    a.gemm = 1; a.gemm_tm = tm; a.nsplit = ns;
    return true;
}

template <int PR>
static void launch_lowp(const ConvArgs& a, int taps, hipStream_t st) {
    // The norm-cotangent term exists in ONE place: the staged epilogue of the per-pixel 1x1 kernels (EPI_COT1), un-split launches of
    // whole cout tiles.  Anything else would drop it silently -- refuse loudly (the engine plans such launches only: run_conv).
    if (a.cot_d && (taps != 1 || a.gemm || a.nsplit > 1 || a.Cin2 > 0 || (a.Cout % conv_bf16_tile_couts(a)) != 0 || a.st_kind == ST_TAN || a.st_kind == ST_COT)) {
        fprintf(stderr, "loco: ConvArgs::cot_d on a launch whose epilogue has no norm-cotangent term (taps %d, gemm %d, nsplit %d, Cout %d, st_kind %d)\n",
                taps, a.gemm, a.nsplit, a.Cout, a.st_kind);
        abort();
    }
    if constexpr (PR == PR_BF16X3) {
        if (taps == 1 && a.gemm) { launch_conv_gemm(a, st); return; }
    }
    if (taps == 9 && a.Cin2 > 0) {       // the caller checked conv_lowp_can_kcat
        if (a.mode == CM_GN_SILU) launch_kcat_b<PR, CM_GN_SILU>(a, st);
        else if (a.mode == CM_GN_GELU) launch_kcat_b<PR, CM_GN_GELU>(a, st);      // forward pass of a GELU network (DeepFloyd IF)
        else launch_kcat_b<PR, CM_TAN_SILU>(a, st);
        return;
    }
#ifdef LOCO_DIAG
    if constexpr (PR == PR_BF16X3) {
        if (taps == 9 && a.dual) {           // conv_lowp_plan checked conv_dual_ok
            switch (a.mode) {
                case CM_NONE: launch_dual_b<PR, CM_NONE>(a, st); break;
                case CM_GN_SILU: launch_dual_b<PR, CM_GN_SILU>(a, st); break;
                case CM_TAN_SILU: launch_dual_b<PR, CM_TAN_SILU>(a, st); break;
                default: launch_dual_b<PR, CM_COT_SILU>(a, st); break;
            }
            return;
        }
    }
#endif
    if constexpr (PR == PR_BF16X3) {
        if (taps == 9 && conv_pair_ok(a)) {
            switch (a.mode) {
                case CM_NONE: launch_pair_b<PR, CM_NONE>(a, st); break;
                case CM_GN_SILU: launch_pair_b<PR, CM_GN_SILU>(a, st); break;
                case CM_TAN_SILU: launch_pair_b<PR, CM_TAN_SILU>(a, st); break;
                default: launch_pair_b<PR, CM_COT_SILU>(a, st); break;
            }
            return;
        }
    }
    if (taps == 9) {
        switch (a.mode) {
            case CM_NONE: launch_tile_b<PR, 9, CM_NONE>(a, st); break;
            case CM_GN_SILU: launch_tile_b<PR, 9, CM_GN_SILU>(a, st); break;
            case CM_GN_GELU: launch_tile_b<PR, 9, CM_GN_GELU>(a, st); break;
            case CM_TAN_SILU: launch_tile_b<PR, 9, CM_TAN_SILU>(a, st); break;
            case CM_COT_SILU: launch_tile_b<PR, 9, CM_COT_SILU>(a, st); break;
            default: launch_tile_b<PR, 9, CM_GN>(a, st); break;
        }
    } else {
        switch (a.mode) {
            case CM_NONE: launch_tile_b<PR, 1, CM_NONE>(a, st); break;
            default: launch_tile_b<PR, 1, CM_GN>(a, st); break;
        }
    }
}
void launch_conv_bf16x3(const ConvArgs& a, int taps, hipStream_t st) { launch_lowp<PR_BF16X3>(a, taps, st); }
void launch_conv_f16(const ConvArgs& a, int taps, hipStream_t st) { launch_lowp<PR_F16>(a, taps, st); }

}  // namespace loco
