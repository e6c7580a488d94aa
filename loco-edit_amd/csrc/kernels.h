// Internal launch interface between the engine (engine.hip) and the gfx950
// kernels.  All tensors are fp32 NCHW with an explicit batch stride so that a
// tensor may live inside a wider (concatenated) buffer and a primal (B=1)
// tensor can be broadcast over a probe batch with stride 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace loco {

// One-time per-DEVICE setup guard for launchers (kernel attributes such as the dynamic-LDS limit belong to the device's
// copy of the code object): true the first time it is called with `flags` on the current device.
struct DeviceOnce { bool seen[64] = {}; };
inline bool first_on_device(DeviceOnce& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    if (f.seen[d]) return false;
    f.seen[d] = true;
    return true;
}


enum ConvMode : int {
    CM_NONE = 0,      // raw input
    CM_GN_SILU = 1,   // a = silu(sc*x + sh)                         (forward, ResnetBlock norm->swish->conv)
    CM_GN = 2,        // a = sc*x + sh                               (forward, AttnBlock norm->q/k/v)
    CM_TAN_SILU = 3,  // a = silu'(y) * sc*(d - m1 - xh*m2)          (tangent of norm->swish)
    CM_COT_SILU = 4,  // a = rstd*(gamma*silu'(y)*d - m1 - xh*m2)    (cotangent of norm->swish, fused into next dgrad)
    CM_GN_GELU = 5,   // a = gelu(sc*x + sh)                         (forward of the DeepFloyd-IF blocks: exact erf GELU)
};
// Activation of the norm -> activation -> conv chains of one network (ConvArgs::act and the `act` argument of the
// elementwise launchers): every "silu" in the mode / kind descriptions below reads as this function.  The low-precision
// conv kernels take it as a MODE (CM_GN_SILU / CM_GN_GELU forward; their tangent / cotangent modes read act'(y) from the
// primal cache of launch_gn_cache), everything else as a wave-uniform runtime switch.
enum ActKind : int { ACT_SILU = 0, ACT_GELU = 1 };
__device__ __forceinline__ float act_fwd(float y, int act) {
    if (act == ACT_GELU) return 0.5f * y * (1.0f + erff(y * 0.70710678118654752f));
    return y * (1.0f / (1.0f + __expf(-y)));
}
__device__ __forceinline__ float act_der(float y, int act) {
    if (act == ACT_GELU) return 0.5f * (1.0f + erff(y * 0.70710678118654752f)) + y * 0.39894228040143268f * __expf(-0.5f * y * y);
    const float sg = 1.0f / (1.0f + __expf(-y));
    return sg * (1.0f + y * (1.0f - sg));
}

// GroupNorm statistics of a conv's OUTPUT tensor for the norm that consumes it (engine.hip StatReq):
//   ST_FWD: mean / rstd -> mr, sc, sh           ST_TAN: m1 = mean(d), m2 = mean(xhat d) -> tst, tc
//   ST_COT: the same means of z = gamma silu'(y) d (cotangent through norm -> SiLU)
// Split-K convs take them in their split-K epilogue (launch_conv_splitk_reduce_stats: one kernel instead of reduce +
// statistics); un-split low-precision convs with whole cout tiles take the FORWARD ones in the conv epilogue as {mean, M2}
// per (cout row, pixel tile) in ConvArgs::st_part, merged by launch_gn_fused_finalize.
enum StatKind : int { ST_NONE = 0, ST_FWD = 1, ST_TAN = 2, ST_COT = 3 };

struct ConvArgs {
    // input activation (or tangent / cotangent) tensor
    const float* in; long in_bs; int Cin, Hin, Win;
    // primal tensor matching `in` (modes TAN/COT), broadcast with prim_bs = 0
    const float* prim; long prim_bs;
    // weights, layout [Cin][TAPS][Cout]
    const float* w;
    // low-precision weights: bf16x3 [Cin/16][TAPS][Cout^32][hi0-7|hi8-15|lo0-7|lo8-15] (swizzled 64-byte records),
    // f16 [Cin/16][TAPS][Cout^32][k0-7|k8-15] (swizzled 32-byte records)
    const void* wb;
    const void* wh;     // f16 records of the same operator (the engine points wb at them in f16 mode)
    // primal cache {S = sc*silu'(y), xhat} per element of `in`'s primal (bf16x3 path, modes TAN/COT)
    const float2* sx;
    float* out; long out_bs; int Cout, Hout, Wout;
    const float* bias;                   // [Cout] or nullptr
    const float* bias2; long bias2_bs;   // [B][Cout] or nullptr (temb projection)
    const float* res; long res_bs;       // residual, same geometry as out, or nullptr (may alias out)
    // GroupNorm prologue data
    const float* sc; const float* sh; long scsh_bs;   // per (b_prim, channel)
    const float* mr; long mr_bs;                      // per (b_prim, group): {mean, rstd}
    const float* gamma_;                              // per channel GroupNorm gain (mode COT)
    const float* tst; long tst_bs;                    // per (b, group): {m1, m2}
    const float* tc; long tc_bs;                      // per (b, channel): {m1, m2} or {rstd*m1, rstd*m2} (bf16x3 path)
    int cpg;                                          // channels per group
    int mode;
    int stride;      // 1 or 2
    int pad;         // top/left zero padding in input coords (1: 3x3 s1, 0: 3x3 s2 / 1x1, 2: zero-insert dgrad)
    int upsample;    // input is read through a nearest x2 upsample
    int zins;        // input is read through stride-2 zero insertion (dgrad of the stride-2 conv)
    int in_padded;   // `in` (and `sx`) live in a padded engine arena: 16-byte loads may start 1 float before / end 3 after a plane
    int accumulate;  // out += result
    int nsplit;      // split-K factor (>1: raw partials go to `partial`, epilogue by conv_splitk_reduce)
    int taps;        // 9 or 1 (0 = not stated)
    int no_deep;     // 1: the 1x1 per-pixel variants keep the two-buffer stage loop (A/B switch of the register ring)
    float* partial;
    int B;
    // statistics sink of the conv epilogue, [B][Cout][pixel tiles][2] per (cout row, pixel tile) of the FINISHED tensor:
    //   st_kind == ST_FWD: {mean, M2};
    //   ST_TAN (round 6): the RAW sums {sum d, sum x d} with x = st_x, the primal of the output tensor (B = 1) -- independent of
    //       the norm, so one set of partials serves the next block's norm AND the up-path norm over a concatenation;
    //   ST_COT (round 6): {sum S d, sum xhat S d} with {S = sc act'(sc x + sh), xhat} = st_sx, the consuming norm's primal cache
    //       records over the output tensor (B = 1); z = (sc / rstd) act'(.) d = S d / rstd: the merge divides by the group's rstd.
    // Merged per (sample, group) by launch_gn_fused_finalize (FWD) / launch_gn_lin_fused_finalize (TAN, COT).
    float* st_part; int st_kind;
    const float* st_x; const float2* st_sx;
    // K-concatenated second operator (conv_bf16_kernel.h conv_lowp_kcat): a 1x1 conv of the RAW tensor in2 [B][Cin2][Hout][Wout]
    // accumulated into the same output tile; its bias travels in `bias2` (bias2_bs = 0).  Cin2 == 0: none.
    const float* in2; long in2_bs; int Cin2; int in2_padded;
    const void* wb2;
    // Norm-cotangent term added in the epilogue of a 1x1 operator (conv_lowp_epilogue, whole cout tiles, no split-K): the cotangent
    // of a GroupNorm + activation whose INPUT is this conv's output tensor,
    //     out += S d - (rstd m1 + xhat rstd m2)
    // with d = cot_d (cotangent behind the norm, [B][Cout][H][W]), {S = sc act'(sc x + sh), xhat} = cot_sx (that norm's primal cache
    // records over the output tensor, B = 1: no transcendental per element here, and the probes of a tile share the records through
    // L2) and {rstd m1, rstd m2} = cot_tc per (sample, channel) (what the cotangent statistics leave for the conv staging) -- what
    // gn_apply_kernel<2> computes as its own pass over the tensor (ResBlock cotangent: g_in = nin^T g_out + norm1^T g_a1 in one
    // write-out).  nullptr: none.
    const float* cot_d; long cot_d_bs; const float2* cot_sx; const float* cot_tc; long cot_tc_bs;
    int act;           // ActKind of the prologue modes (exact-fp32 kernel, split-K statistics epilogue)
    float res_scale;   // out = conv + bias + res_scale * res  (DeepFloyd-IF: (x + h) / sqrt 2 with the conv's weights pre-scaled); conv_defaults: 1
    int dual;          // 1: run on the dual-probe tile of conv_dual_kernel.h (B even; set by conv_lowp_plan, never by the engine)
    // 1x1 operator as a DMA-fed GEMM (conv_gemm_kernel.h; set by conv_gemm_plan): gemm_tm = 2 / 4 (128 / 256 couts per
    // workgroup); the activations' split records are written to the END of the workspace `partial` (partial_floats floats)
    int gemm, gemm_tm;
    size_t partial_floats;
    // 3x3 kernel persistent over the probes of a tile (conv_bf16_kernel.h PHASE 3; set by conv_pers_plan): number of workgroups
    // that share one (pixel tile, cout tile), each walking the probes b, b + pers_groups, ...; 0: one workgroup per (tile, probe)
    int pers_groups;
};

// the conv kernel only; when a.nsplit > 1 the caller follows with launch_conv_splitk_reduce (run_conv does)
void launch_conv(const ConvArgs& a, int taps, hipStream_t st);
void launch_conv_bf16x3(const ConvArgs& a, int taps, hipStream_t st);
void launch_conv_f16(const ConvArgs& a, int taps, hipStream_t st);     // a.wb = the f16 weight records
void launch_conv_splitk_reduce(const ConvArgs& a, hipStream_t st);
// Dual-probe tile (conv_dual_kernel.h): a low-precision launch is split into its even part on the dual tile (parts[0], `dual`
// set) and -- for an odd batch -- the last probe on the 128 x 256 tile (parts[1]); returns the number of parts (1: `a` itself).
bool conv_dual_ok(const ConvArgs& a);
int conv_lowp_plan(const ConvArgs& a, int taps, int prec, ConvArgs parts[2]);
// Compute-shaped 1x1 operators on the DMA-fed GEMM kernel: decides eligibility, the cout tile and the split-K factor (overwrites
// a.nsplit) and sets a.gemm / a.gemm_tm; false: the launch stays on the per-pixel kernel with the split-K factor it had.
// LOCO_CONV_GEMM=0 switches it off (A/B).
bool conv_gemm_plan(ConvArgs& a);
// 3x3 launches whose (pixel tile, cout tile) grid is at least as wide as the chip, or divides it: persistent over the probes
// (sets a.pers_groups; opt-in: LOCO_CONV_PERS=1 | 2, see conv_bf16.hip)
bool conv_pers_plan(ConvArgs& a);
// does the low-precision 3x3 launch `a` (taps, split-K, pers_groups, dual already decided) run on the 16x16x32 tap-pair kernel?
bool conv_pair_ok(const ConvArgs& a);
// can a launch with these arguments feed a.st_part from its epilogue?  (whole cout tiles of the chosen variant, no split-K)
bool conv_lowp_can_fuse_stats(const ConvArgs& a);
int conv_bf16_tile_couts(const ConvArgs& a);
// can the low-precision launch `a` (3x3, its split-K factor already chosen) carry the 1x1 operator a.in2 / a.Cin2 / a.wb2 in the
// same kernel?  (128 x 256 tiles, stride 1, whole 16-channel chunks of both inputs, no split-K)
bool conv_lowp_can_kcat(const ConvArgs& a);
int conv_pick_tile(int Cout, int HW);
extern int g_bf16_tile_override;
void launch_fill_random(float* p, long count, unsigned seed, float scale, hipStream_t st);
// sx[c][hw] = { sc_c * silu'(sc_c*x + sh_c), (x - mean_g) * rstd_g }   (B = 1 primal cache)
void launch_gn_cache(const float* x, int C, int HW, int cpg, const float* sc, const float* sh, const float* mr,
                     float2* sx, hipStream_t st, int act = 0);
// name of the kernel variant launch_conv would pick (for the per-kernel profile)
const char* conv_variant_name(const ConvArgs& a, int taps, int prec);
// workspace (floats) a conv launch with these args needs for split-K partials
size_t conv_partial_floats(const ConvArgs& a);
int conv_pick_nsplit(int Cin, int Cout, int Hout, int Wout, int B, int taps);
int conv_bf16_pick_nsplit(int Cin, int Cout, int Hout, int Wout, int B, int chip_share = 1, int taps = 9, int lanes = 1);   // chip_share: launches running side by side (loco_set_chip_share)
int conv_bf16_pick_tile(int Cout, int HW, int Bsplit);
int conv_bf16_tile_pixels(const ConvArgs& a);

// Generic strided batched GEMM  C[b](m,n) = alpha * sum_k A[b](m,k) B[b](k,n) + beta*C + bias[m] + R[b](m,n)
struct GemmArgs {
    const float* A; long sam, sak, sab;
    const float* Bm; long sbk, sbn, sbb;
    float* C; long scm, scn, scb;
    const float* bias;           // per m or nullptr
    const float* colbias;        // per n or nullptr (mask of padded context columns: 0 / -1e30)
    const float* R; long srb;    // residual with C's (m,n) strides, batch stride srb; or nullptr
    int M, N, K, batch;
    float alpha, beta;
    // second-level batch (attention heads): blockIdx.z = b * batch2 + h, strides added per h
    int batch2; long sah, sbh, sch;
    // optional second contraction segment (K-concatenation): C = alpha * (A B + A2 B2) + ..., same shapes and
    // row / column / k / head strides, own batch strides (the tangent forms dq^T k + q^T dk and dv P^T + v dP^T pair a
    // per-probe operand with a broadcast primal one)
    const float* A2; long sab2; const float* Bm2; long sbb2;
};
void launch_gemm(const GemmArgs& g, hipStream_t st);              // exact fp32 (f32-input MFMA)
void launch_gemm_bf16x3(const GemmArgs& g, hipStream_t st);       // split-bf16 operands on the bf16 matrix pipe
bool gemm_prefers_bf16x3(const GemmArgs& g);                      // long contraction, matrix-rate bound on the f32 MFMA
// round 5 (gemm_rec.hip): the same products with both operands pre-split into records and streamed by LDS-DMA; `ws` must hold
// gemm_rec_ws_bytes(g) bytes (records of the operands + the partial tiles of a K split)
bool gemm_rec_eligible(const GemmArgs& g);
size_t gemm_rec_ws_bytes(const GemmArgs& g);
void launch_gemm_rec(const GemmArgs& g, unsigned char* ws, hipStream_t st);

// ---- tangent / cotangent of multi-head self-attention without per-probe [T x T] matrices (attn_flash.hip) ----
// All tensors [channel][token] with the engine's strides; q / k / v (and their tangents / cotangents) of head h start
// hs floats apart, o (and do / g_o) CH * T floats apart.  Primal q, k, v, P = softmax(scale q^T k) [NH][T][T], o = v P^T: B = 1.
struct AttnFlashArgs {
    int T, NH, B, CH; float scale;             // CH: channels per head (<= 64)
    const float *q, *k, *v; long hs;
    const float* P;
    const float* o;
    const float *dq, *dk, *dv; long bs_d;     // tangent inputs per probe
    float* out; long bs_out;                  // tangent result do per probe
    const float* go; long bs_go;              // cotangent input g_o per probe
    float *gq, *gk, *gv; long bs_g;           // cotangent results per probe
    float* delta;                             // scratch [B][NH][T]: delta_i = <g_o_i, o_i>
    // keys / values = [text ; image] in one softmax (DeepFloyd-IF): Lt text columns (a multiple of 64, zero-padded) ahead of
    // the T image columns in every row of P ([NH][T][Lt + T]); kt / vt [NH][CH][Lt] constants.  Lt = 0: plain self-attention
    int Lt; const float *kt, *vt;
    // round 5: workspace for the operands' pre-split records (attn_flash_ws_bytes); nullptr / too small: the kernels that convert
    // the operands themselves
    unsigned char* ws; size_t ws_bytes;
};
size_t attn_flash_ws_bytes(const AttnFlashArgs& a);
bool attn_flash_supported(int T, int CH);     // heads of <= 96 channels, token counts that are multiples of 128
bool attn_flash_text_supported(int T, int CH, int Lt);   // the [text ; image] form: heads of <= 64 channels
void launch_attn_flash_tangent(const AttnFlashArgs& a, hipStream_t st);
void launch_attn_flash_cotangent(const AttnFlashArgs& a, hipStream_t st);

// ---- token-wise pieces of the latent-diffusion SpatialTransformer (xfmr.hip); tensors [channel][token] ----
// LayerNorm over the C channels of each token; stats = [2][T] {mean, rstd} per sample
void launch_ln_fwd(const float* x, long xbs, int B, int C, int T, const float* gamma, const float* beta, float eps, float* y,
                   long ybs, float* stats, long sbs, hipStream_t st);
void launch_ln_tan(const float* dx, long dbs, const float* xprim, const float* sprim, int B, int C, int T, const float* gamma,
                   float* dy, long ybs, hipStream_t st);
// gx = base + rstd (gamma gy - mean_c(gamma gy) - xhat mean_c(xhat gamma gy))      (base may be null)
void launch_ln_cot(const float* gy, long gbs, const float* xprim, const float* sprim, int B, int C, int T, const float* gamma,
                   const float* base, long base_bs, float* gx, long xbs, hipStream_t st);
// GEGLU on f = [value | gate] (2 * n4 floats per sample): kind 0 forward, 1 tangent (in = df), 2 cotangent (in = g of the output)
// fused text cross-attention of one pass (xattn.hip): scores of X against K1 -> row operation -> values out of K2
struct XAttnArgs {
    int T, NH, B, CH, L, Lp, fwd;     // tokens, heads, samples, channels per head, real / padded context length; fwd: softmax (writes P)
    float alpha;                      // softmax scale
    const float* X; long x_bs;        // [B][C][T]: q (forward), dq (tangent), g_o (cotangent)
    const float* K1; const float* K2; // [C][Lp]: forward / tangent K_ctx, V_ctx; cotangent V_ctx, K_ctx
    float* P; long p_bs;              // [.][NH][T][Lp]: written by the forward (per sample), read by the others (sample 0)
    float* O; long o_bs;              // [B][C][T]
};
bool xattn_fused_supported(int T, int CH, int L, int Lp);
void launch_xattn_fused(const XAttnArgs& a, hipStream_t st);
void launch_geglu(int kind, const float* in, long in_bs, const float* fprim, int B, long n4, float* out, long out_bs,
                  hipStream_t st);

// ---- GroupNorm statistics -------------------------------------------------
// x: [B][C][HW] with batch stride bs; groups of cpg channels (contiguous cpg*HW floats)
// writes mr[b][g] = {mean, rstd}, sc[b][c] = gamma*rstd, sh[b][c] = beta - mean*rstd*gamma
void launch_gn_stats(const float* x, long bs, int B, int C, int HW, int G, float eps,
                     const float* gamma, const float* beta,
                     float* mr, float* sc, float* sh, long stats_bs, double* scratch, hipStream_t st,
                     const float* ss_scale = nullptr, const float* ss_shift = nullptr);
// row-tile partials of a conv epilogue (ConvArgs::st_part, [B][C][ntile]{mean, M2}) -> the arrays of launch_gn_stats
void launch_gn_fused_finalize_cat(const float* partA, int C1, int ntA, const float* partB, int ntB, int B, int C, int HW, int G,
                                  float eps, const float* gamma, const float* beta, float* mr, float* sc, float* sh,
                                  long stats_bs, hipStream_t st);
void launch_gn_fused_finalize(const float* part, int ntile, int B, int C, int HW, int G, float eps, const float* gamma,
                              const float* beta, float* mr, float* sc, float* sh, long stats_bs, const float* ss_scale,
                              const float* ss_shift, hipStream_t st);
// row-tile partials of a conv epilogue in the tangent / cotangent kinds (ConvArgs::st_part, kind ST_TAN: raw {sum d, sum x d},
// ST_COT: {sum z, sum xhat z}) -> tst[b][g] = {m1, m2} and its per-channel expansion tc, as launch_gn_tstats writes them.  The
// channels [0, C1) come from partA (ntA tiles per row), the channels [C1, C) of a concatenation from partB (ntB tiles; nullptr:
// C1 == C).  mr: the primal {mean, rstd} per group of the consuming norm.
void launch_gn_lin_fused_finalize(int kind, const float* partA, int C1, int ntA, const float* partB, int ntB, int B, int C, int HW,
                                  int G, const float* mr, float* tst, float* tc, long tst_bs, hipStream_t st);
// split-K epilogue (sum of the K-slabs + bias / bias2 / residual / accumulate, as launch_conv_splitk_reduce) that also takes
// the statistics of the finished tensor and finalises them: replaces the reduce AND the statistics launch behind a split-K conv
void launch_conv_splitk_reduce_stats(const ConvArgs& a, int kind, int G, float eps, const float* gamma, const float* beta,
                                     float* mr, float* sc, float* sh, long stats_bs, const float* ss_scale,
                                     const float* ss_shift, const float* prim, const float* sc_prim, const float* sh_prim,
                                     const float* mr_prim, float* tst, float* tc, long tst_bs, double* scratch,
                                     hipStream_t st);
// tangent / cotangent group statistics:
//   kind 0 (tangent):            z = d
//   kind 1 (cotangent, silu):    z = gamma * silu'(y) * d
//   kind 2 (cotangent, no silu): z = gamma * d
//   tst[b][g] = { mean_g(z), mean_g(xh * z) },  xh = (x - mean)*rstd of the primal (prim_bs may be 0)
void launch_gn_tstats(const float* d, long d_bs, const float* x, long x_bs, int B, int C, int HW, int G,
                      const float* sc, const float* sh, const float* mr, long pbs_c, long pbs_g,
                      int kind, float* tst, float* tc, long tst_bs, double* scratch, hipStream_t st, int act = 0);
// elementwise GroupNorm applications (no conv behind them):
//   kind 0: out = sc*x + sh                                    (attention norm forward)
//   kind 1: out = sc*(d - m1 - xh*m2)                          (attention norm tangent)
//   kind 2: out (+)= base + rstd*(gamma*silu'(y)*d - m1 - xh*m2)  (resblock norm1 cotangent)
//   kind 3: out (+)= base + rstd*(gamma*d - m1 - xh*m2)           (attention norm cotangent)
//   kind 4: out = silu(sc*x + sh)                                  (activation ahead of an avg-pool, ADM down block)
//   kind 5: out = silu'(y)*sc*(d - m1 - xh*m2)                     (its tangent)
void launch_gn_apply(int kind, const float* d, long d_bs, const float* x, long x_bs,
                     const float* base, long base_bs, float* out, long out_bs, int accumulate,
                     int B, int C, int HW, int G, const float* sc, const float* sh, const float* mr,
                     long pbs_c, long pbs_g, const float* tst, long tst_bs, hipStream_t st, int act = 0,
                     float base_scale = 1.0f);      // kinds 2, 3: base_scale * base

// GroupNorm of encoder states given token-major: tok [L][D] -> out [L][D], statistics per group of D / G channels over all
// L tokens (the `norm_encoder` of the DeepFloyd-IF attention blocks: GroupNorm on [D][L])
void launch_ctx_groupnorm(const float* tok, int L, int D, int G, float eps, const float* gamma, const float* beta,
                          float* out, hipStream_t st);

// ---- attention helpers ------------------------------------------------------
// rows: [R][T] contiguous; softmax over T in place (S -> P)
void launch_softmax_rows(float* S, long rows, int T, hipStream_t st, int B = 1, long bs = 0);
// dP = P * (dS - rowsum(P*dS)) * scale, in place on dS; P has prow rows broadcast: row r uses P row (r % prow_mod)
void launch_softmax_jac(float* dS, const float* P, long rows, int T, long p_rows, float scale, hipStream_t st,
                        int B = 1, long bs = 0);

// ---- small ops ---------------------------------------------------------------
// temb pipeline: sinusoid(t) -> dense0 -> swish -> dense1 -> swish -> all per-block projections
void launch_temb(float t, int ch, int temb_ch, const float* freq, const float* w0, const float* b0,
                 const float* w1, const float* b1, float* scratch, hipStream_t st, int cos_first = 0,
                 const float* add = nullptr,      // add[temb_ch]: conditioning embedding added to emb before the SiLU
                 const float* t_ptr = nullptr,    // non-null: read the timestep from device memory (graph replay)
                 int act = 0);
void launch_set_scalar(float* p, float v, hipStream_t st);
void launch_clock_stamp(unsigned long long* out2, hipStream_t st);   // {s_memtime, s_memrealtime}
void launch_temb_proj(const float* tact, int temb_ch, const float* w, const float* b, int cout,
                      float* out, hipStream_t st);
// out[b][c][y][x] = sum of the 2x2 block of in[b][c][2y..][2x..]   (adjoint of nearest x2)
void launch_pool2x2_sum(const float* in, long in_bs, float* out, long out_bs, int accumulate,
                        int B, int C, int Hout, int Wout, hipStream_t st, float scale = 1.0f);
void launch_upsample2x(const float* in, long in_bs, float* out, long out_bs, int accumulate, float scale, int B,
                       int C, int Hin, int Win, hipStream_t st);
// strided copy / add of [B][C][HW] tensors
void launch_copy(const float* in, long in_bs, float* out, long out_bs, int accumulate,
                 int B, long per_sample, hipStream_t st);

// ---- DDIM / x0 algebra --------------------------------------------------------
void launch_ddim_step(const float* x, const float* eps, const float* noise, float* out, float* x0_out, long count,
                      float c_x0_x, float c_x0_e, float c_next_x0, float c_next_e, float c_noise,
                      hipStream_t st);
// U = mask * (cv*V + ce*dEps); rows >= split use mask2 when it is given (two solves sharing a probe batch)
void launch_latent_sample(const float* mom, const float* noise, float scale, float* z, int B, long zhw, hipStream_t st);
void launch_masked_axpby(const float* V, const float* dE, const uint8_t* mask, float cv, float ce,
                         float* U, int k, long n, hipStream_t st, const uint8_t* mask2 = nullptr, long split = 0);
// G = mask*U (cotangent seed); outputs gE = ce*G  and keeps cv*G in gX0
void launch_cot_seed(const float* U, const uint8_t* mask, float cv, float ce, float* gE, float* gX0,
                     int k, long n, hipStream_t st, const uint8_t* mask2 = nullptr, long split = 0);
void launch_add(const float* a, const float* b, float* out, long count, hipStream_t st);
// out = sum_{i<n} coef[i] * src[i]   (n <= 4, 16-byte aligned tensors of `count` floats; CFG combination of Jacobian products)
void launch_lincomb(const float* const* src, const float* coef, int n, float* out, long count, hipStream_t st);

// ---- solver --------------------------------------------------------------------
// G[k][k] (double) = A A^T, A: [k][n] fp32
void launch_gram(const float* A, int k, long n, double* G, double* scratch, hipStream_t st);
// cross Gram C[k1][k2] (double) = A B^T
void launch_cross_gram(const float* A, int k1, const float* B, int k2, long n, double* C, double* scratch,
                       hipStream_t st);
// symmetric eigen-decomposition of G (k<=64, double, cyclic Jacobi in one workgroup):
// evals descending in w[k], eigenvectors as rows of Q[k][k]
void launch_jacobi_eig(double* G, int k, double* w, double* Q, hipStream_t st);
// A <- diag(scale) * Q * A   (k x n, in place, via temp copy), scale from w: mode 0: 1/sqrt(max(w,tiny)); mode 1: none
void launch_rotate_rows(const float* Ain, float* Aout, int k, long n, const double* Q, const double* w,
                        int mode, hipStream_t st);
void launch_sign_fix(float* A, int k, long n, float* s_out, const double* w, float* scratch, hipStream_t st);   // scratch: k * ceil(n/4096) floats
// Cholesky factor of G (k x k double, lower) in one workgroup, then A <- L^{-1} A
void launch_cholesky(double* G, int k, hipStream_t st);
void launch_trsm_rows(const float* Ain, float* Aout, int k, long n, const double* L, hipStream_t st);
void launch_convergence(const float* a, const float* b, long count, float atol, float rtol, float* out2,
                        double* scratch, hipStream_t st);
// the same per row and up to each row's sign; scratch: k * 64 * 4 doubles
void launch_convergence_rows(const float* a, const float* b, int k, long n, float atol, float rtol, float* out2,
                             double* scratch, hipStream_t st);
// out = Vm - C^T Vn  (C: [k0][k] double = Vn Vm^T), then row-normalise
void launch_project_rows(const float* Vm, int k, const float* Vn, int k0, long n, const double* C,
                         float* out, hipStream_t st);
void launch_normalize_rows(float* A, int k, long n, double* scratch, hipStream_t st);
void launch_edit_axpy(const float* x, const float* v, const float* alphas_dev, int B, long n, float* out,
                      hipStream_t st);
void launch_mask_gather(const float* U, const int* idx, long L, long n, int k, float* out, hipStream_t st);
// idx[0..L) = ascending positions of the non-zero bytes of mask[0..n), *count = L  (one workgroup, ordered scan)
void launch_mask_compact(const uint8_t* mask, long n, int* idx, int* count, hipStream_t st);

}  // namespace loco
