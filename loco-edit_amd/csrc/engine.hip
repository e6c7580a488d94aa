// loco_ctx: the U-Net program (op list), parameter store, activation arenas and
// the three passes (forward / tangent / cotangent) of the PMP-Jacobian operator,
// plus the C ABI of include/loco_hip.h.
//
// Memory plan (HBM): every logical tensor of the network has a fixed offset in
// a per-sample layout; an arena holds max_batch samples of that layout, so all
// tensors share one batch stride.  Skip tensors are placed directly inside the
// concatenation buffer of the up-block that consumes them (torch.cat of
// reference diffusion.py:186 is never materialised).  Arena P holds the primal
// pass (B images for DDIM loops, B=1 cached for the Jacobian), arena T is shared
// by the tangent and the cotangent pass of a probe batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/loco_hip.h"
#ifdef LOCO_DIAG
#include "../../include/loco_hip_diag.h"
#endif
#include "kernels.h"

using namespace loco;

#define HIPCHK(ctx, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

namespace {

struct Tens {
    long off; int C, H, W;
    int cons_op = -1, cons_norm = 0;   // op whose norm (1: n1, 2: nx) takes its statistics over exactly this tensor, or -1
    // channel concatenation [cat_a | cat_b] consumed by a norm (the up-path ResBlocks read torch.cat([h, skip])): the two
    // parts are tensors of their own, each written by its own conv.  A part keeps the {mean, M2} tile partials its producer's
    // epilogue took (keep, keep_ntile > 0 in the forward pass that wrote them) so the norm over the concatenation can be
    // finalised from both parts' partials instead of re-reading the tensor (gn_fused_finalize_cat_kernel).
    int cat_a = -1, cat_b = -1, cat_of = -1;
    float* keep = nullptr; size_t keep_floats = 0; int keep_ntile = 0;
};

struct ConvP {       // one convolution's parameters in kernel layouts
    float* wf = nullptr;    // forward  [Cin][taps][CoutP]
    float* wd = nullptr;    // dgrad    [Cout][taps][CinP]  (flipped taps)
    float* wbf = nullptr;   // split-bf16 forward records [Cin/16][taps][CoutP][32 x u16]
    float* wbd = nullptr;   // split-bf16 dgrad records   [Cout/16][taps][CinP][32 x u16]
    float* whf = nullptr;   // f16 forward records        [Cin/16][taps][CoutP][16 x f16]
    float* whd = nullptr;   // f16 dgrad records          [Cout/16][taps][CinP][16 x f16]
    float* bias = nullptr;
    int cin = 0, cout = 0, taps = 0;
};
struct NormP {
    float* gamma = nullptr; float* beta = nullptr; int C = 0;
    long soff = 0;      // offset into a stats arena
    long sx_off = -1;   // offset (in float2) into the primal {S, xhat} cache, -1: none
    bool ready = false; // statistics of the current pass were delivered with the producing conv (run_conv StatReq)
    float eps = 0.f;    // 0: cfg.gn_eps; the SpatialTransformer's GroupNorm has its own (1e-6)
};
inline float eps_of(const loco_ctx* c, const NormP& n);

enum OpKind { OP_CONV_IN, OP_RES, OP_ATTN, OP_DOWN, OP_UP, OP_OUT, OP_CONV, OP_XFMR };   // OP_CONV: plain 3x3 conv, tensor -> tensor
// OP_XFMR tensors (latent-diffusion SpatialTransformer, depth 1), all [C][T] unless noted
enum XT { X_G0, X_H0, X_A1, X_QKV, X_S, X_O, X_H1, X_A2, X_XQ, X_XS, X_XO, X_H2, X_A3, X_F, X_GG, X_H3, X_LN1, X_LN2, X_LN3, X_NT };

struct Op {
    OpKind kind;
    std::string name;
    int in = -1, out = -1;        // tensor ids
    int h1 = -1, a1 = -1;         // RES: conv1 output; cotangent scratch with the input's shape
    int hn = -1, qkv = -1, S = -1, o = -1;   // ATTN
    int up = -1;                  // UP: cotangent scratch at the upsampled size
    int ap = -1, xu = -1;         // ADM up/down ResBlock: pooled activation (down), resampled shortcut input
    int updown = 0;               // RES: 0 none, 1 down (avg-pool 2x2 on both branches), 2 up (nearest x2)
    bool scale_shift = false;     // RES: GN(h)*(1+scale)+shift from the embedding (ADM); else conv1 += Linear(temb) (DDPM)
    int heads = 1;                // ATTN
    int ksize = 3;                // CONV_IN: 3 (conv_in of the denoisers) or 1 (post_quant_conv of the latent decoder)
    // ATTN with a text cross-attention stage behind it (cfg.context_dim > 0): xmid = output of the self-attention
    // stage, xhn = GN(xmid), xq = q projection [C][T], xS = scores / probabilities [heads][T][Lp], xo = attended values
    bool has_x = false;
    int xmid = -1, xhn = -1, xq = -1, xS = -1, xo = -1;
    NormP nx;
    ConvP xqc, xproj;
    float *xkw = nullptr, *xkb = nullptr, *xvw = nullptr, *xvb = nullptr;   // key / value projections of the context [C][D], [C]
    float *xK = nullptr, *xV = nullptr;                                     // projected context [C][Lp] (loco_set_context)
    // DeepFloyd-IF attention (cfg.added_kv): keys / values = [text ; image] in one softmax.  The text part reuses xkw .. xV
    // (encoder_kv rows of head h: [k_h | v_h]) behind the block's own GroupNorm of the states (xng / xnb, `norm_encoder`);
    // S is [heads][T][Lp + T] with the (padded, masked) text columns first
    bool added_kv = false;
    float *xng = nullptr, *xnb = nullptr;
    bool in_is_skip = false;
    bool has_nin = false;
    bool has_temb = true;         // RES: false for the embedding-free blocks of the decoder (arch 2)
    // parameter name stems in the reference state_dict
    std::string pn_n1, pn_c1, pn_emb, pn_n2, pn_c2, pn_skip, pn_qkv, pn_proj, pn_conv;
    ConvP c1, c2, nin, qkvc, proj, conv;     // conv: CONV_IN / DOWN / UP / OUT
    NormP n1, n2;                 // RES norm1/norm2, ATTN norm (n1), OUT norm_out (n1)
    long tproj_off = 0;           // RES: offset into the concatenated temb projections
    bool sym_down = false;        // DOWN: conv3 stride 2 with symmetric padding 1 (guided-diffusion Downsample) instead of (0,1,0,1)
    // XFMR: tensors, the three LayerNorms, the linear maps (as 1x1 convs over [C][T])
    int xt[X_NT] = {};
    float *lng[3] = {nullptr, nullptr, nullptr}, *lnb[3] = {nullptr, nullptr, nullptr};
    ConvP pj_in, to_out1, to_out2, ff1, ff2, pj_out;   // qkvc = fused [to_q; to_k; to_v] of attn1 (per-head rows), xqc = attn2.to_q
};

struct HostParam { std::vector<float> data; std::vector<int64_t> shape; bool loaded = false; };

}  // namespace

struct loco_ctx {
    loco_unet_cfg cfg;
    std::string err;
    std::map<std::string, HostParam> params;
    std::vector<std::string> param_order;
    bool finalized = false;

    std::vector<Tens> tens;
    std::vector<Op> ops;
    long per_sample = 0;       // floats per sample in an activation arena
    long stats_per_sample = 0; // floats per sample in a stats arena
    long tproj_total = 0;
    int n_in = 0;              // C*H*W of the network input (image / latent)
    int n_out = 0;             // C*H*W of the network output (= n_in for the denoisers; the decoded image for arch 2)
    int eps_t = -1;            // tensor id of the network output

    float *arenaP = nullptr, *arenaT = nullptr;
    float *statsP = nullptr, *statsT = nullptr;
    double* red = nullptr;         // reduction scratch (doubles)
    bool fuse_cot = true;          // norm-cotangent term in the shortcut conv's epilogue (LOCO_FUSE_COT=0: standalone gn_apply pass)
    float *stpart = nullptr, *stpart2 = nullptr;   // row partials of the statistics taken in conv epilogues (lane 0 / lane 1)
    size_t stpart_floats = 0;
    bool fuse_stats = true;        // LOCO_FUSE_STATS=0: every statistics pass as its own kernels (A/B timing)
    bool fuse_xattn = true;        // LOCO_FUSE_XATTN=0: cross-attention as strided GEMM + row kernel + strided GEMM (A/B)
    bool deep1 = true;             // LOCO_DEEP1=0: 1x1 operators on the two-buffer stage loop instead of the register ring (A/B)
    bool flash_attn = true;        // LOCO_FLASH_ATTN=0: tangent / cotangent attention on the generic GEMM + softmax-Jacobian path
    float* attn_delta = nullptr;   // [max_batch][heads][tokens] scratch of the flash cotangent (a lane's samples start at its first sample: LaneSwap)
    long attn_dmax = 0;            // heads * tokens of the largest attention = the per-sample pitch of attn_delta
    float* partial = nullptr;      // split-K workspace
    size_t partial_floats = 0;
    const float* bench_out = nullptr; int64_t bench_out_count = 0;      // diag: output tensor of the last loco_bench_conv
    float *tact = nullptr, *tproj = nullptr, *freq = nullptr;
    float *tp_w = nullptr, *tp_b = nullptr;          // concatenated temb_proj
    float *td0w = nullptr, *td0b = nullptr, *td1w = nullptr, *td1b = nullptr;
    float* eps_buf = nullptr;      // [max_batch][n]
    float* gx0 = nullptr;          // [max_batch][n] direct cotangent term
    float* ge = nullptr;           // [max_batch][n] cotangent seed of eps
    float* tmpA = nullptr;         // [64][n] temp for solver rotations
    double *G = nullptr, *Q = nullptr, *W = nullptr, *gscratch = nullptr;
    float* alphas = nullptr;
    int ctx_Lp = 0;                // context length padded to a multiple of 64 (score row length of the cross-attention)
    float* ctx_colbias = nullptr;  // [Lp]: 0 for real tokens, -1e30 for the padding
    bool has_ctx = false;
    float* cond_add = nullptr;     // [temb_ch] conditioning embedding of the time embedding (loco_set_cond)
    float* ctx_norm = nullptr;     // [context_len][context_dim] scratch: the states behind one block's norm_encoder (added_kv)
    float res_scale = 1.f;         // cfg.res_scale (0 -> 1): ResBlock output = (shortcut + h) * res_scale
    bool has_cond = false;
    float2* sxcache = nullptr;     // primal {S, xhat} per GroupNorm+SiLU input (bf16x3 path)
    long sx_total = 0;
    int prec = 0;                  // 0: exact fp32 MFMA, 1: split-bf16 (bf16x3) MFMA, 2: single f16 MFMA
    std::vector<float*> owned;     // everything to hipFree
    size_t bytes = 0;

    // PMP state
    bool primal_ok = false;
    float p_cv = 0.f, p_ce = 0.f;
    uint8_t* mask = nullptr;       // device, [n] (owned copy)
    bool has_mask = false;
    uint8_t* mask2 = nullptr;      // device, [n] (owned copy): mask of the probe rows >= mask2_from (loco_pmp_set_second_mask)
    bool has_mask2 = false;
    int mask2_from = 0;
    int* mask_idx = nullptr;       // device: indices of the selected elements, ascending
    int* mask_L_dev = nullptr;     // device: their count
    long mask_L = 0;               // host copy, -1 = not read back yet
    hipStream_t mask_stream = nullptr;
    int primal_B = 0;
    double flops = 0.0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second lane (probe groups of one tangent / cotangent pass on two streams, see run_lanes)
    int n_streams = 1;
    hipStream_t st2 = nullptr;       // the context's own second stream
    hipStream_t st2_user = nullptr;  // caller-supplied second stream (loco_set_side_stream): used instead of st2 when set
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double* red2 = nullptr;
    // B = 1..max_batch denoiser evaluations replayed as HIP graphs (loco_unet_forward / loco_ddim_step): the DDIM
    // opt-in (LOCO_GRAPH=1).  Measured: no gain on an idle host -- the eager launch list already runs back to back
    // (B = 1: 363 kernels, 5.29 ms busy of 5.31 ms span per evaluation) -- it only takes the ~360 launches per
    // evaluation off the host thread.
    struct FwdGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int calls = 0; };
    std::map<int, FwdGraph> fwd_graphs;
    bool graph_on = false;
    bool gemm_lowp = true;         // LOCO_GEMM_LOWP=0: every attention product on the exact f32-input kernel (A/B timing)
    // workspace of the record GEMM (gemm_rec.hip: the operands' split records + K-split partial tiles), one per stream lane, grown
    // on demand outside stream capture
    unsigned char* gemm_ws[2] = {nullptr, nullptr}; size_t gemm_ws_bytes[2] = {0, 0}; int lane = 0;
    // Shared parameter store (loco_fork, round 6): a forked context uses the device copies of its root's parameters (all six
    // layouts) and owns only its arenas, statistics, scratch and per-prompt constants.  The root stays allocated until its last
    // fork is destroyed (`forks`, `zombie`).
    loco_ctx* weights_of = nullptr;
    int forks = 0;
    bool zombie = false;
    bool fuse_lin = true;          // tangent / cotangent group means taken in the conv epilogues (LOCO_FUSE_LIN=0: standalone passes)
    int lanes_active = 1;          // 2 while run_lanes enqueues the two probe groups of a pass on two streams (split-K then aims below the whole chip)
    int lane_s0 = 0;               // first sample of the lane being enqueued (run_lanes): where its rows of a kept partial buffer start
    int chip_share = 1;            // contexts whose passes the host enqueues side by side on other streams (loco_set_chip_share): split-K aims at 256 / share workgroups
    hipStream_t cap_st = nullptr;  // capture stream (the caller's may be the legacy default stream)
    float* xin_buf = nullptr;      // [max_batch][n] fixed graph input
    float* t_dev = nullptr;        // timestep read by the captured time-embedding kernel
    // per-launch conv profile
    bool prof_on = false;
    struct ProfRec { const char* name; double flops; hipEvent_t e0, e1; int cin, cout, h, b, ns, mode, taps; };
    bool prof_shapes = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
};

namespace {

template <typename T>
int dalloc(loco_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    HIPCHK(c, hipMalloc(&q, count * sizeof(T)));
    *p = reinterpret_cast<T*>(q);
    c->owned.push_back(reinterpret_cast<float*>(q));
    c->bytes += count * sizeof(T);
    return 0;
}

long align4(long v) { return (v + 63) & ~63L; }

int new_tensor(loco_ctx* c, int C, int H, int W, long off = -1) {
    Tens t;
    t.C = C; t.H = H; t.W = W;
    if (off < 0) {
        t.off = c->per_sample;
        c->per_sample += align4((long)C * H * W);
    } else {
        t.off = off;
    }
    c->tens.push_back(t);
    return (int)c->tens.size() - 1;
}
inline float eps_of(const loco_ctx* c, const NormP& n) { return n.eps > 0.f ? n.eps : c->cfg.gn_eps; }
NormP new_norm(loco_ctx* c, int C) {
    NormP n;
    n.C = C;
    n.soff = c->stats_per_sample;
    // layout per norm: sc[C], sh[C], mr[2G], tst[2G], tc[2C]
    c->stats_per_sample += align4(4L * C + 4L * c->cfg.gn_groups);
    return n;
}
void norm_cache(loco_ctx* c, NormP& n, int HW) {
    n.sx_off = c->sx_total;
    c->sx_total += (long)n.C * HW;
}
bool attn_at(const loco_unet_cfg& cfg, int res) {
    for (int i = 0; i < cfg.num_attn_res; ++i)
        if (cfg.attn_resolutions[i] == res) return true;
    return false;
}

int build_program_adm(loco_ctx* c);
int build_program_dec(loco_ctx* c);
int build_program_enc(loco_ctx* c);

// Build the op list + memory plan (mirrors DDPM.__init__/forward, reference diffusion.py:22-200)
int build_program(loco_ctx* c) {
    if (c->cfg.arch == 1) return build_program_adm(c);
    if (c->cfg.arch == 2) return build_program_dec(c);
    if (c->cfg.arch == 3) return build_program_enc(c);
    const loco_unet_cfg& cfg = c->cfg;
    const int ch = cfg.ch, nres = cfg.num_levels, R = cfg.resolution;
    c->n_in = cfg.in_channels * R * R;
    c->n_out = cfg.out_ch * R * R;
    auto add_attn = [&](const std::string& name, int in_t, int out_t) {
        Op a; a.kind = OP_ATTN; a.name = name; a.in = in_t; a.out = out_t;
        const Tens& t = c->tens[in_t];
        int T = t.H * t.W;
        a.hn = new_tensor(c, t.C, t.H, t.W);
        a.qkv = new_tensor(c, 3 * t.C, t.H, t.W);
        a.S = new_tensor(c, 1, T, T);
        a.o = new_tensor(c, t.C, t.H, t.W);
        a.n1 = new_norm(c, t.C);
        c->ops.push_back(a);
    };
    // ---- pass 1: shapes of the skip stack, to place skips inside concat buffers
    struct Skip { int C, H; };
    std::vector<Skip> hs;
    {
        int res = R;
        hs.push_back({ch, res});
        for (int l = 0; l < nres; ++l) {
            for (int b = 0; b < cfg.num_res_blocks; ++b) hs.push_back({ch * cfg.ch_mult[l], res});
            if (l != nres - 1) { res /= 2; hs.push_back({ch * cfg.ch_mult[l], res}); }
        }
    }
    // up path consumption order: pops from the back
    // concat buffer j (j-th up block) = [h_prev (C1) | skip hs[n-1-j] (C2)]
    const int nskip = (int)hs.size();
    std::vector<int> cat_t(nskip), skip_t(nskip), hprev_t(nskip);
    {
        int res = hs.back().H;
        int block_in = ch * cfg.ch_mult[nres - 1];
        int j = 0;
        for (int l = nres - 1; l >= 0; --l) {
            int block_out = ch * cfg.ch_mult[l];
            for (int b = 0; b < cfg.num_res_blocks + 1; ++b) {
                const Skip& sk = hs[nskip - 1 - j];
                int C1 = block_in, C2 = sk.C;
                if (sk.H != res) { c->err = "internal: skip resolution mismatch"; return -1; }
                int cat = new_tensor(c, C1 + C2, res, res);
                long base = c->tens[cat].off;
                cat_t[j] = cat;
                hprev_t[j] = new_tensor(c, C1, res, res, base);
                skip_t[nskip - 1 - j] = new_tensor(c, C2, res, res, base + (long)C1 * res * res);
                block_in = block_out;
                ++j;
            }
            if (l != 0) res *= 2;
        }
    }
    // ---- pass 2: ops
    int res = R;
    int si = 0;   // skip index
    {
        Op o; o.kind = OP_CONV_IN; o.name = "conv_in"; o.in = -1; o.out = skip_t[si++];
        c->ops.push_back(o);
    }
    int block_in = ch;
    int cur = skip_t[0];
    for (int l = 0; l < nres; ++l) {
        int block_out = ch * cfg.ch_mult[l];
        for (int b = 0; b < cfg.num_res_blocks; ++b) {
            Op r; r.kind = OP_RES; r.name = "down." + std::to_string(l) + ".block." + std::to_string(b);
            r.in = cur; r.in_is_skip = true;
            r.has_nin = (block_in != block_out);
            bool at = attn_at(cfg, res);
            int out_t = at ? new_tensor(c, block_out, res, res) : skip_t[si];
            r.out = out_t;
            r.h1 = new_tensor(c, block_out, res, res);
            r.a1 = new_tensor(c, block_in, res, res);
            r.n1 = new_norm(c, block_in); r.n2 = new_norm(c, block_out);
            c->ops.push_back(r);
            if (at) add_attn("down." + std::to_string(l) + ".attn." + std::to_string(b), out_t, skip_t[si]);
            cur = skip_t[si++];
            block_in = block_out;
        }
        if (l != nres - 1) {
            Op d; d.kind = OP_DOWN; d.name = "down." + std::to_string(l) + ".downsample.conv";
            d.in = cur; d.in_is_skip = true; d.out = skip_t[si];
            c->ops.push_back(d);
            cur = skip_t[si++];
            res /= 2;
        }
    }
    // middle
    {
        Op r; r.kind = OP_RES; r.name = "mid.block_1"; r.in = cur; r.in_is_skip = true; r.has_nin = false;
        r.out = new_tensor(c, block_in, res, res);
        r.h1 = new_tensor(c, block_in, res, res); r.a1 = new_tensor(c, block_in, res, res);
        r.n1 = new_norm(c, block_in); r.n2 = new_norm(c, block_in);
        c->ops.push_back(r);
        int a_out = new_tensor(c, block_in, res, res);
        add_attn("mid.attn_1", r.out, a_out);
        Op r2; r2.kind = OP_RES; r2.name = "mid.block_2"; r2.in = a_out; r2.has_nin = false;
        r2.out = hprev_t[0];
        r2.h1 = new_tensor(c, block_in, res, res); r2.a1 = new_tensor(c, block_in, res, res);
        r2.n1 = new_norm(c, block_in); r2.n2 = new_norm(c, block_in);
        c->ops.push_back(r2);
    }
    // up path
    {
        int j = 0;
        for (int l = nres - 1; l >= 0; --l) {
            int block_out = ch * cfg.ch_mult[l];
            for (int b = 0; b < cfg.num_res_blocks + 1; ++b) {
                Op r; r.kind = OP_RES; r.name = "up." + std::to_string(l) + ".block." + std::to_string(b);
                r.in = cat_t[j];
                int cin = c->tens[cat_t[j]].C;
                r.has_nin = (cin != block_out);
                bool at = attn_at(cfg, res);
                bool last_of_level = (b == cfg.num_res_blocks);
                bool final_block = (l == 0 && last_of_level);
                // where does the block's (or its attention's) output go?
                int dest;
                if (final_block) dest = new_tensor(c, block_out, res, res);
                else if (last_of_level) dest = new_tensor(c, block_out, res, res);   // feeds the upsample conv
                else dest = hprev_t[j + 1];
                int out_t = at ? new_tensor(c, block_out, res, res) : dest;
                r.out = out_t;
                r.h1 = new_tensor(c, block_out, res, res);
                r.a1 = new_tensor(c, cin, res, res);
                r.n1 = new_norm(c, cin); r.n2 = new_norm(c, block_out);
                c->ops.push_back(r);
                if (at) add_attn("up." + std::to_string(l) + ".attn." + std::to_string(b), out_t, dest);
                cur = dest;
                ++j;
                if (last_of_level && l != 0) {
                    Op u; u.kind = OP_UP; u.name = "up." + std::to_string(l) + ".upsample.conv";
                    u.in = cur; u.out = hprev_t[j];
                    u.up = new_tensor(c, block_out, res * 2, res * 2);
                    c->ops.push_back(u);
                    cur = u.out;
                    res *= 2;
                }
            }
        }
    }
    {
        Op o; o.kind = OP_OUT; o.name = "conv_out"; o.in = cur;
        o.out = new_tensor(c, cfg.out_ch, R, R);
        o.a1 = new_tensor(c, c->tens[cur].C, R, R);
        o.n1 = new_norm(c, c->tens[cur].C);
        c->ops.push_back(o);
        c->eps_t = o.out;
    }
    for (auto& op : c->ops) {
        if (op.kind == OP_RES) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
            norm_cache(c, op.n2, ti.H * ti.W);
            op.pn_n1 = op.name + ".norm1"; op.pn_c1 = op.name + ".conv1"; op.pn_emb = op.name + ".temb_proj";
            op.pn_n2 = op.name + ".norm2"; op.pn_c2 = op.name + ".conv2"; op.pn_skip = op.name + ".nin_shortcut";
        } else if (op.kind == OP_ATTN) {
            op.pn_n1 = op.name + ".norm"; op.pn_proj = op.name + ".proj_out";
        } else if (op.kind == OP_OUT) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
            op.pn_n1 = "norm_out"; op.pn_conv = "conv_out";
        } else {
            op.pn_conv = op.name;
        }
    }
    return 0;
}


// guided-diffusion / P2 U-Net (reference guided_diffusion/unet.py:398-684 with P2_DICT script_util.py:166-190):
// input_blocks = conv, then per level {ResBlock [+Attention]} x num_res_blocks and a ResBlock(down) between levels;
// middle = Res, Attn, Res; output_blocks = {ResBlock(cat) [+Attention] [+ResBlock(up)]}; out = GN, SiLU, conv.
int build_program_adm(loco_ctx* c) {
    const loco_unet_cfg& cfg = c->cfg;
    const int mc = cfg.ch, nlev = cfg.num_levels, R = cfg.resolution;
    c->n_in = cfg.in_channels * R * R;
    c->n_out = cfg.out_ch * R * R;
    c->ctx_Lp = cfg.context_dim > 0 ? ((cfg.context_len + 63) / 64) * 64 : 0;
    auto heads_of = [&](int C) { return cfg.num_heads > 0 ? cfg.num_heads : (cfg.num_head_channels > 0 ? C / cfg.num_head_channels : 1); };
    auto add_res = [&](const std::string& name, int in_t, int out_t, int updown, bool in_is_skip) {
        Op r; r.kind = OP_RES; r.name = name; r.in = in_t; r.out = out_t; r.updown = updown;
        r.scale_shift = cfg.scale_shift_norm != 0; r.in_is_skip = in_is_skip;
        const Tens ti = c->tens[in_t];
        const Tens to = c->tens[out_t];
        r.has_nin = (ti.C != to.C);
        r.h1 = new_tensor(c, to.C, to.H, to.W);
        r.a1 = new_tensor(c, ti.C, ti.H, ti.W);
        if (updown) { r.xu = new_tensor(c, ti.C, to.H, to.W); }
        if (updown == 1) { r.ap = new_tensor(c, ti.C, to.H, to.W); }
        r.n1 = new_norm(c, ti.C); r.n2 = new_norm(c, to.C);
        r.pn_n1 = name + ".in_layers.0"; r.pn_c1 = name + ".in_layers.2"; r.pn_emb = name + ".emb_layers.1";
        r.pn_n2 = name + ".out_layers.0"; r.pn_c2 = name + ".out_layers.3"; r.pn_skip = name + ".skip_connection";
        c->ops.push_back(r);
    };
    auto add_xfmr = [&](const std::string& name, int in_t, int out_t) {
        Op a; a.kind = OP_XFMR; a.name = name; a.in = in_t; a.out = out_t;
        const Tens t = c->tens[in_t];
        const int T = t.H * t.W, C = t.C;
        a.heads = heads_of(C);
        a.has_x = true;
        auto ct = [&](int ch) { return new_tensor(c, ch, t.H, t.W); };
        a.xt[X_G0] = ct(C); a.xt[X_H0] = ct(C); a.xt[X_A1] = ct(C); a.xt[X_QKV] = ct(3 * C);
        a.xt[X_S] = new_tensor(c, a.heads, T, T); a.xt[X_O] = ct(C); a.xt[X_H1] = ct(C); a.xt[X_A2] = ct(C);
        a.xt[X_XQ] = ct(C); a.xt[X_XS] = new_tensor(c, a.heads, T, c->ctx_Lp); a.xt[X_XO] = ct(C); a.xt[X_H2] = ct(C);
        a.xt[X_A3] = ct(C); a.xt[X_F] = ct(8 * C); a.xt[X_GG] = ct(4 * C); a.xt[X_H3] = ct(C);
        a.xt[X_LN1] = new_tensor(c, 2, 1, T); a.xt[X_LN2] = new_tensor(c, 2, 1, T); a.xt[X_LN3] = new_tensor(c, 2, 1, T);
        // the attention helpers address the block through the ATTN field names
        a.qkv = a.xt[X_QKV]; a.S = a.xt[X_S]; a.o = a.xt[X_O]; a.xq = a.xt[X_XQ]; a.xS = a.xt[X_XS]; a.xo = a.xt[X_XO];
        a.n1 = new_norm(c, C); a.n1.eps = 1e-6f;
        a.pn_n1 = name + ".norm";
        c->ops.push_back(a);
    };
    auto add_attn = [&](const std::string& name, int in_t, int out_t) {
        if (cfg.transformer_depth > 0) { add_xfmr(name, in_t, out_t); return; }
        Op a; a.kind = OP_ATTN; a.name = name; a.in = in_t; a.out = out_t;
        const Tens t = c->tens[in_t];
        int T = t.H * t.W;
        a.heads = heads_of(t.C);
        a.hn = new_tensor(c, t.C, t.H, t.W);
        a.qkv = new_tensor(c, 3 * t.C, t.H, t.W);
        a.added_kv = cfg.added_kv != 0;
        a.S = new_tensor(c, a.heads, T, a.added_kv ? c->ctx_Lp + T : T);
        a.o = new_tensor(c, t.C, t.H, t.W);
        a.n1 = new_norm(c, t.C);
        a.pn_n1 = name + ".norm"; a.pn_qkv = name + ".qkv"; a.pn_proj = name + ".proj_out";
        if (cfg.context_dim > 0 && !a.added_kv) {
            a.has_x = true;
            a.xmid = new_tensor(c, t.C, t.H, t.W);
            a.xhn = new_tensor(c, t.C, t.H, t.W);
            a.xq = new_tensor(c, t.C, t.H, t.W);
            a.xS = new_tensor(c, a.heads, T, c->ctx_Lp);
            a.xo = new_tensor(c, t.C, t.H, t.W);
            a.nx = new_norm(c, t.C);
        }
        c->ops.push_back(a);
    };
    // pass 1: skip stack (channels, resolution) in push order
    struct Skip { int C, H; };
    std::vector<Skip> hs;
    {
        int ch = mc * cfg.ch_mult[0], res = R;
        hs.push_back({ch, res});
        for (int l = 0; l < nlev; ++l) {
            for (int b = 0; b < cfg.num_res_blocks; ++b) { ch = mc * cfg.ch_mult[l]; hs.push_back({ch, res}); }
            if (l != nlev - 1) { res /= 2; hs.push_back({ch, res}); }
        }
    }
    const int nskip = (int)hs.size();
    std::vector<int> cat_t(nskip), skip_t(nskip), hprev_t(nskip);
    {
        int res = hs.back().H, ch = hs.back().C, j = 0;
        for (int l = nlev - 1; l >= 0; --l) {
            for (int i = 0; i < cfg.num_res_blocks + 1; ++i) {
                const Skip& sk = hs[nskip - 1 - j];
                if (sk.H != res) { c->err = "internal: skip resolution mismatch (adm)"; return -1; }
                int cat = new_tensor(c, ch + sk.C, res, res);
                long base = c->tens[cat].off;
                cat_t[j] = cat;
                hprev_t[j] = new_tensor(c, ch, res, res, base);
                skip_t[nskip - 1 - j] = new_tensor(c, sk.C, res, res, base + (long)ch * res * res);
                ch = mc * cfg.ch_mult[l];
                if (l && i == cfg.num_res_blocks) res *= 2;
                ++j;
            }
        }
    }
    // pass 2: ops
    int si = 0, res = R, ib = 1;
    {
        Op o; o.kind = OP_CONV_IN; o.name = "input_blocks.0.0"; o.out = skip_t[si++]; o.pn_conv = "input_blocks.0.0";
        c->ops.push_back(o);
    }
    int cur = skip_t[0];
    int ch = mc * cfg.ch_mult[0];
    for (int l = 0; l < nlev; ++l) {
        for (int b = 0; b < cfg.num_res_blocks; ++b) {
            int cout = mc * cfg.ch_mult[l];
            bool at = attn_at(cfg, res);
            std::string nm = "input_blocks." + std::to_string(ib);
            int out_t = at ? new_tensor(c, cout, res, res) : skip_t[si];
            add_res(nm + ".0", cur, out_t, 0, true);
            if (at) add_attn(nm + ".1", out_t, skip_t[si]);
            cur = skip_t[si++]; ch = cout; ++ib;
        }
        if (l != nlev - 1) {
            std::string nm = "input_blocks." + std::to_string(ib);
            if (cfg.resblock_updown) {
                add_res(nm + ".0", cur, skip_t[si], 1, true);
            } else {        // Downsample(use_conv=True): conv3 stride 2 padding 1 (unet.py:113-142)
                Op d; d.kind = OP_DOWN; d.name = nm + ".0.op"; d.pn_conv = d.name; d.in = cur; d.in_is_skip = true;
                d.out = skip_t[si]; d.sym_down = true;
                c->ops.push_back(d);
            }
            cur = skip_t[si++]; ++ib; res /= 2;
        }
    }
    {
        int t0 = new_tensor(c, ch, res, res);
        add_res("middle_block.0", cur, t0, 0, true);
        int t1 = new_tensor(c, ch, res, res);
        add_attn("middle_block.1", t0, t1);
        add_res("middle_block.2", t1, hprev_t[0], 0, false);
    }
    {
        int j = 0, ob = 0;
        for (int l = nlev - 1; l >= 0; --l) {
            for (int i = 0; i < cfg.num_res_blocks + 1; ++i) {
                int cout = mc * cfg.ch_mult[l];
                bool at = attn_at(cfg, res);
                bool has_up = (l != 0 && i == cfg.num_res_blocks);
                bool final_block = (l == 0 && i == cfg.num_res_blocks);
                std::string nm = "output_blocks." + std::to_string(ob);
                // destination of the block's last op
                int dest = final_block ? new_tensor(c, cout, res, res) : -1;
                int sub = 1;
                int res_out = (at || has_up) ? new_tensor(c, cout, res, res) : (final_block ? dest : hprev_t[j + 1]);
                add_res(nm + ".0", cat_t[j], res_out, 0, false);
                int last = res_out;
                if (at) {
                    int a_out = has_up ? new_tensor(c, cout, res, res) : (final_block ? dest : hprev_t[j + 1]);
                    add_attn(nm + "." + std::to_string(sub++), last, a_out);
                    last = a_out;
                }
                if (has_up) {
                    if (cfg.resblock_updown) {
                        add_res(nm + "." + std::to_string(sub++), last, hprev_t[j + 1], 2, false);
                    } else {    // Upsample(use_conv=True): nearest x2 + conv3 (unet.py:83-110)
                        Op u; u.kind = OP_UP; u.name = nm + "." + std::to_string(sub++) + ".conv"; u.pn_conv = u.name;
                        u.in = last; u.out = hprev_t[j + 1];
                        u.up = new_tensor(c, cout, res * 2, res * 2);
                        c->ops.push_back(u);
                    }
                    last = hprev_t[j + 1];
                    res *= 2;
                }
                cur = last; ch = cout;
                ++j; ++ob;
            }
        }
    }
    {
        Op o; o.kind = OP_OUT; o.name = "out"; o.in = cur;
        o.out = new_tensor(c, cfg.out_ch, R, R);
        o.a1 = new_tensor(c, c->tens[cur].C, R, R);
        o.n1 = new_norm(c, c->tens[cur].C);
        o.pn_n1 = "out.0"; o.pn_conv = "out.2";
        c->ops.push_back(o);
        c->eps_t = o.out;
    }
    for (auto& op : c->ops) {
        if (op.kind == OP_RES) {
            const Tens& ti = c->tens[op.in];
            const Tens& to = c->tens[op.out];
            norm_cache(c, op.n1, ti.H * ti.W);
            norm_cache(c, op.n2, to.H * to.W);
        } else if (op.kind == OP_OUT) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
        }
    }
    return 0;
}

// Latent decoder (arch 2): the `Decoder` of the latent-diffusion autoencoder that `vae.decode` runs in the reference's
// Stable Diffusion path (edit.py:750, 770-771; diffusers AutoencoderKL, un-vendored -- same module tree as the DDPM
// U-Net's up half without skips and without a time embedding): conv_in (z_channels -> ch*ch_mult[-1]) at the latent
// resolution R; mid.block_1, mid.attn_1, mid.block_2; for each level from the coarsest: num_res_blocks + 1
// ResnetBlocks [+ attention at cfg.attn_resolutions] and, except at level 0, nearest x2 + conv3; norm_out, SiLU,
// conv_out.  Output [out_ch, R * 2^(levels-1), same].
int build_program_dec(loco_ctx* c) {
    const loco_unet_cfg& cfg = c->cfg;
    const int ch = cfg.ch, nlev = cfg.num_levels, R = cfg.resolution;
    const int Rout = R << (nlev - 1);
    c->n_in = cfg.in_channels * R * R;
    c->n_out = cfg.out_ch * Rout * Rout;
    auto add_res = [&](const std::string& name, int in_t, int cout) {
        Op r; r.kind = OP_RES; r.name = name; r.in = in_t; r.has_temb = false;
        const Tens ti = c->tens[in_t];
        r.has_nin = (ti.C != cout);
        r.out = new_tensor(c, cout, ti.H, ti.W);
        r.h1 = new_tensor(c, cout, ti.H, ti.W);
        r.a1 = new_tensor(c, ti.C, ti.H, ti.W);
        r.n1 = new_norm(c, ti.C); r.n2 = new_norm(c, cout);
        c->ops.push_back(r);
        return r.out;
    };
    auto add_attn = [&](const std::string& name, int in_t) {
        Op a; a.kind = OP_ATTN; a.name = name; a.in = in_t;
        const Tens t = c->tens[in_t];
        const int T = t.H * t.W;
        a.out = new_tensor(c, t.C, t.H, t.W);
        a.hn = new_tensor(c, t.C, t.H, t.W);
        a.qkv = new_tensor(c, 3 * t.C, t.H, t.W);
        a.S = new_tensor(c, 1, T, T);
        a.o = new_tensor(c, t.C, t.H, t.W);
        a.n1 = new_norm(c, t.C);
        c->ops.push_back(a);
        return a.out;
    };
    int res = R, block_in = ch * cfg.ch_mult[nlev - 1];
    int cur;
    {   // AutoencoderKL.decode = decoder(post_quant_conv(z)): the 1x1 conv on the latent channels comes first
        Op o; o.kind = OP_CONV_IN; o.name = "post_quant_conv"; o.in = -1; o.ksize = 1;
        o.out = new_tensor(c, cfg.in_channels, res, res);
        c->ops.push_back(o);
        Op ci; ci.kind = OP_CONV; ci.name = "conv_in"; ci.in = o.out; ci.out = new_tensor(c, block_in, res, res);
        c->ops.push_back(ci);
        cur = ci.out;
    }
    cur = add_res("mid.block_1", cur, block_in);
    cur = add_attn("mid.attn_1", cur);
    cur = add_res("mid.block_2", cur, block_in);
    for (int l = nlev - 1; l >= 0; --l) {
        const int block_out = ch * cfg.ch_mult[l];
        for (int b = 0; b < cfg.num_res_blocks + 1; ++b) {
            cur = add_res("up." + std::to_string(l) + ".block." + std::to_string(b), cur, block_out);
            if (attn_at(cfg, res)) cur = add_attn("up." + std::to_string(l) + ".attn." + std::to_string(b), cur);
            block_in = block_out;
        }
        if (l != 0) {
            Op u; u.kind = OP_UP; u.name = "up." + std::to_string(l) + ".upsample.conv";
            u.in = cur; u.out = new_tensor(c, block_in, res * 2, res * 2);
            u.up = new_tensor(c, block_in, res * 2, res * 2);
            c->ops.push_back(u);
            cur = u.out;
            res *= 2;
        }
    }
    {
        Op o; o.kind = OP_OUT; o.name = "conv_out"; o.in = cur;
        o.out = new_tensor(c, cfg.out_ch, Rout, Rout);
        o.a1 = new_tensor(c, c->tens[cur].C, Rout, Rout);
        o.n1 = new_norm(c, c->tens[cur].C);
        c->ops.push_back(o);
        c->eps_t = o.out;
    }
    for (auto& op : c->ops) {
        if (op.kind == OP_RES) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
            norm_cache(c, op.n2, ti.H * ti.W);
            op.pn_n1 = op.name + ".norm1"; op.pn_c1 = op.name + ".conv1";
            op.pn_n2 = op.name + ".norm2"; op.pn_c2 = op.name + ".conv2"; op.pn_skip = op.name + ".nin_shortcut";
        } else if (op.kind == OP_ATTN) {
            op.pn_n1 = op.name + ".norm"; op.pn_proj = op.name + ".proj_out";
        } else if (op.kind == OP_OUT) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
            op.pn_n1 = "norm_out"; op.pn_conv = "conv_out";
        } else {
            op.pn_conv = op.name;
        }
    }
    return 0;
}

// Latent encoder (arch 3): `vae.encode` of the reference's latent inversion (edit.py:594-597; diffusers AutoencoderKL =
// the latent-diffusion `Encoder` + `quant_conv`, un-vendored): conv_in at the image resolution R; per level num_res_blocks
// embedding-free ResnetBlocks and, except on the last level, pad (0,1,0,1) + conv3 stride 2; mid block / attention / block;
// norm_out, SiLU, conv_out (2 z channels: mean | log-variance), 1x1 quant_conv.  Output [out_ch, R >> (levels-1), same].
int build_program_enc(loco_ctx* c) {
    const loco_unet_cfg& cfg = c->cfg;
    const int ch = cfg.ch, nlev = cfg.num_levels, R = cfg.resolution;
    const int Rout = R >> (nlev - 1);
    c->n_in = cfg.in_channels * R * R;
    c->n_out = cfg.out_ch * Rout * Rout;
    auto add_res = [&](const std::string& name, int in_t, int cout) {
        Op r; r.kind = OP_RES; r.name = name; r.in = in_t; r.has_temb = false;
        const Tens ti = c->tens[in_t];
        r.has_nin = (ti.C != cout);
        r.out = new_tensor(c, cout, ti.H, ti.W);
        r.h1 = new_tensor(c, cout, ti.H, ti.W);
        r.a1 = new_tensor(c, ti.C, ti.H, ti.W);
        r.n1 = new_norm(c, ti.C); r.n2 = new_norm(c, cout);
        c->ops.push_back(r);
        return r.out;
    };
    int res = R, cur;
    {
        Op o; o.kind = OP_CONV_IN; o.name = "conv_in"; o.in = -1; o.out = new_tensor(c, ch, res, res);
        c->ops.push_back(o);
        cur = o.out;
    }
    for (int l = 0; l < nlev; ++l) {
        const int block_out = ch * cfg.ch_mult[l];
        for (int b = 0; b < cfg.num_res_blocks; ++b)
            cur = add_res("down." + std::to_string(l) + ".block." + std::to_string(b), cur, block_out);
        if (l != nlev - 1) {
            Op d; d.kind = OP_DOWN; d.name = "down." + std::to_string(l) + ".downsample.conv"; d.in = cur;
            d.out = new_tensor(c, block_out, res / 2, res / 2);
            c->ops.push_back(d);
            cur = d.out; res /= 2;
        }
    }
    const int block_in = c->tens[cur].C;
    cur = add_res("mid.block_1", cur, block_in);
    {
        Op a; a.kind = OP_ATTN; a.name = "mid.attn_1"; a.in = cur;
        const Tens t = c->tens[cur];
        const int T = t.H * t.W;
        a.out = new_tensor(c, t.C, t.H, t.W);
        a.hn = new_tensor(c, t.C, t.H, t.W);
        a.qkv = new_tensor(c, 3 * t.C, t.H, t.W);
        a.S = new_tensor(c, 1, T, T);
        a.o = new_tensor(c, t.C, t.H, t.W);
        a.n1 = new_norm(c, t.C);
        c->ops.push_back(a);
        cur = a.out;
    }
    cur = add_res("mid.block_2", cur, block_in);
    {
        Op o; o.kind = OP_OUT; o.name = "conv_out"; o.in = cur;
        o.out = new_tensor(c, cfg.out_ch, Rout, Rout);
        o.a1 = new_tensor(c, c->tens[cur].C, Rout, Rout);
        o.n1 = new_norm(c, c->tens[cur].C);
        c->ops.push_back(o);
        Op q; q.kind = OP_CONV; q.name = "quant_conv"; q.ksize = 1; q.in = o.out; q.out = new_tensor(c, cfg.out_ch, Rout, Rout);
        c->ops.push_back(q);
        c->eps_t = q.out;
    }
    for (auto& op : c->ops) {
        if (op.kind == OP_RES) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
            norm_cache(c, op.n2, ti.H * ti.W);
            op.pn_n1 = op.name + ".norm1"; op.pn_c1 = op.name + ".conv1";
            op.pn_n2 = op.name + ".norm2"; op.pn_c2 = op.name + ".conv2"; op.pn_skip = op.name + ".nin_shortcut";
        } else if (op.kind == OP_ATTN) {
            op.pn_n1 = op.name + ".norm"; op.pn_proj = op.name + ".proj_out";
        } else if (op.kind == OP_OUT) {
            const Tens& ti = c->tens[op.in];
            norm_cache(c, op.n1, ti.H * ti.W);
            op.pn_n1 = "norm_out"; op.pn_conv = "conv_out";
        } else {
            op.pn_conv = op.name;
        }
    }
    return 0;
}

void declare_param(loco_ctx* c, const std::string& name, std::vector<int64_t> shape) {
    HostParam hp; hp.shape = shape;
    c->params[name] = hp;
    c->param_order.push_back(name);
}
void declare_conv(loco_ctx* c, const std::string& n, int cin, int cout, int k) {
    declare_param(c, n + ".weight", {cout, cin, k, k});
    declare_param(c, n + ".bias", {cout});
}
void declare_norm(loco_ctx* c, const std::string& n, int C) {
    declare_param(c, n + ".weight", {C});
    declare_param(c, n + ".bias", {C});
}
void declare_lin(loco_ctx* c, const std::string& n, int cin, int cout) {
    declare_param(c, n + ".weight", {cout, cin});
    declare_param(c, n + ".bias", {cout});
}

void declare_all(loco_ctx* c) {
    const loco_unet_cfg& cfg = c->cfg;
    const bool adm = cfg.arch == 1;
    int temb_ch = cfg.ch * 4;
    if (cfg.arch < 2) {
        declare_lin(c, adm ? "time_embed.0" : "temb.dense.0", cfg.ch, temb_ch);
        declare_lin(c, adm ? "time_embed.2" : "temb.dense.1", temb_ch, temb_ch);
    }
    for (auto& op : c->ops) {
        switch (op.kind) {
            case OP_CONV_IN: declare_conv(c, op.pn_conv, cfg.in_channels, c->tens[op.out].C, op.ksize); break;
            case OP_CONV: declare_conv(c, op.pn_conv, c->tens[op.in].C, c->tens[op.out].C, op.ksize); break;
            case OP_RES: {
                int cin = c->tens[op.in].C, cout = c->tens[op.out].C;
                declare_norm(c, op.pn_n1, cin);
                declare_conv(c, op.pn_c1, cin, cout, 3);
                if (op.has_temb) declare_lin(c, op.pn_emb, temb_ch, op.scale_shift ? 2 * cout : cout);
                declare_norm(c, op.pn_n2, cout);
                declare_conv(c, op.pn_c2, cout, cout, 3);
                if (op.has_nin) declare_conv(c, op.pn_skip, cin, cout, 1);
                break;
            }
            case OP_ATTN: {
                int C = c->tens[op.in].C;
                declare_norm(c, op.pn_n1, C);
                if (adm) {   // Conv1d weights [3C, C, 1] / [C, C, 1] (unet.py:286,296)
                    declare_param(c, op.pn_qkv + ".weight", {3 * C, C, 1});
                    declare_param(c, op.pn_qkv + ".bias", {3 * C});
                    declare_param(c, op.pn_proj + ".weight", {C, C, 1});
                    declare_param(c, op.pn_proj + ".bias", {C});
                    if (op.added_kv) {   // deepfloyd_if AttentionBlock: norm_encoder (GroupNorm over the states), encoder_kv Conv1d
                        declare_norm(c, op.name + ".norm_encoder", cfg.context_dim);
                        declare_param(c, op.name + ".encoder_kv.weight", {2 * C, cfg.context_dim, 1});
                        declare_param(c, op.name + ".encoder_kv.bias", {2 * C});
                    }
                    if (op.has_x) {
                        const std::string x = op.name + ".xattn";
                        declare_norm(c, x + ".norm", C);
                        declare_param(c, x + ".q.weight", {C, C, 1}); declare_param(c, x + ".q.bias", {C});
                        declare_lin(c, x + ".k", cfg.context_dim, C);
                        declare_lin(c, x + ".v", cfg.context_dim, C);
                        declare_param(c, x + ".proj_out.weight", {C, C, 1}); declare_param(c, x + ".proj_out.bias", {C});
                    }
                } else {
                    for (const char* p : {"q", "k", "v"}) declare_conv(c, op.name + "." + p, C, C, 1);
                    declare_conv(c, op.pn_proj, C, C, 1);
                }
                break;
            }
            case OP_DOWN: case OP_UP: {
                int C = c->tens[op.in].C;
                declare_conv(c, op.pn_conv, C, C, 3);
                break;
            }
            case OP_XFMR: {      // latent-diffusion SpatialTransformer, depth 1 (ldm/modules/attention.py parameter names)
                const int C = c->tens[op.in].C, D = cfg.context_dim;
                const std::string b = op.name + ".transformer_blocks.0";
                declare_norm(c, op.pn_n1, C);
                declare_conv(c, op.name + ".proj_in", C, C, 1);
                for (const char* n : {".norm1", ".norm2", ".norm3"}) declare_norm(c, b + n, C);
                for (const char* n : {".attn1.to_q", ".attn1.to_k", ".attn1.to_v"}) declare_param(c, b + n + ".weight", {C, C});
                declare_lin(c, b + ".attn1.to_out.0", C, C);
                declare_param(c, b + ".attn2.to_q.weight", {C, C});
                declare_param(c, b + ".attn2.to_k.weight", {C, D});
                declare_param(c, b + ".attn2.to_v.weight", {C, D});
                declare_lin(c, b + ".attn2.to_out.0", C, C);
                declare_lin(c, b + ".ff.net.0.proj", C, 8 * C);
                declare_lin(c, b + ".ff.net.2", 4 * C, C);
                declare_conv(c, op.name + ".proj_out", C, C, 1);
                break;
            }
            case OP_OUT:
                declare_norm(c, op.pn_n1, c->tens[op.in].C);
                declare_conv(c, op.pn_conv, c->tens[op.in].C, cfg.out_ch * (cfg.learn_sigma ? 2 : 1), 3);
                break;
        }
    }
}

int upload(loco_ctx* c, float** dst, const std::vector<float>& h) {
    if (dalloc(c, dst, h.size() ? h.size() : 1)) return -1;
    HIPCHK(c, hipMemcpy(*dst, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

static inline uint16_t f2bf(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f;
}
// Host-side layout building is split over threads (independent output rows): an 860 M-parameter denoiser has six layouts of
// every operator to build, single-threaded 22 s per engine context (three contexts per T-LOCO object).
template <typename F>
static void parallel_for(int n, size_t work, F fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if ((int)nt > n) nt = (unsigned)n;
    if (nt <= 1 || work < ((size_t)1 << 18)) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&]() { int i; while ((i = next.fetch_add(1)) < n) fn(i); });
    for (auto& t : th) t.join();
}

// split-bf16 records: W(o, i, t) for o < nout, i < nin -> [ceil(nin/16)][taps][noutP][32 x u16], 64-byte record
// per (chunk, tap, o): logical 16-byte pieces {hi k0-7, hi k8-15, lo k0-7, lo k8-15} stored at piece ^ ((o>>2)&3)
template <typename F>
static std::vector<float> build_records(int nin, int nout, int taps, F get) {
    int nch = (nin + 15) / 16, noutP = (nout + 31) & ~31;
    std::vector<uint16_t> rec((size_t)nch * taps * noutP * 32, 0);
    parallel_for(nch, (size_t)nin * nout * taps, [&](int ck) {
        for (int t = 0; t < taps; ++t)
            for (int o = 0; o < nout; ++o) {
                uint16_t* r = rec.data() + (((size_t)ck * taps + t) * noutP + o) * 32;
                int sw = (o >> 2) & 3;
                for (int k = 0; k < 16; ++k) {
                    int i = ck * 16 + k;
                    float v = (i < nin) ? get(o, i, t) : 0.f;
                    uint16_t h = f2bf(v);
                    uint16_t l = f2bf(v - bf2f(h));
                    int qh = (k >> 3) ^ sw, ql = (2 + (k >> 3)) ^ sw;
                    r[qh * 8 + (k & 7)] = h;
                    r[ql * 8 + (k & 7)] = l;
                }
            }
    });
    std::vector<float> out(rec.size() / 2);
    std::memcpy(out.data(), rec.data(), rec.size() * 2);
    return out;
}

// f16 records: W(o, i, t) -> [ceil(nin/16)][taps][noutP][16 x f16], 32-byte record per (chunk, tap, o): logical 16-byte
// pieces {k0-7, k8-15} stored at piece ^ ((o>>3)&1)
template <typename F>
static std::vector<float> build_records_f16(int nin, int nout, int taps, F get) {
    int nch = (nin + 15) / 16, noutP = (nout + 31) & ~31;
    std::vector<_Float16> rec((size_t)nch * taps * noutP * 16, (_Float16)0.f);
    parallel_for(nch, (size_t)nin * nout * taps, [&](int ck) {
        for (int t = 0; t < taps; ++t)
            for (int o = 0; o < nout; ++o) {
                _Float16* r = rec.data() + (((size_t)ck * taps + t) * noutP + o) * 16;
                int sw = (o >> 3) & 1;
                for (int k = 0; k < 16; ++k) {
                    int i = ck * 16 + k;
                    float v = (i < nin) ? get(o, i, t) : 0.f;
                    r[((k >> 3) ^ sw) * 8 + (k & 7)] = (_Float16)v;     // round to nearest even
                }
            }
    });
    std::vector<float> out(rec.size() / 2);
    std::memcpy(out.data(), rec.data(), rec.size() * 2);
    return out;
}

// weights [cout][cin][k][k] -> forward [cin][taps][coutP], dgrad [cout][taps][cinP] with flipped taps
int make_conv(loco_ctx* c, const std::vector<const HostParam*>& ws, const std::vector<const HostParam*>& bs,
              ConvP* out, int row_limit = -1) {
    int cin = (int)ws[0]->shape[1], k = ws[0]->shape.size() > 2 ? (int)ws[0]->shape[2] : 1;
    int taps = ws[0]->shape.size() == 4 ? k * k : k;      // Conv2d, Conv1d (k=1) or Linear
    int cout = 0;
    for (auto* w : ws) cout += (int)w->shape[0];
    if (row_limit > 0 && ws.size() == 1 && row_limit < cout) cout = row_limit;   // eps half of a learn_sigma head
    int coutP = (cout + 31) & ~31, cinP = (cin + 31) & ~31;
    std::vector<float> wf((size_t)cin * taps * coutP, 0.f), wd((size_t)cout * taps * cinP, 0.f), bias(cout);
    int co0 = 0;
    for (size_t wi = 0; wi < ws.size(); ++wi) {
        const HostParam* w = ws[wi];
        int co_n = (int)w->shape[0];
        if (co_n > cout - co0) co_n = cout - co0;
        parallel_for(co_n, (size_t)co_n * cin * taps, [&](int co) {      // dgrad layout: one row block per cout
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    wd[((size_t)(co0 + co) * taps + (taps - 1 - t)) * cinP + ci] = w->data[((size_t)co * cin + ci) * taps + t];
        });
        parallel_for(cin, (size_t)co_n * cin * taps, [&](int ci) {       // forward layout: one row block per cin
            for (int t = 0; t < taps; ++t)
                for (int co = 0; co < co_n; ++co)
                    wf[((size_t)ci * taps + t) * coutP + co0 + co] = w->data[((size_t)co * cin + ci) * taps + t];
        });
        for (int co = 0; co < co_n; ++co) bias[co0 + co] = bs[wi]->data[co];
        co0 += co_n;
    }
    out->cin = cin; out->cout = cout; out->taps = taps;
    if (upload(c, &out->wf, wf) || upload(c, &out->wd, wd) || upload(c, &out->bias, bias)) return -1;
    // the same two operators as split-bf16 records (read back from the fp32 layouts built above)
    auto fwd = [&](int o, int i, int t) { return wf[((size_t)i * taps + t) * coutP + o]; };
    auto dgr = [&](int o, int i, int t) { return wd[((size_t)i * taps + t) * cinP + o]; };   // o: cin index, i: cout index
    std::vector<float> rf = build_records(cin, cout, taps, fwd);
    std::vector<float> rd = build_records(cout, cin, taps, dgr);
    if (upload(c, &out->wbf, rf) || upload(c, &out->wbd, rd)) return -1;
    std::vector<float> hf = build_records_f16(cin, cout, taps, fwd);
    std::vector<float> hd = build_records_f16(cout, cin, taps, dgr);
    if (upload(c, &out->whf, hf) || upload(c, &out->whd, hd)) return -1;
    return 0;
}
int make_conv1(loco_ctx* c, const std::string& name, ConvP* out, int row_limit = -1) {
    return make_conv(c, {&c->params[name + ".weight"]}, {&c->params[name + ".bias"]}, out, row_limit);
}
// the same operator with weight and bias multiplied by `s` (the ResBlock output scale of cfg.res_scale lives in conv2 / the shortcut)
int make_conv1_scaled(loco_ctx* c, const std::string& name, ConvP* out, float s) {
    if (s == 1.f) return make_conv1(c, name, out);
    HostParam w = c->params[name + ".weight"], b = c->params[name + ".bias"];
    for (float& v : w.data) v *= s;
    for (float& v : b.data) v *= s;
    return make_conv(c, {&w}, {&b}, out);
}
int make_norm(loco_ctx* c, const std::string& name, NormP* n) {
    if (upload(c, &n->gamma, c->params[name + ".weight"].data)) return -1;
    if (upload(c, &n->beta, c->params[name + ".bias"].data)) return -1;
    return 0;
}

int finalize_params(loco_ctx* c) {
    if (c->finalized) return 0;
    for (auto& n : c->param_order)
        if (!c->params[n].loaded) { c->err = "parameter not loaded: " + n; return -1; }
    const loco_unet_cfg& cfg = c->cfg;
    int temb_ch = cfg.ch * 4;
    const bool adm = cfg.arch == 1;
    const std::string te0 = adm ? "time_embed.0" : "temb.dense.0", te1 = adm ? "time_embed.2" : "temb.dense.1";
    if (cfg.arch < 2) {
        if (upload(c, &c->td0w, c->params[te0 + ".weight"].data)) return -1;
        if (upload(c, &c->td0b, c->params[te0 + ".bias"].data)) return -1;
        if (upload(c, &c->td1w, c->params[te1 + ".weight"].data)) return -1;
        if (upload(c, &c->td1b, c->params[te1 + ".bias"].data)) return -1;
    }
    // sinusoid frequencies exactly as torch computes them in fp32:
    //   DDPM: exp(float32(i) * float32(-ln(1e4)/(half-1)))            (diffusion.py:797-798)
    //   ADM : exp(float32(-ln(1e4)) * float32(i) / float32(half))     (guided_diffusion/nn.py:113-115)
    {
        int half = cfg.ch / 2;
        std::vector<float> f(half);
        for (int i = 0; i < half; ++i) {
            float x;
            if (adm) x = ((float)(-std::log(10000.0)) * (float)i) / (float)half;
            else x = (float)i * (float)(-(std::log(10000.0) / (double)(half - 1)));
            f[i] = (float)std::exp((double)x);
        }
        if (upload(c, &c->freq, f)) return -1;
    }
    std::vector<float> tpw, tpb;
    for (auto& op : c->ops) {
        switch (op.kind) {
            case OP_CONV_IN: case OP_CONV: if (make_conv1(c, op.pn_conv, &op.conv)) return -1; break;
            case OP_RES: {
                if (make_norm(c, op.pn_n1, &op.n1) || make_norm(c, op.pn_n2, &op.n2)) return -1;
                if (make_conv1(c, op.pn_c1, &op.c1) || make_conv1_scaled(c, op.pn_c2, &op.c2, c->res_scale)) return -1;
                if (op.has_nin && make_conv1_scaled(c, op.pn_skip, &op.nin, c->res_scale)) return -1;
                if (!op.has_temb) break;
                op.tproj_off = (long)tpb.size();
                auto& w = c->params[op.pn_emb + ".weight"].data;
                auto& b = c->params[op.pn_emb + ".bias"].data;
                tpw.insert(tpw.end(), w.begin(), w.end());
                tpb.insert(tpb.end(), b.begin(), b.end());
                break;
            }
            case OP_ATTN: {
                if (make_norm(c, op.pn_n1, &op.n1)) return -1;
                if (adm) {
                    if (make_conv1(c, op.pn_qkv, &op.qkvc)) return -1;
                } else if (make_conv(c, {&c->params[op.name + ".q.weight"], &c->params[op.name + ".k.weight"],
                                         &c->params[op.name + ".v.weight"]},
                                     {&c->params[op.name + ".q.bias"], &c->params[op.name + ".k.bias"],
                                      &c->params[op.name + ".v.bias"]}, &op.qkvc)) return -1;
                if (make_conv1(c, op.pn_proj, &op.proj)) return -1;
                if (op.added_kv) {
                    // encoder_kv rows of head h are [k_h (CH) | v_h (CH)] (QKVAttention of deepfloyd_if: the states' projection is
                    // split per head like qkv): un-interleave into the key and the value map, rows ordered (head, channel)
                    const int C = c->tens[op.in].C, NH = op.heads, CH = C / NH, D = c->cfg.context_dim;
                    const HostParam& w = c->params[op.name + ".encoder_kv.weight"];
                    const HostParam& b = c->params[op.name + ".encoder_kv.bias"];
                    std::vector<float> kw((size_t)C * D), vw((size_t)C * D), kb(C), vb(C);
                    for (int h = 0; h < NH; ++h)
                        for (int i = 0; i < CH; ++i) {
                            const size_t rk = (size_t)h * 2 * CH + i, rv = rk + CH, r = (size_t)h * CH + i;
                            std::copy(w.data.begin() + rk * D, w.data.begin() + (rk + 1) * D, kw.begin() + r * D);
                            std::copy(w.data.begin() + rv * D, w.data.begin() + (rv + 1) * D, vw.begin() + r * D);
                            kb[r] = b.data[rk]; vb[r] = b.data[rv];
                        }
                    if (upload(c, &op.xkw, kw) || upload(c, &op.xkb, kb) || upload(c, &op.xvw, vw) || upload(c, &op.xvb, vb) ||
                        upload(c, &op.xng, c->params[op.name + ".norm_encoder.weight"].data) ||
                        upload(c, &op.xnb, c->params[op.name + ".norm_encoder.bias"].data)) return -1;
                    const size_t kv = (size_t)C * c->ctx_Lp;
                    if (dalloc(c, &op.xK, kv) || dalloc(c, &op.xV, kv)) return -1;
                }
                if (op.has_x) {
                    const std::string x = op.name + ".xattn";
                    if (make_norm(c, x + ".norm", &op.nx) || make_conv1(c, x + ".q", &op.xqc) ||
                        make_conv1(c, x + ".proj_out", &op.xproj)) return -1;
                    if (upload(c, &op.xkw, c->params[x + ".k.weight"].data) || upload(c, &op.xkb, c->params[x + ".k.bias"].data) ||
                        upload(c, &op.xvw, c->params[x + ".v.weight"].data) || upload(c, &op.xvb, c->params[x + ".v.bias"].data))
                        return -1;
                    const size_t kv = (size_t)c->tens[op.in].C * c->ctx_Lp;
                    if (dalloc(c, &op.xK, kv) || dalloc(c, &op.xV, kv)) return -1;
                }
                break;
            }
            case OP_DOWN: case OP_UP: if (make_conv1(c, op.pn_conv, &op.conv)) return -1; break;
            case OP_XFMR: {
                const int C = c->tens[op.in].C, NH = op.heads, CH = C / NH;
                const std::string b = op.name + ".transformer_blocks.0";
                HostParam zb; zb.shape = {8 * C}; zb.data.assign((size_t)8 * C, 0.f);       // bias of the bias-free maps
                auto nobias = [&](const std::string& w, ConvP* out) { return make_conv(c, {&c->params[w + ".weight"]}, {&zb}, out); };
                if (make_norm(c, op.pn_n1, &op.n1) || make_conv1(c, op.name + ".proj_in", &op.pj_in) ||
                    make_conv1(c, op.name + ".proj_out", &op.pj_out)) return -1;
                const char* lnn[3] = {".norm1", ".norm2", ".norm3"};
                for (int i = 0; i < 3; ++i)
                    if (upload(c, &op.lng[i], c->params[b + lnn[i] + ".weight"].data) ||
                        upload(c, &op.lnb[i], c->params[b + lnn[i] + ".bias"].data)) return -1;
                {   // attn1: to_q / to_k / to_v fused into one [3C][C] map whose rows follow the per-head [q_h | k_h | v_h] layout
                    // the attention products address (head h of 'b n (h d)' = channels h*d .. h*d+d-1)
                    HostParam w; w.shape = {3 * C, C}; w.data.resize((size_t)3 * C * C);
                    const char* nm[3] = {".attn1.to_q.weight", ".attn1.to_k.weight", ".attn1.to_v.weight"};
                    for (int which = 0; which < 3; ++which) {
                        const std::vector<float>& src = c->params[b + nm[which]].data;
                        for (int h = 0; h < NH; ++h)
                            for (int j = 0; j < CH; ++j)
                                std::memcpy(&w.data[((size_t)(h * 3 + which) * CH + j) * C], &src[(size_t)(h * CH + j) * C], (size_t)C * 4);
                    }
                    if (make_conv(c, {&w}, {&zb}, &op.qkvc)) return -1;
                }
                if (make_conv1(c, b + ".attn1.to_out.0", &op.to_out1) || nobias(b + ".attn2.to_q", &op.xqc) ||
                    make_conv1(c, b + ".attn2.to_out.0", &op.to_out2) || make_conv1(c, b + ".ff.net.0.proj", &op.ff1) ||
                    make_conv1(c, b + ".ff.net.2", &op.ff2)) return -1;
                std::vector<float> z0((size_t)C, 0.f);
                if (upload(c, &op.xkw, c->params[b + ".attn2.to_k.weight"].data) || upload(c, &op.xkb, z0) ||
                    upload(c, &op.xvw, c->params[b + ".attn2.to_v.weight"].data) || upload(c, &op.xvb, z0)) return -1;
                const size_t kv = (size_t)C * c->ctx_Lp;
                if (dalloc(c, &op.xK, kv) || dalloc(c, &op.xV, kv)) return -1;
                break;
            }
            case OP_OUT:
                if (make_norm(c, op.pn_n1, &op.n1) || make_conv1(c, op.pn_conv, &op.conv, cfg.out_ch)) return -1;
                break;
        }
    }
    c->tproj_total = (long)tpb.size();
    if (upload(c, &c->tp_w, tpw) || upload(c, &c->tp_b, tpb)) return -1;
    if (dalloc(c, &c->tact, (size_t)temb_ch) || dalloc(c, &c->tproj, (size_t)(c->tproj_total > 0 ? c->tproj_total : 1))) return -1;
    // FLOP model (2*MAC): convolutions + attention products
    double fl = 0.0;
    for (auto& op : c->ops) {
        auto convf = [&](const ConvP& p, int H, int W) { fl += 2.0 * p.cin * p.cout * p.taps * (double)H * W; };
        const Tens& to = c->tens[op.out];
        switch (op.kind) {
            case OP_CONV_IN: case OP_CONV: case OP_DOWN: case OP_UP: case OP_OUT: convf(op.conv, to.H, to.W); break;
            case OP_RES:
                convf(op.c1, to.H, to.W); convf(op.c2, to.H, to.W);
                if (op.has_nin) convf(op.nin, to.H, to.W);
                break;
            case OP_XFMR: {
                for (const ConvP* p : {&op.pj_in, &op.qkvc, &op.to_out1, &op.xqc, &op.to_out2, &op.ff1, &op.ff2, &op.pj_out}) convf(*p, to.H, to.W);
                double T = (double)to.H * to.W;
                fl += 2.0 * 2.0 * T * T * to.C + 2.0 * 2.0 * T * c->cfg.context_len * to.C;
                break;
            }
            case OP_ATTN: {
                convf(op.qkvc, to.H, to.W); convf(op.proj, to.H, to.W);
                double T = (double)to.H * to.W;
                fl += 2.0 * 2.0 * T * T * to.C;
                if (op.has_x) {
                    convf(op.xqc, to.H, to.W); convf(op.xproj, to.H, to.W);
                    fl += 2.0 * 2.0 * T * c->cfg.context_len * to.C;
                }
                break;
            }
        }
    }
    c->flops = fl;
    // host copies are no longer needed
    for (auto& kv : c->params) { std::vector<float>().swap(kv.second.data); }
    c->finalized = true;
    return 0;
}

// ---------------------------------------------------------------------------
struct Pass {
    loco_ctx* c;
    hipStream_t st;
    int B;
    float* arena;      // activation arena of this pass
    float* stats;      // stats arena of this pass
    float* T(int id) const { return arena + c->tens[id].off; }
    long bs() const { return c->per_sample; }
};

// attention products: exact fp32 MFMA in the f32 mode and for the short / small ones, split-bf16 on the bf16 matrix
// pipe for the long contractions of the low-precision modes (decoder mid attention at 4096 tokens)
// the captured forward graphs (LOCO_GRAPH=1) hold raw pointers: dropped whenever a buffer they may name is replaced
static void drop_graphs(loco_ctx* c) {
    for (auto& kv : c->fwd_graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    c->fwd_graphs.clear();
}
inline unsigned char* gemm_ws_get(loco_ctx* c, size_t need, hipStream_t st) {
    const int l = c->lane;
    if (c->gemm_ws_bytes[l] >= need) return c->gemm_ws[l];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    if (c->gemm_ws[l]) {
        // a captured forward graph (LOCO_GRAPH=1) may hold this pointer: the graphs go before the buffer does;
        // the context's byte count follows the buffer it replaces
        (void)hipDeviceSynchronize();
        drop_graphs(c);
        (void)hipFree(c->gemm_ws[l]);
        c->bytes -= c->gemm_ws_bytes[l];
        c->gemm_ws[l] = nullptr; c->gemm_ws_bytes[l] = 0;
    }
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->gemm_ws[l] = static_cast<unsigned char*>(p); c->gemm_ws_bytes[l] = need; c->bytes += need;
    return c->gemm_ws[l];
}
inline bool attn_gemm_rec(const loco_ctx* c, const GemmArgs& g) { return c->prec >= 1 && c->gemm_lowp && gemm_rec_eligible(g); }
inline void attn_gemm(loco_ctx* c, const GemmArgs& g, hipStream_t st) {
    if (attn_gemm_rec(c, g)) {        // both operands pre-split into records, LDS-DMA fed (gemm_rec.hip)
        if (unsigned char* ws = gemm_ws_get(c, gemm_rec_ws_bytes(g), st)) { launch_gemm_rec(g, ws, st); return; }
    }
    if (c->prec >= 1 && c->gemm_lowp && gemm_prefers_bf16x3(g)) launch_gemm_bf16x3(g, st); else launch_gemm(g, st);
}

void conv_defaults(ConvArgs& a) {
    std::memset(&a, 0, sizeof(a));
    a.stride = 1; a.pad = 1; a.nsplit = 1; a.mode = CM_NONE; a.res_scale = 1.f;
}

// sc / sh / mr / tst pointers of a norm inside a stats arena
struct NS { float *sc, *sh, *mr, *tst, *tc; };
NS nstats(const loco_ctx* c, float* base, const NormP& n) {
    NS s;
    s.sc = base + n.soff; s.sh = s.sc + n.C; s.mr = s.sh + n.C; s.tst = s.mr + 2 * c->cfg.gn_groups;
    s.tc = s.tst + 2 * c->cfg.gn_groups;
    return s;
}

// Statistics the NEXT consumer of a conv's output needs (GroupNorm forward statistics, or the tangent / cotangent group
// means), handed to run_conv with the conv that finishes the tensor: taken in the conv epilogue (whole cout tiles), in the
// split-K epilogue (reduce + statistics in one kernel), or -- where neither applies -- by the standalone kernels behind the
// conv.  Either way they are complete when run_conv returns, and the norm is marked `ready` for its consumer.
struct StatReq {
    int kind = ST_NONE;                       // ST_FWD / ST_TAN / ST_COT
    NormP* n = nullptr;
    float* stats = nullptr;                   // stats arena receiving the result (FWD: this pass's; TAN / COT: statsT)
    const float* prim = nullptr;              // primal of the conv's output tensor, B = 1 (TAN / COT)
    const float* ss_scale = nullptr; const float* ss_shift = nullptr;   // FWD: scale-shift norm (ADM)
    // FWD, output tensor is one part of a concatenation: its tile partials go to (and stay in) `keep`; *keep_ntile receives
    // the tile count, 0 when this launch could not take them.  n may be nullptr (no norm over the part alone).
    float* keep = nullptr; size_t keep_floats = 0; int* keep_ntile = nullptr;
};

void stats_standalone(loco_ctx* c, const StatReq& rq, const float* x, long xbs, int B, int HW, int s0, hipStream_t st) {
    const NormP& n = *rq.n;
    const long SB = c->stats_per_sample;
    const int G = c->cfg.gn_groups;
    if (rq.kind == ST_FWD) {
        NS s = nstats(c, rq.stats + (long)s0 * SB, n);
        launch_gn_stats(x, xbs, B, n.C, HW, G, eps_of(c, n), n.gamma, n.beta, s.mr, s.sc, s.sh, SB, c->red, st, rq.ss_scale,
                        rq.ss_shift);
    } else {
        NS sp = nstats(c, c->statsP, n);
        NS stt = nstats(c, rq.stats + (long)s0 * SB, n);
        launch_gn_tstats(x, xbs, rq.prim, 0, B, n.C, HW, G, sp.sc, sp.sh, sp.mr, 0, 0, rq.kind == ST_TAN ? 0 : 1, stt.tst,
                         stt.tc, SB, c->red, st, c->cfg.act);
    }
}

// run a conv with automatic split-K selection
// `second`: a prepared 1x1 operator on the same output tensor (the ResBlock shortcut, `second->out == a.out`) that `a` then
// reads as its residual.  Where the launch allows it is K-concatenated into `a`'s kernel (conv_lowp_kcat: one write-out, no
// read-modify-write of the block output, one launch less); otherwise it runs first and `a` takes its result as residual.
void run_conv(loco_ctx* c, ConvArgs& a, int taps, hipStream_t st, const StatReq* rq = nullptr, ConvArgs* second = nullptr) {
    a.act = c->cfg.act;
    if (a.mode == CM_GN_SILU && c->cfg.act == ACT_GELU) a.mode = CM_GN_GELU;     // the forward prologue of a GELU network
    if (second) {
        const size_t span = (size_t)c->cfg.max_batch * c->per_sample;
        ConvArgs t = a;
        t.taps = taps;
        t.nsplit = c->prec >= 1 ? conv_bf16_pick_nsplit(a.Cin, a.Cout, a.Hout, a.Wout, a.B, c->chip_share, 9, c->lanes_active) : 2;
        t.in_padded = (a.in >= c->arenaP && a.in < c->arenaP + span) || (a.in >= c->arenaT && a.in < c->arenaT + span);
        t.in2 = second->in; t.in2_bs = second->in_bs; t.Cin2 = second->Cin;
        const long per_probe = (long)((a.Hout * a.Wout) / 256) * ((a.Cout + 127) / 128), total = per_probe * a.B;
        const bool whole_rounds = total <= 256 || total % 256 == 0 || total % 256 > 160;      // no tail-probe split (below)
        if (c->prec >= 1 && taps == 9 && whole_rounds && !a.bias2 && conv_lowp_can_kcat(t)) {
            a.in2 = second->in; a.in2_bs = second->in_bs; a.Cin2 = second->Cin;
            a.in2_padded = (second->in >= c->arenaP && second->in < c->arenaP + span) ||
                           (second->in >= c->arenaT && second->in < c->arenaT + span);
            a.wb2 = c->prec == 2 ? second->wh : second->wb;
            a.bias2 = second->bias; a.bias2_bs = 0;          // the shortcut's bias (the forward pass; tangents carry none)
            a.res = nullptr;
        } else {
            run_conv(c, *second, 1, st);
            a.res = second->out; a.res_bs = second->out_bs;
        }
    }
    if (c->prec == 2) a.wb = a.wh;
    a.taps = taps;
    a.no_deep = c->deep1 ? 0 : 1;
    a.nsplit = c->prec >= 1 ? conv_bf16_pick_nsplit(a.Cin, a.Cout, a.Hout, a.Wout, a.B, c->chip_share, taps, c->lanes_active)
                            : conv_pick_nsplit(a.Cin, a.Cout, a.Hout, a.Wout, a.B, taps);
    while (a.nsplit > 1 && (size_t)a.nsplit * a.B * a.Cout * a.Hout * a.Wout > c->partial_floats) a.nsplit >>= 1;
    if (a.nsplit < 1) a.nsplit = 1;
    a.partial = c->partial;
    a.partial_floats = c->partial_floats;
    a.gemm = 0;
    // compute-shaped 1x1 operators (the transformer's linear layers): the DMA-fed GEMM with its own split-K choice
    if (c->prec == 1 && taps == 1) conv_gemm_plan(a);
    {
        const size_t span = (size_t)c->cfg.max_batch * c->per_sample;
        a.in_padded = (a.in >= c->arenaP && a.in < c->arenaP + span) || (a.in >= c->arenaT && a.in < c->arenaT + span);
    }
    // 3x3 launches with several probes per (pixel tile, cout tile) workgroup slot: one workgroup walks them (conv_pers_plan)
    a.pers_groups = 0;
    if (c->prec == 1 && taps == 9) conv_pers_plan(a);
    // Tail-probe split (bf16x3): when the workgroups of the launch fill whole rounds of the 256 CUs plus a short
    // tail made of the last probes (5 probes x 64 tiles = 320 = 256 + 64), those probes are launched separately
    // with split-K so the tail round is as wide as the chip: 1 + 1/s rounds + a one-sample reduce instead of 2
    // (measured 331.0 vs 334.1 ms per step).
    int tail_probes = 0, tail_split = 1;
    if (c->prec >= 1 && a.nsplit == 1 && a.B >= 2 && !a.gemm && !a.pers_groups) {
        const long per_probe = (long)((a.Hout * a.Wout) / conv_bf16_tile_pixels(a)) * ((a.Cout + 127) / 128);
        const long total = per_probe * a.B, r = total % 256;
        const int nchunks = (a.Cin + 15) / 16;
        if (total > 256 && r > 0 && r <= 160 && per_probe > 0 && r % per_probe == 0 && r / per_probe < a.B) {
            int s = (int)(256 / r);
            if (s > 4) s = 4;
            while (s > 1 && nchunks / s < 4) --s;
            const size_t one = (size_t)(r / per_probe) * a.Cout * a.Hout * a.Wout;
            while (s > 1 && (size_t)s * one > c->partial_floats) --s;
            if (s > 1) { tail_probes = (int)(r / per_probe); tail_split = s; }
        }
    }
    // one conv kernel (+ its split-K reduce), each timed as its own profile record so the per-kernel averages agree
    // with rocprofv3's
    const bool want = rq && rq->kind != ST_NONE && c->prec >= 1 && c->fuse_stats;
    const int HWo = a.Hout * a.Wout, Gn = c->cfg.gn_groups;
    const long SBs = c->stats_per_sample;
    // statistics of samples [s0, s0 + x.B) of the launch `x`, by the cheapest route that applies
    auto stats_of = [&](ConvArgs& x, int s0) -> int {      // 0: none asked; 1: in the conv epilogue; 2: in the split-K epilogue; 3: standalone; 4: partials only
        if (!rq || rq->kind == ST_NONE) return 0;
        if (rq->keep_ntile) *rq->keep_ntile = 0;
        if (!want) return rq->n ? 3 : 0;
        if (x.nsplit > 1) return !rq->n ? 0 : ((HWo % 4 == 0) ? 2 : 3);
        const int ntile = HWo / conv_bf16_tile_pixels(x);
        const size_t need = (size_t)x.B * x.Cout * ntile * 2;
        const bool kept = rq->keep && s0 == 0 && x.B == a.B && need <= rq->keep_floats;      // the whole batch in one launch
        if (rq->kind == ST_FWD && conv_lowp_can_fuse_stats(x) && (kept || (rq->n && need <= c->stpart_floats))) {
            x.st_part = kept ? rq->keep : c->stpart; x.st_kind = ST_FWD;
            if (kept) *rq->keep_ntile = ntile;
            return rq->n ? 1 : 4;
        }
        // Tangent / cotangent group means in the conv epilogue (round 6, LOCO_FUSE_LIN=0: off): whole cout tiles, no split-K, not the
        // opt-in persistent / dual-probe kernels (their epilogues take the forward statistics only).  A tangent launch that finishes
        // one part of a concatenation keeps its (norm-independent) raw sums in the part's buffer, at this lane's samples.
        const bool cot_cache = rq->kind == ST_COT && rq->n && rq->n->sx_off >= 0 && c->sxcache;      // (its {S, xhat} records exist)
        // (not next to a norm-cotangent term: the epilogue holds one of the two in its record registers, ConvArgs::cot_d)
        if ((rq->kind == ST_TAN || cot_cache) && c->fuse_lin && !x.cot_d && conv_lowp_can_fuse_stats(x) && !x.pers_groups && !conv_dual_ok(x)) {
            const size_t lane_off = (size_t)c->lane_s0 * x.Cout * ntile * 2;
            const bool keptl = rq->kind == ST_TAN && rq->keep && s0 == 0 && x.B == a.B && lane_off + need <= rq->keep_floats;
            if (keptl || (rq->n && need <= c->stpart_floats)) {
                x.st_part = keptl ? rq->keep + lane_off : c->stpart; x.st_kind = rq->kind; x.st_x = rq->prim;
                if (rq->kind == ST_COT) x.st_sx = c->sxcache + rq->n->sx_off;
                if (keptl) *rq->keep_ntile = ntile;
                return rq->n ? 1 : 4;
            }
        }
        return rq->n ? 3 : 0;
    };
    auto stats_after = [&](const ConvArgs& x, int s0, int how) {
        if (how == 0 || how == 2 || how == 4) return;
        const NormP& n = *rq->n;
        if (how == 3) { stats_standalone(c, *rq, x.out, x.out_bs, x.B, HWo, s0, st); return; }
        NS so = nstats(c, rq->stats + (long)s0 * SBs, n);
        if (rq->kind != ST_FWD) {
            NS sp = nstats(c, c->statsP, n);
            launch_gn_lin_fused_finalize(rq->kind, x.st_part, n.C, HWo / conv_bf16_tile_pixels(x), nullptr, 0, x.B, n.C, HWo, Gn,
                                         sp.mr, so.tst, so.tc, SBs, st);
            return;
        }
        launch_gn_fused_finalize(x.st_part, HWo / conv_bf16_tile_pixels(x), x.B, n.C, HWo, Gn, eps_of(c, n), n.gamma, n.beta,
                                 so.mr, so.sc, so.sh, SBs, rq->ss_scale, rq->ss_shift, st);
    };
    auto reduce_with_stats = [&](const ConvArgs& x, int s0) {
        const NormP& n = *rq->n;
        NS sp = nstats(c, c->statsP, n);
        NS so = nstats(c, rq->stats + (long)s0 * SBs, n);
        launch_conv_splitk_reduce_stats(x, rq->kind, Gn, eps_of(c, n), n.gamma, n.beta, so.mr, so.sc, so.sh, SBs, rq->ss_scale,
                                        rq->ss_shift, rq->prim, sp.sc, sp.sh, sp.mr, so.tst, so.tc, SBs, c->red, st);
    };
    auto one = [&](ConvArgs& x, int s0) {
        const int how = stats_of(x, s0);
        // the launch as the low-precision dispatcher runs it: its even part on the dual-probe tile + an odd last probe on the
        // 128 x 256 tile (conv_lowp_plan), each kernel with its own profile record
        ConvArgs parts[2];
        const int nparts = conv_lowp_plan(x, taps, c->prec, parts);
        for (int pi = 0; pi < nparts; ++pi) {
            ConvArgs& y = parts[pi];
            loco_ctx::ProfRec r;
            if (c->prof_on) {
                r.name = conv_variant_name(y, taps, c->prec);
                r.flops = 2.0 * (y.Cin * taps + y.Cin2) * y.Cout * (double)y.Hout * y.Wout * y.B;
                if (y.zins) r.flops *= 0.25;     // algorithmic work of the stride-2 data gradient
                r.cin = y.Cin; r.cout = y.Cout; r.h = y.Hout; r.b = y.B; r.ns = y.nsplit; r.mode = y.mode; r.taps = taps;
                r.e0 = c->next_event(); r.e1 = c->next_event();
                (void)hipEventRecord(r.e0, st);
            }
            if (c->prec == 1) launch_conv_bf16x3(y, taps, st);
            else if (c->prec == 2) launch_conv_f16(y, taps, st);
            else launch_conv(y, taps, st);
            if (c->prof_on) { (void)hipEventRecord(r.e1, st); c->prof.push_back(r); }
        }
        if (x.nsplit > 1) {
            loco_ctx::ProfRec rr;
            if (c->prof_on) {
                rr.name = "conv_splitk_reduce"; rr.flops = 0.0;
                rr.cin = x.Cin; rr.cout = x.Cout; rr.h = x.Hout; rr.b = x.B; rr.ns = x.nsplit; rr.mode = x.mode; rr.taps = taps;
                rr.e0 = c->next_event(); rr.e1 = c->next_event();
                (void)hipEventRecord(rr.e0, st);
            }
            if (how == 2) reduce_with_stats(x, s0); else launch_conv_splitk_reduce(x, st);
            if (c->prof_on) { (void)hipEventRecord(rr.e1, st); c->prof.push_back(rr); }
        }
        stats_after(x, s0, how);
    };
    if (rq && rq->kind != ST_NONE && rq->n) rq->n->ready = true;
    if (!tail_probes) { one(a, 0); return; }
    // tail-probe split: two launches finish the tensor.  Tangent / cotangent statistics are taken once over the whole batch
    // behind them (separate statistics for the few tail probes would add a reduce-with-statistics and a finalize launch per
    // conv: LOCO_FUSE_LIN=0).  FORWARD statistics are per sample and come for free with both launches: the main launch's epilogue
    // partials (+ one merge launch for its samples), the tail's split-K epilogue -- no pass over the finished tensor.
    // (round 6: the tangent / cotangent means likewise -- they are per probe too: the main launch's epilogue partials + one merge
    // launch, the tail's split-K epilogue)
    const bool per_part = want && rq && rq->n && (rq->kind == ST_FWD || (c->fuse_lin && HWo % 4 == 0));
    const StatReq* rq_all = (!per_part && rq && rq->kind != ST_NONE && rq->n) ? rq : nullptr;
    if (rq && rq->keep_ntile) *rq->keep_ntile = 0;
    if (!per_part) rq = nullptr;
    ConvArgs m = a, t = a;
    const int nb = a.B - tail_probes;
    m.B = nb;
    t.B = tail_probes; t.nsplit = tail_split;
    t.in += (long)nb * a.in_bs; t.out += (long)nb * a.out_bs;
    if (t.prim) t.prim += (long)nb * a.prim_bs;
    if (t.bias2) t.bias2 += (long)nb * a.bias2_bs;
    if (t.res) t.res += (long)nb * a.res_bs;
    if (t.sc) { t.sc += (long)nb * a.scsh_bs; t.sh += (long)nb * a.scsh_bs; }
    if (t.mr) t.mr += (long)nb * a.mr_bs;
    if (t.tst) t.tst += (long)nb * a.tst_bs;
    if (t.tc) t.tc += (long)nb * a.tc_bs;
    one(m, 0);
    one(t, nb);
    if (rq_all) stats_standalone(c, *rq_all, a.out, a.out_bs, a.B, HWo, 0, st);
}

inline void setw(ConvArgs& a, const ConvP& p, bool dgrad) {
    a.w = dgrad ? p.wd : p.wf;
    a.wb = dgrad ? (const void*)p.wbd : (const void*)p.wbf;
    a.wh = dgrad ? (const void*)p.whd : (const void*)p.whf;
}

// the norm (if any) that takes its statistics over exactly tensor `tid`
NormP* consumer_norm(loco_ctx* c, int tid) {
    const Tens& t = c->tens[tid];
    if (t.cons_op < 0) return nullptr;
    Op& op = c->ops[t.cons_op];
    return t.cons_norm == 2 ? &op.nx : &op.n1;
}
StatReq req_fwd(NormP* n, float* stats, const float* ss_scale = nullptr, const float* ss_shift = nullptr) {
    StatReq r;
    if (n) { r.kind = ST_FWD; r.n = n; r.stats = stats; r.ss_scale = ss_scale; r.ss_shift = ss_shift; }
    return r;
}
StatReq req_lin(loco_ctx* c, int kind, NormP* n, const float* prim) {       // tangent / cotangent statistics into statsT
    StatReq r;
    if (n) { r.kind = kind; r.n = n; r.stats = c->statsT; r.prim = prim; }
    return r;
}
void clear_ready(loco_ctx* c) {
    for (auto& op : c->ops) op.n1.ready = op.n2.ready = op.nx.ready = false;
    for (auto& t : c->tens) t.keep_ntile = 0;
}
// forward statistics for the consumer of tensor `tid` and / or, when `tid` is one part of a concatenation, its tile partials
StatReq req_fwd_of(loco_ctx* c, int tid, float* stats) {
    StatReq r = req_fwd(consumer_norm(c, tid), stats);
    Tens& t = c->tens[tid];
    if (t.cat_of >= 0 && t.keep) {
        r.kind = ST_FWD; r.stats = stats;
        r.keep = t.keep; r.keep_floats = t.keep_floats; r.keep_ntile = &t.keep_ntile;
    }
    return r;
}
// norm over a concatenation whose two parts both kept their producers' tile partials in this pass: finalise from those
bool cat_fused_stats(loco_ctx* c, const NormP& n, int tid, float* stats, int B, hipStream_t st) {
    const Tens& q = c->tens[tid];
    static int on = -1;                      // LOCO_FUSE_CAT=0: A/B switch (the parts' partials are still taken, not used)
    if (on < 0) { const char* e = getenv("LOCO_FUSE_CAT"); on = e ? (atoi(e) != 0) : 1; }
    if (q.cat_a < 0 || c->prec < 1 || !c->fuse_stats || !on) return false;
    const Tens& A = c->tens[q.cat_a];
    const Tens& Bt = c->tens[q.cat_b];
    if (A.keep_ntile <= 0 || Bt.keep_ntile <= 0) return false;
    NS s = nstats(c, stats, n);
    launch_gn_fused_finalize_cat(A.keep, A.C, A.keep_ntile, Bt.keep, Bt.keep_ntile, B, q.C, q.H * q.W, c->cfg.gn_groups,
                                 eps_of(c, n), n.gamma, n.beta, s.mr, s.sc, s.sh, c->stats_per_sample, st);
    return true;
}

// the tangent group means of a norm over a concatenation whose two producers both kept their raw {sum d, sum x d} tile partials
// in this pass (run_conv, ST_TAN): finalised from those instead of a pass over the concatenation (round 6)
bool cat_fused_tstats(loco_ctx* c, const NormP& n, int tid, int B, hipStream_t st) {
    const Tens& q = c->tens[tid];
    if (q.cat_a < 0 || c->prec < 1 || !c->fuse_stats || !c->fuse_lin) return false;
    const Tens& A = c->tens[q.cat_a];
    const Tens& Bt = c->tens[q.cat_b];
    if (A.keep_ntile <= 0 || Bt.keep_ntile <= 0) return false;
    NS sp = nstats(c, c->statsP, n);
    NS stt = nstats(c, c->statsT, n);
    launch_gn_lin_fused_finalize(ST_TAN, A.keep + (size_t)c->lane_s0 * A.C * A.keep_ntile * 2, A.C, A.keep_ntile,
                                 Bt.keep + (size_t)c->lane_s0 * Bt.C * Bt.keep_ntile * 2, Bt.keep_ntile, B, q.C, q.H * q.W,
                                 c->cfg.gn_groups, sp.mr, stt.tst, stt.tc, c->stats_per_sample, st);
    return true;
}

void gn_forward_stats(const Pass& p, const NormP& n, const float* x, long xbs, int HW) {
    NS s = nstats(p.c, p.stats, n);
    launch_gn_stats(x, xbs, p.B, n.C, HW, p.c->cfg.gn_groups, eps_of(p.c, n), n.gamma, n.beta, s.mr, s.sc, s.sh,
                    p.c->stats_per_sample, p.c->red, p.st);
}


// ------------------------------ text cross-attention stage of an attention block ------------------------------
// out = xmid + proj(o),  o = V_ctx P^T,  P = softmax_l(scale q^T K_ctx),  q = Wq GN(xmid); K_ctx / V_ctx [C][Lp] are the
// projected encoder states (loco_set_context), constant with respect to the image, so the tangent / cotangent only
// travel through q.  Generic GEMM + row kernels: the score rows are only Lp (<= 128) long.
struct XA {                     // tensors of one pass: `a` = the arena holding this pass's values, `ap` = the primal arena
    loco_ctx* c; const Op* op; hipStream_t st; int B; int C, T, NH, CH, Lp;
    float scale;
};
XA xa_of(loco_ctx* c, const Op& op, int B, hipStream_t st) {
    XA x; x.c = c; x.op = &op; x.st = st; x.B = B;
    const Tens& t = c->tens[op.in];
    x.C = t.C; x.T = t.H * t.W; x.NH = op.heads; x.CH = x.C / x.NH; x.Lp = c->ctx_Lp;
    x.scale = 1.0f / std::sqrt((float)x.CH);
    return x;
}
// S[b][h][i][l] = alpha * sum_c X[b][h*CH+c][i] K[h*CH+c][l]      (X: q / dq / g_o, K: K_ctx or V_ctx)
void xa_scores(const XA& x, const float* X, long x_bs, const float* K, float* S, long s_bs, float alpha, bool mask) {
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = X; g.sam = 1; g.sak = x.T; g.sab = x_bs; g.sah = (long)x.CH * x.T;
    g.Bm = K; g.sbk = x.Lp; g.sbn = 1; g.sbb = 0; g.sbh = (long)x.CH * x.Lp;
    g.C = S; g.scm = x.Lp; g.scn = 1; g.scb = s_bs; g.sch = (long)x.T * x.Lp;
    g.M = x.T; g.N = x.Lp; g.K = x.CH; g.batch = x.B; g.batch2 = x.NH; g.alpha = alpha; g.beta = 0.f;
    g.colbias = mask ? x.c->ctx_colbias : nullptr;
    launch_gemm(g, x.st);
}
// O[b][h*CH+c][i] = sum_l K[h*CH+c][l] S[b][h][i][l]              (K: V_ctx or K_ctx, S: P / dP / g_S)
void xa_values(const XA& x, const float* K, const float* S, long s_bs, float* O, long o_bs) {
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = K; g.sam = x.Lp; g.sak = 1; g.sab = 0; g.sah = (long)x.CH * x.Lp;
    g.Bm = S; g.sbk = 1; g.sbn = x.Lp; g.sbb = s_bs; g.sbh = (long)x.T * x.Lp;
    g.C = O; g.scm = x.T; g.scn = 1; g.scb = o_bs; g.sch = (long)x.CH * x.T;
    g.M = x.CH; g.N = x.T; g.K = x.Lp; g.batch = x.B; g.batch2 = x.NH; g.alpha = 1.f; g.beta = 0.f;
    launch_gemm(g, x.st);
}
// the three steps of one pass (scores, row operation, values) as one kernel where the shape allows (LOCO_FUSE_XATTN=0: off)
//   kind 0 forward: P = softmax(scale X^T K1) -> Pout (per sample), O = K2 P^T
//   kind 1 tangent / cotangent: R = scale P o (X^T K1 - <P, X^T K1>) with the primal P, O = K2 R^T
bool xa_fused(const XA& x, int kind, const float* X, long x_bs, const float* K1, const float* K2, float* P, long p_bs,
              float* O, long o_bs) {
    loco_ctx* c = x.c;
    if (!c->fuse_xattn || !xattn_fused_supported(x.T, x.CH, c->cfg.context_len, x.Lp)) return false;
    XAttnArgs a; std::memset(&a, 0, sizeof(a));
    a.T = x.T; a.NH = x.NH; a.B = x.B; a.CH = x.CH; a.L = c->cfg.context_len; a.Lp = x.Lp; a.fwd = kind == 0;
    a.alpha = x.scale; a.X = X; a.x_bs = x_bs; a.K1 = K1; a.K2 = K2; a.P = P; a.p_bs = p_bs; a.O = O; a.o_bs = o_bs;
    launch_xattn_fused(a, x.st);
    return true;
}
void xa_conv1x1(loco_ctx* c, const ConvP& w, bool dgrad, const float* in, long in_bs, float* out, long out_bs, int C, int H,
                int W, int B, const float* res, long res_bs, bool with_bias, hipStream_t st, const StatReq* rq = nullptr) {
    ConvArgs a; conv_defaults(a);
    a.in = in; a.in_bs = in_bs; a.Cin = C; a.Hin = H; a.Win = W;
    setw(a, w, dgrad); a.pad = 0;
    if (with_bias) a.bias = w.bias;
    a.res = res; a.res_bs = res_bs;
    a.out = out; a.out_bs = out_bs; a.Cout = C; a.Hout = H; a.Wout = W; a.B = B;
    run_conv(c, a, 1, st, rq);
}

// ------------------------------ self-attention core of a SpatialTransformer block ------------------------------
// Between the fused q/k/v map (per-head rows [q_h | k_h | v_h] of `op.qkv`) and the attended values `op.o`; the same
// strided products and row kernels as the AttentionBlock (and its flash tangent / cotangent where the head width allows).
struct SA { loco_ctx* c; const Op* op; int B, T, NH, CH; hipStream_t st; };
void sa_forward(const SA& s, float* ar) {
    loco_ctx* c = s.c;
    const long PS = c->per_sample, HS = 3L * s.CH * s.T, SS = (long)s.T * s.T;
    float* q = ar + c->tens[s.op->qkv].off; float* k = q + (long)s.CH * s.T; float* v = k + (long)s.CH * s.T;
    float* S = ar + c->tens[s.op->S].off; float* o = ar + c->tens[s.op->o].off;
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = q; g.sam = 1; g.sak = s.T; g.sab = PS; g.sah = HS;
    g.Bm = k; g.sbk = s.T; g.sbn = 1; g.sbb = PS; g.sbh = HS;
    g.C = S; g.scm = s.T; g.scn = 1; g.scb = PS; g.sch = SS;
    g.M = s.T; g.N = s.T; g.K = s.CH; g.batch = s.B; g.batch2 = s.NH; g.alpha = 1.0f / std::sqrt((float)s.CH); g.beta = 0.f;
    attn_gemm(c, g, s.st);
    launch_softmax_rows(S, (long)s.NH * s.T, s.T, s.st, s.B, PS);
    GemmArgs h; std::memset(&h, 0, sizeof(h));
    h.A = v; h.sam = s.T; h.sak = 1; h.sab = PS; h.sah = HS;
    h.Bm = S; h.sbk = 1; h.sbn = s.T; h.sbb = PS; h.sbh = SS;
    h.C = o; h.scm = s.T; h.scn = 1; h.scb = PS; h.sch = (long)s.CH * s.T;
    h.M = s.CH; h.N = s.T; h.K = s.T; h.batch = s.B; h.batch2 = s.NH; h.alpha = 1.f; h.beta = 0.f;
    attn_gemm(c, h, s.st);
}
void sa_tangent(const SA& s) {       // dq, dk, dv in arenaT(qkv) -> do in arenaT(o); primal in arenaP (B = 1)
    loco_ctx* c = s.c;
    const int T = s.T, CH = s.CH, NH = s.NH, B = s.B;
    const long PS = c->per_sample, HS = 3L * CH * T, SS = (long)T * T;
    float* q = c->arenaP + c->tens[s.op->qkv].off; float* k = q + (long)CH * T; float* v = k + (long)CH * T;
    float* dq = c->arenaT + c->tens[s.op->qkv].off; float* dk = dq + (long)CH * T; float* dv = dk + (long)CH * T;
    float* SP = c->arenaP + c->tens[s.op->S].off; float* ST = c->arenaT + c->tens[s.op->S].off;
    float* oP = c->arenaP + c->tens[s.op->o].off; float* oT = c->arenaT + c->tens[s.op->o].off;
    const float scale = 1.0f / std::sqrt((float)CH);
    if (c->flash_attn && c->prec >= 1 && attn_flash_supported(T, CH)) {
        AttnFlashArgs fa; std::memset(&fa, 0, sizeof(fa));
        fa.ws = reinterpret_cast<unsigned char*>(c->partial); fa.ws_bytes = c->partial_floats * sizeof(float);      // (records of the operands: attn_flash.hip)
        fa.T = T; fa.NH = NH; fa.B = B; fa.CH = CH; fa.scale = scale; fa.q = q; fa.k = k; fa.v = v; fa.hs = HS; fa.P = SP; fa.o = oP;
        fa.dq = dq; fa.dk = dk; fa.dv = dv; fa.bs_d = PS; fa.out = oT; fa.bs_out = PS;
        launch_attn_flash_tangent(fa, s.st);
        return;
    }
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = dq; g.sam = 1; g.sak = T; g.sab = PS; g.sah = HS;
    g.Bm = k; g.sbk = T; g.sbn = 1; g.sbb = 0; g.sbh = HS;
    g.C = ST; g.scm = T; g.scn = 1; g.scb = PS; g.sch = SS;
    g.M = T; g.N = T; g.K = CH; g.batch = B; g.batch2 = NH; g.alpha = 1.f; g.beta = 0.f;
    attn_gemm(c, g, s.st);
    g.A = q; g.sab = 0; g.Bm = dk; g.sbb = PS; g.beta = 1.f;
    attn_gemm(c, g, s.st);
    launch_softmax_jac(ST, SP, (long)NH * T, T, (long)NH * T, scale, s.st, B, PS);
    GemmArgs h; std::memset(&h, 0, sizeof(h));
    h.A = dv; h.sam = T; h.sak = 1; h.sab = PS; h.sah = HS;
    h.Bm = SP; h.sbk = 1; h.sbn = T; h.sbb = 0; h.sbh = SS;
    h.C = oT; h.scm = T; h.scn = 1; h.scb = PS; h.sch = (long)CH * T;
    h.M = CH; h.N = T; h.K = T; h.batch = B; h.batch2 = NH; h.alpha = 1.f; h.beta = 0.f;
    attn_gemm(c, h, s.st);
    h.A = v; h.sab = 0; h.Bm = ST; h.sbb = PS; h.beta = 1.f;
    attn_gemm(c, h, s.st);
}
void sa_cotangent(const SA& s) {     // g_o in arenaT(o) -> g_q, g_k, g_v in arenaT(qkv)
    loco_ctx* c = s.c;
    const int T = s.T, CH = s.CH, NH = s.NH, B = s.B;
    const long PS = c->per_sample, HS = 3L * CH * T, SS = (long)T * T, OS = (long)CH * T;
    float* q = c->arenaP + c->tens[s.op->qkv].off; float* k = q + (long)CH * T; float* v = k + (long)CH * T;
    float* gq = c->arenaT + c->tens[s.op->qkv].off; float* gk = gq + (long)CH * T; float* gv = gk + (long)CH * T;
    float* SP = c->arenaP + c->tens[s.op->S].off; float* SG = c->arenaT + c->tens[s.op->S].off;
    float* oP = c->arenaP + c->tens[s.op->o].off; float* oG = c->arenaT + c->tens[s.op->o].off;
    const float scale = 1.0f / std::sqrt((float)CH);
    if (c->flash_attn && c->prec >= 1 && attn_flash_supported(T, CH)) {
        AttnFlashArgs fa; std::memset(&fa, 0, sizeof(fa));
        fa.ws = reinterpret_cast<unsigned char*>(c->partial); fa.ws_bytes = c->partial_floats * sizeof(float);      // (records of the operands: attn_flash.hip)
        fa.T = T; fa.NH = NH; fa.B = B; fa.CH = CH; fa.scale = scale; fa.q = q; fa.k = k; fa.v = v; fa.hs = HS; fa.P = SP; fa.o = oP;
        fa.go = oG; fa.bs_go = PS; fa.gq = gq; fa.gk = gk; fa.gv = gv; fa.bs_g = PS; fa.delta = c->attn_delta;
        launch_attn_flash_cotangent(fa, s.st);
        return;
    }
    GemmArgs g; std::memset(&g, 0, sizeof(g));                      // g_v[c][j] = sum_i g_o[c][i] P[i][j]
    g.A = oG; g.sam = T; g.sak = 1; g.sab = PS; g.sah = OS;
    g.Bm = SP; g.sbk = T; g.sbn = 1; g.sbb = 0; g.sbh = SS;
    g.C = gv; g.scm = T; g.scn = 1; g.scb = PS; g.sch = HS;
    g.M = CH; g.N = T; g.K = T; g.batch = B; g.batch2 = NH; g.alpha = 1.f; g.beta = 0.f;
    attn_gemm(c, g, s.st);
    GemmArgs h; std::memset(&h, 0, sizeof(h));                      // g_P[i][j] = sum_c g_o[c][i] v[c][j]
    h.A = oG; h.sam = 1; h.sak = T; h.sab = PS; h.sah = OS;
    h.Bm = v; h.sbk = T; h.sbn = 1; h.sbb = 0; h.sbh = HS;
    h.C = SG; h.scm = T; h.scn = 1; h.scb = PS; h.sch = SS;
    h.M = T; h.N = T; h.K = CH; h.batch = B; h.batch2 = NH; h.alpha = 1.f; h.beta = 0.f;
    attn_gemm(c, h, s.st);
    launch_softmax_jac(SG, SP, (long)NH * T, T, (long)NH * T, scale, s.st, B, PS);
    GemmArgs a; std::memset(&a, 0, sizeof(a));                      // g_q[c][i] = sum_j k[c][j] g_S[i][j]
    a.A = k; a.sam = T; a.sak = 1; a.sab = 0; a.sah = HS;
    a.Bm = SG; a.sbk = 1; a.sbn = T; a.sbb = PS; a.sbh = SS;
    a.C = gq; a.scm = T; a.scn = 1; a.scb = PS; a.sch = HS;
    a.M = CH; a.N = T; a.K = T; a.batch = B; a.batch2 = NH; a.alpha = 1.f; a.beta = 0.f;
    attn_gemm(c, a, s.st);
    GemmArgs b = a;                                                 // g_k[c][j] = sum_i q[c][i] g_S[i][j]
    b.A = q; b.Bm = SG; b.sbk = T; b.sbn = 1; b.C = gk;
    attn_gemm(c, b, s.st);
}

// ------------------------------ attention over [text ; image] keys (cfg.added_kv) ------------------------------
// The DeepFloyd-IF AttentionBlock (diffusers AttnAddedKVProcessor): P = softmax_j(scale q^T [K_text | k]) over Lp + T
// columns (the Lp - L padding columns carry -1e30), o = [V_text | v] P^T.  K_text / V_text [C][Lp] are constants of the
// prompt (loco_set_context), so tangents / cotangents reach them through q only.  Strided products on column ranges of
// one score matrix S [NH][T][Lp + T] per sample; q / k / v of head h start HS floats apart inside `op.qkv`.
struct AKV {
    loco_ctx* c; const Op* op; int B, T, NH, CH, Lp, Tk; long HS, SS, KS; float scale; hipStream_t st;
};
AKV akv_of(loco_ctx* c, const Op& op, int B, hipStream_t st) {
    AKV s; s.c = c; s.op = &op; s.B = B; s.st = st;
    const Tens& t = c->tens[op.in];
    s.T = t.H * t.W; s.NH = op.heads; s.CH = t.C / op.heads; s.Lp = c->ctx_Lp; s.Tk = s.Lp + s.T;
    s.HS = 3L * s.CH * s.T; s.SS = (long)s.T * s.Tk; s.KS = (long)s.CH * s.Lp;
    s.scale = 1.0f / std::sqrt((float)s.CH);
    return s;
}
// S[b][h][i][col0 + j] (+)= alpha * sum_c X[b][h][c][i] K[.][h][c][j]     X: q-like [CH][T] per head (head stride xh)
void akv_scores(const AKV& s, const float* X, long xb, long xh, const float* K, long kb, long kh, int kcols, float* S, long sb,
                int col0, int N, float alpha, float beta, bool mask, bool exact) {
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = X; g.sam = 1; g.sak = s.T; g.sab = xb; g.sah = xh;
    g.Bm = K; g.sbk = kcols; g.sbn = 1; g.sbb = kb; g.sbh = kh;
    g.C = S + col0; g.scm = s.Tk; g.scn = 1; g.scb = sb; g.sch = s.SS;
    g.M = s.T; g.N = N; g.K = s.CH; g.batch = s.B; g.batch2 = s.NH; g.alpha = alpha; g.beta = beta;
    g.colbias = mask ? s.c->ctx_colbias : nullptr;
    if (exact) launch_gemm(g, s.st); else attn_gemm(s.c, g, s.st);
}
// O[b][h][c][i] (+)= sum_j V[.][h][c][j] S[b][h][i][col0 + j]
void akv_values(const AKV& s, const float* V, long vb, long vh, int vcols, const float* S, long sb, int col0, int K, float* O,
                long ob, long oh, float beta, bool exact) {
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = V; g.sam = vcols; g.sak = 1; g.sab = vb; g.sah = vh;
    g.Bm = S + col0; g.sbk = 1; g.sbn = s.Tk; g.sbb = sb; g.sbh = s.SS;
    g.C = O; g.scm = s.T; g.scn = 1; g.scb = ob; g.sch = oh;
    g.M = s.CH; g.N = s.T; g.K = K; g.batch = s.B; g.batch2 = s.NH; g.alpha = 1.f; g.beta = beta;
    if (exact) launch_gemm(g, s.st); else attn_gemm(s.c, g, s.st);
}
// G[b][h][c][j] = sum_i X[.][h][c][i] S[b][h][i][col0 + j]            (g_v = g_o P, g_k = q g_S)
void akv_keys(const AKV& s, const float* X, long xb, long xh, const float* S, long sb, int col0, float* G, long gb, long gh) {
    GemmArgs g; std::memset(&g, 0, sizeof(g));
    g.A = X; g.sam = s.T; g.sak = 1; g.sab = xb; g.sah = xh;
    g.Bm = S + col0; g.sbk = s.Tk; g.sbn = 1; g.sbb = sb; g.sbh = s.SS;
    g.C = G; g.scm = s.T; g.scn = 1; g.scb = gb; g.sch = gh;
    g.M = s.CH; g.N = s.T; g.K = s.T; g.batch = s.B; g.batch2 = s.NH; g.alpha = 1.f; g.beta = 0.f;
    attn_gemm(s.c, g, s.st);
}
void akv_forward(const AKV& s, float* ar, long bs) {           // q, k, v in ar(qkv) -> P in ar(S), o in ar(o)
    loco_ctx* c = s.c; const Op& op = *s.op;
    float* q = ar + c->tens[op.qkv].off; float* k = q + (long)s.CH * s.T; float* v = k + (long)s.CH * s.T;
    float* S = ar + c->tens[op.S].off; float* o = ar + c->tens[op.o].off;
    akv_scores(s, q, bs, s.HS, op.xK, 0, s.KS, s.Lp, S, bs, 0, s.Lp, s.scale, 0.f, true, true);
    akv_scores(s, q, bs, s.HS, k, bs, s.HS, s.T, S, bs, s.Lp, s.T, s.scale, 0.f, false, false);
    launch_softmax_rows(S, (long)s.NH * s.T, s.Tk, s.st, s.B, bs);
    akv_values(s, v, bs, s.HS, s.T, S, bs, s.Lp, s.T, o, bs, (long)s.CH * s.T, 0.f, false);
    akv_values(s, op.xV, 0, s.KS, s.Lp, S, bs, 0, s.Lp, o, bs, (long)s.CH * s.T, 1.f, true);
}
void akv_tangent(const AKV& s) {        // dq, dk, dv in arenaT(qkv) -> do in arenaT(o); primal in arenaP (B = 1)
    loco_ctx* c = s.c; const Op& op = *s.op;
    const long PS = c->per_sample, OS = (long)s.CH * s.T;
    float* q = c->arenaP + c->tens[op.qkv].off; float* k = q + OS; float* v = k + OS;
    float* dq = c->arenaT + c->tens[op.qkv].off; float* dk = dq + OS; float* dv = dk + OS;
    float* SP = c->arenaP + c->tens[op.S].off; float* ST = c->arenaT + c->tens[op.S].off;
    float* oT = c->arenaT + c->tens[op.o].off;
    if (c->flash_attn && c->prec >= 1 && attn_flash_text_supported(s.T, s.CH, s.Lp)) {      // no [T x (Lp + T)] tangent (attn_flash.hip TXT)
        AttnFlashArgs fa; std::memset(&fa, 0, sizeof(fa));
        fa.ws = reinterpret_cast<unsigned char*>(c->partial); fa.ws_bytes = c->partial_floats * sizeof(float);      // (records of the operands: attn_flash.hip)
        fa.T = s.T; fa.NH = s.NH; fa.B = s.B; fa.CH = s.CH; fa.scale = s.scale; fa.q = q; fa.k = k; fa.v = v; fa.hs = s.HS;
        fa.P = SP; fa.o = c->arenaP + c->tens[op.o].off; fa.Lt = s.Lp; fa.kt = op.xK; fa.vt = op.xV;
        fa.dq = dq; fa.dk = dk; fa.dv = dv; fa.bs_d = PS; fa.out = oT; fa.bs_out = PS;
        launch_attn_flash_tangent(fa, s.st);
        return;
    }
    akv_scores(s, dq, PS, s.HS, op.xK, 0, s.KS, s.Lp, ST, PS, 0, s.Lp, 1.f, 0.f, false, true);       // dS_text = dq^T K_text
    akv_scores(s, dq, PS, s.HS, k, 0, s.HS, s.T, ST, PS, s.Lp, s.T, 1.f, 0.f, false, false);         // dS_img = dq^T k
    akv_scores(s, q, 0, s.HS, dk, PS, s.HS, s.T, ST, PS, s.Lp, s.T, 1.f, 1.f, false, false);         //        + q^T dk
    launch_softmax_jac(ST, SP, (long)s.NH * s.T, s.Tk, (long)s.NH * s.T, s.scale, s.st, s.B, PS);
    akv_values(s, dv, PS, s.HS, s.T, SP, 0, s.Lp, s.T, oT, PS, OS, 0.f, false);                      // do = dv P_img^T
    akv_values(s, v, 0, s.HS, s.T, ST, PS, s.Lp, s.T, oT, PS, OS, 1.f, false);                       //    + v dP_img^T
    akv_values(s, op.xV, 0, s.KS, s.Lp, ST, PS, 0, s.Lp, oT, PS, OS, 1.f, true);                     //    + V_text dP_text^T
}
void akv_cotangent(const AKV& s) {      // g_o in arenaT(o) -> g_q, g_k, g_v in arenaT(qkv)
    loco_ctx* c = s.c; const Op& op = *s.op;
    const long PS = c->per_sample, OS = (long)s.CH * s.T;
    float* q = c->arenaP + c->tens[op.qkv].off; float* k = q + OS; float* v = k + OS;
    float* gq = c->arenaT + c->tens[op.qkv].off; float* gk = gq + OS; float* gv = gk + OS;
    float* SP = c->arenaP + c->tens[op.S].off; float* SG = c->arenaT + c->tens[op.S].off;
    float* oG = c->arenaT + c->tens[op.o].off;
    if (c->flash_attn && c->prec >= 1 && attn_flash_text_supported(s.T, s.CH, s.Lp)) {
        AttnFlashArgs fa; std::memset(&fa, 0, sizeof(fa));
        fa.ws = reinterpret_cast<unsigned char*>(c->partial); fa.ws_bytes = c->partial_floats * sizeof(float);      // (records of the operands: attn_flash.hip)
        fa.T = s.T; fa.NH = s.NH; fa.B = s.B; fa.CH = s.CH; fa.scale = s.scale; fa.q = q; fa.k = k; fa.v = v; fa.hs = s.HS;
        fa.P = SP; fa.o = c->arenaP + c->tens[op.o].off; fa.Lt = s.Lp; fa.kt = op.xK; fa.vt = op.xV;
        fa.go = oG; fa.bs_go = PS; fa.gq = gq; fa.gk = gk; fa.gv = gv; fa.bs_g = PS; fa.delta = c->attn_delta;
        launch_attn_flash_cotangent(fa, s.st);
        return;
    }
    akv_keys(s, oG, PS, OS, SP, 0, s.Lp, gv, PS, s.HS);                                              // g_v = g_o P_img
    akv_scores(s, oG, PS, OS, op.xV, 0, s.KS, s.Lp, SG, PS, 0, s.Lp, 1.f, 0.f, false, true);         // g_P_text = g_o^T V_text
    akv_scores(s, oG, PS, OS, v, 0, s.HS, s.T, SG, PS, s.Lp, s.T, 1.f, 0.f, false, false);           // g_P_img = g_o^T v
    launch_softmax_jac(SG, SP, (long)s.NH * s.T, s.Tk, (long)s.NH * s.T, s.scale, s.st, s.B, PS);
    akv_values(s, k, 0, s.HS, s.T, SG, PS, s.Lp, s.T, gq, PS, s.HS, 0.f, false);                     // g_q = k g_S_img^T
    akv_values(s, op.xK, 0, s.KS, s.Lp, SG, PS, 0, s.Lp, gq, PS, s.HS, 1.f, true);                   //     + K_text g_S_text^T
    akv_keys(s, q, 0, s.HS, SG, PS, s.Lp, gk, PS, s.HS);                                             // g_k = q g_S_img
}

// y[Cout][T] (+ bias, + residual) = W x[Cin][T] as a 1x1 conv; dgrad: the transposed map
void lin1x1(loco_ctx* c, const ConvP& w, bool dgrad, const float* in, long in_bs, int Cin, float* out, long out_bs, int Cout,
            int H, int W, int B, const float* res, long res_bs, bool with_bias, hipStream_t st, const StatReq* rq = nullptr) {
    ConvArgs a; conv_defaults(a);
    a.in = in; a.in_bs = in_bs; a.Cin = Cin; a.Hin = H; a.Win = W;
    setw(a, w, dgrad); a.pad = 0;
    if (with_bias) a.bias = w.bias;
    a.res = res; a.res_bs = res_bs;
    a.out = out; a.out_bs = out_bs; a.Cout = Cout; a.Hout = H; a.Wout = W; a.B = B;
    run_conv(c, a, 1, st, rq);
}

// ------------------------------ forward ------------------------------------
int forward_pass(loco_ctx* c, const float* x, float t, int B, float* arena, float* stats, hipStream_t st,
                 const float* t_ptr = nullptr) {
    const loco_unet_cfg& cfg = c->cfg;
    Pass p{c, st, B, arena, stats};
    const long SB = c->stats_per_sample;
    clear_ready(c);
    auto next_fwd = [&](int tid) { return req_fwd_of(c, tid, stats); };   // forward statistics for the consumer of `tid` (+ kept partials of a concatenation part)
    if (cfg.arch < 2) {
        launch_temb(t, cfg.ch, cfg.ch * 4, c->freq, c->td0w, c->td0b, c->td1w, c->td1b, c->tact, st, cfg.arch == 1,
                    c->has_cond ? c->cond_add : nullptr, t_ptr, cfg.act);
        launch_temb_proj(c->tact, cfg.ch * 4, c->tp_w, c->tp_b, (int)c->tproj_total, c->tproj, st);
    }
    for (auto& op : c->ops) {
        const Tens& to = c->tens[op.out];
        const int HW = to.H * to.W;
        switch (op.kind) {
            case OP_CONV_IN: {
                ConvArgs a; conv_defaults(a);
                a.in = x; a.in_bs = c->n_in; a.Cin = cfg.in_channels; a.Hin = cfg.resolution; a.Win = cfg.resolution;
                setw(a, op.conv, false); a.bias = op.conv.bias;
                a.out = p.T(op.out); a.out_bs = p.bs(); a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                if (op.ksize == 1) a.pad = 0;
                const StatReq rq = next_fwd(op.out);
                run_conv(c, a, op.ksize == 1 ? 1 : 9, st, &rq);
                break;
            }
            case OP_CONV: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = p.T(op.in); a.in_bs = p.bs(); a.Cin = ti.C; a.Hin = ti.H; a.Win = ti.W;
                setw(a, op.conv, false); a.bias = op.conv.bias;
                a.out = p.T(op.out); a.out_bs = p.bs(); a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                if (op.ksize == 1) a.pad = 0;
                const StatReq rq = next_fwd(op.out);
                run_conv(c, a, op.ksize == 1 ? 1 : 9, st, &rq);
                break;
            }
            case OP_RES: {
                const Tens& ti = c->tens[op.in];
                const int HWi = ti.H * ti.W;
                if (!op.n1.ready && !cat_fused_stats(c, op.n1, op.in, stats, B, st)) gn_forward_stats(p, op.n1, p.T(op.in), p.bs(), HWi);
                op.n1.ready = false;
                NS s1 = nstats(c, stats, op.n1);
                ConvArgs a; conv_defaults(a);
                a.Cin = ti.C;
                setw(a, op.c1, false); a.bias = op.c1.bias;
                if (!op.scale_shift && op.has_temb) { a.bias2 = c->tproj + op.tproj_off; a.bias2_bs = 0; }
                if (op.updown == 1) {
                    // h = conv(avg_pool(silu(gn(x)))), x' = avg_pool(x)      (unet.py:198-200, 239-244)
                    launch_gn_apply(4, nullptr, 0, p.T(op.in), p.bs(), nullptr, 0, p.T(op.a1), p.bs(), 0, B, ti.C, HWi,
                                    cfg.gn_groups, s1.sc, s1.sh, s1.mr, SB, SB, nullptr, 0, st, cfg.act);
                    launch_pool2x2_sum(p.T(op.a1), p.bs(), p.T(op.ap), p.bs(), 0, B, ti.C, to.H, to.W, st, 0.25f);
                    launch_pool2x2_sum(p.T(op.in), p.bs(), p.T(op.xu), p.bs(), 0, B, ti.C, to.H, to.W, st, 0.25f);
                    a.in = p.T(op.ap); a.in_bs = p.bs(); a.Hin = to.H; a.Win = to.W;
                } else {
                    a.in = p.T(op.in); a.in_bs = p.bs(); a.Hin = ti.H; a.Win = ti.W;
                    a.mode = CM_GN_SILU; a.sc = s1.sc; a.sh = s1.sh; a.scsh_bs = SB;
                    if (op.updown == 2) {
                        a.upsample = 1;
                        launch_upsample2x(p.T(op.in), p.bs(), p.T(op.xu), p.bs(), 0, 1.0f, B, ti.C, ti.H, ti.W, st);
                    }
                }
                a.out = p.T(op.h1); a.out_bs = p.bs(); a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                {   // conv1 also delivers norm2's statistics of its output (scale-shift folded in for the ADM blocks)
                    const float* ssc = op.scale_shift ? c->tproj + op.tproj_off : nullptr;
                    const StatReq rq2 = req_fwd(&op.n2, stats, ssc, ssc ? ssc + to.C : nullptr);
                    run_conv(c, a, 9, st, &rq2);
                    op.n2.ready = false;
                }
                NS s2 = nstats(c, stats, op.n2);
                const float* xin = op.updown ? p.T(op.xu) : p.T(op.in);   // shortcut input after x_upd
                ConvArgs n; conv_defaults(n);                             // the 1x1 shortcut (diffusion.py:887-912), if any
                if (op.has_nin) {
                    n.in = xin; n.in_bs = p.bs(); n.Cin = ti.C; n.Hin = to.H; n.Win = to.W;
                    setw(n, op.nin, false); n.bias = op.nin.bias; n.pad = 0;
                    n.out = p.T(op.out); n.out_bs = p.bs(); n.Cout = to.C; n.Hout = to.H; n.Wout = to.W; n.B = B;
                }
                ConvArgs b; conv_defaults(b);
                b.in = p.T(op.h1); b.in_bs = p.bs(); b.Cin = to.C; b.Hin = to.H; b.Win = to.W;
                setw(b, op.c2, false); b.bias = op.c2.bias; b.res = xin; b.res_bs = p.bs();
                b.res_scale = op.has_nin ? 1.f : c->res_scale;        // (shortcut + h) * res_scale: c2 / nin carry it in their weights
                b.mode = CM_GN_SILU; b.sc = s2.sc; b.sh = s2.sh; b.scsh_bs = SB;
                b.out = p.T(op.out); b.out_bs = p.bs(); b.Cout = to.C; b.Hout = to.H; b.Wout = to.W; b.B = B;
                const StatReq rq = next_fwd(op.out);
                run_conv(c, b, 9, st, &rq, op.has_nin ? &n : nullptr);     // shortcut K-concatenated, or run first as the residual
                break;
            }
            case OP_ATTN: {
                const int C = to.C, T = HW, NH = op.heads, CH = C / NH;
                if (!op.n1.ready) gn_forward_stats(p, op.n1, p.T(op.in), p.bs(), HW);
                op.n1.ready = false;
                NS s = nstats(c, stats, op.n1);
                ConvArgs a; conv_defaults(a);
                a.in = p.T(op.in); a.in_bs = p.bs(); a.Cin = C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.qkvc, false); a.bias = op.qkvc.bias; a.pad = 0;
                a.mode = CM_GN; a.sc = s.sc; a.sh = s.sh; a.scsh_bs = SB;
                a.out = p.T(op.qkv); a.out_bs = p.bs(); a.Cout = 3 * C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                run_conv(c, a, 1, st);
                // per head h: q = qkv[h*3*CH ...], k = +CH, v = +2CH channels (QKVAttentionLegacy, unet.py:346;
                // with one head this is the [q|k|v] stacking of the fused DDPM projection)
                float* q = p.T(op.qkv); float* k = q + (long)CH * T; float* v = k + (long)CH * T;
                if (op.added_kv) {
                    if (!c->has_ctx) { c->err = "this architecture attends over the prompt's states: call loco_set_context first"; return -1; }
                    akv_forward(akv_of(c, op, B, st), arena, p.bs());
                } else {
                GemmArgs g; std::memset(&g, 0, sizeof(g));
                g.A = q; g.sam = 1; g.sak = T; g.sab = p.bs(); g.sah = 3L * CH * T;
                g.Bm = k; g.sbk = T; g.sbn = 1; g.sbb = p.bs(); g.sbh = 3L * CH * T;
                g.C = p.T(op.S); g.scm = T; g.scn = 1; g.scb = p.bs(); g.sch = (long)T * T;
                g.M = T; g.N = T; g.K = CH; g.batch = B; g.batch2 = NH; g.alpha = 1.0f / std::sqrt((float)CH); g.beta = 0.f;
                attn_gemm(c, g, st);
                launch_softmax_rows(p.T(op.S), (long)NH * T, T, st, B, p.bs());
                GemmArgs h; std::memset(&h, 0, sizeof(h));
                h.A = v; h.sam = T; h.sak = 1; h.sab = p.bs(); h.sah = 3L * CH * T;
                h.Bm = p.T(op.S); h.sbk = 1; h.sbn = T; h.sbb = p.bs(); h.sbh = (long)T * T;
                h.C = p.T(op.o); h.scm = T; h.scn = 1; h.scb = p.bs(); h.sch = (long)CH * T;
                h.M = CH; h.N = T; h.K = T; h.batch = B; h.batch2 = NH; h.alpha = 1.f; h.beta = 0.f;
                attn_gemm(c, h, st);
                }
                ConvArgs pr; conv_defaults(pr);
                pr.in = p.T(op.o); pr.in_bs = p.bs(); pr.Cin = C; pr.Hin = to.H; pr.Win = to.W;
                setw(pr, op.proj, false); pr.bias = op.proj.bias; pr.pad = 0;
                pr.res = p.T(op.in); pr.res_bs = p.bs();
                pr.out = p.T(op.has_x ? op.xmid : op.out); pr.out_bs = p.bs(); pr.Cout = C; pr.Hout = to.H; pr.Wout = to.W; pr.B = B;
                {
                    const StatReq rq = next_fwd(op.has_x ? op.xmid : op.out);
                    run_conv(c, pr, 1, st, &rq);
                }
                if (op.has_x) {
                    if (!c->has_ctx) { c->err = "this architecture has cross-attention stages: call loco_set_context first"; return -1; }
                    const XA x = xa_of(c, op, B, st);
                    if (!op.nx.ready) gn_forward_stats(p, op.nx, p.T(op.xmid), p.bs(), HW);
                    op.nx.ready = false;
                    NS sx = nstats(c, stats, op.nx);
                    ConvArgs qa; conv_defaults(qa);
                    qa.in = p.T(op.xmid); qa.in_bs = p.bs(); qa.Cin = C; qa.Hin = to.H; qa.Win = to.W;
                    setw(qa, op.xqc, false); qa.bias = op.xqc.bias; qa.pad = 0;
                    qa.mode = CM_GN; qa.sc = sx.sc; qa.sh = sx.sh; qa.scsh_bs = SB;
                    qa.out = p.T(op.xq); qa.out_bs = p.bs(); qa.Cout = C; qa.Hout = to.H; qa.Wout = to.W; qa.B = B;
                    run_conv(c, qa, 1, st);
                    if (!xa_fused(x, 0, p.T(op.xq), p.bs(), op.xK, op.xV, p.T(op.xS), p.bs(), p.T(op.xo), p.bs())) {
                        xa_scores(x, p.T(op.xq), p.bs(), op.xK, p.T(op.xS), p.bs(), x.scale, true);
                        launch_softmax_rows(p.T(op.xS), (long)NH * T, x.Lp, st, B, p.bs());
                        xa_values(x, op.xV, p.T(op.xS), p.bs(), p.T(op.xo), p.bs());
                    }
                    const StatReq rqx = next_fwd(op.out);
                    xa_conv1x1(c, op.xproj, false, p.T(op.xo), p.bs(), p.T(op.out), p.bs(), C, to.H, to.W, B,
                               p.T(op.xmid), p.bs(), true, st, &rqx);
                }
                break;
            }
            case OP_XFMR: {      // latent-diffusion SpatialTransformer (oracle/loco_oracle.py _ldm_spatial_transformer)
                if (!c->has_ctx) { c->err = "this architecture has cross-attention stages: call loco_set_context first"; return -1; }
                const int C = to.C, T = HW, NH = op.heads, H = to.H, W = to.W;
                const long PSb = p.bs();
                auto X = [&](int i) { return p.T(op.xt[i]); };
                if (!op.n1.ready) gn_forward_stats(p, op.n1, p.T(op.in), PSb, HW);
                op.n1.ready = false;
                NS s = nstats(c, stats, op.n1);
                {   // h0 = proj_in(GN(x))
                    ConvArgs a; conv_defaults(a);
                    a.in = p.T(op.in); a.in_bs = PSb; a.Cin = C; a.Hin = H; a.Win = W;
                    setw(a, op.pj_in, false); a.bias = op.pj_in.bias; a.pad = 0;
                    a.mode = CM_GN; a.sc = s.sc; a.sh = s.sh; a.scsh_bs = SB;
                    a.out = X(X_H0); a.out_bs = PSb; a.Cout = C; a.Hout = H; a.Wout = W; a.B = B;
                    run_conv(c, a, 1, st);
                }
                const SA sa{c, &op, B, T, NH, C / NH, st};
                const XA xa = xa_of(c, op, B, st);
                // x = x + attn1(LN1(x))
                launch_ln_fwd(X(X_H0), PSb, B, C, T, op.lng[0], op.lnb[0], 1e-5f, X(X_A1), PSb, X(X_LN1), PSb, st);
                lin1x1(c, op.qkvc, false, X(X_A1), PSb, C, X(X_QKV), PSb, 3 * C, H, W, B, nullptr, 0, false, st);
                sa_forward(sa, arena);
                lin1x1(c, op.to_out1, false, X(X_O), PSb, C, X(X_H1), PSb, C, H, W, B, X(X_H0), PSb, true, st);
                // x = x + attn2(LN2(x), context)
                launch_ln_fwd(X(X_H1), PSb, B, C, T, op.lng[1], op.lnb[1], 1e-5f, X(X_A2), PSb, X(X_LN2), PSb, st);
                lin1x1(c, op.xqc, false, X(X_A2), PSb, C, X(X_XQ), PSb, C, H, W, B, nullptr, 0, false, st);
                if (!xa_fused(xa, 0, X(X_XQ), PSb, op.xK, op.xV, X(X_XS), PSb, X(X_XO), PSb)) {
                    xa_scores(xa, X(X_XQ), PSb, op.xK, X(X_XS), PSb, xa.scale, true);
                    launch_softmax_rows(X(X_XS), (long)NH * T, xa.Lp, st, B, PSb);
                    xa_values(xa, op.xV, X(X_XS), PSb, X(X_XO), PSb);
                }
                lin1x1(c, op.to_out2, false, X(X_XO), PSb, C, X(X_H2), PSb, C, H, W, B, X(X_H1), PSb, true, st);
                // x = x + Linear(GEGLU(LN3(x)))
                launch_ln_fwd(X(X_H2), PSb, B, C, T, op.lng[2], op.lnb[2], 1e-5f, X(X_A3), PSb, X(X_LN3), PSb, st);
                lin1x1(c, op.ff1, false, X(X_A3), PSb, C, X(X_F), PSb, 8 * C, H, W, B, nullptr, 0, true, st);
                launch_geglu(0, X(X_F), PSb, nullptr, B, 4L * C * T, X(X_GG), PSb, st);
                lin1x1(c, op.ff2, false, X(X_GG), PSb, 4 * C, X(X_H3), PSb, C, H, W, B, X(X_H2), PSb, true, st);
                // out = input + proj_out(x)
                const StatReq rq = next_fwd(op.out);
                lin1x1(c, op.pj_out, false, X(X_H3), PSb, C, p.T(op.out), PSb, C, H, W, B, p.T(op.in), PSb, true, st, &rq);
                break;
            }
            case OP_DOWN: case OP_UP: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = p.T(op.in); a.in_bs = p.bs(); a.Cin = ti.C; a.Hin = ti.H; a.Win = ti.W;
                setw(a, op.conv, false); a.bias = op.conv.bias;
                if (op.kind == OP_DOWN) { a.stride = 2; a.pad = op.sym_down ? 1 : 0; } else { a.upsample = 1; }
                a.out = p.T(op.out); a.out_bs = p.bs(); a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                const StatReq rq = next_fwd(op.out);
                run_conv(c, a, 9, st, &rq);
                break;
            }
            case OP_OUT: {
                const Tens& ti = c->tens[op.in];
                if (!op.n1.ready) gn_forward_stats(p, op.n1, p.T(op.in), p.bs(), ti.H * ti.W);
                op.n1.ready = false;
                NS s = nstats(c, stats, op.n1);
                ConvArgs a; conv_defaults(a);
                a.in = p.T(op.in); a.in_bs = p.bs(); a.Cin = ti.C; a.Hin = ti.H; a.Win = ti.W;
                setw(a, op.conv, false); a.bias = op.conv.bias;
                a.mode = CM_GN_SILU; a.sc = s.sc; a.sh = s.sh; a.scsh_bs = SB;
                a.out = p.T(op.out); a.out_bs = p.bs(); a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                run_conv(c, a, 9, st);
                break;
            }
        }
    }
    return 0;
}

// ------------------------------ tangent ------------------------------------
// primal: arenaP/statsP with B = 1 (broadcast); tangents: arenaT/statsT with B = k
void tangent_stats(loco_ctx* c, const NormP& n, const float* d, long dbs, const float* x, int HW, int B,
                   hipStream_t st) {
    NS sp = nstats(c, c->statsP, n);
    NS stt = nstats(c, c->statsT, n);
    launch_gn_tstats(d, dbs, x, 0, B, n.C, HW, c->cfg.gn_groups, sp.sc, sp.sh, sp.mr, 0, 0, 0, stt.tst,
                     stt.tc, c->stats_per_sample, c->red, st);
}
void set_tan(loco_ctx* c, ConvArgs& a, const NormP& n, const float* prim) {
    NS sp = nstats(c, c->statsP, n);
    NS stt = nstats(c, c->statsT, n);
    a.mode = CM_TAN_SILU; a.prim = prim; a.prim_bs = 0;
    a.sc = sp.sc; a.sh = sp.sh; a.scsh_bs = 0; a.mr = sp.mr; a.mr_bs = 0; a.gamma_ = n.gamma;
    a.tst = stt.tst; a.tst_bs = c->stats_per_sample; a.cpg = n.C / c->cfg.gn_groups;
    a.tc = stt.tc; a.tc_bs = c->stats_per_sample;
    a.sx = (n.sx_off >= 0 && c->sxcache) ? c->sxcache + n.sx_off : nullptr;
}

constexpr size_t RED_BYTES = (size_t)4 << 20;   // reduction scratch (GN partial sums)

int tangent_pass(loco_ctx* c, const float* V, int B, hipStream_t st) {
    const loco_unet_cfg& cfg = c->cfg;
    const long PS = c->per_sample;
    auto TP = [&](int id) { return c->arenaP + c->tens[id].off; };   // primal
    auto TT = [&](int id) { return c->arenaT + c->tens[id].off; };   // tangent
    clear_ready(c);
    // tangent group means for the norm that consumes tensor `tid`, delivered with the conv that finishes it
    // (+ the kept raw partials of a concatenation part: the up-path norm over torch.cat([h, skip]) is finalised from both parts')
    auto next_tan = [&](int tid) {
        StatReq r = req_lin(c, ST_TAN, consumer_norm(c, tid), TP(tid));
        Tens& t = c->tens[tid];
        if (t.cat_of >= 0 && t.keep && c->fuse_lin) {
            r.kind = ST_TAN; r.stats = c->statsT; r.prim = TP(tid);
            r.keep = t.keep; r.keep_floats = t.keep_floats; r.keep_ntile = &t.keep_ntile;
        }
        return r;
    };
    for (auto& op : c->ops) {
        const Tens& to = c->tens[op.out];
        const int HW = to.H * to.W;
        switch (op.kind) {
            case OP_CONV_IN: {
                ConvArgs a; conv_defaults(a);
                a.in = V; a.in_bs = c->n_in; a.Cin = cfg.in_channels; a.Hin = cfg.resolution; a.Win = cfg.resolution;
                setw(a, op.conv, false);
                a.out = TT(op.out); a.out_bs = PS; a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                if (op.ksize == 1) a.pad = 0;
                const StatReq rq = next_tan(op.out);
                run_conv(c, a, op.ksize == 1 ? 1 : 9, st, &rq);
                break;
            }
            case OP_CONV: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = TT(op.in); a.in_bs = PS; a.Cin = ti.C; a.Hin = ti.H; a.Win = ti.W;
                setw(a, op.conv, false);
                a.out = TT(op.out); a.out_bs = PS; a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                if (op.ksize == 1) a.pad = 0;
                const StatReq rq = next_tan(op.out);
                run_conv(c, a, op.ksize == 1 ? 1 : 9, st, &rq);
                break;
            }
            case OP_RES: {
                const Tens& ti = c->tens[op.in];
                const int HWi = ti.H * ti.W;
                if (!op.n1.ready && !cat_fused_tstats(c, op.n1, op.in, B, st)) tangent_stats(c, op.n1, TT(op.in), PS, TP(op.in), HWi, B, st);
                op.n1.ready = false;
                ConvArgs a; conv_defaults(a);
                a.Cin = ti.C;
                setw(a, op.c1, false);
                if (op.updown == 1) {
                    NS sp = nstats(c, c->statsP, op.n1);
                    NS stt = nstats(c, c->statsT, op.n1);
                    launch_gn_apply(5, TT(op.in), PS, TP(op.in), 0, nullptr, 0, TT(op.a1), PS, 0, B, ti.C, HWi,
                                    cfg.gn_groups, sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st, cfg.act);
                    launch_pool2x2_sum(TT(op.a1), PS, TT(op.ap), PS, 0, B, ti.C, to.H, to.W, st, 0.25f);
                    launch_pool2x2_sum(TT(op.in), PS, TT(op.xu), PS, 0, B, ti.C, to.H, to.W, st, 0.25f);
                    a.in = TT(op.ap); a.in_bs = PS; a.Hin = to.H; a.Win = to.W;
                } else {
                    a.in = TT(op.in); a.in_bs = PS; a.Hin = ti.H; a.Win = ti.W;
                    set_tan(c, a, op.n1, TP(op.in));
                    if (op.updown == 2) {
                        a.upsample = 1;
                        launch_upsample2x(TT(op.in), PS, TT(op.xu), PS, 0, 1.0f, B, ti.C, ti.H, ti.W, st);
                    }
                }
                a.out = TT(op.h1); a.out_bs = PS; a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                {
                    const StatReq rq2 = req_lin(c, ST_TAN, &op.n2, TP(op.h1));       // norm2's group means with conv1
                    run_conv(c, a, 9, st, &rq2);
                    op.n2.ready = false;
                }
                const float* xin = op.updown ? TT(op.xu) : TT(op.in);
                ConvArgs n; conv_defaults(n);
                if (op.has_nin) {
                    n.in = xin; n.in_bs = PS; n.Cin = ti.C; n.Hin = to.H; n.Win = to.W;
                    setw(n, op.nin, false); n.pad = 0;
                    n.out = TT(op.out); n.out_bs = PS; n.Cout = to.C; n.Hout = to.H; n.Wout = to.W; n.B = B;
                }
                ConvArgs b; conv_defaults(b);
                b.in = TT(op.h1); b.in_bs = PS; b.Cin = to.C; b.Hin = to.H; b.Win = to.W;
                setw(b, op.c2, false); b.res = xin; b.res_bs = PS;
                b.res_scale = op.has_nin ? 1.f : c->res_scale;
                set_tan(c, b, op.n2, TP(op.h1));
                b.out = TT(op.out); b.out_bs = PS; b.Cout = to.C; b.Hout = to.H; b.Wout = to.W; b.B = B;
                const StatReq rq = next_tan(op.out);
                run_conv(c, b, 9, st, &rq, op.has_nin ? &n : nullptr);
                break;
            }
            case OP_ATTN: {
                const int C = to.C, T = HW, NH = op.heads, CH = C / NH;
                if (!op.n1.ready) tangent_stats(c, op.n1, TT(op.in), PS, TP(op.in), HW, B, st);
                op.n1.ready = false;
                NS sp = nstats(c, c->statsP, op.n1);
                NS stt = nstats(c, c->statsT, op.n1);
                launch_gn_apply(1, TT(op.in), PS, TP(op.in), 0, nullptr, 0, TT(op.hn), PS, 0, B, C, HW,
                                cfg.gn_groups, sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st);
                ConvArgs a; conv_defaults(a);
                a.in = TT(op.hn); a.in_bs = PS; a.Cin = C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.qkvc, false); a.pad = 0;
                a.out = TT(op.qkv); a.out_bs = PS; a.Cout = 3 * C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                run_conv(c, a, 1, st);
                const long HS = 3L * CH * T, SS = (long)T * T;
                float* q = TP(op.qkv); float* k = q + (long)CH * T; float* v = k + (long)CH * T;
                float* dq = TT(op.qkv); float* dk = dq + (long)CH * T; float* dv = dk + (long)CH * T;
                const bool flash = (c->flash_attn && c->prec >= 1 && attn_flash_supported(T, CH)) || op.added_kv;   // added_kv: its own products
                if (op.added_kv) akv_tangent(akv_of(c, op, B, st));
                else if (flash) {   // do from dq, dk, dv and the primal P / o in one kernel, no [T x T] tangent (attn_flash.hip)
                    AttnFlashArgs fa; std::memset(&fa, 0, sizeof(fa));
                    fa.ws = reinterpret_cast<unsigned char*>(c->partial); fa.ws_bytes = c->partial_floats * sizeof(float);      // (records of the operands: attn_flash.hip)
                    fa.T = T; fa.NH = NH; fa.B = B; fa.CH = CH; fa.scale = 1.0f / std::sqrt((float)CH);
                    fa.q = q; fa.k = k; fa.v = v; fa.hs = HS; fa.P = TP(op.S); fa.o = TP(op.o);
                    fa.dq = dq; fa.dk = dk; fa.dv = dv; fa.bs_d = PS; fa.out = TT(op.o); fa.bs_out = PS;
                    launch_attn_flash_tangent(fa, st);
                }
                GemmArgs g; std::memset(&g, 0, sizeof(g));
                g.A = dq; g.sam = 1; g.sak = T; g.sab = PS; g.sah = HS;
                g.Bm = k; g.sbk = T; g.sbn = 1; g.sbb = 0; g.sbh = HS;
                g.C = TT(op.S); g.scm = T; g.scn = 1; g.scb = PS; g.sch = SS;
                g.M = T; g.N = T; g.K = CH; g.batch = B; g.batch2 = NH; g.alpha = 1.f; g.beta = 0.f;
                // dS = dq^T k + q^T dk and do = dv P^T + v dP^T: one launch per pair (K-concatenation) where the launches
                // are latency-shaped (T <= 256: 327.7 vs 330.0 ms per headline step); at 1024 tokens two launches with
                // beta = 1 measured faster (616 vs 639 ms per tloco_if64 step)
                // -- and where the record GEMM takes the product (the decoder's 4096-token head): one write of dS instead of a write
                // and a read-modify-write (500 vs 276 + 586 us, tests/diag/gemm_rec_bench.hip)
                bool kcat = T <= 256;
                if (!kcat && !flash) { GemmArgs g2 = g; g2.A2 = q; g2.Bm2 = dk; g2.sbb2 = PS; kcat = attn_gemm_rec(c, g2); }
                if (kcat) { g.A2 = q; g.sab2 = 0; g.Bm2 = dk; g.sbb2 = PS; }
                if (!flash) {
                attn_gemm(c, g, st);
                if (!kcat) { g.A = q; g.sab = 0; g.Bm = dk; g.sbb = PS; g.beta = 1.f; attn_gemm(c, g, st); }
                launch_softmax_jac(TT(op.S), TP(op.S), (long)NH * T, T, (long)NH * T, 1.0f / std::sqrt((float)CH), st, B, PS);
                }
                GemmArgs h; std::memset(&h, 0, sizeof(h));
                h.A = dv; h.sam = T; h.sak = 1; h.sab = PS; h.sah = HS;
                h.Bm = TP(op.S); h.sbk = 1; h.sbn = T; h.sbb = 0; h.sbh = SS;
                h.C = TT(op.o); h.scm = T; h.scn = 1; h.scb = PS; h.sch = (long)CH * T;
                h.M = CH; h.N = T; h.K = T; h.batch = B; h.batch2 = NH; h.alpha = 1.f; h.beta = 0.f;
                if (kcat) { h.A2 = v; h.sab2 = 0; h.Bm2 = TT(op.S); h.sbb2 = PS; }
                if (!flash) {
                attn_gemm(c, h, st);
                if (!kcat) { h.A = v; h.sab = 0; h.Bm = TT(op.S); h.sbb = PS; h.beta = 1.f; attn_gemm(c, h, st); }
                }
                ConvArgs pr; conv_defaults(pr);
                pr.in = TT(op.o); pr.in_bs = PS; pr.Cin = C; pr.Hin = to.H; pr.Win = to.W;
                setw(pr, op.proj, false); pr.pad = 0; pr.res = TT(op.in); pr.res_bs = PS;
                pr.out = TT(op.has_x ? op.xmid : op.out); pr.out_bs = PS; pr.Cout = C; pr.Hout = to.H; pr.Wout = to.W; pr.B = B;
                {
                    const StatReq rq = next_tan(op.has_x ? op.xmid : op.out);
                    run_conv(c, pr, 1, st, &rq);
                }
                if (op.has_x) {
                    const XA x = xa_of(c, op, B, st);
                    if (!op.nx.ready) tangent_stats(c, op.nx, TT(op.xmid), PS, TP(op.xmid), HW, B, st);
                    op.nx.ready = false;
                    NS spx = nstats(c, c->statsP, op.nx);
                    NS stx = nstats(c, c->statsT, op.nx);
                    launch_gn_apply(1, TT(op.xmid), PS, TP(op.xmid), 0, nullptr, 0, TT(op.xhn), PS, 0, B, C, HW,
                                    cfg.gn_groups, spx.sc, spx.sh, spx.mr, 0, 0, stx.tst, c->stats_per_sample, st);
                    xa_conv1x1(c, op.xqc, false, TT(op.xhn), PS, TT(op.xq), PS, C, to.H, to.W, B, nullptr, 0, false, st);
                    if (!xa_fused(x, 1, TT(op.xq), PS, op.xK, op.xV, TP(op.xS), 0, TT(op.xo), PS)) {
                        xa_scores(x, TT(op.xq), PS, op.xK, TT(op.xS), PS, 1.f, false);                  // dS = dq^T K
                        launch_softmax_jac(TT(op.xS), TP(op.xS), (long)NH * T, x.Lp, (long)NH * T, x.scale, st, B, PS);
                        xa_values(x, op.xV, TT(op.xS), PS, TT(op.xo), PS);                              // do = V dP^T
                    }
                    const StatReq rqx = next_tan(op.out);
                    xa_conv1x1(c, op.xproj, false, TT(op.xo), PS, TT(op.out), PS, C, to.H, to.W, B, TT(op.xmid), PS, false, st, &rqx);
                }
                break;
            }
            case OP_XFMR: {
                const int C = to.C, T = HW, NH = op.heads, H = to.H, W = to.W;
                auto XP = [&](int i) { return TP(op.xt[i]); };
                auto XT_ = [&](int i) { return TT(op.xt[i]); };
                if (!op.n1.ready) tangent_stats(c, op.n1, TT(op.in), PS, TP(op.in), HW, B, st);
                op.n1.ready = false;
                NS sp = nstats(c, c->statsP, op.n1);
                NS stt = nstats(c, c->statsT, op.n1);
                launch_gn_apply(1, TT(op.in), PS, TP(op.in), 0, nullptr, 0, XT_(X_G0), PS, 0, B, C, HW,
                                cfg.gn_groups, sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st);
                lin1x1(c, op.pj_in, false, XT_(X_G0), PS, C, XT_(X_H0), PS, C, H, W, B, nullptr, 0, false, st);
                const SA sa{c, &op, B, T, NH, C / NH, st};
                const XA xa = xa_of(c, op, B, st);
                launch_ln_tan(XT_(X_H0), PS, XP(X_H0), XP(X_LN1), B, C, T, op.lng[0], XT_(X_A1), PS, st);
                lin1x1(c, op.qkvc, false, XT_(X_A1), PS, C, XT_(X_QKV), PS, 3 * C, H, W, B, nullptr, 0, false, st);
                sa_tangent(sa);
                lin1x1(c, op.to_out1, false, XT_(X_O), PS, C, XT_(X_H1), PS, C, H, W, B, XT_(X_H0), PS, false, st);
                launch_ln_tan(XT_(X_H1), PS, XP(X_H1), XP(X_LN2), B, C, T, op.lng[1], XT_(X_A2), PS, st);
                lin1x1(c, op.xqc, false, XT_(X_A2), PS, C, XT_(X_XQ), PS, C, H, W, B, nullptr, 0, false, st);
                if (!xa_fused(xa, 1, XT_(X_XQ), PS, op.xK, op.xV, XP(X_XS), 0, XT_(X_XO), PS)) {
                    xa_scores(xa, XT_(X_XQ), PS, op.xK, XT_(X_XS), PS, 1.f, false);                   // dS = dq^T K
                    launch_softmax_jac(XT_(X_XS), XP(X_XS), (long)NH * T, xa.Lp, (long)NH * T, xa.scale, st, B, PS);
                    xa_values(xa, op.xV, XT_(X_XS), PS, XT_(X_XO), PS);                              // do = V dP^T
                }
                lin1x1(c, op.to_out2, false, XT_(X_XO), PS, C, XT_(X_H2), PS, C, H, W, B, XT_(X_H1), PS, false, st);
                launch_ln_tan(XT_(X_H2), PS, XP(X_H2), XP(X_LN3), B, C, T, op.lng[2], XT_(X_A3), PS, st);
                lin1x1(c, op.ff1, false, XT_(X_A3), PS, C, XT_(X_F), PS, 8 * C, H, W, B, nullptr, 0, false, st);
                launch_geglu(1, XT_(X_F), PS, XP(X_F), B, 4L * C * T, XT_(X_GG), PS, st);
                lin1x1(c, op.ff2, false, XT_(X_GG), PS, 4 * C, XT_(X_H3), PS, C, H, W, B, XT_(X_H2), PS, false, st);
                const StatReq rq = next_tan(op.out);
                lin1x1(c, op.pj_out, false, XT_(X_H3), PS, C, TT(op.out), PS, C, H, W, B, TT(op.in), PS, false, st, &rq);
                break;
            }
            case OP_DOWN: case OP_UP: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = TT(op.in); a.in_bs = PS; a.Cin = ti.C; a.Hin = ti.H; a.Win = ti.W;
                setw(a, op.conv, false);
                if (op.kind == OP_DOWN) { a.stride = 2; a.pad = op.sym_down ? 1 : 0; } else { a.upsample = 1; }
                a.out = TT(op.out); a.out_bs = PS; a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                const StatReq rq = next_tan(op.out);
                run_conv(c, a, 9, st, &rq);
                break;
            }
            case OP_OUT: {
                const Tens& ti = c->tens[op.in];
                if (!op.n1.ready) tangent_stats(c, op.n1, TT(op.in), PS, TP(op.in), ti.H * ti.W, B, st);
                op.n1.ready = false;
                ConvArgs a; conv_defaults(a);
                a.in = TT(op.in); a.in_bs = PS; a.Cin = ti.C; a.Hin = ti.H; a.Win = ti.W;
                setw(a, op.conv, false);
                set_tan(c, a, op.n1, TP(op.in));
                a.out = TT(op.out); a.out_bs = PS; a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                run_conv(c, a, 9, st);
                break;
            }
        }
    }
    return 0;
}

// ------------------------------ cotangent ----------------------------------
void cot_stats(loco_ctx* c, const NormP& n, const float* d, long dbs, const float* x, int HW, int B, int kind,
               hipStream_t st) {
    NS sp = nstats(c, c->statsP, n);
    NS stt = nstats(c, c->statsT, n);
    launch_gn_tstats(d, dbs, x, 0, B, n.C, HW, c->cfg.gn_groups, sp.sc, sp.sh, sp.mr, 0, 0, kind, stt.tst,
                     stt.tc, c->stats_per_sample, c->red, st, c->cfg.act);
}

// ge: cotangent of eps [B][n]; result A[B][n] = conv_in^T(...) + gx0
int cotangent_pass(loco_ctx* c, const float* ge, const float* gx0, float* Aout, int B, hipStream_t st) {
    const loco_unet_cfg& cfg = c->cfg;
    const long PS = c->per_sample;
    const int G = cfg.gn_groups;
    auto TP = [&](int id) { return c->arenaP + c->tens[id].off; };
    auto TG = [&](int id) { return c->arenaT + c->tens[id].off; };   // cotangent of tensor id
    // the op that produces the network output takes the caller's cotangent (conv_out; the encoder's quant_conv)
    auto GO = [&](const Op& o) -> const float* { return o.out == c->eps_t ? ge : TG(o.out); };
    auto GOS = [&](const Op& o) -> long { return o.out == c->eps_t ? (long)c->n_out : PS; };
    clear_ready(c);
    for (int oi = (int)c->ops.size() - 1; oi >= 0; --oi) {
        Op& op = c->ops[oi];
        const Tens& to = c->tens[op.out];
        const int HW = to.H * to.W;
        switch (op.kind) {
            case OP_OUT: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = GO(op); a.in_bs = GOS(op); a.Cin = to.C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.conv, true);
                a.out = TG(op.a1); a.out_bs = PS; a.Cout = ti.C; a.Hout = ti.H; a.Wout = ti.W; a.B = B;
                {
                    const StatReq rq = req_lin(c, ST_COT, &op.n1, TP(op.in));        // the norm's cotangent means with the conv
                    run_conv(c, a, 9, st, &rq);
                    op.n1.ready = false;
                }
                NS sp = nstats(c, c->statsP, op.n1);
                NS stt = nstats(c, c->statsT, op.n1);
                launch_gn_apply(2, TG(op.a1), PS, TP(op.in), 0, nullptr, 0, TG(op.in), PS, 0, B, ti.C, ti.H * ti.W, G,
                                sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st, cfg.act);
                break;
            }
            case OP_UP: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = TG(op.out); a.in_bs = PS; a.Cin = to.C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.conv, true);
                a.out = TG(op.up); a.out_bs = PS; a.Cout = ti.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                run_conv(c, a, 9, st);
                launch_pool2x2_sum(TG(op.up), PS, TG(op.in), PS, 0, B, ti.C, ti.H, ti.W, st);
                break;
            }
            case OP_DOWN: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = TG(op.out); a.in_bs = PS; a.Cin = to.C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.conv, true); a.zins = 1; a.pad = op.sym_down ? 1 : 2; a.accumulate = op.in_is_skip ? 1 : 0;
                a.out = TG(op.in); a.out_bs = PS; a.Cout = ti.C; a.Hout = ti.H; a.Wout = ti.W; a.B = B;
                run_conv(c, a, 9, st);
                break;
            }
            case OP_RES: {
                const Tens& ti = c->tens[op.in];
                const int HWi = ti.H * ti.W;
                // g_a2 = dgrad conv2 (g_out)  -> slot h1
                ConvArgs a; conv_defaults(a);
                a.in = TG(op.out); a.in_bs = PS; a.Cin = to.C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.c2, true);
                a.out = TG(op.h1); a.out_bs = PS; a.Cout = to.C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                {
                    const StatReq rq2 = req_lin(c, ST_COT, &op.n2, TP(op.h1));
                    run_conv(c, a, 9, st, &rq2);
                    op.n2.ready = false;
                }
                // dgrad conv1 of the norm2/silu cotangent of g_a2 (fused in the staging), at conv1's resolution
                ConvArgs b; conv_defaults(b);
                b.in = TG(op.h1); b.in_bs = PS; b.Cin = to.C; b.Hin = to.H; b.Win = to.W;
                setw(b, op.c1, true);
                set_tan(c, b, op.n2, TP(op.h1));
                b.mode = CM_COT_SILU;
                b.out_bs = PS; b.Cout = ti.C; b.Hout = to.H; b.Wout = to.W; b.B = B;
                if (op.updown == 0) {
                    b.out = TG(op.a1);
                    const StatReq rq1 = req_lin(c, ST_COT, &op.n1, TP(op.in));
                    run_conv(c, b, 9, st, &rq1);
                } else if (op.updown == 1) {
                    b.out = TG(op.ap);                                     // cotangent of the pooled activation
                    run_conv(c, b, 9, st);
                    launch_upsample2x(TG(op.ap), PS, TG(op.a1), PS, 0, 0.25f, B, ti.C, to.H, to.W, st);   // avg-pool^T
                } else {
                    b.out = TG(op.xu);                                     // cotangent of the nearest-upsampled activation
                    run_conv(c, b, 9, st);
                    launch_pool2x2_sum(TG(op.xu), PS, TG(op.a1), PS, 0, B, ti.C, ti.H, ti.W, st);           // nearest^T
                }
                if (!op.n1.ready) cot_stats(c, op.n1, TG(op.a1), PS, TP(op.in), HWi, B, 1, st);
                op.n1.ready = false;
                NS sp = nstats(c, c->statsP, op.n1);
                NS stt = nstats(c, c->statsT, op.n1);
                int acc = op.in_is_skip ? 1 : 0;
                // shortcut cotangent at the block's output resolution: identity or nin^T
                const float* gsk = TG(op.out);
                bool cot_in_epilogue = false;
                if (op.has_nin) {
                    ConvArgs n; conv_defaults(n);
                    n.in = TG(op.out); n.in_bs = PS; n.Cin = to.C; n.Hin = to.H; n.Win = to.W;
                    setw(n, op.nin, true); n.pad = 0;
                    n.Cout = ti.C; n.Hout = to.H; n.Wout = to.W; n.B = B; n.out_bs = PS;
                    n.out = TG(op.in); n.accumulate = acc;      // (resampling blocks never change channels: no nin there)
                    if (op.updown == 0 && c->prec >= 1 && c->fuse_cot) {
                        // g_in = nin^T g_out + norm1^T g_a1: the norm-cotangent term in the shortcut conv's epilogue instead of
                        // a read-modify-write pass of gn_apply_kernel<2> over g_in -- one unsplit launch of whole cout tiles only
                        ConvArgs t = n;
                        t.taps = 1;
                        t.nsplit = conv_bf16_pick_nsplit(n.Cin, n.Cout, n.Hout, n.Wout, n.B, c->chip_share, 1, c->lanes_active);
                        const long per_probe = (long)((n.Hout * n.Wout) / conv_bf16_tile_pixels(t)) * ((n.Cout + 127) / 128);
                        const long total = per_probe * n.B, r = total % 256;
                        if (t.nsplit == 1 && (total <= 256 || r == 0 || r > 160) && conv_lowp_can_fuse_stats(t) && op.n1.sx_off >= 0 &&
                            c->sxcache) {
                            // (from norm1's {S, xhat} records and the per-channel {rstd m1, rstd m2} its cotangent statistics left)
                            n.cot_d = TG(op.a1); n.cot_d_bs = PS; n.cot_sx = c->sxcache + op.n1.sx_off;
                            n.cot_tc = stt.tc; n.cot_tc_bs = c->stats_per_sample;
                            cot_in_epilogue = true;
                        }
                    }
                    run_conv(c, n, 1, st);
                    gsk = nullptr;
                }
                if (op.updown == 0) {
                    if (cot_in_epilogue) { /* done by the shortcut conv */ }
                    else if (op.has_nin)
                        launch_gn_apply(2, TG(op.a1), PS, TP(op.in), 0, nullptr, 0, TG(op.in), PS, 1, B, ti.C, HWi, G,
                                        sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st, cfg.act);
                    else        // identity shortcut: g_in = res_scale * g_out + norm1^T g_a1
                        launch_gn_apply(2, TG(op.a1), PS, TP(op.in), 0, gsk, PS, TG(op.in), PS, acc, B, ti.C, HWi, G,
                                        sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st, cfg.act, c->res_scale);
                } else {
                    launch_gn_apply(2, TG(op.a1), PS, TP(op.in), 0, nullptr, 0, TG(op.in), PS, acc, B, ti.C, HWi, G,
                                    sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st, cfg.act);
                    if (op.updown == 1) launch_upsample2x(gsk, PS, TG(op.in), PS, 1, 0.25f * c->res_scale, B, ti.C, to.H, to.W, st);
                    else launch_pool2x2_sum(gsk, PS, TG(op.in), PS, 1, B, ti.C, ti.H, ti.W, st, c->res_scale);
                }
                break;
            }
            case OP_ATTN: {
                const int C = to.C, T = HW, NH = op.heads, CH = C / NH;
                const long HS = 3L * CH * T, SS = (long)T * T, OS = (long)CH * T;
                float* q = TP(op.qkv); float* k = q + (long)CH * T; float* v = k + (long)CH * T;
                float* gq = TG(op.qkv); float* gk = gq + (long)CH * T; float* gv = gk + (long)CH * T;
                const int so = op.has_x ? op.xmid : op.out;       // output tensor of the self-attention stage
                if (op.has_x) {
                    // cotangent of the cross-attention stage: g_out -> g_xmid (residual + the q path through GN)
                    const XA x = xa_of(c, op, B, st);
                    xa_conv1x1(c, op.xproj, true, TG(op.out), PS, TG(op.xo), PS, C, to.H, to.W, B, nullptr, 0, false, st);
                    if (!xa_fused(x, 1, TG(op.xo), PS, op.xV, op.xK, TP(op.xS), 0, TG(op.xq), PS)) {
                        xa_scores(x, TG(op.xo), PS, op.xV, TG(op.xS), PS, 1.f, false);                  // g_P = g_o^T V
                        launch_softmax_jac(TG(op.xS), TP(op.xS), (long)NH * T, x.Lp, (long)NH * T, x.scale, st, B, PS);
                        xa_values(x, op.xK, TG(op.xS), PS, TG(op.xq), PS);                              // g_q = K g_S^T
                    }
                    xa_conv1x1(c, op.xqc, true, TG(op.xq), PS, TG(op.xhn), PS, C, to.H, to.W, B, nullptr, 0, false, st);
                    cot_stats(c, op.nx, TG(op.xhn), PS, TP(op.xmid), HW, B, 2, st);
                    NS spx = nstats(c, c->statsP, op.nx);
                    NS stx = nstats(c, c->statsT, op.nx);
                    launch_gn_apply(3, TG(op.xhn), PS, TP(op.xmid), 0, TG(op.out), PS, TG(op.xmid), PS, 0, B, C, HW, G,
                                    spx.sc, spx.sh, spx.mr, 0, 0, stx.tst, c->stats_per_sample, st);
                }
                // g_o = proj^T g_out
                ConvArgs pr; conv_defaults(pr);
                pr.in = TG(so); pr.in_bs = PS; pr.Cin = C; pr.Hin = to.H; pr.Win = to.W;
                setw(pr, op.proj, true); pr.pad = 0;
                pr.out = TG(op.o); pr.out_bs = PS; pr.Cout = C; pr.Hout = to.H; pr.Wout = to.W; pr.B = B;
                run_conv(c, pr, 1, st);
                const bool flash = (c->flash_attn && c->prec >= 1 && attn_flash_supported(T, CH)) || op.added_kv;   // added_kv: its own products
                if (op.added_kv) akv_cotangent(akv_of(c, op, B, st));
                else if (flash) {   // g_q, g_k, g_v from g_o and the primal q / k / v / P / o, no [T x T] cotangent (attn_flash.hip)
                    AttnFlashArgs fa; std::memset(&fa, 0, sizeof(fa));
                    fa.ws = reinterpret_cast<unsigned char*>(c->partial); fa.ws_bytes = c->partial_floats * sizeof(float);      // (records of the operands: attn_flash.hip)
                    fa.T = T; fa.NH = NH; fa.B = B; fa.CH = CH; fa.scale = 1.0f / std::sqrt((float)CH);
                    fa.q = q; fa.k = k; fa.v = v; fa.hs = HS; fa.P = TP(op.S); fa.o = TP(op.o);
                    fa.go = TG(op.o); fa.bs_go = PS; fa.gq = gq; fa.gk = gk; fa.gv = gv; fa.bs_g = PS; fa.delta = c->attn_delta;
                    launch_attn_flash_cotangent(fa, st);
                }
                // g_v[c][j] = sum_i g_o[c][i] P[i][j]
                GemmArgs g; std::memset(&g, 0, sizeof(g));
                g.A = TG(op.o); g.sam = T; g.sak = 1; g.sab = PS; g.sah = OS;
                g.Bm = TP(op.S); g.sbk = T; g.sbn = 1; g.sbb = 0; g.sbh = SS;
                g.C = gv; g.scm = T; g.scn = 1; g.scb = PS; g.sch = HS;
                g.M = CH; g.N = T; g.K = T; g.batch = B; g.batch2 = NH; g.alpha = 1.f; g.beta = 0.f;
                if (!flash) attn_gemm(c, g, st);
                // g_P[i][j] = sum_c g_o[c][i] v[c][j]
                GemmArgs h; std::memset(&h, 0, sizeof(h));
                h.A = TG(op.o); h.sam = 1; h.sak = T; h.sab = PS; h.sah = OS;
                h.Bm = v; h.sbk = T; h.sbn = 1; h.sbb = 0; h.sbh = HS;
                h.C = TG(op.S); h.scm = T; h.scn = 1; h.scb = PS; h.sch = SS;
                h.M = T; h.N = T; h.K = CH; h.batch = B; h.batch2 = NH; h.alpha = 1.f; h.beta = 0.f;
                if (!flash) {
                attn_gemm(c, h, st);
                launch_softmax_jac(TG(op.S), TP(op.S), (long)NH * T, T, (long)NH * T, 1.0f / std::sqrt((float)CH), st, B, PS);
                }
                // g_q[c][i] = sum_j k[c][j] g_S[i][j]
                GemmArgs gq_; std::memset(&gq_, 0, sizeof(gq_));
                gq_.A = k; gq_.sam = T; gq_.sak = 1; gq_.sab = 0; gq_.sah = HS;
                gq_.Bm = TG(op.S); gq_.sbk = 1; gq_.sbn = T; gq_.sbb = PS; gq_.sbh = SS;
                gq_.C = gq; gq_.scm = T; gq_.scn = 1; gq_.scb = PS; gq_.sch = HS;
                gq_.M = CH; gq_.N = T; gq_.K = T; gq_.batch = B; gq_.batch2 = NH; gq_.alpha = 1.f; gq_.beta = 0.f;
                if (!flash) attn_gemm(c, gq_, st);
                // g_k[c][j] = sum_i q[c][i] g_S[i][j]
                GemmArgs gk_ = gq_;
                gk_.A = q; gk_.Bm = TG(op.S); gk_.sbk = T; gk_.sbn = 1; gk_.C = gk;
                if (!flash) attn_gemm(c, gk_, st);
                // g_hn = Wqkv^T g_qkv
                ConvArgs a; conv_defaults(a);
                a.in = TG(op.qkv); a.in_bs = PS; a.Cin = 3 * C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.qkvc, true); a.pad = 0;
                a.out = TG(op.hn); a.out_bs = PS; a.Cout = C; a.Hout = to.H; a.Wout = to.W; a.B = B;
                run_conv(c, a, 1, st);
                cot_stats(c, op.n1, TG(op.hn), PS, TP(op.in), HW, B, 2, st);
                NS sp = nstats(c, c->statsP, op.n1);
                NS stt = nstats(c, c->statsT, op.n1);
                launch_gn_apply(3, TG(op.hn), PS, TP(op.in), 0, TG(so), PS, TG(op.in), PS,
                                op.in_is_skip ? 1 : 0, B, C, HW, G, sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st);
                break;
            }
            case OP_XFMR: {      // transposes of the forward chain, last map first
                const int C = to.C, T = HW, NH = op.heads, H = to.H, W = to.W;
                auto XP = [&](int i) { return TP(op.xt[i]); };
                auto XG = [&](int i) { return TG(op.xt[i]); };
                const SA sa{c, &op, B, T, NH, C / NH, st};
                const XA xa = xa_of(c, op, B, st);
                lin1x1(c, op.pj_out, true, TG(op.out), PS, C, XG(X_H3), PS, C, H, W, B, nullptr, 0, false, st);          // g_h3
                // feed-forward: h3 = h2 + ff2(geglu(ff1(LN3(h2))))
                lin1x1(c, op.ff2, true, XG(X_H3), PS, C, XG(X_GG), PS, 4 * C, H, W, B, nullptr, 0, false, st);
                launch_geglu(2, XG(X_GG), PS, XP(X_F), B, 4L * C * T, XG(X_F), PS, st);
                lin1x1(c, op.ff1, true, XG(X_F), PS, 8 * C, XG(X_A3), PS, C, H, W, B, nullptr, 0, false, st);
                launch_ln_cot(XG(X_A3), PS, XP(X_H2), XP(X_LN3), B, C, T, op.lng[2], XG(X_H3), PS, XG(X_H2), PS, st);  // g_h2
                // cross-attention: h2 = h1 + to_out2(V P^T), q = to_q(LN2(h1))
                lin1x1(c, op.to_out2, true, XG(X_H2), PS, C, XG(X_XO), PS, C, H, W, B, nullptr, 0, false, st);
                if (!xa_fused(xa, 1, XG(X_XO), PS, op.xV, op.xK, XP(X_XS), 0, XG(X_XQ), PS)) {
                    xa_scores(xa, XG(X_XO), PS, op.xV, XG(X_XS), PS, 1.f, false);                     // g_P = g_o^T V
                    launch_softmax_jac(XG(X_XS), XP(X_XS), (long)NH * T, xa.Lp, (long)NH * T, xa.scale, st, B, PS);
                    xa_values(xa, op.xK, XG(X_XS), PS, XG(X_XQ), PS);                                // g_q = K g_S^T
                }
                lin1x1(c, op.xqc, true, XG(X_XQ), PS, C, XG(X_A2), PS, C, H, W, B, nullptr, 0, false, st);
                launch_ln_cot(XG(X_A2), PS, XP(X_H1), XP(X_LN2), B, C, T, op.lng[1], XG(X_H2), PS, XG(X_H1), PS, st);  // g_h1
                // self-attention: h1 = h0 + to_out1(attn(qkv(LN1(h0))))
                lin1x1(c, op.to_out1, true, XG(X_H1), PS, C, XG(X_O), PS, C, H, W, B, nullptr, 0, false, st);
                sa_cotangent(sa);
                lin1x1(c, op.qkvc, true, XG(X_QKV), PS, 3 * C, XG(X_A1), PS, C, H, W, B, nullptr, 0, false, st);
                launch_ln_cot(XG(X_A1), PS, XP(X_H0), XP(X_LN1), B, C, T, op.lng[0], XG(X_H1), PS, XG(X_H0), PS, st);  // g_h0
                // proj_in of GN(x), and the residual  out = x + ...
                lin1x1(c, op.pj_in, true, XG(X_H0), PS, C, XG(X_G0), PS, C, H, W, B, nullptr, 0, false, st);
                cot_stats(c, op.n1, XG(X_G0), PS, TP(op.in), HW, B, 2, st);
                NS sp = nstats(c, c->statsP, op.n1);
                NS stt = nstats(c, c->statsT, op.n1);
                launch_gn_apply(3, XG(X_G0), PS, TP(op.in), 0, TG(op.out), PS, TG(op.in), PS,
                                op.in_is_skip ? 1 : 0, B, C, HW, G, sp.sc, sp.sh, sp.mr, 0, 0, stt.tst, c->stats_per_sample, st);
                break;
            }
            case OP_CONV_IN: {
                ConvArgs a; conv_defaults(a);
                a.in = TG(op.out); a.in_bs = PS; a.Cin = to.C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.conv, true); a.res = gx0; a.res_bs = c->n_in;
                a.out = Aout; a.out_bs = c->n_in; a.Cout = cfg.in_channels; a.Hout = cfg.resolution;
                a.Wout = cfg.resolution; a.B = B;
                if (op.ksize == 1) a.pad = 0;
                run_conv(c, a, op.ksize == 1 ? 1 : 9, st);
                break;
            }
            case OP_CONV: {
                const Tens& ti = c->tens[op.in];
                ConvArgs a; conv_defaults(a);
                a.in = GO(op); a.in_bs = GOS(op); a.Cin = to.C; a.Hin = to.H; a.Win = to.W;
                setw(a, op.conv, true);
                a.out = TG(op.in); a.out_bs = PS; a.Cout = ti.C; a.Hout = ti.H; a.Wout = ti.W; a.B = B;
                if (op.ksize == 1) a.pad = 0;
                run_conv(c, a, op.ksize == 1 ? 1 : 9, st);
                break;
            }
        }
    }
    return 0;
}

}  // namespace

// Probe groups on two streams.  The probes of a batch are independent, so a batch of B >= 4 is cut in two groups whose
// passes are enqueued on two streams: the MFMA-bound convolutions of one group then run beside the bandwidth-bound
// statistics / apply / reduce kernels of the other, and the two groups' 256-workgroup rounds drift out of phase (the
// write burst of one group's epilogue overlaps the other's stage loops).  Lane 1 works on its own samples of the T
// arena and its own reduction / split-K scratch; the host enqueues lane 0 completely, then lane 1.
namespace {
struct LaneSwap {
    loco_ctx* c; float *arenaT, *statsT, *partial, *eps_buf, *ge, *gx0, *stpart, *attn_delta; double* red; size_t partial_floats;
    LaneSwap(loco_ctx* c_, int s0) : c(c_) {
        arenaT = c->arenaT; statsT = c->statsT; partial = c->partial; eps_buf = c->eps_buf; ge = c->ge; gx0 = c->gx0;
        // the flash cotangent indexes its delta scratch by the lane-local sample: lane 1 gets the slots of ITS samples
        attn_delta = c->attn_delta; c->attn_delta += (long)s0 * c->attn_dmax;
        red = c->red; partial_floats = c->partial_floats; stpart = c->stpart; c->stpart = c->stpart2;
        c->arenaT += (long)s0 * c->per_sample; c->statsT += (long)s0 * c->stats_per_sample;
        c->eps_buf += (long)s0 * c->n_out; c->ge += (long)s0 * c->n_out; c->gx0 += (long)s0 * c->n_in;
        c->partial += partial_floats / 2; c->partial_floats = partial_floats / 2;
        c->red = c->red2;
        c->lane = 1; c->lane_s0 = s0;
    }
    ~LaneSwap() {
        c->arenaT = arenaT; c->statsT = statsT; c->partial = partial; c->eps_buf = eps_buf; c->ge = ge; c->gx0 = gx0;
        c->red = red; c->partial_floats = partial_floats; c->stpart = stpart; c->attn_delta = attn_delta;
        c->lane = 0; c->lane_s0 = 0;
    }
};
template <typename F>
int run_lanes(loco_ctx* c, int B, hipStream_t st, F body) {      // body(first sample, count, stream)
    if (c->n_streams < 2 || c->prof_on || B < 4) return body(0, B, st);
    const int nA = (B + 1) / 2;
    const size_t pf = c->partial_floats;
    HIPCHK(c, hipEventRecord(c->ev_fork, st));
    c->partial_floats = pf / 2;                    // lane 0 keeps the lower half of the split-K workspace
    c->lanes_active = 2;
    int rc = body(0, nA, st);
    c->partial_floats = pf;
    if (rc) { c->lanes_active = 1; return rc; }
    hipStream_t s2 = c->st2_user ? c->st2_user : c->st2;
    if (hipStreamWaitEvent(s2, c->ev_fork, 0) != hipSuccess) { c->lanes_active = 1; c->err = "run_lanes: hipStreamWaitEvent"; return -1; }
    {
        LaneSwap sw(c, nA);
        rc = body(nA, B - nA, s2);
    }
    c->lanes_active = 1;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_join, s2));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
    return 0;
}
}  // namespace

// =============================== C ABI =======================================
extern "C" {

const char* loco_version(void) { return "loco_hip 0.2 (gfx950; conv arithmetic: f32 MFMA | bf16x3 split-bf16 MFMA | f16 MFMA)"; }

int loco_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int loco_create(const loco_unet_cfg* cfg, loco_ctx** out) {
    if (!cfg || !out) return -2;
    loco_ctx* c = new loco_ctx();
    *out = c;
    if (cfg->struct_size != (int32_t)sizeof(loco_unet_cfg)) {
        // a binding built against another revision of include/loco_hip.h: refuse before reading past its struct
        c->err = "loco_unet_cfg.struct_size = " + std::to_string(cfg->struct_size) + ", this library expects " +
                 std::to_string(sizeof(loco_unet_cfg)) + " (binding and library built from different headers)";
        return -2;
    }
    c->cfg = *cfg;
    if (cfg->max_batch < 1 || cfg->num_levels < 1 || cfg->num_levels > 8) { c->err = "bad config"; return -2; }
    {
        int r = cfg->resolution;
        for (int l = 0; l < cfg->num_levels - 1; ++l) r /= 2;
        if (cfg->arch == 2) r = cfg->resolution;      // decoder: `resolution` is the coarsest (latent) level
        if (cfg->arch > 3 || cfg->arch < 0) { c->err = "arch must be 0 (Ho-DDPM), 1 (guided-diffusion family), 2 (latent decoder) or 3 (latent encoder)"; return -2; }
        if (r < 8 || (cfg->resolution & (cfg->resolution - 1))) {
            c->err = "resolution must be a power of two with >= 8x8 at the coarsest level";
            return -2;
        }
        if (cfg->ch % 32) { c->err = "ch must be a multiple of 32"; return -2; }
    }
    if (cfg->act != ACT_SILU && cfg->act != ACT_GELU) { c->err = "act must be 0 (SiLU) or 1 (GELU)"; return -2; }
    c->res_scale = cfg->res_scale == 0.f ? 1.f : cfg->res_scale;
    if ((cfg->act != ACT_SILU || c->res_scale != 1.f || cfg->added_kv) && cfg->arch != 1) {
        c->err = "act / res_scale / added_kv belong to the guided-diffusion family (arch 1)"; return -2;
    }
    if (cfg->added_kv && (cfg->context_dim <= 0 || cfg->context_len <= 0 || cfg->transformer_depth != 0 ||
                          cfg->context_dim % cfg->gn_groups)) {
        c->err = "added_kv needs context_dim (a multiple of gn_groups) and context_len > 0 and transformer_depth = 0"; return -2;
    }
    if (build_program(c)) return -2;
    for (auto& op : c->ops)
        if (op.kind == OP_RES && op.updown && op.has_nin) { c->err = "resampling ResBlock with a channel change is not supported"; return -2; }
    // which norm takes its statistics over exactly which tensor: the conv that finishes that tensor delivers them
    for (size_t i = 0; i < c->ops.size(); ++i) {
        const Op& op = c->ops[i];
        if ((op.kind == OP_RES || op.kind == OP_ATTN || op.kind == OP_OUT || op.kind == OP_XFMR) && op.in >= 0) {
            c->tens[op.in].cons_op = (int)i; c->tens[op.in].cons_norm = 1;
        }
        if (op.kind == OP_ATTN && op.has_x) { c->tens[op.xmid].cons_op = (int)i; c->tens[op.xmid].cons_norm = 2; }
    }
    // concatenations consumed by a norm: Q = [P1 | P2] with P1, P2 tensors of their own inside Q's storage
    for (size_t q = 0; q < c->tens.size(); ++q) {
        Tens& Q = c->tens[q];
        if (Q.cons_op < 0) continue;
        int p1 = -1, p2 = -1;
        for (size_t i = 0; i < c->tens.size(); ++i) {
            const Tens& P = c->tens[i];
            if (i == q || P.H != Q.H || P.W != Q.W || P.C >= Q.C) continue;
            if (P.off == Q.off) p1 = (int)i;
        }
        if (p1 < 0) continue;
        for (size_t i = 0; i < c->tens.size(); ++i) {
            const Tens& P = c->tens[i];
            if (i != q && P.H == Q.H && P.W == Q.W && P.C == Q.C - c->tens[p1].C &&
                P.off == Q.off + (long)c->tens[p1].C * Q.H * Q.W) p2 = (int)i;
        }
        if (p2 < 0) continue;
        Q.cat_a = p1; Q.cat_b = p2;
        c->tens[p1].cat_of = (int)q; c->tens[p2].cat_of = (int)q;
    }
    declare_all(c);
    const size_t MB = (size_t)cfg->max_batch;
    {   // 64-float guard bands: the vector halo loads of the convs may touch 1 float before / 3 after a tensor
        float *p0 = nullptr, *p1 = nullptr;
        if (dalloc(c, &p0, MB * c->per_sample + 128) || dalloc(c, &p1, MB * c->per_sample + 128)) return -1;
        c->arenaP = p0 + 64; c->arenaT = p1 + 64;
    }
    if (dalloc(c, &c->statsP, MB * c->stats_per_sample) || dalloc(c, &c->statsT, MB * c->stats_per_sample)) return -1;
    if (dalloc(c, &c->red, RED_BYTES) || dalloc(c, &c->red2, RED_BYTES)) return -1;
    {   // row partials of the epilogue statistics: [B][C][pixel tiles >= HW / 64][2] of the largest tensor
        long big = 0;
        for (const Tens& t : c->tens) big = std::max(big, (long)t.C * t.H * t.W);
        c->stpart_floats = MB * (size_t)(big / 64 + 1) * 2;
        if (dalloc(c, &c->stpart, c->stpart_floats) || dalloc(c, &c->stpart2, c->stpart_floats)) return -1;
        for (Tens& t : c->tens)          // kept tile partials of the concatenation parts: [B][C][<= HW / 64 tiles][2]
            if (t.cat_of >= 0) {
                t.keep_floats = MB * (size_t)t.C * ((size_t)t.H * t.W / 64 + 1) * 2;
                if (dalloc(c, &t.keep, t.keep_floats)) return -1;
            }
        const char* e = getenv("LOCO_FUSE_STATS");
        c->fuse_stats = !(e && atoi(e) == 0);
        e = getenv("LOCO_FUSE_LIN");
        c->fuse_lin = !(e && atoi(e) == 0);
        e = getenv("LOCO_FUSE_COT");
        c->fuse_cot = !(e && atoi(e) == 0) && cfg->act == ACT_SILU;     // the epilogue term is written for SiLU
        e = getenv("LOCO_DEEP1");
        c->deep1 = !(e && atoi(e) == 0);
        e = getenv("LOCO_FUSE_XATTN");
        c->fuse_xattn = !(e && atoi(e) == 0);
        const char* fa = getenv("LOCO_FLASH_ATTN");
        c->flash_attn = !(fa && atoi(fa) == 0);
        long dmax = 1;
        for (const Op& op : c->ops)
            if (op.kind == OP_ATTN || op.kind == OP_XFMR) dmax = std::max(dmax, (long)op.heads * c->tens[op.in].H * c->tens[op.in].W);
        c->attn_dmax = dmax;
        if (dalloc(c, &c->attn_delta, MB * (size_t)dmax)) return -1;
    }
    c->partial_floats = (size_t)64 << 20;   // 256 MB split-K workspace
    if (dalloc(c, &c->partial, c->partial_floats)) return -1;
    if (dalloc(c, &c->xin_buf, MB * c->n_in) || dalloc(c, &c->t_dev, 4)) return -1;
    if (dalloc(c, &c->eps_buf, MB * c->n_out) || dalloc(c, &c->gx0, MB * c->n_in) || dalloc(c, &c->ge, MB * c->n_out))
        return -1;
    if (dalloc(c, &c->tmpA, (size_t)64 * c->n_in)) return -1;
    if (dalloc(c, &c->G, 64 * 64) || dalloc(c, &c->Q, 64 * 64) || dalloc(c, &c->W, 64)) return -1;
    {
        size_t nblk = ((size_t)c->n_in + 255) / 256;
        if (dalloc(c, &c->gscratch, nblk * 64 * 64 + 4096)) return -1;
    }
    if (dalloc(c, &c->alphas, 256) || dalloc(c, &c->cond_add, (size_t)cfg->ch * 4)) return -1;
    {
        float2* sx0 = nullptr;
        if (dalloc(c, &sx0, (size_t)c->sx_total + 64)) return -1;
        c->sxcache = sx0 + 32;
    }
    {
        const char* e = getenv("LOCO_PRECISION");
        c->prec = (e && std::string(e) == "f32") ? 0 : (e && std::string(e) == "f16") ? 2 : 1;   // default: split-bf16
        const char* t = getenv("LOCO_BF16_TILE");
        if (t) g_bf16_tile_override = atoi(t);
    }
    if (dalloc(c, &c->mask2, (size_t)c->n_out)) return -1;
    if (dalloc(c, &c->mask, (size_t)c->n_out) || dalloc(c, &c->mask_idx, (size_t)c->n_out) ||
        dalloc(c, &c->mask_L_dev, 4)) return -1;
    HIPCHK(c, hipEventCreate(&c->ev0));
    HIPCHK(c, hipEventCreate(&c->ev1));
    {
        const char* e = getenv("LOCO_STREAMS");
        c->n_streams = (e && atoi(e) == 2) ? 2 : 1;
        HIPCHK(c, hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->cap_st, hipStreamNonBlocking));
        const char* gl = getenv("LOCO_GEMM_LOWP");
        if (gl) c->gemm_lowp = atoi(gl) != 0;
        const char* gr = getenv("LOCO_GRAPH");
        c->graph_on = gr && atoi(gr) == 1;
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    return 0;
}


int loco_fork(loco_ctx* parent, int32_t max_batch, loco_ctx** out) {
    if (!parent || !out) return -2;
    *out = nullptr;
    if (finalize_params(parent)) return -3;
    loco_ctx* root = parent->weights_of ? parent->weights_of : parent;      // forks of a fork share the root's store
    loco_unet_cfg cfg = parent->cfg;
    if (max_batch > 0) cfg.max_batch = max_batch;
    loco_ctx* c = nullptr;
    const int rc = loco_create(&cfg, &c);
    *out = c;
    if (rc) return rc;
    if (c->ops.size() != root->ops.size()) { c->err = "loco_fork: programs differ"; return -2; }
    // the parameter pointers of every operator (program-dependent fields -- tensor ids, statistics offsets -- are the child's own,
    // identical by construction: same configuration, same build_program)
    for (size_t i = 0; i < c->ops.size(); ++i) {
        Op& d = c->ops[i];
        const Op& s_ = root->ops[i];
        d.c1 = s_.c1; d.c2 = s_.c2; d.nin = s_.nin; d.qkvc = s_.qkvc; d.proj = s_.proj; d.conv = s_.conv;
        d.xqc = s_.xqc; d.xproj = s_.xproj;
        d.pj_in = s_.pj_in; d.to_out1 = s_.to_out1; d.to_out2 = s_.to_out2; d.ff1 = s_.ff1; d.ff2 = s_.ff2; d.pj_out = s_.pj_out;
        d.n1.gamma = s_.n1.gamma; d.n1.beta = s_.n1.beta; d.n2.gamma = s_.n2.gamma; d.n2.beta = s_.n2.beta;
        d.nx.gamma = s_.nx.gamma; d.nx.beta = s_.nx.beta;
        d.xkw = s_.xkw; d.xkb = s_.xkb; d.xvw = s_.xvw; d.xvb = s_.xvb; d.xng = s_.xng; d.xnb = s_.xnb;
        for (int k = 0; k < 3; ++k) { d.lng[k] = s_.lng[k]; d.lnb[k] = s_.lnb[k]; }
        d.tproj_off = s_.tproj_off;
        if (s_.xK) {      // the projected prompt states are per context (loco_set_context)
            const size_t kv = (size_t)c->tens[d.in].C * c->ctx_Lp;
            if (dalloc(c, &d.xK, kv) || dalloc(c, &d.xV, kv)) return -1;
        }
    }
    c->td0w = root->td0w; c->td0b = root->td0b; c->td1w = root->td1w; c->td1b = root->td1b; c->freq = root->freq;
    c->tp_w = root->tp_w; c->tp_b = root->tp_b; c->tproj_total = root->tproj_total;
    if (dalloc(c, &c->tact, (size_t)cfg.ch * 4) || dalloc(c, &c->tproj, (size_t)(c->tproj_total > 0 ? c->tproj_total : 1))) return -1;
    c->flops = root->flops;
    c->prec = parent->prec;
    c->params.clear(); c->param_order.clear();
    c->finalized = true;
    c->weights_of = root;
    ++root->forks;
    return 0;
}

static void destroy_now(loco_ctx* c);
void loco_destroy(loco_ctx* c) {
    if (!c) return;
    if (c->forks > 0) { c->zombie = true; return; }      // its forks still read its parameters: freed with the last of them
    loco_ctx* root = c->weights_of;
    destroy_now(c);
    if (root && --root->forks == 0 && root->zombie) destroy_now(root);
}
static void destroy_now(loco_ctx* c) {
    for (float* p : c->owned) (void)hipFree(p);
    for (unsigned char* p : c->gemm_ws) if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->st2) (void)hipStreamDestroy(c->st2);
    drop_graphs(c);
    if (c->cap_st) (void)hipStreamDestroy(c->cap_st);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
}

const char* loco_last_error(loco_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int loco_load_param(loco_ctx* c, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                    int32_t is_device) {
    if (!c) return -2;
    if (c->finalized) { c->err = "parameters already finalised"; return -2; }
    auto it = c->params.find(name);
    if (it == c->params.end()) { c->err = std::string("unknown parameter: ") + name; return -3; }
    HostParam& hp = it->second;
    if ((int)hp.shape.size() != ndim) { c->err = std::string("rank mismatch for ") + name; return -3; }
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        if (hp.shape[i] != shape[i]) { c->err = std::string("shape mismatch for ") + name; return -3; }
        cnt *= (size_t)shape[i];
    }
    hp.data.resize(cnt);
    if (is_device) HIPCHK(c, hipMemcpy(hp.data.data(), data, cnt * sizeof(float), hipMemcpyDeviceToHost));
    else std::memcpy(hp.data.data(), data, cnt * sizeof(float));
    hp.loaded = true;
    return 0;
}

int loco_params_missing(loco_ctx* c) {
    if (!c) return -2;
    if (c->finalized) return 0;
    int miss = 0;
    for (auto& n : c->param_order)
        if (!c->params[n].loaded) { if (!miss) c->err = "missing parameter: " + n; ++miss; }
    return miss;
}

// Capture one denoiser evaluation of batch B (fixed input xin_buf, timestep read from t_dev, output in the primal
// arena) on the engine's own stream; nothing executes during the capture.
static int capture_forward(loco_ctx* c, int B, loco_ctx::FwdGraph& g) {
    HIPCHK(c, hipStreamBeginCapture(c->cap_st, hipStreamCaptureModeThreadLocal));
    int rc = forward_pass(c, c->xin_buf, 0.f, B, c->arenaP, c->statsP, c->cap_st, c->t_dev);
    hipError_t e = hipStreamEndCapture(c->cap_st, &g.graph);
    if (rc) { if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; } return rc; }
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    return 0;
}

int loco_unet_forward(loco_ctx* c, const float* x, float t, int32_t B, float* eps, void* stream) {
    if (!c) return -2;
    if (B < 1 || B > c->cfg.max_batch) { c->err = "batch exceeds max_batch"; return -2; }
    if (finalize_params(c)) return -3;
    hipStream_t st = (hipStream_t)stream;
    c->primal_ok = false;
    if (c->graph_on && !c->prof_on) {
        // key: batch, condition on/off, arithmetic (all three change the launch list)
        loco_ctx::FwdGraph& g = c->fwd_graphs[(B << 3) | (c->has_cond ? 4 : 0) | c->prec];
        // the first evaluation of a key runs eagerly (one-time lazy setup inside the launchers), the second is captured
        if (!g.exec && g.calls++ >= 1 && capture_forward(c, B, g)) return -1;
        if (g.exec) {
            launch_copy(x, c->n_in, c->xin_buf, c->n_in, 0, B, c->n_in, st);
            launch_set_scalar(c->t_dev, t, st);
            HIPCHK(c, hipGraphLaunch(g.exec, st));
            c->primal_B = B;
            launch_copy(c->arenaP + c->tens[c->eps_t].off, c->per_sample, eps, c->n_out, 0, B, c->n_out, st);
            HIPCHK(c, hipGetLastError());
            return 0;
        }
    }
    if (forward_pass(c, x, t, B, c->arenaP, c->statsP, st)) return -1;
    c->primal_B = B;
    launch_copy(c->arenaP + c->tens[c->eps_t].off, c->per_sample, eps, c->n_out, 0, B, c->n_out, st);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static void sched_coeffs(float at, float at_next, float eta, float* s1, float* s2, float* s3, float* ce, float* cn) {
    // fp32 arithmetic in the op order of reference utils.py:362-374
    *s1 = std::sqrt(1.0f - at); *s2 = std::sqrt(at); *s3 = std::sqrt(at_next);
    *cn = 0.f;
    if (eta == 0.f) {
        *ce = std::sqrt(1.0f - at_next);
    } else {
        float sigma = std::sqrt((1.0f - at / at_next) * (1.0f - at_next) / (1.0f - at));
        *ce = std::sqrt(1.0f - at_next - eta * (sigma * sigma));
        *cn = eta * sigma;
    }
}

int loco_sched_step(loco_ctx* c, const float* x, const float* et, float at, float at_next, float eta,
                    const float* noise, int64_t count, float* x_next, float* x0_out, void* stream) {
    if (!c) return -2;
    if (eta != 0.f && !noise) { c->err = "eta != 0 needs a noise tensor"; return -2; }
    float s1, s2, s3, ce, cn;
    sched_coeffs(at, at_next, eta, &s1, &s2, &s3, &ce, &cn);
    launch_ddim_step(x, et, eta == 0.f ? nullptr : noise, x_next, x0_out, (long)count, s2, s1, s3, ce, cn,
                     (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_ddim_step(loco_ctx* c, const float* x, float t, float at, float at_next, float eta, const float* noise,
                   int32_t B, float* x_next, void* stream) {
    if (!c) return -2;
    if (eta != 0.f && !noise) { c->err = "eta != 0 needs a noise tensor"; return -2; }
    if (c->n_out != c->n_in) { c->err = "loco_ddim_step: this network is not a denoiser (output and input sizes differ)"; return -2; }
    int rc = loco_unet_forward(c, x, t, B, c->eps_buf, stream);
    if (rc) return rc;
    return loco_sched_step(c, x, c->eps_buf, at, at_next, eta, noise, (int64_t)B * c->n_in, x_next, nullptr, stream);
}

int loco_pmp_primal(loco_ctx* c, const float* x, float t, float at, const uint8_t* mask, int32_t use_et,
                    void* stream) {
    if (!c) return -2;
    if (finalize_params(c)) return -3;
    if (c->n_out != c->n_in && !use_et) {
        c->err = "loco_pmp_primal: a network whose output size differs from its input has no x0 combination; use_et = 1 "
                 "(raw network Jacobian)";
        return -2;
    }
    hipStream_t st = (hipStream_t)stream;
    if (forward_pass(c, x, t, 1, c->arenaP, c->statsP, st)) return -1;
    c->primal_B = 1;
    if (c->prec >= 1) {
        // {S, xhat} cache for the tangent / cotangent staging of the split-bf16 convs
        for (auto& op : c->ops) {
            if (op.kind != OP_RES && op.kind != OP_OUT) continue;
            const Tens& ti = c->tens[op.in];
            const int HW = ti.H * ti.W, G = c->cfg.gn_groups;
            NS s1 = nstats(c, c->statsP, op.n1);
            launch_gn_cache(c->arenaP + ti.off, op.n1.C, HW, op.n1.C / G, s1.sc, s1.sh, s1.mr,
                            c->sxcache + op.n1.sx_off, st, c->cfg.act);
            if (op.kind == OP_RES) {
                const Tens& th = c->tens[op.h1];
                NS s2 = nstats(c, c->statsP, op.n2);
                launch_gn_cache(c->arenaP + th.off, op.n2.C, th.H * th.W, op.n2.C / G, s2.sc, s2.sh, s2.mr,
                                c->sxcache + op.n2.sx_off, st, c->cfg.act);
            }
        }
    }
    if (use_et) { c->p_cv = 0.f; c->p_ce = 1.f; }
    else { c->p_cv = 1.0f / std::sqrt(at); c->p_ce = -std::sqrt(1.0f - at) / std::sqrt(at); }
    c->has_mask = (mask != nullptr);
    c->has_mask2 = false;
    c->mask_L = c->n_out;
    if (mask) {
        // masked-latent gather list built on the device (ordered prefix-sum compaction, one launch, no host sync);
        // L is read back lazily by loco_mask_count / loco_mask_gather
        HIPCHK(c, hipMemcpyAsync(c->mask, mask, (size_t)c->n_out, hipMemcpyDeviceToDevice, st));
        launch_mask_compact(c->mask, c->n_out, c->mask_idx, c->mask_L_dev, st);
        c->mask_L = -1;
        c->mask_stream = st;
    }
    HIPCHK(c, hipGetLastError());
    c->primal_ok = true;
    return 0;
}

int loco_pmp_set_second_mask(loco_ctx* c, const uint8_t* mask2, int32_t from_row, void* stream) {
    if (!c) return -2;
    if (!c->primal_ok) { c->err = "loco_pmp_primal has not been called"; return -2; }
    if (!mask2) { c->has_mask2 = false; return 0; }
    if (!c->has_mask) { c->err = "a second mask needs a first one (loco_pmp_primal with mask)"; return -2; }
    if (from_row < 0) { c->err = "from_row must be >= 0"; return -2; }
    HIPCHK(c, hipMemcpyAsync(c->mask2, mask2, (size_t)c->n_out, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    c->has_mask2 = true;
    c->mask2_from = from_row;
    return 0;
}

int loco_pmp_jvp(loco_ctx* c, const float* V, int32_t k, float* U, void* stream) {
    if (!c) return -2;
    if (!c->primal_ok) { c->err = "loco_pmp_primal has not been called"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    const int MB = c->cfg.max_batch;
    for (int b0 = 0; b0 < k; b0 += MB) {
        int B = (k - b0 < MB) ? k - b0 : MB;
        int rc = run_lanes(c, B, st, [&](int s0, int nb, hipStream_t ls) -> int {
            const float* Vc = V + (long)(b0 + s0) * c->n_in;
            if (tangent_pass(c, Vc, nb, ls)) return -1;
            // U = mask * (cv*V + ce*dEps); dEps lives strided in arena T -> gather through eps_buf
            launch_copy(c->arenaT + c->tens[c->eps_t].off, c->per_sample, c->eps_buf, c->n_out, 0, nb, c->n_out, ls);
            launch_masked_axpby(c->n_out == c->n_in ? Vc : nullptr, c->eps_buf, c->has_mask ? c->mask : nullptr, c->p_cv,
                                c->p_ce, U + (long)(b0 + s0) * c->n_out, nb, c->n_out, ls,
                                c->has_mask2 ? c->mask2 : nullptr, (long)c->mask2_from - (b0 + s0));
            return 0;
        });
        if (rc) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_pmp_vjp(loco_ctx* c, const float* U, int32_t k, float* A, void* stream) {
    if (!c) return -2;
    if (!c->primal_ok) { c->err = "loco_pmp_primal has not been called"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    const int MB = c->cfg.max_batch;
    for (int b0 = 0; b0 < k; b0 += MB) {
        int B = (k - b0 < MB) ? k - b0 : MB;
        int rc = run_lanes(c, B, st, [&](int s0, int nb, hipStream_t ls) -> int {
            const bool same = c->n_out == c->n_in;
            launch_cot_seed(U + (long)(b0 + s0) * c->n_out, c->has_mask ? c->mask : nullptr, c->p_cv, c->p_ce, c->ge,
                            same ? c->gx0 : nullptr, nb, c->n_out, ls, c->has_mask2 ? c->mask2 : nullptr,
                            (long)c->mask2_from - (b0 + s0));
            return cotangent_pass(c, c->ge, same ? c->gx0 : nullptr, A + (long)(b0 + s0) * c->n_in, nb, ls);
        });
        if (rc) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_orthonormalize(loco_ctx* c, float* A, int32_t k, int64_t n, float* s, void* stream) {
    if (!c) return -2;
    if (k < 1 || k > 64 || n > c->n_in) { c->err = "orthonormalize: need k <= 64 and n <= C*H*W"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    launch_gram(A, k, n, c->G, c->gscratch, st);
    launch_jacobi_eig(c->G, k, c->W, c->Q, st);
    launch_rotate_rows(A, c->tmpA, k, n, c->Q, c->W, 0, st);
    // singular values of the input = sqrt(eigenvalues of A A^T)
    HIPCHK(c, hipMemcpyAsync(A, c->tmpA, (size_t)k * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_sign_fix(A, k, n, s, c->W, reinterpret_cast<float*>(c->red), st);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_qr_rows(loco_ctx* c, float* A, int32_t k, int64_t n, void* stream) {
    if (!c) return -2;
    if (k < 1 || k > 64 || n > c->n_in) { c->err = "qr_rows: need k <= 64 and n <= C*H*W"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    for (int rep = 0; rep < 2; ++rep) {   // CholeskyQR2
        launch_gram(A, k, n, c->G, c->gscratch, st);
        launch_cholesky(c->G, k, st);
        launch_trsm_rows(A, c->tmpA, k, n, c->G, st);
        HIPCHK(c, hipMemcpyAsync(A, c->tmpA, (size_t)k * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_convergence(loco_ctx* c, const float* Vprev, const float* V, int64_t count, float atol, float* out2,
                     void* stream) {
    if (!c) return -2;
    launch_convergence(Vprev, V, count, atol, 1e-5f, out2, c->gscratch, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_convergence_rows(loco_ctx* c, const float* Vprev, const float* V, int32_t k, int64_t n, float atol,
                          float* out2, void* stream) {
    if (!c) return -2;
    if (k < 1 || k > 64 || n < 1) { c->err = "convergence_rows: need 1 <= k <= 64"; return -2; }
    // the kernel leaves k * min(64, ceil(n / 256)) * 4 doubles in gscratch, which is sized for rows of the context's own width
    if (n > c->n_in) { c->err = "convergence_rows: rows longer than the context's input"; return -2; }
    {
        const size_t nseg = (size_t)((n + 255) / 256) < 64 ? (size_t)((n + 255) / 256) : 64;
        const size_t cap = (((size_t)c->n_in + 255) / 256) * 64 * 64 + 4096;       // doubles in gscratch (finalize_params)
        if ((size_t)k * nseg * 4 > cap) { c->err = "convergence_rows: k rows of this length exceed the reduction workspace"; return -2; }
    }
    launch_convergence_rows(Vprev, V, k, n, atol, 1e-5f, out2, c->gscratch, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_null_project(loco_ctx* c, const float* Vm, int32_t k, const float* Vn, int32_t k0, int64_t n, float* out,
                      void* stream) {
    if (!c) return -2;
    if (k < 1 || k > 64 || k0 > 64 || n > c->n_in) { c->err = "null_project: bad sizes"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    if (Vn && k0 > 0) {
        launch_cross_gram(Vn, k0, Vm, k, n, c->G, c->gscratch, st);
        launch_project_rows(Vm, k, Vn, k0, n, c->G, out, st);
    } else if (out != Vm) {
        HIPCHK(c, hipMemcpyAsync(out, Vm, (size_t)k * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    launch_normalize_rows(out, k, n, c->gscratch, st);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_edit_axpy(loco_ctx* c, const float* x, const float* v, const float* alphas, int32_t B, int64_t n, float* out,
                   void* stream) {
    if (!c) return -2;
    if (B > 256) { c->err = "edit_axpy: B > 256"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(c, hipMemcpyAsync(c->alphas, alphas, B * sizeof(float), hipMemcpyHostToDevice, st));
    launch_edit_axpy(x, v, c->alphas, B, n, out, st);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static long mask_count_lazy(loco_ctx* c) {
    if (c->mask_L < 0) {
        int L = 0;
        if (hipMemcpyAsync(&L, c->mask_L_dev, sizeof(int), hipMemcpyDeviceToHost, c->mask_stream) != hipSuccess ||
            hipStreamSynchronize(c->mask_stream) != hipSuccess) return -1;
        c->mask_L = L;
    }
    return c->mask_L;
}
int64_t loco_mask_count(loco_ctx* c) { return c ? mask_count_lazy(c) : -1; }

int loco_mask_gather(loco_ctx* c, const float* U, int32_t k, float* out, void* stream) {
    if (!c) return -2;
    if (!c->primal_ok) { c->err = "loco_pmp_primal has not been called"; return -2; }
    hipStream_t st = (hipStream_t)stream;
    if (!c->has_mask) {
        HIPCHK(c, hipMemcpyAsync(out, U, (size_t)k * c->n_out * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    const long L = mask_count_lazy(c);
    if (L < 0) { c->err = "mask count readback failed"; return -1; }
    if (L > 0) launch_mask_gather(U, c->mask_idx, L, c->n_out, k, out, st);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_clock_stamp(loco_ctx* c, uint64_t* out2, void* stream) {
    if (!c || !out2) return -2;
    launch_clock_stamp(reinterpret_cast<unsigned long long*>(out2), (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

double loco_unet_flops(loco_ctx* c) {
    if (!c) return 0.0;
    if (!c->finalized && finalize_params(c)) return 0.0;
    return c->flops;
}
int64_t loco_workspace_bytes(loco_ctx* c) { return c ? (int64_t)c->bytes : 0; }

int loco_timer_start(loco_ctx* c, void* stream) {
    if (!c) return -2;
    HIPCHK(c, hipEventRecord(c->ev0, (hipStream_t)stream));
    return 0;
}
int loco_timer_stop(loco_ctx* c, void* stream, float* ms) {
    if (!c) return -2;
    HIPCHK(c, hipEventRecord(c->ev1, (hipStream_t)stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

int loco_set_precision(loco_ctx* c, int32_t mode) {
    if (!c) return -2;
    if (mode < 0 || mode > 2) { c->err = "precision mode must be 0 (f32), 1 (bf16x3) or 2 (f16)"; return -2; }
    c->prec = mode;
    c->primal_ok = false;   // the {S, xhat} cache belongs to the mode it was built in
    return 0;
}
int loco_get_precision(loco_ctx* c) { return c ? c->prec : -2; }

int loco_set_cond(loco_ctx* c, const float* emb_add, void* stream) {
    if (!c) return -2;
    c->primal_ok = false;           // cached activations belong to the previous condition
    if (!emb_add) { c->has_cond = false; return 0; }
    HIPCHK(c, hipMemcpyAsync(c->cond_add, emb_add, (size_t)c->cfg.ch * 4 * sizeof(float), hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    c->has_cond = true;
    return 0;
}

int loco_set_context(loco_ctx* c, const float* tokens, void* stream) {
    if (!c) return -2;
    if (c->cfg.context_dim <= 0) { c->err = "this architecture has no cross-attention stages (context_dim = 0)"; return -2; }
    if (!tokens) { c->err = "loco_set_context: null tokens"; return -2; }
    if (finalize_params(c)) return -3;
    hipStream_t st = (hipStream_t)stream;
    c->primal_ok = false;
    const int L = c->cfg.context_len, D = c->cfg.context_dim, Lp = c->ctx_Lp;
    if (!c->ctx_colbias) {
        std::vector<float> cb(Lp, -1e30f);
        for (int l = 0; l < L; ++l) cb[l] = 0.f;
        if (upload(c, &c->ctx_colbias, cb)) return -1;
    }
    if (c->cfg.added_kv && !c->ctx_norm && dalloc(c, &c->ctx_norm, (size_t)L * D)) return -1;
    for (auto& op : c->ops) {
        if (!((op.kind == OP_ATTN && (op.has_x || op.added_kv)) || op.kind == OP_XFMR)) continue;
        const int C = c->tens[op.in].C;
        const float* tok = tokens;
        if (op.added_kv) {      // the block's own GroupNorm over the states [D][L] (norm_encoder), then encoder_kv
            launch_ctx_groupnorm(tokens, L, D, c->cfg.gn_groups, c->cfg.gn_eps, op.xng, op.xnb, c->ctx_norm, st);
            tok = c->ctx_norm;
        }
        for (int w = 0; w < 2; ++w) {
            float* dst = w ? op.xV : op.xK;
            HIPCHK(c, hipMemsetAsync(dst, 0, (size_t)C * Lp * sizeof(float), st));
            // dst[c][l] = sum_d W[c][d] tokens[l][d] + b[c]
            GemmArgs g; std::memset(&g, 0, sizeof(g));
            g.A = w ? op.xvw : op.xkw; g.sam = D; g.sak = 1;
            g.Bm = tok; g.sbk = 1; g.sbn = D;
            g.C = dst; g.scm = Lp; g.scn = 1;
            g.bias = w ? op.xvb : op.xkb;
            g.M = C; g.N = L; g.K = D; g.batch = 1; g.alpha = 1.f; g.beta = 0.f;
            launch_gemm(g, st);
        }
    }
    HIPCHK(c, hipGetLastError());
    c->has_ctx = true;
    return 0;
}

int loco_masked_axpby(loco_ctx* c, const float* V, const float* E, float cv, float ce, int32_t k, float* out,
                      void* stream) {
    if (!c) return -2;
    if (!c->primal_ok) { c->err = "loco_pmp_primal has not been called"; return -2; }
    launch_masked_axpby(V, E, c->has_mask ? c->mask : nullptr, cv, ce, out, k, c->n_out, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_latent_sample(loco_ctx* c, const float* moments, const float* noise, float scale, int32_t B, float* z, void* stream) {
    if (!c) return -2;
    if (c->cfg.arch != 3 || (c->cfg.out_ch & 1)) { c->err = "latent_sample: the context must be a latent encoder (arch 3) with out_ch = 2 z"; return -2; }
    if (B < 1 || !moments || !z) { c->err = "latent_sample: bad arguments"; return -2; }
    launch_latent_sample(moments, noise, scale, z, B, c->n_out / 2, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int loco_lincomb(loco_ctx* c, const float* const* src, const float* coef, int32_t n, float* out, int64_t count,
                 void* stream) {
    if (!c) return -2;
    if (n < 1 || n > 4) { c->err = "lincomb: 1..4 terms"; return -2; }
    launch_lincomb(src, coef, n, out, (long)count, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Tuning hook: time one convolution shape on scratch data (random inputs), avg ms over `iters` launches.
#ifdef LOCO_DIAG     /* include/loco_hip_diag.h: tuning / bring-up hooks, only in the diag build */
}  // extern "C"
// well-formed split-bf16 weight records of random values (64 bytes: [hi k0-7|hi k8-15|lo k0-7|lo k8-15], pieces XOR-swizzled by
// (record index inside its tap block >> 2) & 3), so that diagnostic launches produce comparable numbers, not Inf / NaN
__global__ void diag_fill_records(unsigned char* rec, long nrec, int wpitch, unsigned seed, float scale) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const int idx = (int)(r % wpitch);
    unsigned short hi[16], lo[16];
    for (int k = 0; k < 16; ++k) {
        unsigned h = (unsigned)(r * 16 + k) * 2654435761u + seed * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float v = scale * ((float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f);
        const __bf16 hb = (__bf16)v;
        const __bf16 lb = (__bf16)(v - (float)hb);
        hi[k] = __builtin_bit_cast(unsigned short, hb);
        lo[k] = __builtin_bit_cast(unsigned short, lb);
    }
    const int sw = (idx >> 2) & 3;
    unsigned short* base = reinterpret_cast<unsigned short*>(rec + r * 64);
    for (int k = 0; k < 8; ++k) {
        base[((0 ^ sw) << 3) + k] = hi[k];
        base[((1 ^ sw) << 3) + k] = hi[8 + k];
        base[((2 ^ sw) << 3) + k] = lo[k];
        base[((3 ^ sw) << 3) + k] = lo[8 + k];
    }
}
extern "C" {
int loco_bench_conv(loco_ctx* c, int32_t cin, int32_t cout, int32_t H, int32_t W, int32_t B, int32_t mode,
                    int32_t taps, int32_t tile, int32_t iters, float* ms_avg, void* stream) {
    if (!c) return -2;
    if (finalize_params(c)) return -3;
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const long in_e = (long)cin * HW, out_e = (long)cout * HW;
    if ((in_e + out_e) * B > (long)c->cfg.max_batch * c->per_sample || 2 * in_e > (long)c->per_sample ||
        in_e > c->sx_total) { c->err = "bench_conv: shape exceeds the arenas"; return -2; }
    float* in = c->arenaT;
    float* out = c->arenaT + in_e * B;
    // LOCO_BENCH_ZERO (bit 1: activations, bit 2: weights): all-zero operands draw less power in the matrix pipe -- the time
    // against random data says how much of a launch the chip's clock management decides (MI355X_MICROARCH.md, DVFS give-back)
    const int bz = getenv("LOCO_BENCH_ZERO") ? atoi(getenv("LOCO_BENCH_ZERO")) : 0;
    const float zs = (bz & 1) ? 0.0f : 1.0f, zw = (bz & 2) ? 0.0f : 1.0f;
    launch_fill_random(in, in_e * B, 1u, zs, st);
    launch_fill_random(c->arenaP, in_e, 2u, zs, st);
    launch_fill_random(reinterpret_cast<float*>(c->sxcache), 2 * in_e, 3u, zs, st);
    launch_fill_random(c->statsP, c->stats_per_sample, 4u, 1.0f, st);
    launch_fill_random(c->statsT, c->stats_per_sample * B, 5u, 0.01f, st);
    // weights: any conv of matching size is fine for timing; synthesise records in the split-K workspace
    size_t wfl = (size_t)((cin + 15) / 16) * 16 * taps * ((cout + 31) & ~31);
    if (wfl * 2 > c->partial_floats) { c->err = "bench_conv: weights exceed workspace"; return -2; }
    float* wf = c->partial + c->partial_floats - wfl * 2;
    launch_fill_random(wf, (long)wfl * 2, 6u, 0.05f * zw, st);
    if (c->prec == 1) {      // split-bf16 records: well-formed ones
        const long nrec = (long)((cin + 15) / 16) * taps * ((cout + 31) & ~31);
        hipLaunchKernelGGL(diag_fill_records, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<unsigned char*>(wf), nrec, (cout + 31) & ~31, 7u, 0.05f * zw);
    }
    ConvArgs a; conv_defaults(a);
    a.in = in; a.in_bs = in_e; a.Cin = cin; a.Hin = H; a.Win = W;
    a.prim = c->arenaP; a.prim_bs = 0; a.sx = c->sxcache;
    a.w = wf; a.wb = wf; a.wh = wf;
    a.out = out; a.out_bs = out_e; a.Cout = cout; a.Hout = H; a.Wout = W; a.B = B;
    a.mode = mode; a.pad = taps == 9 ? 1 : 0; a.in_padded = 1; a.taps = taps;
    a.sc = c->statsP; a.sh = c->statsP + cin; a.scsh_bs = 0; a.mr = c->statsP + 2 * cin; a.mr_bs = 0;
    a.gamma_ = c->statsP; a.tst = c->statsT; a.tst_bs = c->stats_per_sample; a.cpg = cin / c->cfg.gn_groups;
    a.tc = c->statsT + 64; a.tc_bs = c->stats_per_sample;
    if (a.cpg < 1) a.cpg = 1;
    a.nsplit = 1; a.partial = c->partial;
    a.partial_floats = c->partial_floats - wfl * 2;                            // (the synthetic weights sit at the end of the workspace)
    if (getenv("LOCO_BENCH_COT") && atoi(getenv("LOCO_BENCH_COT")) && taps == 1) {
        // the ResBlock shortcut's cotangent form: norm-cotangent term in the epilogue (synthetic operands behind the output tensor)
        if ((in_e + 2 * out_e) * B > (long)c->cfg.max_batch * c->per_sample || out_e > c->sx_total ||
            2L * cout > c->stats_per_sample) { c->err = "bench_conv: cot operands exceed the arenas"; return -2; }
        launch_fill_random(out + out_e * B, out_e * B, 8u, zs, st);
        if (out_e > in_e) launch_fill_random(reinterpret_cast<float*>(c->sxcache), 2 * out_e, 3u, zs, st);
        a.cot_d = out + out_e * B; a.cot_d_bs = out_e; a.cot_sx = c->sxcache; a.cot_tc = c->statsT; a.cot_tc_bs = c->stats_per_sample;
    }
    if (getenv("LOCO_BENCH_ACC") && atoi(getenv("LOCO_BENCH_ACC"))) a.accumulate = 1;
    if (c->prec == 1 && taps == 1) conv_gemm_plan(a);
    if (c->prec == 1 && taps == 9) conv_pers_plan(a);
    if (const char* e = getenv("LOCO_DUAL_WHATIF")) a.no_deep = atoi(e);      // stamp build of the dual tile only (bits 2 / 4)
    int saved = g_bf16_tile_override;
    g_bf16_tile_override = tile;
    ConvArgs parts[2];
    const int nparts = conv_lowp_plan(a, taps, c->prec, parts);
    auto run = [&]() {
        for (int pi = 0; pi < nparts; ++pi) {
            if (c->prec == 1) launch_conv_bf16x3(parts[pi], taps, st);
            else if (c->prec == 2) launch_conv_f16(parts[pi], taps, st);
            else launch_conv(parts[pi], taps, st);
        }
        launch_conv_splitk_reduce(a, st);
    };
    run();
    HIPCHK(c, hipEventRecord(c->ev0, st));
    for (int i = 0; i < iters; ++i) run();
    HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    g_bf16_tile_override = saved;
    c->bench_out = out; c->bench_out_count = (int64_t)out_e * B;
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_avg = ms / iters;
    HIPCHK(c, hipGetLastError());
    return 0;
}

#endif

int loco_set_streams(loco_ctx* c, int32_t n) {
    if (!c || (n != 1 && n != 2)) return -2;
    c->n_streams = n;
    return 0;
}

int loco_set_chip_share(loco_ctx* c, int32_t n) {
    if (!c || n < 1 || n > 8) return -2;
    c->chip_share = n;
    return 0;
}

int loco_set_side_stream(loco_ctx* c, void* stream) {
    if (!c) return -2;
    c->st2_user = (hipStream_t)stream;          // nullptr: back to the context's own stream
    return 0;
}

int loco_profile_enable(loco_ctx* c, int32_t on) {
    if (!c) return -2;
    c->prof_on = on != 0;
    c->prof_shapes = on == 2;     // 2: aggregate per layer shape instead of per kernel variant
    if (on) { c->prof.clear(); c->ev_used = 0; }
    return 0;
}

int loco_profile_report(loco_ctx* c, char* buf, int64_t cap) {
    if (!c || !buf || cap < 1) return -2;
    HIPCHK(c, hipDeviceSynchronize());
    std::map<std::string, std::array<double, 3>> agg;   // launches, ms, flops
    for (auto& r : c->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        std::string key = r.name;
        if (c->prof_shapes) {
            char sb[96];
            snprintf(sb, sizeof(sb), "|t%d_m%d_ci%d_co%d_h%d_b%d_s%d", r.taps, r.mode, r.cin, r.cout, r.h, r.b, r.ns);
            key += sb;
        }
        auto& a = agg[key];
        a[0] += 1.0; a[1] += ms; a[2] += r.flops;
    }
    std::string out;
    char line[256];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s %.0f %.6f %.6e\n", kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2]);
        out += line;
    }
    if ((int64_t)out.size() + 1 > cap) { c->err = "profile buffer too small"; return -4; }
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

#ifdef LOCO_DIAG
int64_t loco_debug_tensor(loco_ctx* c, const char* name, float* dst, int64_t cap, void* stream) {
    if (!c) return -2;
    std::string nm(name);
    if (nm == "bench_out") {     // the output tensor of the last loco_bench_conv ([B][Cout][H][W])
        if (!c->bench_out) { c->err = "no loco_bench_conv yet"; return -3; }
        int64_t cnt = cap < c->bench_out_count ? cap : c->bench_out_count;
        HIPCHK(c, hipMemcpyAsync(dst, c->bench_out, (size_t)cnt * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return cnt;
    }
    if (nm == "workspace") {     // head of the split-K workspace (diagnostic kernels leave their cycle stamps there)
        int64_t cnt = cap < (int64_t)c->partial_floats ? cap : (int64_t)c->partial_floats;
        HIPCHK(c, hipMemcpyAsync(dst, c->partial, (size_t)cnt * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return cnt;
    }
    int arena = 0;   // "T:" prefix selects the tangent/cotangent arena, "h1:" the conv1 output of a block
    if (nm.rfind("T:", 0) == 0) { arena = 1; nm = nm.substr(2); }
    bool want_h1 = false;
    if (nm.rfind("h1:", 0) == 0) { want_h1 = true; nm = nm.substr(3); }
    for (auto& op : c->ops) {
        if (op.name != nm) continue;
        int id = want_h1 ? op.h1 : op.out;
        if (id < 0) return -3;
        const Tens& t = c->tens[id];
        int64_t cnt = (int64_t)t.C * t.H * t.W;
        int B = arena ? 1 : (c->primal_B > 0 ? c->primal_B : 1);
        if (cnt * B > cap) { c->err = "debug buffer too small"; return -4; }
        float* base = (arena ? c->arenaT : c->arenaP) + t.off;
        launch_copy(base, c->per_sample, dst, cnt, 0, B, cnt, (hipStream_t)stream);
        return cnt * B;
    }
    c->err = "no such op: " + nm;
    return -3;
}

#endif

}  // extern "C"
