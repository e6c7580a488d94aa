/*
 * loco_hip.h -- C ABI of libloco_hip.so, the MI355X (gfx950) engine behind the
 * LOCO-Edit null-space-projection hot path.
 *
 * The reference (ChicyChen/LOCO-Edit) has no FFI layer: its boundary is Python
 * duck typing.  Each entry point below names the
 * reference interface it replaces; the Python host (loco-edit_amd/) binds them
 * with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: every tensor pointer is a DEVICE pointer owned by the caller
 * (torch `tensor.data_ptr()`), fp32, contiguous NCHW unless stated.  The ctx
 * owns only its parameter copies, workspace and activation caches.  All work is
 * enqueued on the caller-supplied hipStream_t (passed as void*).  Return 0 on
 * success, negative on error (message via loco_last_error).  One ctx per
 * (process, GPU); not thread-safe.  No exceptions cross the ABI.
 */
#ifndef LOCO_HIP_H
#define LOCO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct loco_ctx loco_ctx;

/* Architecture of the denoiser: reference src/configs/custom_celeba_ddpm.yml
 * `model:` block + DDPM.__init__ (src/models/ddpm/diffusion.py:22-126). */
typedef struct loco_unet_cfg {
    int32_t struct_size;       /* = sizeof(loco_unet_cfg) of the header the caller was built against; loco_create
                                  refuses a mismatch instead of reading past an older binding's struct */
    int32_t resolution;        /* data.image_size */
    int32_t in_channels;       /* model.in_channels */
    int32_t out_ch;            /* model.out_ch */
    int32_t ch;                /* model.ch */
    int32_t num_levels;        /* len(model.ch_mult) */
    int32_t ch_mult[8];        /* model.ch_mult */
    int32_t num_res_blocks;    /* model.num_res_blocks */
    int32_t num_attn_res;      /* len(model.attn_resolutions) */
    int32_t attn_resolutions[8];
    int32_t gn_groups;         /* 32  (diffusion.py:810) */
    float   gn_eps;            /* 1e-6 */
    int32_t max_batch;         /* largest image / probe batch one call may carry */
    /* 0: Ho-DDPM U-Net (models/ddpm/diffusion.py); 1: guided-diffusion / P2 U-Net
     * (models/guided_diffusion/unet.py:398-684 with P2_DICT, script_util.py:166-190:
     * scale-shift norm, ResBlock up/down, legacy multi-head attention, [cos,sin] embedding);
     * 2: latent decoder -- the network behind `self.vae.decode(z).sample` of the Stable Diffusion path
     * (src/modules/edit.py:750, 770-771; diffusers AutoencoderKL decoder, un-vendored): conv_in, mid block/attn/block,
     * up levels of num_res_blocks+1 ResnetBlocks + nearest-x2 conv, norm_out/SiLU/conv_out; no skips, no time
     * embedding.  `resolution` is the latent resolution, the output is [out_ch, resolution << (num_levels-1), same];
     * loco_unet_forward ignores t; loco_pmp_primal needs use_et = 1 (raw network Jacobian, mask on the OUTPUT) and
     * loco_pmp_jvp / _vjp then map [k, in] -> [k, out] / [k, out] -> [k, in]; loco_ddim_step is refused;
     * 3: latent encoder -- the network behind `self.vae.encode(x0).latent_dist` of the latent inversion
     * (src/modules/edit.py:594-597): conv_in, down levels of num_res_blocks ResnetBlocks + pad (0,1,0,1) conv stride 2,
     * mid block/attn/block, norm_out/SiLU/conv_out, 1x1 quant_conv.  `resolution` is the IMAGE resolution, the output is
     * the posterior's moments [out_ch = 2 z, resolution >> (num_levels-1), same]; used through loco_unet_forward (t ignored) */
    int32_t arch;
    int32_t num_head_channels; /* arch 1: channels per attention head (P2: 64) */
    int32_t learn_sigma;       /* arch 1: the head emits 2*out_ch channels, eps = first out_ch (unet.py:680-684) */
    /* arch 1, text-to-image stand-ins: when context_dim > 0 every attention block is followed by a text
     * cross-attention stage  h += proj(softmax(q(GN(h))^T k(ctx) / sqrt(d)) applied to v(ctx))  over the
     * context_len x context_dim encoder states given to loco_set_context (the role of encoder_hidden_states in
     * edit.py:636-674 / 1286-1373; the reference's cross-attention lives in un-vendored diffusers blocks) */
    int32_t context_dim;
    int32_t context_len;
    /* arch 1 variants of the same guided-diffusion skeleton (unet.py constructor switches).  The latent-diffusion /
     * Stable Diffusion v1 denoiser is arch 1 with scale_shift_norm = 0, resblock_updown = 0, num_heads = 8,
     * transformer_depth = 1, context_dim = 768 (859 520 964 parameters at 320 x (1,2,4,4)):
     *   scale_shift_norm  1: GN(h) * (1 + scale) + shift (unet.py:250-254, P2)   0: GN(h + emb_out) (unet.py:255-257)
     *   resblock_updown   1: ResBlock(up/down=True) between levels (P2)            0: Downsample / Upsample with a 3x3 conv
     *                        (stride 2 padding 1 / nearest x2 + conv; unet.py:83-142)
     *   num_heads         > 0: that many heads in every attention (head width = C / num_heads); else num_head_channels
     *   transformer_depth 0: AttentionBlock (unet.py:261-307) [+ the cross-attention stage when context_dim > 0]
     *                     1: SpatialTransformer of latent-diffusion (GroupNorm eps 1e-6 -> 1x1 proj_in -> LayerNorm ->
     *                        self-attention, LayerNorm -> cross-attention over the loco_set_context states, LayerNorm ->
     *                        GEGLU feed-forward, residuals -> 1x1 proj_out -> + input); needs context_dim > 0 */
    int32_t scale_shift_norm;
    int32_t resblock_updown;
    int32_t num_heads;
    int32_t transformer_depth;
    /* arch 1, the DeepFloyd-IF stage-I denoiser (`self.unet = self.stage_1.unet`, src/modules/edit.py:1213-1222: diffusers
     * UNet2DConditionModel with ResnetDownsampleBlock2D / SimpleCrossAttn*Block2D, un-vendored; the same tree as the
     * UNetModel of the deepfloyd_if package) = the guided-diffusion skeleton with scale-shift norm and ResBlock resampling plus:
     *   act        0: SiLU   1: exact (erf) GELU in every norm -> activation -> conv chain and in the time embedding
     *                 (`act_fn = "gelu"`; the per-block embedding projections read act(emb) once: `resnet_skip_time_act`)
     *   res_scale  ResBlock output = (shortcut + h) * res_scale (`resnet_out_scale_factor` = sqrt 2 -> 0.70710678);
     *                 0 reads as 1
     *   added_kv   1: every AttentionBlock attends over [text ; image] keys / values in ONE softmax
     *                 (`AttnAddedKVProcessor`: key = cat([add_k_proj(GN(ctx)), to_k(h)])): the context_len x context_dim
     *                 states of loco_set_context (after the host's `encoder_hid_proj`) go through the block's own
     *                 GroupNorm (`norm_encoder`) and `encoder_kv` projection; needs context_dim > 0, transformer_depth = 0 */
    int32_t act;
    float   res_scale;
    int32_t added_kv;
} loco_unet_cfg;

/* Library / device probes (no ctx). */
const char* loco_version(void);
int  loco_device_count(void);

/* Replaces PullBackDDPM(args) construction (diffusion.py:128-143). */
int  loco_create(const loco_unet_cfg* cfg, loco_ctx** out);
void loco_destroy(loco_ctx* ctx);
/* A second context on the SAME parameters (round 6): the reference runs all classifier-free-guidance branches through ONE
 * U-Net object -- one set of weights (src/modules/edit.py:1319-1322, :655-667).  The fork shares the device copies of the
 * parent's parameters in every layout and owns only its activation arenas (max_batch samples; <= 0: the parent's), statistics,
 * scratch and per-prompt constants (loco_set_context / loco_set_cond); results are bit-identical to an independent context
 * loaded with the same state_dict.  The parent must have all parameters loaded; it may be destroyed first (its parameters are
 * freed with the last fork). */
int  loco_fork(loco_ctx* parent, int32_t max_batch, loco_ctx** out);
const char* loco_last_error(loco_ctx* ctx);

/* Replaces model.load_state_dict (src/utils/utils.py:102-105): one call per
 * state_dict entry, names exactly as in the reference module tree.  `data` is a
 * host OR device pointer to fp32 values (is_device says which). */
int  loco_load_param(loco_ctx* ctx, const char* name, const void* data,
                     const int64_t* shape, int32_t ndim, int32_t is_device);
/* 0 when every parameter of the architecture has been loaded, else the count
 * still missing (first missing name in loco_last_error). */
int  loco_params_missing(loco_ctx* ctx);

/* eps = unet(x, t): PullBackDDPM.forward (diffusion.py:145-200) as called at
 * edit.py:2151, 2375, 2572.  x, eps: [B,C,H,W].  t is the float timestep fed
 * to the time embedding. */
int  loco_unet_forward(loco_ctx* ctx, const float* x, float t, int32_t B,
                       float* eps, void* stream);

/* One DDIM update fused with the denoiser call: scheduler.step
 * (src/utils/utils.py:342-383) after unet(xt,t), the body of HOT LOOPs A/A'/C
 * (edit.py:2146-2160, 2568-2584).  at/at_next are alpha-bar at floor(t),
 * floor(t_next) (utils.py:444-461).  eta==0: deterministic; eta!=0 needs
 * `noise` [B,C,H,W] (the randn_like draw of utils.py:374).  x_next may alias x. */
int  loco_ddim_step(loco_ctx* ctx, const float* x, float t, float at, float at_next,
                    float eta, const float* noise, int32_t B, float* x_next, void* stream);

/* The scheduler update alone, for callers that hold eps already (seam 3 of
 * scheduler.step(et, t, xt, eta).prev_sample / .x0, utils.py:342-383).
 * x0_out (optional) receives P_xt = (xt - et*sqrt(1-at))/sqrt(at). */
int  loco_sched_step(loco_ctx* ctx, const float* x, const float* et, float at, float at_next,
                     float eta, const float* noise, int64_t count, float* x_next, float* x0_out,
                     void* stream);

/* --- PMP-Jacobian operator J = d x0_hat[mask] / d x_t  (edit.py:2369-2391) ---
 * loco_pmp_primal evaluates the denoiser once at (x,t), caching what the
 * tangent and cotangent passes need; `mask` is uint8 [C*H*W] (nullptr = all
 * ones), use_et!=0 selects get_et (edit.py:2394-2403) instead of get_x0. */
int  loco_pmp_primal(loco_ctx* ctx, const float* x, float t, float at,
                     const uint8_t* mask, int32_t use_et, void* stream);
/* Two subspace solves on the same (x, t) with different masks -- the modify-space solve on `mask` and the null-space
 * solve on `~mask` of run_edit_null_space_projection (edit.py:2290-2310) -- can share one probe batch: after
 * loco_pmp_primal(mask), rows >= from_row of every later loco_pmp_jvp / loco_pmp_vjp call use `mask2` (device, [n])
 * instead.  The Jacobian products of the rows are independent, so each solve's iterates are what it would compute
 * alone; the wider batch fills the deep levels of the network better (5 + 5 probes: 14 % less time per probe).
 * mask2 == NULL switches it off; the next loco_pmp_primal also does.  loco_mask_count / loco_mask_gather keep
 * referring to the first mask. */
int  loco_pmp_set_second_mask(loco_ctx* ctx, const uint8_t* mask2, int32_t from_row, void* stream);
/* U = J V  (replaces torch.func.jacfwd at edit.py:2451-2455).  V: [k, n];
 * U: dense [k, n] with zeros outside the mask. */
int  loco_pmp_jvp(loco_ctx* ctx, const float* V, int32_t k, float* U, void* stream);
/* A = U^T J (replaces torch.autograd.functional.jacobian at edit.py:2460-2480).
 * U: dense [k, n] (entries outside the mask are ignored); A: [k, n]. */
int  loco_pmp_vjp(loco_ctx* ctx, const float* U, int32_t k, float* A, void* stream);

/* Thin SVD re-orthonormalisation of the k x n block (replaces torch.linalg.svd
 * at edit.py:2482): on return A holds Vh (orthonormal rows, descending
 * singular value, sign: largest-|.| entry of each row positive), s[k] the
 * singular values of the input. k <= 64. */
int  loco_orthonormalize(loco_ctx* ctx, float* A, int32_t k, int64_t n, float* s, void* stream);
/* Q = thin-QR orthonormal basis of the rows of A (replaces torch.linalg.qr at
 * edit.py:2436 on the transposed layout): A [k,n] in, orthonormal rows out,
 * row i in span(rows 0..i) with positive pivot. */
int  loco_qr_rows(loco_ctx* ctx, float* A, int32_t k, int64_t n, void* stream);
/* Convergence test of edit.py:2489-2492: out[0] = ||Vp - V||_F,
 * out[1] = 1.0 if allclose(Vp, V, atol, rtol=1e-5) else 0.0 (device floats). */
int  loco_convergence(loco_ctx* ctx, const float* Vprev, const float* V, int64_t count,
                      float atol, float* out2, void* stream);
/* The same test row by row and up to each row's sign: the reference compares LAPACK's singular vectors
 * (edit.py:2482-2492), whose signs are LAPACK's choice and, for k >= 2, change from one iteration to the next
 * (tests/golden/converge.pt records it); the rows computed here carry no sign of their own.  Vprev, V: [k, n];
 * out[0] = sqrt(sum_rows min(||vp - v||^2, ||vp + v||^2)), out[1] = 1.0 if every row is allclose(atol, rtol=1e-5)
 * to +-its predecessor. k <= 64. */
int  loco_convergence_rows(loco_ctx* ctx, const float* Vprev, const float* V, int32_t k, int64_t n,
                           float atol, float* out2, void* stream);

/* Null-space projection + row normalisation (edit.py:2317-2323):
 * out = normalize_rows(Vm - (Vn^T (Vn Vm^T))^T); Vn==nullptr: normalise only. */
int  loco_null_project(loco_ctx* ctx, const float* Vm, int32_t k, const float* Vn, int32_t k0,
                       int64_t n, float* out, void* stream);
/* Edit step x + alpha*v (x_space_guidance_direct, edit.py:2618-2625), batched:
 * out[b] = x + alphas[b]*v for b < B (alphas on host). */
int  loco_edit_axpy(loco_ctx* ctx, const float* x, const float* v, const float* alphas,
                    int32_t B, int64_t n, float* out, void* stream);
/* Compact the masked entries: out[k, L] = U[k, mask] (P_xt[:, mask], edit.py:2390). */
int  loco_mask_gather(loco_ctx* ctx, const float* U, int32_t k, float* out, void* stream);
/* L = number of selected mask elements of the last loco_pmp_primal (C*H*W without a mask).  The gather list is built
 * on the device; the first call after a primal reads L back (one 4-byte copy + stream sync). */
int64_t loco_mask_count(loco_ctx* ctx);

/* Measurement helper for bench.py: enqueue a one-lane kernel that writes {shader-clock counter (s_memtime),
 * 100 MHz counter (s_memrealtime)} to out2 (device, 2 x uint64).  Two stamps around a timed region give the average
 * shader clock over it: (d s_memtime / d s_memrealtime) x 100 MHz -- MI355X lowers its clock under matrix load and
 * boxes differ by several percent, which otherwise hides kernel changes in box-to-box comparisons. */
int  loco_clock_stamp(loco_ctx* ctx, uint64_t* out2, void* stream);

/* Work model helpers for bench.py: 2*MAC of one denoiser evaluation (B=1). */
double loco_unet_flops(loco_ctx* ctx);
/* Bytes of device memory the ctx holds. */
int64_t loco_workspace_bytes(loco_ctx* ctx);

/* HIP-event timing on the stream the kernels run on (bench.py roofline leg):
 * average duration in ms of kernels whose name contains `substr` is not
 * available from the API; instead these bracket a region. */
int  loco_timer_start(loco_ctx* ctx, void* stream);
int  loco_timer_stop(loco_ctx* ctx, void* stream, float* ms);

/* Arithmetic of the convolutions: 0 = exact fp32 (v_mfma_f32_32x32x2_f32, the parity
 * anchor), 1 = split-bf16 "bf16x3" (3 x v_mfma_f32_32x32x16_bf16 per product, fp32
 * accumulate, fp32-faithful to ~2^-16; the default), 2 = "f16" (one v_mfma_f32_32x32x16_f16
 * per product: operands rounded to 11 significant bits -- the precision class of the TF32
 * convolutions the reference's CUDA path runs by PyTorch default --, fp32 accumulate and
 * fp32 tensors in HBM).  Env LOCO_PRECISION=f32|bf16x3|f16 sets the initial mode.
 * Invalidates the cached primal. */
int  loco_set_precision(loco_ctx* ctx, int32_t mode);
/* Probe groups of one tangent / cotangent pass on n = 1 (default) or 2 HIP streams: the probes of a batch are independent,
 * so with n = 2 a batch of >= 4 is cut in two groups enqueued on two streams (the bandwidth-bound statistics / apply
 * kernels of one group run beside the convolutions of the other; results identical).  Env LOCO_STREAMS sets the initial
 * value.  Per-kernel durations measured while two streams overlap are not kernel properties: bench.py keeps n = 1 for
 * the headline and its roofline, and reports n = 2 as an extra line. */
int  loco_set_streams(loco_ctx* ctx, int32_t n);
/* The second stream of the n = 2 mode, supplied by the caller (nullptr: the context's own).  HIP hands hardware queues out
 * round-robin over a few (GPU_MAX_HW_QUEUES, 4 by default): a stream created by the library may land on the queue of the
 * caller's stream and then runs strictly behind it -- same results, no overlap.  The host can measure which of its streams
 * runs beside its current one (loco_edit_amd.tloco.BranchStreams._pick: two spin kernels) and hand that one over. */
int  loco_set_side_stream(loco_ctx* ctx, void* stream);
/* n = number of engine contexts whose passes the host enqueues SIDE BY SIDE on different streams (the guidance branches of
 * T-LOCO: loco_edit_amd.tloco.BranchStreams; reference: the prompts of one batched U-Net call, edit.py:1319-1322).  A launch
 * then has about 1 / n of the chip, and the split-K choice of the small-image convolutions aims at 256 / n workgroups instead
 * of 256: fewer partial tiles and less reduce work for the same occupancy (config 5: 315 -> 296 ms per solve at n = 2).
 * Default 1.  Results change by the summation order of the split only. */
int  loco_set_chip_share(loco_ctx* ctx, int32_t n);
int  loco_get_precision(loco_ctx* ctx);

/* Conditional denoisers (T-LOCO, reference edit.py:1286-1373 `self.unet(x, t, encoder_hidden_states=...)`): a
 * conditioning embedding of temb_ch = 4*ch floats (device pointer) that is added to the time embedding before its
 * SiLU -- the slot guided-diffusion uses for `label_emb(y)` (unet.py:660-662) and diffusers' UNet2DConditionModel for
 * `addition_embed_type="text"`.  NULL clears it.  Invalidates the cached primal. */
int  loco_set_cond(loco_ctx* ctx, const float* emb_add, void* stream);
/* out[k, n] = mask * (cv * V + ce * E) with the mask of the last loco_pmp_primal (all ones without one): the
 * x0_hat = (x - eps sqrt(1-at)) / sqrt(at) algebra of edit.py:1574 / 2385 applied to tangents or cotangents when eps is a
 * CFG combination assembled by the caller.  V, E: [k, n]; out may alias either. */
int  loco_masked_axpby(loco_ctx* ctx, const float* V, const float* E, float cv, float ce, int32_t k, float* out,
                       void* stream);
/* Latent encoder contexts (arch 3): z[B, Z, h, w] = scale * (mean + exp(0.5 clamp(logvar, -30, 20)) * noise) from the
 * moments [B, 2 Z, h, w] loco_unet_forward returned (mean | logvar) -- `self.vae.encode(x0).latent_dist.sample() * 0.18215`
 * of the latent inversion (src/modules/edit.py:594-597; DiagonalGaussianDistribution of diffusers, un-vendored).
 * noise: [B, Z, h, w] standard normal, or NULL for the posterior mean (`.mode()`). */
int  loco_latent_sample(loco_ctx* ctx, const float* moments, const float* noise, float scale, int32_t B, float* z, void* stream);
/* out = sum_{i<n} coef[i] * src[i], n <= 4: the classifier-free-guidance combination of eps / J V / J^T U terms of
 * several conditions (edit.py:1324-1372).  src: host array of device pointers, coef: host array; out may alias a src. */
/* Encoder states of the prompt for the cross-attention stages (context_dim > 0): tokens = device pointer to
 * [context_len][context_dim] fp32.  Projects them to the per-block keys / values once; the cached primal is
 * invalidated.  Replaces `encoder_hidden_states=prompt_emb` of self.unet(...) (edit.py:664-667, 1319-1322).
 * With cfg.added_kv (the DeepFloyd-IF U-Net) the tokens are the states AFTER the model's `encoder_hid_proj` (the host computes
 * that projection and the pooled `add_embedding` for loco_set_cond once per prompt): every attention block passes them
 * through its own GroupNorm (`norm_cross`) and key / value projections (`add_k_proj`, `add_v_proj`) here, and attends over
 * [these ; its image tokens] in one softmax. */
int  loco_set_context(loco_ctx* ctx, const float* tokens, void* stream);

int  loco_lincomb(loco_ctx* ctx, const float* const* src, const float* coef, int32_t n, float* out, int64_t count,
                  void* stream);

/* Per-kernel HIP-event profile of the convolution launches (bench.py roofline
 * leg).  While enabled every conv launch is bracketed by two events on the
 * caller's stream; loco_profile_report synchronises, then writes one line per
 * kernel variant: "name launches total_ms total_flops" (algorithmic 2*MAC). */
int  loco_profile_enable(loco_ctx* ctx, int32_t on);
int  loco_profile_report(loco_ctx* ctx, char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* LOCO_HIP_H */
