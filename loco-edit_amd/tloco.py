"""Text-supervised T-LOCO in pixel space (DeepFloyd-IF stage-I path) on the MI355X engine: the class
``EditDeepFloydIF`` with the method names, argument order and file names of the reference
(``src/modules/edit.py:1193-2028``).

Built here (BASELINE config 5): the classifier-free-guidance combination of the
conditional noise predictions (``_classifer_free_guidance`` :1286-1373), ``get_x0`` (:1566-1587), the CFG-combined
PMP-Jacobian subspace solver (``local_encoder_decoder_pullback_xt`` :1589-1676), the edit direction through the
Jacobian and directly in noise space (``get_delta_xt_via_grad`` :1680-1717, ``get_v_modify`` :1720-1741), the sampler
(``DDPMforwardsteps`` :1412-1481) and the two drivers (``run_edit_null_space_projection_xt`` :1745-1868,
``..._xt_semantic`` :1871-2018, ablations ``null-space-proj`` and ``sega``).

Why several engines: J = d x0_hat / d x_t with eps = sum_c w_c eps_c(x, t | prompt_c) is the same linear combination of
the per-prompt Jacobians, J V = mask (V/sqrt(at) - sqrt(1-at)/sqrt(at) sum_c w_c dEps_c V).  Each prompt gets its own
engine context ("branch") holding that prompt's primal activations; a probe batch runs one tangent and one cotangent
pass per branch, the branches are combined by ``loco_lincomb`` / ``loco_masked_axpby``.  The reference pays the same
2-3 U-Net evaluations per product as a batch of 2-3.

The denoiser: ``config.IF_I_M_UNET`` (round 4), the stage-I architecture the shipped scripts name -- the engine's
guided-diffusion family with GELU, (skip + h) / sqrt 2 and attention over [text ; image] keys; the image-independent text
conditioning (``encoder_proj``, ``encoder_pooling``) runs on the host (``IFTextConditioner``) and reaches the engine through
``loco_set_context`` / ``loco_set_cond``.  Written from the published module trees: neither diffusers nor deepfloyd_if is
installed and there are no weights, so the architecture's parity is unpinned, while the orchestration above is pinned
against the reference's own methods (oracle/make_golden_tloco.py).  The round-2 / 3 stand-ins stay selectable: the
guided-diffusion U-Net whose time embedding receives ``cond_proj(mean_tokens(prompt_emb))``, and the same with text
cross-attention stages.  Prompt embeddings are inputs (``--prompt_emb_path``: a dict of [1, tokens, D] tensors; the T5
encoder is out of scope) or seeded; stages II/III, SAM and the ``diffedit`` ablation are not on this path.
"""
from __future__ import annotations

import os
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import solver
from .config import UNetConfig, synth_params
from .dist import ProbeSharder
from .hip import LocoEngine
from .scheduler import SchedulerOutput
from .utils import save_image as _save_image

CFG_MODES = ("null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)", "(for-null)", "(edit-null)")


def cond_params(cfg: UNetConfig, cond_dim: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic ``cond_proj`` (Linear cond_dim -> 4*ch) of the stand-in conditional denoiser."""
    rng = np.random.default_rng(zlib.crc32(f"cond_proj:{seed}:{cond_dim}".encode()))
    ted = cfg.ch * 4
    return {"cond_proj.weight": (rng.standard_normal((ted, cond_dim)) / np.sqrt(cond_dim)).astype(np.float32),
            "cond_proj.bias": (0.1 * rng.standard_normal(ted)).astype(np.float32)}


class IFTextConditioner:
    """The part of the IF U-Net that does not see the image: from the text encoder's states [1, L, E] (T5-XXL: 77 x 4096)
    the context [L, D] its attention blocks read (`encoder_proj`; diffusers `encoder_hid_proj`) and the embedding [4 ch]
    added to the time embedding (`encoder_pooling` = LayerNorm -> AttentionPooling -> Linear -> LayerNorm; diffusers
    `add_embedding` = TextTimeEmbedding).  Once per prompt, a few small torch products on the device (host plumbing: the
    engine takes the results through ``loco_set_context`` / ``loco_set_cond``).  Parameters: the ``encoder_proj.*`` /
    ``encoder_pooling.*`` entries of the U-Net's state_dict (config.adm_param_shapes)."""
    PREFIXES = ("encoder_proj.", "encoder_pooling.")

    def __init__(self, params: Dict[str, "np.ndarray | torch.Tensor"], cfg: UNetConfig, device):
        self.cfg = cfg
        self.p = {k: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=torch.float32).to(device)
                  for k, v in params.items() if k.startswith(self.PREFIXES)}
        missing = [k for k in ("encoder_proj.weight", "encoder_pooling.1.q_proj.weight", "encoder_pooling.3.weight") if k not in self.p]
        if missing:
            raise KeyError(f"text conditioning parameters missing from the checkpoint: {missing}")

    def __call__(self, states: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        import math
        F = torch.nn.functional
        p = self.p
        x = states.to(p["encoder_proj.weight"].device, torch.float32)
        if x.dim() == 2:
            x = x[None]
        E = x.shape[-1]
        context = F.linear(x, p["encoder_proj.weight"], p["encoder_proj.bias"])[0]
        x = F.layer_norm(x, (E,), p["encoder_pooling.0.weight"], p["encoder_pooling.0.bias"], 1e-5)
        d = min(64, E // 2)                  # 64 channels per pooling head (att_pool_heads = 64 at E = 4096)
        heads = E // d
        token = x.mean(dim=1, keepdim=True) + p["encoder_pooling.1.positional_embedding"]
        seq = torch.cat([token, x], dim=1)

        def split(z):                        # [1, N, E] -> [heads, N, d]
            return z.reshape(-1, heads, d).transpose(0, 1)
        q = split(F.linear(token, p["encoder_pooling.1.q_proj.weight"], p["encoder_pooling.1.q_proj.bias"]))
        k = split(F.linear(seq, p["encoder_pooling.1.k_proj.weight"], p["encoder_pooling.1.k_proj.bias"]))
        v = split(F.linear(seq, p["encoder_pooling.1.v_proj.weight"], p["encoder_pooling.1.v_proj.bias"]))
        w = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(d), dim=-1)          # [heads, 1, N]
        pooled = (w @ v).transpose(0, 1).reshape(1, E)
        pooled = F.linear(pooled, p["encoder_pooling.2.weight"], p["encoder_pooling.2.bias"])
        aug = F.layer_norm(pooled, (pooled.shape[-1],), p["encoder_pooling.3.weight"], p["encoder_pooling.3.bias"], 1e-5)[0]
        return context.contiguous(), aug.contiguous()


class IFScheduler(object):
    """The scheduler the reference patches onto the IF pipeline (utils.py:159-213): squared-cosine alpha-bar
    (``betas_for_alpha_bar`` :425-441, float32), float timesteps ``linspace(0,1,N)*990``, alpha-bar looked up at
    ``floor(t)``, DDIM update (HIP kernel behind ``loco_sched_step``)."""
    t_max = 990

    def __init__(self, engine=None):
        import math
        ab = lambda ts: math.cos((ts + 0.008) / 1.008 * math.pi / 2) ** 2
        betas = torch.tensor([min(1 - ab((i + 1) / 1000) / ab(i / 1000), 0.999) for i in range(1000)], dtype=torch.float32)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.timesteps = self.timesteps_next = None
        self.engine = engine

    def set_timesteps(self, num_inferences, device=None, is_inversion=False):
        seq = torch.linspace(0, 1, num_inferences) * self.t_max
        if is_inversion:
            seq = seq + 1e-6
            seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]], dim=0)
            self.timesteps, self.timesteps_next = seq_prev[1:], seq[1:]
        else:
            seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]], dim=0)
            self.timesteps, self.timesteps_next = torch.flip(seq[1:], dims=[0]), torch.flip(seq_prev[1:], dims=[0])

    def scale_model_input(self, x, t):
        return x

    def index_of(self, t) -> int:
        return self.timesteps.tolist().index(float(t))

    def alpha_at(self, t) -> float:
        return float(self.alphas_cumprod[int(torch.as_tensor(float(t)).long())])

    def step(self, et, t, xt, eta=0.0, **kwargs):
        if eta != 0.0:
            raise ValueError("the T-LOCO sampler is deterministic (edit.py:1466 eta=0)")
        t_next = self.timesteps_next[self.index_of(t)]
        nxt, x0 = self.engine.sched_step(xt.contiguous(), et.contiguous(), self.alpha_at(t), self.alpha_at(t_next), 0.0,
                                         None, want_x0=True)
        return SchedulerOutput(nxt, x0)


def cfg_weights(mode: str, g: float, ge: float, do_cfg: bool = True) -> List[Tuple[str, float]]:
    """eps = sum_c w_c eps_c for the modes of edit.py:1324-1356 -> [(branch, w_c)]."""
    if not do_cfg:
        return [("for", 1.0)]
    table = {
        "null+(for-null)+(edit-null)": [("for", g), ("edit", ge), ("null", 1.0 - g - ge)],
        "null+(for-null)": [("for", g), ("null", 1.0 - g)],
        "null+(edit-null)": [("edit", g), ("null", 1.0 - g)],
        "(for-edit)": [("for", g), ("edit", -g)],
        "(for-null)": [("for", g), ("null", -g)],
        "(edit-null)": [("edit", g), ("null", -g)],
    }
    if mode not in table:
        # "edit-proj[for](edit)" reads an undefined variable in the reference (edit.py:1359) and, like
        # "null+for+edit-proj[for](edit)", is not reachable from the shipped scripts
        raise NotImplementedError(f"CFG mode {mode!r}")
    return table[mode]


class BranchStreams:
    """The CFG branches are independent engine contexts evaluating the same input under different prompts: their passes
    run side by side on separate HIP streams and join before the combination.  At 64x64 a 5-probe pass leaves most launches
    below one workgroup per CU, so two or three branches fill the chip where one does not (the reference batches the
    prompts through one U-Net call, edit.py:1319-1322; here each prompt owns a context with its own cached primal).
    ``LOCO_CFG_STREAMS=0``: one branch after the other on the current stream.
    The callables must not allocate device memory: outputs are allocated by the caller on its own stream and passed in
    (``out=``) -- an allocation under a side stream belongs to that stream's pool, and when the pool has to grow (hipMalloc
    synchronises the device) the overlap is lost: measured 474 vs 334 ms per config-5 solve inside a process whose memory
    was mostly taken."""

    def __init__(self, n: int, device):
        self.enabled = os.environ.get("LOCO_CFG_STREAMS", "1") != "0" and torch.device(device).type == "cuda" and n > 1
        self.side = self._pick(n - 1, device) if self.enabled else []
        if self.enabled and len(self.side) < n - 1:
            self.enabled = False          # no stream found that runs beside the current one: stay serial

    @staticmethod
    def _pick(count: int, device, candidates: int = 12):
        """HIP streams share a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), handed out round-robin at creation:
        a side stream that lands on the queue of the current stream (or of another side stream) runs strictly after it
        and the overlap is lost without a trace -- whether that happens depends on how many streams the process created
        before (measured: the same config-5 solve 326 ms in a fresh process, 468 ms after an unrelated engine had run).
        So the streams are chosen by measurement: two spin kernels, one per stream; streams that finish them in the time of
        one are on different queues."""
        if not hasattr(torch.cuda, "_sleep"):
            return [torch.cuda.Stream(device=device) for _ in range(count)]
        import time
        main = torch.cuda.current_stream(device)
        cyc = 1_500_000

        def spin(streams):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for s_ in streams:
                with torch.cuda.stream(s_):
                    torch.cuda._sleep(cyc)
            torch.cuda.synchronize(device)
            return time.perf_counter() - t0
        spin([main])
        one = min(spin([main]) for _ in range(3))
        chosen = []
        for _ in range(candidates):
            if len(chosen) == count:
                break
            c = torch.cuda.Stream(device=device)
            spin([main, c])
            if all(min(spin([o, c]) for _ in range(2)) < 1.5 * one for o in [main] + chosen):
                chosen.append(c)
        return chosen

    def run(self, fns):
        if not self.enabled or len(fns) < 2:
            return [f() for f in fns]
        main = torch.cuda.current_stream()
        outs = [None] * len(fns)
        for i, f in enumerate(fns[1:]):
            self.side[i].wait_stream(main)                 # the inputs were produced on the current stream
            with torch.cuda.stream(self.side[i]):
                outs[i + 1] = f()
        outs[0] = fns[0]()
        for i in range(len(fns) - 1):
            main.wait_stream(self.side[i])
        return outs


class CFGJacobianOperator:
    """J and J^T of x0_hat(x_t) under classifier-free guidance, assembled from the per-prompt engines."""

    def __init__(self, branches: Dict[str, LocoEngine], weights, x, t, at, mask, streams: Optional[BranchStreams] = None):
        self.w = [(branches[name], w) for name, w in weights if w != 0.0]
        self.lead = self.w[0][0]
        self.n = self.lead.n
        self.n_out = self.lead.n_out
        self.masked = mask is not None
        self.cv = 1.0 / float(np.sqrt(np.float32(at)))
        self.ce = -float(np.sqrt(np.float32(1.0) - np.float32(at))) / float(np.sqrt(np.float32(at)))
        self.streams = streams or BranchStreams(1, "cpu")
        xc = x.contiguous()
        m8 = None if mask is None else mask.to(device=xc.device, dtype=torch.uint8).contiguous()   # converted on this stream
        # dEps products; x0 algebra applied here
        self.streams.run([(lambda e=eng: e.pmp_primal(xc, float(t), at, m8, use_et=True)) for eng, _ in self.w])

    def check_mask(self):
        if self.masked and self.lead.mask_count() == 0:
            raise ValueError("empty mask: J = d x0_hat[mask] / d x_t has no rows")

    def _outs(self, rows, width):
        return [torch.empty(rows, width, device=self.lead.device, dtype=torch.float32) for _ in self.w]

    def jvp(self, V):
        bufs = self._outs(V.shape[0], self.n_out)
        outs = self.streams.run([(lambda e=eng, o=o: e.pmp_jvp(V, out=o)) for (eng, _), o in zip(self.w, bufs)])
        terms = [(w, o) for (_, w), o in zip(self.w, outs)]
        dE = terms[0][1] if (len(terms) == 1 and terms[0][0] == 1.0) else self.lead.lincomb(terms)
        return self.lead.masked_axpby(V, dE, self.cv, self.ce)

    def vjp(self, U):
        bufs = self._outs(U.shape[0], self.n)
        outs = self.streams.run([(lambda e=eng, o=o: e.pmp_vjp(U, out=o)) for (eng, _), o in zip(self.w, bufs)])
        terms = [(self.ce * w, o) for (_, w), o in zip(self.w, outs)]
        terms.append((1.0, self.lead.masked_axpby(U, U, self.cv, 0.0)))
        return self.lead.lincomb(terms)

    def gather(self, U):
        return self.lead.mask_gather(U)


class EditDeepFloydIF(object):
    def __init__(self, args):
        self.device, self.dtype = args.device, args.dtype
        if self.dtype == torch.float16:
            # `--dtype fp16` (the reference loads its IF / SD pipelines with torch_dtype=float16, utils.py:260-283, and casts to
            # fp32 before the SVD, edit.py:1653): tensors stay fp32 on this engine, the request selects the f16 conv arithmetic
            # (f16 MFMA operands, fp32 accumulate) unless --precision says otherwise
            if not (getattr(args, "precision", None) or os.environ.get("LOCO_PRECISION")):
                args.precision = "f16"
            print(f"dtype fp16: fp32 tensors with conv arithmetic {getattr(args, 'precision', None) or os.environ.get('LOCO_PRECISION')}")
            self.dtype = torch.float32
        elif self.dtype != torch.float32:
            raise ValueError("tensors are fp32 on this engine; choose the conv arithmetic with --precision")
        self.buffer_device, self.memory_bound = getattr(args, "buffer_device", "cpu"), getattr(args, "memory_bound", 5)
        self.seed = args.seed
        cfg: UNetConfig = args.unet_config
        self.cfg = cfg
        self.c_in, self.image_size = cfg.in_channels, cfg.resolution
        self.for_steps, self.use_yh_custom_scheduler = args.for_steps, args.use_yh_custom_scheduler
        self.guidance_scale, self.guidance_scale_edit = args.guidance_scale, args.guidance_scale_edit
        self.dataset_name = args.dataset_name
        # ---- prompt embeddings: inputs of this path (the T5 encoder is out of scope)
        pe = getattr(args, "prompt_emb", None)
        if pe is None and getattr(args, "prompt_emb_path", ""):
            pe = torch.load(args.prompt_emb_path)
        # text enters through cross-attention stages when the architecture has them (context_dim > 0: tokens
        # [context_len, context_dim] via loco_set_context), otherwise pooled through the time embedding (loco_set_cond)
        self.use_context = cfg.context_dim > 0
        self.use_text_cond = cfg.encoder_dim > 0      # the IF U-Net: states -> (context, pooled embedding) on the host
        if pe is None:
            g = torch.Generator().manual_seed(int(getattr(args, "prompt_emb_seed", 31)))
            ntok, D = (cfg.context_len, cfg.context_dim) if self.use_context else (7, int(getattr(args, "cond_dim", 16)))
            if self.use_text_cond:
                D = cfg.encoder_dim
            pe = {k: torch.randn(1, ntok, D, generator=g) for k in ("for", "edit", "null")}
        self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb = pe["for"], pe["edit"], pe["null"]
        self.for_prompt, self.edit_prompt, self.null_prompt = args.for_prompt, args.edit_prompt, ""
        # ---- the conditional denoiser: one engine context per prompt
        params = getattr(args, "params", None)
        if params is None:
            if getattr(args, "ckpt_path", ""):
                # vendored / latent-diffusion names load as they are; a CompVis pipeline file (`model.diffusion_model.*`
                # next to the autoencoder and text encoder) and the diffusers UNet2DConditionModel naming are recognised
                # and renamed, anything else is refused with the list of foreign keys (checkpoints.py)
                from .checkpoints import normalize_unet_state_dict
                params = normalize_unet_state_dict(torch.load(args.ckpt_path, map_location="cpu"), cfg)
            else:
                seed = getattr(args, "synthetic_weights", None)
                if seed is None:
                    raise ValueError("no checkpoint: pass --ckpt_path or --synthetic_weights SEED")
                params = dict(synth_params(cfg, seed=int(seed)))
                if not self.use_context:
                    params.update(cond_params(cfg, self.for_prompt_emb.shape[-1], seed=int(seed)))
        if not self.use_context:
            self.cond_w = torch.as_tensor(np.asarray(params["cond_proj.weight"]), dtype=torch.float32)
            self.cond_b = torch.as_tensor(np.asarray(params["cond_proj.bias"]), dtype=torch.float32)
        self.text_cond = IFTextConditioner(params, cfg, self.device) if self.use_text_cond else None
        unet_params = {k: v for k, v in params.items() if not k.startswith(("cond_proj.",) + IFTextConditioner.PREFIXES)}
        self.branches: Dict[str, LocoEngine] = {}
        prec = getattr(args, "precision", None) or os.environ.get("LOCO_PRECISION")
        # ONE set of weights for the three prompts, as the reference's single U-Net object (edit.py:1319-1322, :655-667): the first
        # context loads the parameters (six device layouts), the others are forks of it (loco_fork: own arenas, statistics and
        # prompt constants only; bit-identical to independently loaded contexts).  LOCO_CFG_FORK=0: three independent contexts
        share_weights = os.environ.get("LOCO_CFG_FORK", "1") != "0"
        for name in ("for", "edit", "null"):
            if share_weights and self.branches:
                eng = self.branches["for"].fork()
            else:
                eng = LocoEngine(cfg, max_batch=getattr(args, "max_batch", 8), device=self.device)
                eng.load_state_dict(unet_params)
            if prec:
                eng.set_precision(prec)
            self.branches[name] = eng
        self.engine = self.branches["for"]
        self.branch_streams = BranchStreams(len(self.branches), self.device)
        if self.branch_streams.enabled:
            # two guidance branches run every J V / J^T U side by side: each engine sizes its split-K for half of the chip
            # (include/loco_hip.h loco_set_chip_share; LOCO_CFG_SHARE overrides, 1 = off)
            share = int(os.environ.get("LOCO_CFG_SHARE", "2"))
            for eng in self.branches.values():
                eng.set_chip_share(share)
        self._cond_of: Dict[str, int] = {}
        for name, e in (("for", self.for_prompt_emb), ("edit", self.edit_prompt_emb), ("null", self.null_prompt_emb)):
            self._bind(name, e)
        print(f'engine : {self.engine.version()}, conv arithmetic : {self.engine.get_precision()}, branches : for / edit / null')
        self.scheduler = IFScheduler(engine=self.engine)
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        self.edit_t = args.edit_t
        self.edit_t_idx = int((self.scheduler.timesteps - self.edit_t * 1000).abs().argmin())
        self.no_edit_t = getattr(args, "no_edit_t", 0.5)
        self.sampling_mode = args.sampling_mode
        self.tilda_v_score_type = args.tilda_v_score_type
        self.ablation_method = args.ablation_method
        self.mask_type = args.mask_type
        self.vT_path = args.vT_path
        self.x_space_guidance_edit_step = args.x_space_guidance_edit_step
        self.x_space_guidance_scale = args.x_space_guidance_scale
        self.x_space_guidance_num_step = args.x_space_guidance_num_step
        # path (edit.py:1205-1208)
        # the reference ends the name with the model size of "DeepFloyd/IF-I-<size>-v1.0" (edit.py:1204-1206); the stand-in
        # denoisers (no IF architecture behind the name) say so instead
        parts = str(getattr(args, "model_name", "")).split("-")
        size = (parts[2] if len(parts) > 2 else "M") if self.use_text_cond else "standin"
        self.result_folder = os.path.join(args.result_folder, f"for_prompt_{args.for_prompt}_cfg{args.guidance_scale}_seed{args.seed}_{size}")
        os.makedirs(self.result_folder, exist_ok=True)
        self.sharder = ProbeSharder("world")
        self.EXP_NAME = "exp"
        self.args = args

    # ------------------------------------------------------------------ conditioning
    def cond_embedding(self, prompt_emb: torch.Tensor) -> torch.Tensor:
        """[1, tokens, D] -> [4*ch] (host GEMV of 4*ch x D: negligible, once per prompt)."""
        return torch.nn.functional.linear(prompt_emb.float().mean(dim=1), self.cond_w, self.cond_b)[0]

    def _bind(self, name: str, prompt_emb: torch.Tensor):
        key = (prompt_emb.data_ptr(), tuple(prompt_emb.shape), float(prompt_emb.sum()))
        if self._cond_of.get(name) != key:
            if self.use_text_cond:
                context, aug = self.text_cond(prompt_emb)
                self.branches[name].set_context(context)
                self.branches[name].set_cond(aug)
            elif self.use_context:
                self.branches[name].set_context(prompt_emb[0].to(self.device, torch.float32).contiguous())
            else:
                self.branches[name].set_cond(self.cond_embedding(prompt_emb).to(self.device).contiguous())
            self._cond_of[name] = key

    def _bind_all(self, for_e, edit_e, null_e):
        self._bind("for", for_e); self._bind("edit", edit_e); self._bind("null", null_e)

    def _get_prompt_emb(self, prompt):
        raise NotImplementedError("the T5 text encoder is outside this path: pass prompt embeddings (--prompt_emb_path)")

    # ------------------------------------------------------------------ CFG noise (edit.py:1286-1373)
    def _classifer_free_guidance(self, latents, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mode,
                                 do_classifier_free_guidance):
        self._bind_all(for_prompt_emb, edit_prompt_emb, null_prompt_emb)
        x = latents.to(self.device, torch.float32).contiguous()
        weights = cfg_weights(mode, self.guidance_scale, self.guidance_scale_edit, do_classifier_free_guidance)
        mb = self.engine.max_batch
        out = torch.empty_like(x)
        for b0 in range(0, x.shape[0], mb):
            xs = x[b0:b0 + mb].contiguous()
            bufs = [torch.empty_like(xs) for _ in weights]          # allocated here: nothing is allocated under a side stream
            outs = self.branch_streams.run([(lambda n=name, o=o: self.branches[n].unet_forward(xs, float(t), out=o))
                                            for (name, _), o in zip(weights, bufs)])
            terms = [(w, o) for (_, w), o in zip(weights, outs)]
            out[b0:b0 + mb] = terms[0][1] if (len(terms) == 1 and terms[0][0] == 1.0) else self.engine.lincomb(terms)
        return out

    # ------------------------------------------------------------------ x0 (edit.py:1566-1587)
    def get_x0(self, xt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mask=None,
               mode="null+(for-null)+(edit-null)", flatten=False):
        do_cfg = self.guidance_scale > 1.0
        noise_pred = self._classifer_free_guidance(xt, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mode=mode,
                                                   do_classifier_free_guidance=do_cfg)
        at = self.scheduler.alpha_at(t)
        _, x0_hat = self.engine.sched_step(xt.to(self.device, torch.float32).contiguous(), noise_pred, at, at, 0.0, None,
                                           want_x0=True)
        if mask is not None:
            return x0_hat[:, mask.to(x0_hat.device)]
        if flatten:
            x0_hat = x0_hat.view(x0_hat.shape[0], -1)
        return x0_hat

    def _operator(self, xt, t, mask, mode):
        weights = cfg_weights(mode, self.guidance_scale, self.guidance_scale_edit, self.guidance_scale > 1.0)
        return CFGJacobianOperator(self.branches, weights, xt.to(self.device, torch.float32).contiguous(), t,
                                   self.scheduler.alpha_at(t), mask, streams=self.branch_streams)

    # ------------------------------------------------------------------ solver (edit.py:1589-1676)
    def local_encoder_decoder_pullback_xt(self, xt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, op=None,
                                          block_idx=None, pca_rank=50, chunk_size=25, min_iter=10, max_iter=100,
                                          convergence_threshold=1e-3, mask=None, mode="null+(for-null)+(edit-null)",
                                          v0=None, verbose=True):
        assert mode in ["null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)"]
        self._bind_all(for_prompt_emb, edit_prompt_emb, null_prompt_emb)
        n = self.engine.n
        if v0 is None:
            v0 = torch.randn(n, pca_rank, device=self.device, dtype=torch.float)          # edit.py:1616
        V = v0.to(self.device, torch.float32).T.contiguous()
        self.engine.qr_rows_(V)                                                            # :1617
        opj = self._operator(xt, t, mask, mode)
        U, s, V, self.last_n_iter = solver.subspace_iteration(opj, self.engine, V, min_iter, max_iter,
                                                              convergence_threshold, sharder=self.sharder, verbose=verbose)
        opj.check_mask()
        u = opj.gather(U).T.contiguous()
        return u, s.sqrt(), V

    # ------------------------------------------------------------------ directions (edit.py:1680-1741)
    @torch.no_grad()
    def get_delta_xt_via_grad(self, xt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mask=None,
                              mode="null+(for-null)+(edit-null)"):
        """Unit-norm J_mode^T (x0_hat[mode] - x0_hat["null+(for-null)"]) restricted to the mask."""
        do_cfg = self.guidance_scale > 1.0
        e0 = self._classifer_free_guidance(xt, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, "null+(for-null)", do_cfg)
        e1 = self._classifer_free_guidance(xt, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mode, do_cfg)
        opj = self._operator(xt, t, mask, mode)
        # x0_hat_after - x0_hat = ce (e1 - e0) (the x_t terms cancel), masked by the operator's cotangent seed
        d = self.engine.lincomb([(opj.ce, e1.view(1, -1).contiguous()), (-opj.ce, e0.view(1, -1).contiguous())])
        v_ = opj.vjp(d)
        return self.engine.null_project(v_, None)            # v_ / v_.norm(dim=1)

    @torch.no_grad()
    def get_v_modify(self, xt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mask=None,
                     mode="(for-edit)-direct", jacobian=False):
        """Directions in noise space.  The reference returns them un-normalised and normalises in the caller
        (edit.py:1952); here every mode returns unit rows (same direction, the caller's normalisation is a no-op)."""
        if jacobian:
            return self.get_delta_xt_via_grad(xt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mask=mask,
                                              mode=self.tilda_v_score_type)
        cfgn = lambda m: self._classifer_free_guidance(xt, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mode=m,
                                                       do_classifier_free_guidance=True).view(1, -1).contiguous()
        if mode == "(for-edit)-direct":
            return self.engine.null_project(cfgn("(for-edit)"), None)
        if mode == "(edit-null)-direct":
            return self.engine.null_project(self.engine.lincomb([(-1.0, cfgn("(edit-null)"))]), None)
        if mode == "proj_null[for-null](edit-null)-direct":
            e1n = self.engine.null_project(cfgn("(for-null)"), None)
            perp = self.engine.null_project(cfgn("(edit-null)"), e1n)        # eps_2 - <eps_2, e1> e1 / <e1, e1>, unit norm
            return self.engine.lincomb([(-1.0, perp)])
        raise ValueError(mode)

    # ------------------------------------------------------------------ sampler (edit.py:1412-1481)
    @torch.no_grad()
    def DDPMforwardsteps(self, xt, t_start_idx, t_end_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb,
                         mode="null+(for-null)", **kwargs):
        assert mode in ["null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)"]
        do_cfg = self.guidance_scale > 1.0
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        xt = xt.to(self.device, torch.float32).contiguous()
        for t_idx, t in enumerate(self.scheduler.timesteps):
            if t_idx < t_start_idx:
                continue
            elif t_start_idx == t_idx:
                pass
            elif t_idx == t_end_idx:
                return xt, t, t_idx
            xt = self.scheduler.scale_model_input(xt, t)
            noise_pred = self._classifer_free_guidance(xt, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mode=mode,
                                                       do_classifier_free_guidance=do_cfg)
            xt = self.scheduler.step(noise_pred, t, xt, eta=0).prev_sample
        xt = (xt / 2 + 0.5).clamp(0, 1)
        if self.sharder.is_main:
            _save_image(xt, os.path.join(self.result_folder, f'{self.EXP_NAME}_stage1.png'), nrow=xt.size(0))
        return (xt * 255).to(torch.uint8).permute(0, 2, 3, 1)

    @torch.no_grad()
    def x_space_guidance_direct(self, xt, t_idx, vk, single_edit_step):
        return self.engine.edit_axpy(xt.contiguous(), vk.contiguous().view(-1), [self.x_space_guidance_scale * single_edit_step])

    # ------------------------------------------------------------------ multi-rank file discipline (as EditUncondDiffusion)
    # Under torchrun every rank runs the same flow: whether a file exists is rank 0's decision (every rank takes the same
    # branch or they hang in the next all-gather), and files are read by rank 0 and broadcast (the ranks need not share a
    # file system, and a failed read raises on every rank together).
    def _exists(self, path):
        return self.sharder.agree(bool(path) and os.path.exists(path))

    def _load(self, path, **kw):
        if not self.sharder.active:
            return torch.load(path, **kw)
        obj, err = None, None
        if self.sharder.is_main:
            try:
                obj = torch.load(path, map_location="cpu")
            except Exception as ex:              # agreed below: every rank raises, none is left waiting in a collective
                err = repr(ex)
        obj, err = self.sharder.agree((obj, err))
        if err is not None:
            raise RuntimeError(f"rank 0 could not read {path}: {err}")
        return obj

    # ------------------------------------------------------------------ drivers
    def _masks(self):
        mpath = os.path.join(self.result_folder, "mask/mask.pt")
        if not self._exists(mpath):
            raise FileNotFoundError(f"{mpath} missing: stage-II super-resolution + SAM (edit.py:1768-1776) are outside this "
                                    "path; provide mask.pt (bool [N,res,res])")
        print("Loading masks......")
        return self._load(mpath)

    def _walk(self, original_xt, v_row, vis_num):
        """+/- walk of edit.py:1840-1860 in one kernel (frames x + j*scale*step*v)."""
        S = self.x_space_guidance_num_step
        idxs = [0, S] if vis_num == 1 else list(range(0, S + 1, (S + 1) // vis_num))
        step = self.x_space_guidance_scale * self.x_space_guidance_edit_step
        alphas = [-j * step for j in reversed(idxs)][:-1] + [j * step for j in idxs]
        return self.engine.edit_axpy(original_xt.contiguous(), v_row.contiguous().view(-1), alphas)

    def _xT(self):
        if self.dataset_name != 'Random':
            raise ValueError("T-LOCO runs from x_T ~ N(0, I) (dataset_name 'Random', edit.py:1763)")
        return torch.randn(1, self.c_in, self.image_size, self.image_size, dtype=self.dtype, device=self.device)

    @torch.no_grad()
    def run_edit_null_space_projection_xt(self, op, block_idx, vis_num, mask_index=0, vis_num_pc=1, vis_vT=False, pca_rank=50,
                                          edit_prompt=None, null_space_projection=False, pca_rank_null=50):
        """edit.py:1745-1868: unsupervised (non-semantic) directions of the CFG denoiser, null-space projected."""
        self.scheduler.set_timesteps(self.for_steps)
        xT = self._xT()
        self.EXP_NAME = "original"
        masks = self._masks()
        if self.sampling_mode:
            return None
        mask = masks[mask_index].squeeze(dim=0).repeat(3, 1, 1)
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        xt, t, t_idx = self.DDPMforwardsteps(xT, t_start_idx=0, t_end_idx=self.edit_t_idx, for_prompt_emb=F,
                                             edit_prompt_emb=E, null_prompt_emb=N, mode="null+(for-null)")
        assert t_idx == self.edit_t_idx
        save_dir = os.path.join(self.result_folder, "basis", f'local_basis-{self.edit_t}T-pca-rank-{pca_rank}-select-mask{mask_index}')
        os.makedirs(save_dir, exist_ok=True)
        paths = {k: os.path.join(save_dir, f) for k, f in (("um", 'u-modify.pt'), ("vm", 'vT-modify.pt'),
                 ("un", f'u-null-null_space_rank_{pca_rank_null}.pt'), ("vn", f'vT-null-null_space_rank_{pca_rank_null}.pt'))}
        vT_null = None
        if self.sharder.agree(all(os.path.exists(p) for p in paths.values())):
            print('!!!Load CALCULATED BASIS!!!')
            vT_modify = self._load(paths["vm"], map_location=self.device).to(self.device).type(self.dtype)
            vT_null = self._load(paths["vn"], map_location=self.device).to(self.device).type(self.dtype)
        else:
            print('subspace solve: CFG-combined Jacobian')
            u_modify, s_modify, vT_modify = self.local_encoder_decoder_pullback_xt(
                xt, t, t_idx, F, E, N, op=op, block_idx=block_idx, pca_rank=pca_rank, chunk_size=5, min_iter=10, max_iter=50,
                convergence_threshold=1e-3, mask=mask, mode="null+(for-null)")
            if self.sharder.is_main:
                torch.save(u_modify, paths["um"]); torch.save(vT_modify, paths["vm"])
            if null_space_projection:
                u_null, s_null, vT_null = self.local_encoder_decoder_pullback_xt(
                    xt, t, t_idx, F, E, N, op=op, block_idx=block_idx, pca_rank=pca_rank_null, chunk_size=5, min_iter=10,
                    max_iter=50, convergence_threshold=1e-3, mask=~mask, mode="null+(for-null)")
                if self.sharder.is_main:
                    torch.save(u_null, paths["un"]); torch.save(vT_null, paths["vn"])
        vT = self.engine.null_project(vT_modify.contiguous(),
                                      vT_null[:pca_rank_null, :].contiguous() if null_space_projection else None)
        original_xt = xt.clone()
        x0 = None
        for pc_idx in range(vis_num_pc):
            self.EXP_NAME = (f'Non-semantic_Edit_xt-edit_{self.edit_t}T-select_mask{mask_index}-edit_space_rank-{pc_idx}-'
                             f'null_space_projection_{null_space_projection}-null_space_rank_{pca_rank_null}_{self.tilda_v_score_type}')
            xb = self._walk(original_xt, vT[pc_idx, :], vis_num)
            x0 = self.DDPMforwardsteps(xb, t_start_idx=self.edit_t_idx, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E,
                                       null_prompt_emb=N, mode="null+(for-null)")
        return x0

    @torch.no_grad()
    def run_edit_null_space_projection_xt_semantic(self, op, block_idx, vis_num, mask_index=0, vis_num_pc=1, vis_vT=False,
                                                   pca_rank=50, edit_prompt=None, null_space_projection=False,
                                                   pca_rank_null=50, jacobian=False):
        """edit.py:1871-2018: text-supervised direction (through the Jacobian or directly), projected onto the null
        space of the complement-mask Jacobian; ablations 'null-space-proj' and 'sega'."""
        self.scheduler.set_timesteps(self.for_steps)
        xT = self._xT()
        if self.mask_type != "SAM":
            raise NotImplementedError("mask_type 'diffedit' (edit.py:1395-1409) is not on this path")
        masks = self._masks()
        mask = masks[mask_index].squeeze(dim=0).repeat(3, 1, 1)
        if self.sampling_mode:
            return None
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        xt, t, t_idx = self.DDPMforwardsteps(xT, t_start_idx=0, t_end_idx=self.edit_t_idx, for_prompt_emb=F,
                                             edit_prompt_emb=E, null_prompt_emb=N, mode="null+(for-null)")
        assert t_idx == self.edit_t_idx
        save_dir = os.path.join(self.result_folder, "basis")
        os.makedirs(save_dir, exist_ok=True)
        if self.ablation_method == "null-space-proj":
            if not self._exists(self.vT_path):
                vT_modify = self.get_v_modify(xt, t, t_idx, F, E, N, mask=mask, mode=self.tilda_v_score_type, jacobian=jacobian)
                vT_null = None
                if null_space_projection:
                    print('subspace solve: CFG-combined Jacobian')
                    _, _, vT_null = self.local_encoder_decoder_pullback_xt(
                        xt, t, t_idx, F, E, N, op=op, block_idx=block_idx, pca_rank=pca_rank_null, chunk_size=5, min_iter=10,
                        max_iter=50, convergence_threshold=1e-3, mask=~mask, mode="null+(for-null)")
                    vT_null = vT_null[:pca_rank_null, :].contiguous()
                vT = self.engine.null_project(vT_modify.contiguous(), vT_null)
                BASIS_NAME = (f"edit-{self.edit_t}T-edit_prompt-{self.edit_prompt}-select_mask{mask_index}-null_space_projection_"
                              f"{null_space_projection}_null_space_rank_{pca_rank_null}_{self.tilda_v_score_type}")
                print(BASIS_NAME)
                for pc_idx in range(min(vT.shape[0], vis_num_pc)):
                    self.EXP_NAME = f'Semantic_Edit_xt-{BASIS_NAME}-pc_{pc_idx:0=3d}'
                    if self.sharder.is_main:
                        torch.save(vT[[pc_idx], :], os.path.join(save_dir, f'{self.EXP_NAME}-vT.pt'))
            else:
                print('loading the basis from --vT_path')
                vT = self._load(self.vT_path).to(self.device, torch.float32)
                BASIS_NAME = f"load-basis-'{os.path.basename(self.vT_path)}'"
            original_xt = xt.clone()
            xb = None
            for pc_idx in range(vis_num_pc):
                self.EXP_NAME = f'Semantic_Edit_xt-{BASIS_NAME}_scale_{self.x_space_guidance_scale}'
                xb = self._walk(original_xt, vT[pc_idx, :], vis_num)
            x0 = self.DDPMforwardsteps(xb, t_start_idx=self.edit_t_idx, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E,
                                       null_prompt_emb=N, mode="null+(for-null)")
        elif self.ablation_method == "sega":
            self.EXP_NAME = f'sega-edit_prompt-{self.edit_prompt}-mask_type-{self.mask_type}-select_mask{mask_index}'
            x0 = self.DDPMforwardsteps(xt, t_start_idx=self.edit_t_idx, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E,
                                       null_prompt_emb=N, mode="null+(for-null)+(edit-null)")
        else:
            raise NotImplementedError(f"ablation_method {self.ablation_method!r} (diffedit needs MaskedDDPMforwardsteps)")
        return x0
