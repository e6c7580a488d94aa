"""Checkpoint key maps.

The engine consumes the state_dict layout of the reference's vendored modules.  The shipped scripts load
``google/ddpm-ema-celebahq-256`` through diffusers (``utils.py:93-100,122-125``); diffusers is not vendored and
there is no network here, so the map below is written from the published ``UNet2DModel`` naming (the inverse of
diffusers' ``convert_ddpm_original_checkpoint_to_diffusers.py``) and is only self-consistency tested
(**parity unpinned** until a real checkpoint is available).
"""
from __future__ import annotations

from typing import Dict

import torch

from .config import UNetConfig, param_shapes

_RES = {"norm1": "norm1", "conv1": "conv1", "time_emb_proj": "temb_proj", "norm2": "norm2", "conv2": "conv2",
        "conv_shortcut": "nin_shortcut"}
_ATTN = {"group_norm": "norm", "query": "q", "key": "k", "value": "v", "proj_attn": "proj_out",
         "to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out"}


def is_hf_unet2d(sd: Dict[str, torch.Tensor]) -> bool:
    return any(k.startswith("down_blocks.") for k in sd) and "conv_norm_out.weight" in sd


def hf_unet2d_to_vendored(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """diffusers ``UNet2DModel`` (DDPM flavour) keys -> ``models/ddpm/diffusion.py`` keys.
    Attention projections are ``nn.Linear`` [C,C] in diffusers and 1x1 ``Conv2d`` [C,C,1,1] in the vendored module."""
    nlev = len(cfg.ch_mult)
    out: Dict[str, torch.Tensor] = {}

    def put(name, v):
        out[name] = v

    for k, v in sd.items():
        p = k.split(".")
        suffix = p[-1]                      # weight / bias
        if k.startswith("time_embedding.linear_1."):
            put(f"temb.dense.0.{suffix}", v)
        elif k.startswith("time_embedding.linear_2."):
            put(f"temb.dense.1.{suffix}", v)
        elif p[0] in ("conv_in", "conv_out"):
            put(k, v)
        elif p[0] == "conv_norm_out":
            put(f"norm_out.{suffix}", v)
        elif p[0] in ("down_blocks", "up_blocks"):
            lvl = int(p[1]) if p[0] == "down_blocks" else nlev - 1 - int(p[1])
            side = "down" if p[0] == "down_blocks" else "up"
            if p[2] == "resnets":
                put(f"{side}.{lvl}.block.{p[3]}.{_RES[p[4]]}.{suffix}", v)
            elif p[2] == "attentions":
                sub = ".".join(p[4:-1])
                name = _ATTN[sub]
                if name != "norm" and suffix == "weight" and v.dim() == 2:
                    v = v[:, :, None, None]
                put(f"{side}.{lvl}.attn.{p[3]}.{name}.{suffix}", v)
            elif p[2] == "downsamplers":
                put(f"down.{lvl}.downsample.conv.{suffix}", v)
            elif p[2] == "upsamplers":
                put(f"up.{lvl}.upsample.conv.{suffix}", v)
            else:
                raise KeyError(k)
        elif p[0] == "mid_block":
            if p[1] == "resnets":
                put(f"mid.block_{int(p[2]) + 1}.{_RES[p[3]]}.{suffix}", v)
            elif p[1] == "attentions":
                sub = ".".join(p[3:-1])
                name = _ATTN[sub]
                if name != "norm" and suffix == "weight" and v.dim() == 2:
                    v = v[:, :, None, None]
                put(f"mid.attn_1.{name}.{suffix}", v)
            else:
                raise KeyError(k)
        else:
            raise KeyError(f"unexpected key {k}")
    want = param_shapes(cfg)
    missing = [n for n in want if n not in out]
    if missing:
        raise KeyError(f"{len(missing)} parameters missing after conversion, first: {missing[0]}")
    for n, shp in want.items():
        if tuple(out[n].shape) != tuple(shp):
            raise ValueError(f"shape mismatch for {n}: {tuple(out[n].shape)} vs {tuple(shp)}")
    return out


def vendored_to_hf_unet2d(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Inverse map (used by the round-trip test and to export checkpoints for a diffusers user)."""
    nlev = len(cfg.ch_mult)
    inv_res = {v: k for k, v in _RES.items()}
    inv_attn = {"norm": "group_norm", "q": "query", "k": "key", "v": "value", "proj_out": "proj_attn"}
    out = {}
    for k, v in sd.items():
        p = k.split(".")
        suffix = p[-1]
        if k.startswith("temb.dense.0."):
            out[f"time_embedding.linear_1.{suffix}"] = v
        elif k.startswith("temb.dense.1."):
            out[f"time_embedding.linear_2.{suffix}"] = v
        elif p[0] in ("conv_in", "conv_out"):
            out[k] = v
        elif p[0] == "norm_out":
            out[f"conv_norm_out.{suffix}"] = v
        elif p[0] in ("down", "up"):
            blk = "down_blocks" if p[0] == "down" else "up_blocks"
            idx = int(p[1]) if p[0] == "down" else nlev - 1 - int(p[1])
            if p[2] == "block":
                out[f"{blk}.{idx}.resnets.{p[3]}.{inv_res[p[4]]}.{suffix}"] = v
            elif p[2] == "attn":
                if p[4] != "norm" and suffix == "weight":
                    v = v.reshape(v.shape[0], v.shape[1])
                out[f"{blk}.{idx}.attentions.{p[3]}.{inv_attn[p[4]]}.{suffix}"] = v
            elif p[2] == "downsample":
                out[f"{blk}.{idx}.downsamplers.0.conv.{suffix}"] = v
            elif p[2] == "upsample":
                out[f"{blk}.{idx}.upsamplers.0.conv.{suffix}"] = v
        elif p[0] == "mid":
            if p[1].startswith("block_"):
                out[f"mid_block.resnets.{int(p[1][-1]) - 1}.{inv_res[p[2]]}.{suffix}"] = v
            else:
                if p[2] != "norm" and suffix == "weight":
                    v = v.reshape(v.shape[0], v.shape[1])
                out[f"mid_block.attentions.0.{inv_attn[p[2]]}.{suffix}"] = v
    return out


# ---------------------------------------------------------------------------------------------------------------------
# diffusers ``AutoencoderKL`` (the ``vae`` of a Stable Diffusion pipeline, reference edit.py:498) -> the decoder engine's
# latent-diffusion ``Decoder`` naming (config.dec_param_shapes).  Only ``post_quant_conv`` and ``decoder.*`` are
# consumed (``vae.decode``); written from the published layout, **parity unpinned** like the U-Net map above.
def is_hf_autoencoder_kl(sd: Dict[str, torch.Tensor]) -> bool:
    return "decoder.conv_norm_out.weight" in sd and any(k.startswith("decoder.up_blocks.") for k in sd)


def hf_autoencoder_kl_to_decoder(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    nlev = len(cfg.ch_mult)
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        p = k.split(".")
        suffix = p[-1]
        if p[0] == "post_quant_conv":
            out[k] = v
            continue
        if p[0] != "decoder":
            continue                                    # encoder.*, quant_conv.*: vae.encode is not on the path
        p = p[1:]
        if p[0] in ("conv_in", "conv_out"):
            out[".".join(p)] = v
        elif p[0] == "conv_norm_out":
            out[f"norm_out.{suffix}"] = v
        elif p[0] == "up_blocks":
            lvl = nlev - 1 - int(p[1])                  # diffusers counts up blocks from the coarsest
            if p[2] == "resnets":
                out[f"up.{lvl}.block.{p[3]}.{_RES[p[4]]}.{suffix}"] = v
            elif p[2] == "upsamplers":
                out[f"up.{lvl}.upsample.conv.{suffix}"] = v
            else:
                raise KeyError(k)
        elif p[0] == "mid_block":
            if p[1] == "resnets":
                out[f"mid.block_{int(p[2]) + 1}.{_RES[p[3]]}.{suffix}"] = v
            elif p[1] == "attentions":
                name = _ATTN[".".join(p[3:-1])]
                if name != "norm" and suffix == "weight" and v.dim() == 2:
                    v = v[:, :, None, None]             # nn.Linear [C,C] -> 1x1 conv
                out[f"mid.attn_1.{name}.{suffix}"] = v
            else:
                raise KeyError(k)
        else:
            raise KeyError(f"unexpected key {k}")
    want = param_shapes(cfg)
    missing = [n for n in want if n not in out]
    if missing:
        raise KeyError(f"{len(missing)} decoder parameters missing after conversion, first: {missing[0]}")
    for n, shp in want.items():
        if tuple(out[n].shape) != tuple(shp):
            raise ValueError(f"shape mismatch for {n}: {tuple(out[n].shape)} vs {tuple(shp)}")
    return {n: out[n] for n in want}


def decoder_to_hf_autoencoder_kl(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Inverse map (round-trip test; export for a diffusers user), attention projections as nn.Linear."""
    nlev = len(cfg.ch_mult)
    inv_res = {v: k for k, v in _RES.items()}
    inv_attn = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}
    out = {}
    for k, v in sd.items():
        p = k.split(".")
        suffix = p[-1]
        if p[0] == "post_quant_conv":
            out[k] = v
        elif p[0] in ("conv_in", "conv_out"):
            out["decoder." + k] = v
        elif p[0] == "norm_out":
            out[f"decoder.conv_norm_out.{suffix}"] = v
        elif p[0] == "up":
            idx = nlev - 1 - int(p[1])
            if p[2] == "block":
                out[f"decoder.up_blocks.{idx}.resnets.{p[3]}.{inv_res[p[4]]}.{suffix}"] = v
            elif p[2] == "upsample":
                out[f"decoder.up_blocks.{idx}.upsamplers.0.conv.{suffix}"] = v
        elif p[0] == "mid":
            if p[1].startswith("block_"):
                out[f"decoder.mid_block.resnets.{int(p[1][-1]) - 1}.{inv_res[p[2]]}.{suffix}"] = v
            else:
                if p[2] != "norm" and suffix == "weight":
                    v = v.reshape(v.shape[0], v.shape[1])
                out[f"decoder.mid_block.attentions.0.{inv_attn[p[2]]}.{suffix}"] = v
    return out


def hf_autoencoder_kl_to_encoder(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """diffusers AutoencoderKL state_dict -> the encoder engine's naming (arch "enc": `quant_conv` + `encoder.*`, the half
    behind `vae.encode` of the latent inversion, reference edit.py:594-597).  Key map written from the published layout
    (`encoder.down_blocks.L.resnets.B`, `.downsamplers.0.conv`, `mid_block`, `conv_norm_out`), unpinned like the decoder's."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        p = k.split(".")
        suffix = p[-1]
        if p[0] == "quant_conv":
            out[k] = v
            continue
        if p[0] != "encoder":
            continue                                    # decoder.*, post_quant_conv.*: the other half
        p = p[1:]
        if p[0] in ("conv_in", "conv_out"):
            out[".".join(p)] = v
        elif p[0] == "conv_norm_out":
            out[f"norm_out.{suffix}"] = v
        elif p[0] == "down_blocks":
            if p[2] == "resnets":
                out[f"down.{p[1]}.block.{p[3]}.{_RES[p[4]]}.{suffix}"] = v
            elif p[2] == "downsamplers":
                out[f"down.{p[1]}.downsample.conv.{suffix}"] = v
            else:
                raise KeyError(k)
        elif p[0] == "mid_block":
            if p[1] == "resnets":
                out[f"mid.block_{int(p[2]) + 1}.{_RES[p[3]]}.{suffix}"] = v
            elif p[1] == "attentions":
                name = _ATTN[".".join(p[3:-1])]
                if name != "norm" and suffix == "weight" and v.dim() == 2:
                    v = v[:, :, None, None]             # nn.Linear [C,C] -> 1x1 conv
                out[f"mid.attn_1.{name}.{suffix}"] = v
            else:
                raise KeyError(k)
        else:
            raise KeyError(f"unexpected key {k}")
    want = param_shapes(cfg)
    missing = [n for n in want if n not in out]
    if missing:
        raise KeyError(f"{len(missing)} encoder parameters missing after conversion, first: {missing[0]}")
    for n, shp in want.items():
        if tuple(out[n].shape) != tuple(shp):
            raise ValueError(f"shape mismatch for {n}: {tuple(out[n].shape)} vs {tuple(shp)}")
    return {n: out[n] for n in want}


def encoder_to_hf_autoencoder_kl(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Inverse of ``hf_autoencoder_kl_to_encoder`` (round-trip test; export), attention projections as nn.Linear."""
    inv_res = {v: k for k, v in _RES.items()}
    inv_attn = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}
    out = {}
    for k, v in sd.items():
        p = k.split(".")
        suffix = p[-1]
        if p[0] == "quant_conv":
            out[k] = v
        elif p[0] in ("conv_in", "conv_out"):
            out["encoder." + k] = v
        elif p[0] == "norm_out":
            out[f"encoder.conv_norm_out.{suffix}"] = v
        elif p[0] == "down":
            if p[2] == "block":
                out[f"encoder.down_blocks.{p[1]}.resnets.{p[3]}.{inv_res[p[4]]}.{suffix}"] = v
            elif p[2] == "downsample":
                out[f"encoder.down_blocks.{p[1]}.downsamplers.0.conv.{suffix}"] = v
        elif p[0] == "mid":
            if p[1].startswith("block_"):
                out[f"encoder.mid_block.resnets.{int(p[1][-1]) - 1}.{inv_res[p[2]]}.{suffix}"] = v
            else:
                if p[2] != "norm" and suffix == "weight":
                    v = v.reshape(v.shape[0], v.shape[1])
                out[f"encoder.mid_block.attentions.0.{inv_attn[p[2]]}.{suffix}"] = v
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Stable Diffusion denoiser checkpoints (the reference gets the network from
# ``StableDiffusionPipeline.from_pretrained(...).unet``, utils.py:140-144 / edit.py:619-623).  The engine keeps the
# latent-diffusion ``UNetModel`` names (config.adm_param_shapes).  Two on-disk layouts are recognised:
#   * CompVis / Stability ``*.ckpt``: the whole pipeline in one state_dict -- the denoiser under
#     ``model.diffusion_model.``, next to ``first_stage_model.*`` (autoencoder), ``cond_stage_model.*`` (text encoder)
#     and the schedule buffers (``betas``, ``alphas_cumprod`` ...): keep the denoiser keys, strip the prefix;
#   * diffusers ``UNet2DConditionModel`` (``unet/diffusion_pytorch_model.*``): renamed by the inverse of diffusers'
#     ``convert_from_ckpt`` block map.  Written from the published naming; **parity unpinned** (no diffusers, no weights):
#     what is tested is that the map is a bijection onto the engine's parameter list with the right shapes.
# Stable Diffusion 2.x stores the transformer's proj_in / proj_out as ``nn.Linear`` [C, C] (``use_linear_in_transformer``);
# on [C][T] activations that is the same operator as v1's 1x1 conv [C, C, 1, 1], so the weights are reshaped.
_COMPVIS_PREFIX = "model.diffusion_model."
_LDM_RES = {"norm1": "in_layers.0", "conv1": "in_layers.2", "time_emb_proj": "emb_layers.1", "norm2": "out_layers.0",
            "conv2": "out_layers.3", "conv_shortcut": "skip_connection"}


def is_compvis_sd(sd: Dict[str, torch.Tensor]) -> bool:
    return any(k.startswith(_COMPVIS_PREFIX) for k in sd)


def is_hf_unet2d_condition(sd: Dict[str, torch.Tensor]) -> bool:
    return "time_embedding.linear_1.weight" in sd and any(".transformer_blocks." in k for k in sd) \
        and any(k.startswith("down_blocks.") for k in sd)


def compvis_sd_to_ldm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``model.diffusion_model.*`` of an ``sd-v1-x.ckpt`` / ``v2-1_512-ema-pruned.ckpt`` state_dict -> denoiser keys."""
    return {k[len(_COMPVIS_PREFIX):]: v for k, v in sd.items() if k.startswith(_COMPVIS_PREFIX)}


def _ldm_block_index(cfg: UNetConfig):
    """(level, block) -> input_blocks index / output_blocks index, and whether the level carries transformer blocks."""
    nrb, nlev = cfg.num_res_blocks, len(cfg.ch_mult)
    has_attn = [(cfg.resolution >> lvl) in cfg.attn_resolutions for lvl in range(nlev)]
    return nrb, nlev, has_attn


def hf_unet2d_condition_to_ldm(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """diffusers ``UNet2DConditionModel`` keys -> latent-diffusion ``UNetModel`` keys (the layout of
    ``config.adm_param_shapes`` for the SpatialTransformer presets)."""
    nrb, nlev, has_attn = _ldm_block_index(cfg)
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        p = k.split(".")
        if p[0] == "time_embedding":
            out[f"time_embed.{0 if p[1] == 'linear_1' else 2}.{p[-1]}"] = v
        elif p[0] == "conv_in":
            out[f"input_blocks.0.0.{p[-1]}"] = v
        elif p[0] == "conv_norm_out":
            out[f"out.0.{p[-1]}"] = v
        elif p[0] == "conv_out":
            out[f"out.2.{p[-1]}"] = v
        elif p[0] == "down_blocks":
            lvl = int(p[1])
            if p[2] == "resnets":
                out[f"input_blocks.{(nrb + 1) * lvl + int(p[3]) + 1}.0.{_LDM_RES[p[4]]}.{p[-1]}"] = v
            elif p[2] == "attentions":
                out[f"input_blocks.{(nrb + 1) * lvl + int(p[3]) + 1}.1." + ".".join(p[4:])] = v
            elif p[2] == "downsamplers":
                out[f"input_blocks.{(nrb + 1) * (lvl + 1)}.0.op.{p[-1]}"] = v
            else:
                raise KeyError(k)
        elif p[0] == "mid_block":
            if p[1] == "resnets":
                out[f"middle_block.{0 if p[2] == '0' else 2}.{_LDM_RES[p[3]]}.{p[-1]}"] = v
            elif p[1] == "attentions":
                out["middle_block.1." + ".".join(p[3:])] = v
            else:
                raise KeyError(k)
        elif p[0] == "up_blocks":
            i = int(p[1])
            lvl = nlev - 1 - i
            if p[2] == "resnets":
                out[f"output_blocks.{(nrb + 1) * i + int(p[3])}.0.{_LDM_RES[p[4]]}.{p[-1]}"] = v
            elif p[2] == "attentions":
                out[f"output_blocks.{(nrb + 1) * i + int(p[3])}.1." + ".".join(p[4:])] = v
            elif p[2] == "upsamplers":
                out[f"output_blocks.{(nrb + 1) * i + nrb}.{2 if has_attn[lvl] else 1}.conv.{p[-1]}"] = v
            else:
                raise KeyError(k)
        else:
            raise KeyError(f"not a UNet2DConditionModel key: {k}")
    return out


def ldm_to_hf_unet2d_condition(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Inverse of ``hf_unet2d_condition_to_ldm`` (round-trip tests; exporting a synthetic checkpoint in diffusers naming)."""
    nrb, nlev, has_attn = _ldm_block_index(cfg)
    inv_res = {v: k for k, v in _LDM_RES.items()}
    out: Dict[str, torch.Tensor] = {}

    def res_name(rest):          # "in_layers.0.weight" -> "norm1.weight"
        q = rest.split(".")
        return inv_res[".".join(q[:-1])] + "." + q[-1]

    for k, v in sd.items():
        p = k.split(".")
        if p[0] == "time_embed":
            out[f"time_embedding.linear_{1 if p[1] == '0' else 2}.{p[-1]}"] = v
        elif p[0] == "out":
            out[f"{'conv_norm_out' if p[1] == '0' else 'conv_out'}.{p[-1]}"] = v
        elif p[0] == "input_blocks":
            ib = int(p[1])
            if ib == 0:
                out[f"conv_in.{p[-1]}"] = v
                continue
            lvl, j = divmod(ib - 1, nrb + 1)
            if p[3] == "op":
                out[f"down_blocks.{lvl}.downsamplers.0.conv.{p[-1]}"] = v
            elif p[2] == "0":
                out[f"down_blocks.{lvl}.resnets.{j}." + res_name(".".join(p[3:]))] = v
            else:
                out[f"down_blocks.{lvl}.attentions.{j}." + ".".join(p[3:])] = v
        elif p[0] == "middle_block":
            if p[1] == "1":
                out["mid_block.attentions.0." + ".".join(p[2:])] = v
            else:
                out[f"mid_block.resnets.{0 if p[1] == '0' else 1}." + res_name(".".join(p[2:]))] = v
        elif p[0] == "output_blocks":
            i, j = divmod(int(p[1]), nrb + 1)
            if p[3] == "conv":
                out[f"up_blocks.{i}.upsamplers.0.conv.{p[-1]}"] = v
            elif p[2] == "0":
                out[f"up_blocks.{i}.resnets.{j}." + res_name(".".join(p[3:]))] = v
            else:
                out[f"up_blocks.{i}.attentions.{j}." + ".".join(p[3:])] = v
        else:
            raise KeyError(k)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# DeepFloyd IF stage I: the diffusers ``UNet2DConditionModel`` of the IF pipelines (`self.stage_1.unet`, reference
# src/utils/utils.py:260-283, src/modules/edit.py:1212-1222) <-> the names of ``config.adm_param_shapes`` for the
# ``added_kv`` presets (the module tree of the deepfloyd_if package: guided-diffusion names + ``encoder_kv`` /
# ``norm_encoder`` in the attention blocks, ``encoder_proj`` / ``encoder_pooling`` for the text conditioning).  Written
# from the published layouts -- ResnetDownsampleBlock2D / SimpleCrossAttn{Down,Up}Block2D / UNetMidBlock2DSimpleCrossAttn
# with ``Attention(added_kv_proj_dim=...)``: ``group_norm``, ``to_q / to_k / to_v`` (Linear with bias), ``add_k_proj /
# add_v_proj``, ``norm_cross`` (GroupNorm over the states), ``to_out.0``; resampling ResnetBlock2Ds under ``downsamplers.0``
# / ``upsamplers.0``; ``add_embedding`` = TextTimeEmbedding, ``encoder_hid_proj`` -- **parity unpinned** (no diffusers, no
# weights); tested: the map is a bijection onto the parameter list with the right shapes, and the per-head interleave of
# q / k / v (qkv rows of head h = [q_h | k_h | v_h], encoder_kv rows = [k_h | v_h]) round-trips.
_IF_POOL = {"norm1": "0", "pool": "1", "proj": "2", "norm2": "3"}


def is_hf_if_unet(sd: Dict[str, torch.Tensor]) -> bool:
    return "time_embedding.linear_1.weight" in sd and any(".add_k_proj." in k for k in sd)


def _interleave_heads(parts, heads):
    """[to_q | to_k | to_v] (each [C, ...], heads = contiguous blocks of C / heads rows) -> rows of head h = [q_h | k_h | v_h]."""
    ch = parts[0].shape[0] // heads
    return torch.cat([p_[h * ch:(h + 1) * ch] for h in range(heads) for p_ in parts], dim=0)


def _split_heads(w, heads, n):
    """Inverse of ``_interleave_heads``: n tensors of [C, ...] from [n C, ...] with per-head row blocks."""
    ch = w.shape[0] // (heads * n)
    return [torch.cat([w[(h * n + i) * ch:(h * n + i + 1) * ch] for h in range(heads)], dim=0) for i in range(n)]


def hf_if_unet_to_native(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    if not (cfg.added_kv and cfg.num_head_channels > 0):
        raise ValueError("a DeepFloyd-IF checkpoint needs an added-KV architecture (config.if_stage1_config), "
                         f"not {cfg.arch!r} with added_kv={cfg.added_kv}")
    nrb, nlev, has_attn = _ldm_block_index(cfg)
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    out: Dict[str, torch.Tensor] = {}
    attn: Dict[str, Dict[str, torch.Tensor]] = {}

    def res(dst, p, v):          # p: [..., "norm1", "weight"]
        out[f"{dst}.{_LDM_RES[p[-2]]}.{p[-1]}"] = v

    for k, v in sd.items():
        p = k.split(".")
        if p[0] == "time_embedding":
            out[f"time_embed.{0 if p[1] == 'linear_1' else 2}.{p[-1]}"] = v
        elif p[0] == "encoder_hid_proj":
            out[f"encoder_proj.{p[-1]}"] = v
        elif p[0] == "add_embedding":
            out["encoder_pooling." + _IF_POOL[p[1]] + "." + ".".join(p[2:])] = v
        elif p[0] == "conv_in":
            out[f"input_blocks.0.0.{p[-1]}"] = v
        elif p[0] == "conv_norm_out":
            out[f"out.0.{p[-1]}"] = v
        elif p[0] == "conv_out":
            out[f"out.2.{p[-1]}"] = v
        elif p[0] == "down_blocks":
            lvl = int(p[1])
            if p[2] == "resnets":
                res(f"input_blocks.{(nrb + 1) * lvl + int(p[3]) + 1}.0", p, v)
            elif p[2] == "downsamplers":
                res(f"input_blocks.{(nrb + 1) * (lvl + 1)}.0", p, v)
            elif p[2] == "attentions":
                attn.setdefault(f"input_blocks.{(nrb + 1) * lvl + int(p[3]) + 1}.1", {})[".".join(p[4:])] = v
            else:
                raise KeyError(k)
        elif p[0] == "mid_block":
            if p[1] == "resnets":
                res(f"middle_block.{0 if p[2] == '0' else 2}", p, v)
            elif p[1] == "attentions":
                attn.setdefault("middle_block.1", {})[".".join(p[3:])] = v
            else:
                raise KeyError(k)
        elif p[0] == "up_blocks":
            i = int(p[1])
            lvl = nlev - 1 - i
            if p[2] == "resnets":
                res(f"output_blocks.{(nrb + 1) * i + int(p[3])}.0", p, v)
            elif p[2] == "upsamplers":
                res(f"output_blocks.{(nrb + 1) * i + nrb}.{2 if has_attn[lvl] else 1}", p, v)
            elif p[2] == "attentions":
                attn.setdefault(f"output_blocks.{(nrb + 1) * i + int(p[3])}.1", {})[".".join(p[4:])] = v
            else:
                raise KeyError(k)
        else:
            raise KeyError(f"not a key of the IF UNet2DConditionModel: {k}")
    for name, a in attn.items():
        C = a["to_q.weight"].shape[0]
        heads = C // cfg.num_head_channels
        for wb in ("weight", "bias"):
            out[f"{name}.norm.{wb}"] = a[f"group_norm.{wb}"]
            out[f"{name}.norm_encoder.{wb}"] = a[f"norm_cross.{wb}"]
            qkv = _interleave_heads([a[f"to_q.{wb}"], a[f"to_k.{wb}"], a[f"to_v.{wb}"]], heads)
            ekv = _interleave_heads([a[f"add_k_proj.{wb}"], a[f"add_v_proj.{wb}"]], heads)
            po = a[f"to_out.0.{wb}"]
            out[f"{name}.qkv.{wb}"] = qkv[..., None] if wb == "weight" else qkv
            out[f"{name}.encoder_kv.{wb}"] = ekv[..., None] if wb == "weight" else ekv
            out[f"{name}.proj_out.{wb}"] = po[..., None] if wb == "weight" else po
    return out


def native_to_hf_if_unet(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Inverse of ``hf_if_unet_to_native`` (round-trip tests; exporting a synthetic checkpoint in diffusers naming)."""
    nrb, nlev, has_attn = _ldm_block_index(cfg)
    inv_res = {v: k for k, v in _LDM_RES.items()}
    inv_pool = {v: k for k, v in _IF_POOL.items()}
    out: Dict[str, torch.Tensor] = {}

    def res(dst, rest, v):       # rest: ["in_layers", "0", "weight"]
        out[f"{dst}.{inv_res['.'.join(rest[:-1])]}.{rest[-1]}"] = v

    def att(dst, rest, v):
        v = torch.as_tensor(v)
        what, wb = rest[0], rest[-1]
        if what == "norm":
            out[f"{dst}.group_norm.{wb}"] = v
        elif what == "norm_encoder":
            out[f"{dst}.norm_cross.{wb}"] = v
        elif what == "proj_out":
            out[f"{dst}.to_out.0.{wb}"] = v[..., 0] if wb == "weight" else v
        else:
            n = 3 if what == "qkv" else 2
            w = v[..., 0] if wb == "weight" else v
            heads = (w.shape[0] // n) // cfg.num_head_channels
            names = ("to_q", "to_k", "to_v") if what == "qkv" else ("add_k_proj", "add_v_proj")
            for nm, part in zip(names, _split_heads(w, heads, n)):
                out[f"{dst}.{nm}.{wb}"] = part

    for k, v in sd.items():
        p = k.split(".")
        if p[0] == "time_embed":
            out[f"time_embedding.linear_{1 if p[1] == '0' else 2}.{p[-1]}"] = v
        elif p[0] == "encoder_proj":
            out[f"encoder_hid_proj.{p[-1]}"] = v
        elif p[0] == "encoder_pooling":
            out["add_embedding." + inv_pool[p[1]] + "." + ".".join(p[2:])] = v
        elif p[0] == "out":
            out[f"{'conv_norm_out' if p[1] == '0' else 'conv_out'}.{p[-1]}"] = v
        elif p[0] == "input_blocks":
            ib = int(p[1])
            if ib == 0:
                out[f"conv_in.{p[-1]}"] = v
                continue
            lvl, j = divmod(ib - 1, nrb + 1)
            if j == nrb:
                res(f"down_blocks.{lvl}.downsamplers.0", p[3:], v)
            elif p[2] == "0":
                res(f"down_blocks.{lvl}.resnets.{j}", p[3:], v)
            else:
                att(f"down_blocks.{lvl}.attentions.{j}", p[3:], v)
        elif p[0] == "middle_block":
            if p[1] == "1":
                att("mid_block.attentions.0", p[2:], v)
            else:
                res(f"mid_block.resnets.{0 if p[1] == '0' else 1}", p[2:], v)
        elif p[0] == "output_blocks":
            i, j = divmod(int(p[1]), nrb + 1)
            lvl = nlev - 1 - i
            if p[2] == "0":
                res(f"up_blocks.{i}.resnets.{j}", p[3:], v)
            elif p[2] == "1" and has_attn[lvl]:
                att(f"up_blocks.{i}.attentions.{j}", p[3:], v)
            else:
                res(f"up_blocks.{i}.upsamplers.0", p[3:], v)
        else:
            raise KeyError(k)
    return out


def normalize_unet_state_dict(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """Whatever ``--ckpt_path`` held -> the engine's parameter names and shapes for ``cfg``.

    Unwraps ``{"state_dict": ...}``, recognises the CompVis pipeline layout and the diffusers ``UNet2DConditionModel``
    layout (above) and the diffusers ``UNet2DModel`` layout of the unconditional models, reshapes ``nn.Linear``
    proj_in / proj_out weights of Stable Diffusion 2.x to the 1x1-conv shape, and refuses what is left over with the
    list of foreign keys instead of failing on the first one inside ``loco_load_param``."""
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    if is_compvis_sd(sd):
        sd = compvis_sd_to_ldm(sd)
    elif is_hf_if_unet(sd):
        sd = hf_if_unet_to_native(sd, cfg)
    elif is_hf_unet2d_condition(sd):
        sd = hf_unet2d_condition_to_ldm(sd, cfg)
    elif is_hf_unet2d(sd) and cfg.arch == "ddpm":
        sd = hf_unet2d_to_vendored(sd, cfg)
    want = param_shapes(cfg)
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k in want and len(want[k]) == 4 and getattr(v, "ndim", 0) == 2 and tuple(v.shape) == tuple(want[k][:2]) \
                and tuple(want[k][2:]) == (1, 1):
            v = v[:, :, None, None]
        out[k] = v
    foreign = [k for k in out if k not in want and not k.startswith("cond_proj.")]
    if foreign:
        raise ValueError(f"{len(foreign)} checkpoint keys are not parameters of this architecture "
                         f"(first: {foreign[:3]}); expected e.g. {list(want)[:2]}")
    return out
