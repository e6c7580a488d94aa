"""CPU: the latent T-LOCO oracle (oracle/tloco_sd_oracle.py) against the fixture the reference's own EditStableDiffusion
methods produced on the stand-in networks (tests/golden/tloco_sd_tiny.pt, oracle/make_golden_tloco_sd.py), plus the
host logic of the product module (scheduler table, preset routing)."""
import os

import pytest
import torch

import loco_oracle as orc
import tloco_sd_oracle as tsd
import loco_edit_amd  # noqa: F401
from loco_edit_amd.config import (TINY_DECODER, TINY_LATENT, SD64_STANDIN, SD64_XATTN_STANDIN, SD_VAE_DECODER, param_shapes,
                                  synth_params)
from loco_edit_amd.tloco import cond_params
from loco_edit_amd.tloco_sd import SDScheduler


@pytest.fixture(scope="module")
def setup(golden):
    g = golden("tloco_sd_tiny")
    p = orc.to_torch(synth_params(TINY_LATENT, 0))
    p.update({k: torch.from_numpy(v) for k, v in cond_params(TINY_LATENT, g["cond_dim"], 0).items()})
    ot = tsd.OracleTLocoSD(p, TINY_LATENT, orc.to_torch(synth_params(TINY_DECODER, 0)), TINY_DECODER,
                           guidance_scale=g["guidance_scale"], guidance_scale_edit=g["guidance_scale_edit"])
    return g, ot


def test_sd_scheduler_tables_and_shapes(setup):
    g, ot = setup
    s = SDScheduler()
    assert torch.equal(s.alphas_cumprod, g["alphas_cumprod"]) and torch.equal(ot.sched.alphas_cumprod, g["alphas_cumprod"])
    s.set_timesteps(100)
    assert torch.equal(s.timesteps, g["timesteps"]) and float(s.timesteps[0]) == 999.0
    assert g["edit_t_idx"] == int((s.timesteps - 700.0).abs().argmin()) == ot.edit_t_idx
    # the decoder preset has the published size of the Stable Diffusion autoencoder's decoder
    n_dec = sum(int(torch.tensor(sh).prod()) for sh in param_shapes(SD_VAE_DECODER).values())
    assert n_dec == 49_490_179 + 20 and SD_VAE_DECODER.n == 4 * 64 * 64 and SD_VAE_DECODER.n_out == 3 * 512 * 512
    assert SD64_STANDIN.n == SD64_STANDIN.n_out == 4 * 64 * 64 and TINY_DECODER.out_resolution == 64
    assert "mid.block_1.temb_proj.weight" not in param_shapes(TINY_DECODER) and "temb.dense.0.weight" not in param_shapes(TINY_DECODER)


def test_oracle_decoder_cfg_noise_and_x0(setup):
    g, ot = setup
    z, t = g["z"], g["t"]
    F, E, N = g["for_e"], g["edit_e"], g["null_e"]
    with torch.no_grad():
        assert torch.allclose(ot.decode(g["dec_in"]), g["dec_out"], rtol=1e-4, atol=1e-4)
        zb = torch.cat([z, z.flip(-1)], dim=0)
        for mode, ref in g["eps_modes"].items():
            assert torch.allclose(ot.cfg_noise(zb, t, F, E, N, mode), ref, rtol=1e-4, atol=1e-4)
        assert torch.allclose(ot.get_x0(z, t, F, E, N, mask=g["mask"]), g["x0_masked"], rtol=1e-4, atol=1e-3)
        assert torch.allclose(ot.get_x0(z, t, F, E, N, mode="null+(for-null)"), g["x0_full"], rtol=1e-4, atol=1e-3)
    v = ot.delta_zt_via_grad(z, t, F, E, N, g["mask"])
    assert torch.allclose(v, g["v_grad"], rtol=1e-3, atol=1e-6) and abs(float(v.norm()) - 1.0) < 1e-5


def test_oracle_latent_solver(setup):
    g, ot = setup
    sv = g["solver"]["modify"]
    u, s, vT = ot.pullback(g["z"], g["t"], g["for_e"], g["edit_e"], g["null_e"], 3, g["v0"], min_iter=sv["n_iter"],
                           max_iter=sv["n_iter"], mask=sv["mask"], mode=sv["mode"])
    assert torch.allclose(s, sv["s"], rtol=1e-3)
    assert (vT.double() * sv["vT"].double()).sum(dim=1).abs().min() > 0.9999
    assert u.shape == (int(sv["mask"].sum()), 3) and vT.shape == (3, TINY_LATENT.n)


def test_preset_routes_stable_diffusion_to_the_latent_path(tmp_path, monkeypatch):
    from loco_edit_amd import define_argparser
    monkeypatch.chdir(tmp_path)
    a = define_argparser.parse_args(["--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Random", "--note", "n",
                                     "--seed", "3", "--device", "cpu", "--run_edit_null_space_projection_zt", "True"])
    a = define_argparser.preset(a)
    assert a.is_stable_diffusion and not a.is_DeepFloyd_IF_diffusion and not a.is_LCM
    assert a.exp == "Stable_Diffusion-Random-n" and (a.c_in, a.image_size, a.memory_bound) == (4, 64, 5)
    from loco_edit_amd.config import SD15_UNET
    assert a.unet_config is SD15_UNET and a.vae_config is SD_VAE_DECODER      # the Stable Diffusion v1 denoiser architecture
    assert (a.unet_config.context_len, a.unet_config.context_dim) == (77, 768)
    assert (a.unet_config.transformer_depth, a.unet_config.num_heads, a.unet_config.scale_shift_norm) == (1, 8, False)
    c = define_argparser.parse_args(["--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Random", "--note", "n",
                                     "--seed", "3", "--device", "cpu", "--unet_preset", "sd64_xattn_standin"])
    assert define_argparser.preset(c).unet_config is SD64_XATTN_STANDIN            # the round-2 stand-in stays selectable
    # an image dataset (latent inversion, edit.py:568-633): read at the autoencoder's resolution, encoder preset attached
    d = define_argparser.parse_args(["--model_name", "runwayml/stable-diffusion-v1-5", "--dataset_name", "Synthetic", "--note", "n",
                                     "--seed", "3", "--device", "cpu", "--run_ddim_inversion", "True"])
    d = define_argparser.preset(d)
    from loco_edit_amd.config import SD_VAE_ENCODER
    assert d.vae_encoder_config is SD_VAE_ENCODER and tuple(d.dataset[0].shape) == (1, 3, 512, 512)
    assert getattr(a, "dataset", None) is None and a.vae_encoder_config is SD_VAE_ENCODER      # 'Random': no images
    b = define_argparser.parse_args(["--model_name", "SimianLuo/LCM_Dreamshaper_v7", "--seed", "3", "--device", "cpu"])
    with pytest.raises(NotImplementedError):
        define_argparser.preset(b)


def test_autoencoder_kl_key_map_round_trip():
    """diffusers AutoencoderKL naming <-> the decoder engine's naming: every decoder parameter survives the round trip,
    encoder / quant_conv entries are ignored, Linear attention projections become 1x1 convs."""
    from loco_edit_amd.checkpoints import decoder_to_hf_autoencoder_kl, hf_autoencoder_kl_to_decoder, is_hf_autoencoder_kl
    sd = {k: torch.from_numpy(v) for k, v in synth_params(TINY_DECODER, 3).items()}
    hf = decoder_to_hf_autoencoder_kl(sd, TINY_DECODER)
    assert is_hf_autoencoder_kl(hf) and not is_hf_autoencoder_kl(sd)
    assert "decoder.up_blocks.0.resnets.0.conv1.weight" in hf and "decoder.mid_block.attentions.0.to_q.weight" in hf
    assert hf["decoder.mid_block.attentions.0.to_q.weight"].dim() == 2 and "post_quant_conv.weight" in hf
    # the coarsest level is up_blocks.0 in diffusers, up.<levels-1> here
    assert torch.equal(hf["decoder.up_blocks.0.resnets.0.conv1.weight"], sd[f"up.{len(TINY_DECODER.ch_mult) - 1}.block.0.conv1.weight"])
    hf["encoder.conv_in.weight"] = torch.zeros(1); hf["quant_conv.weight"] = torch.zeros(1)
    back = hf_autoencoder_kl_to_decoder(hf, TINY_DECODER)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    del hf["decoder.conv_out.bias"]
    with pytest.raises(KeyError):
        hf_autoencoder_kl_to_decoder(hf, TINY_DECODER)


def test_oracle_latent_inversion_vs_reference_fixture(golden):
    """tests/golden/tloco_sd_inv.pt = the reference's own run_DDIMinversion (edit.py:568-633) on the stand-ins."""
    from loco_edit_amd.config import TINY_ENCODER, SD_VAE_ENCODER
    gi, gt = golden("tloco_sd_inv"), golden("tloco_sd_tiny")
    p = orc.to_torch(synth_params(TINY_LATENT, 0))
    p.update({k: torch.from_numpy(v) for k, v in cond_params(TINY_LATENT, gi["cond_dim"], 0).items()})
    ot = tsd.OracleTLocoSD(p, TINY_LATENT, orc.to_torch(synth_params(TINY_DECODER, 0)), TINY_DECODER,
                           guidance_scale=gi["guidance_scale"], guidance_scale_edit=gt["guidance_scale_edit"])
    ep = orc.to_torch(synth_params(TINY_ENCODER, 0))
    with torch.no_grad():
        assert torch.allclose(orc.encoder_forward(ep, TINY_ENCODER, gi["x0"]), gi["moments"], rtol=1e-4, atol=1e-5)
        for key, guidance in (("plain", None), ("cfg", True)):
            zT, z0 = ot.inversion(gi["x0"], gi[key]["noise"], ep, TINY_ENCODER, gi["inv_e"], gi["null_e"], gi["inv_steps"],
                                  guidance=guidance, return_z0=True)
            assert torch.allclose(z0, gi[key]["z0"], rtol=1e-4, atol=1e-5)
            assert torch.allclose(zT, gi[key]["zT"], rtol=1e-3, atol=1e-3), key
    # product scheduler: the inversion timesteps of utils.py:172-179
    s = SDScheduler()
    s.set_timesteps(gi["inv_steps"], is_inversion=True)
    assert torch.equal(s.timesteps, gi["timesteps"]) and torch.equal(s.timesteps_next, gi["timesteps_next"])
    assert float(s.timesteps[0]) == pytest.approx(1e-6) and s.alpha_at(s.timesteps[0]) == float(s.alphas_cumprod[0])
    # the encoder preset has the published size of the Stable Diffusion autoencoder's encoder (+ quant_conv)
    n_enc = sum(int(torch.tensor(sh).prod()) for sh in param_shapes(SD_VAE_ENCODER).values())
    assert n_enc == 34_163_592 + 72 and SD_VAE_ENCODER.n == 3 * 512 * 512 and SD_VAE_ENCODER.n_out == 8 * 64 * 64
    assert TINY_ENCODER.out_resolution == TINY_LATENT.resolution and TINY_ENCODER.out_ch == 2 * TINY_LATENT.in_channels


def test_autoencoder_kl_encoder_key_map_round_trip():
    from loco_edit_amd.checkpoints import encoder_to_hf_autoencoder_kl, hf_autoencoder_kl_to_encoder
    from loco_edit_amd.config import TINY_ENCODER
    sd = {k: torch.from_numpy(v) for k, v in synth_params(TINY_ENCODER, 3).items()}
    hf = encoder_to_hf_autoencoder_kl(sd, TINY_ENCODER)
    assert "encoder.down_blocks.0.resnets.0.conv1.weight" in hf and "encoder.down_blocks.0.downsamplers.0.conv.weight" in hf
    assert hf["encoder.mid_block.attentions.0.to_q.weight"].dim() == 2 and "quant_conv.weight" in hf
    assert len(hf) == len(sd)
    hf["decoder.conv_in.weight"] = torch.zeros(1); hf["post_quant_conv.weight"] = torch.zeros(1)
    back = hf_autoencoder_kl_to_encoder(hf, TINY_ENCODER)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    del hf["encoder.conv_out.bias"]
    with pytest.raises(KeyError):
        hf_autoencoder_kl_to_encoder(hf, TINY_ENCODER)


def test_shipped_stable_diffusion_model_id_gets_the_2_1_base_architecture(tmp_path, monkeypatch):
    """scripts/main_T2I_StableDiffusion_null_space_projection*.sh:4 name `stabilityai/stable-diffusion-2-1-base`: names of
    the 2.x family build config.SD21_BASE_UNET (1024-wide prompt states, 64-channel heads = 5/10/20/20 heads, the published
    865 910 724 parameters), v1 names keep SD15_UNET."""
    import json
    import numpy as np
    from loco_edit_amd import define_argparser
    from loco_edit_amd.config import SD15_UNET, SD21_BASE_UNET, param_shapes
    monkeypatch.chdir(tmp_path)
    scripts = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "script_args.json")))
    ids = {a[a.index("--model_name") + 1] for a in scripts.values() if "--model_name" in a and "stable-diffusion" in a[a.index("--model_name") + 1]}
    assert ids == {"stabilityai/stable-diffusion-2-1-base"}
    a = define_argparser.preset(define_argparser.parse_args(
        ["--model_name", "stabilityai/stable-diffusion-2-1-base", "--dataset_name", "Random", "--note", "n", "--seed", "3",
         "--device", "cpu", "--run_edit_null_space_projection_zt", "True"]))
    assert a.unet_config is SD21_BASE_UNET and (a.unet_config.context_len, a.unet_config.context_dim) == (77, 1024)
    assert a.unet_config.num_heads == -1 and a.unet_config.num_head_channels == 64
    assert sum(int(np.prod(v)) for v in param_shapes(SD21_BASE_UNET).values()) == 865_910_724
    b = define_argparser.preset(define_argparser.parse_args(
        ["--model_name", "CompVis/stable-diffusion-v1-4", "--dataset_name", "Random", "--note", "n", "--seed", "3", "--device", "cpu"]))
    assert b.unet_config is SD15_UNET


def test_stable_diffusion_checkpoint_layouts_load_by_name():
    """checkpoints.normalize_unet_state_dict: (1) a CompVis pipeline file (`model.diffusion_model.*` next to
    `first_stage_model.*`, `cond_stage_model.*`, schedule buffers, wrapped in {"state_dict": ...}); (2) the diffusers
    UNet2DConditionModel naming, with nn.Linear proj_in / proj_out as Stable Diffusion 2.x stores them; both come back as
    exactly the engine's parameter list with its shapes and values.  Foreign keys are refused with their names."""
    from loco_edit_amd.checkpoints import (hf_unet2d_condition_to_ldm, is_compvis_sd, is_hf_unet2d_condition,
                                           ldm_to_hf_unet2d_condition, normalize_unet_state_dict)
    from loco_edit_amd.config import SD21_BASE_UNET, TINY_LDM, param_shapes
    cfg = TINY_LDM
    sd = {k: torch.from_numpy(v) for k, v in synth_params(cfg, 5).items()}
    want = param_shapes(cfg)
    ck = {"model.diffusion_model." + k: v for k, v in sd.items()}
    ck.update({"first_stage_model.decoder.conv_in.weight": torch.zeros(2), "cond_stage_model.transformer.x": torch.zeros(1),
               "betas": torch.zeros(3), "model_ema.decay": torch.zeros(())})
    assert is_compvis_sd(ck)
    got = normalize_unet_state_dict({"state_dict": ck}, cfg)
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    hf = ldm_to_hf_unet2d_condition(sd, cfg)
    assert is_hf_unet2d_condition(hf) and not is_hf_unet2d_condition(sd) and len(hf) == len(sd)
    for k in ("time_embedding.linear_1.weight", "conv_in.weight", "down_blocks.0.resnets.0.time_emb_proj.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight", "down_blocks.0.downsamplers.0.conv.weight",
              "mid_block.attentions.0.proj_in.weight", "mid_block.resnets.1.conv2.weight", "up_blocks.0.resnets.1.conv_shortcut.weight",
              "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.1.attentions.1.transformer_blocks.0.ff.net.0.proj.weight",
              "conv_norm_out.weight", "conv_out.bias"):
        assert k in hf, k
    hf2 = {k: (v[:, :, 0, 0] if k.endswith(("proj_in.weight", "proj_out.weight")) else v) for k, v in hf.items()}   # 2.x: nn.Linear
    got = normalize_unet_state_dict(hf2, cfg)
    assert set(got) == set(sd) and all(tuple(got[k].shape) == tuple(want[k]) and torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(ValueError, match="not parameters of this architecture"):
        normalize_unet_state_dict({**sd, "lora.up.weight": torch.zeros(1)}, cfg)
    # the full-size layout: the map is a bijection onto the 686 tensors of the 2.1-base architecture (names only)
    big = param_shapes(SD21_BASE_UNET)
    names = ldm_to_hf_unet2d_condition({k: None for k in big}, SD21_BASE_UNET)
    assert len(names) == len(big) and "up_blocks.0.upsamplers.0.conv.weight" in names and "up_blocks.3.attentions.2.proj_out.bias" in names
    assert set(hf_unet2d_condition_to_ldm(names, SD21_BASE_UNET)) == set(big)
