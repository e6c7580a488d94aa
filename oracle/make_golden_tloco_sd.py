"""Generate tests/golden/tloco_sd_tiny.pt by running the REFERENCE's ``EditStableDiffusion`` methods (imported read-only
from /root/reference with the stub modules of make_golden.py) on stand-in networks, and pin oracle/tloco_sd_oracle.py
against them.  Runs only in the build container; the fixture is data (inputs + expected outputs).

Stand-ins (diffusers' UNet2DConditionModel / AutoencoderKL are un-vendored): ``unet`` = the reference's guided-diffusion
``UNetModel`` on 4 latent channels with ``emb = time_embed(t) + cond_proj(mean_tokens(prompt_emb))`` (as in
make_golden_tloco.py); ``vae.decode`` = a decoder assembled from the reference's own DDPM modules
(models/ddpm/diffusion.py ``ResnetBlock`` with a zeroed ``temb_proj``, ``AttnBlock``, ``Upsample``, ``Normalize``) in the
latent-diffusion ``Decoder`` order.

    python oracle/make_golden_tloco_sd.py
"""
from __future__ import annotations

import os
import sys
import tempfile
import types
from argparse import Namespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import make_golden as mg  # noqa: E402
import make_golden_tloco as mgt  # noqa: E402


def ref_latent_unet(cfg, params):
    from models.guided_diffusion.unet import UNetModel
    ds = tuple(cfg.resolution // r for r in cfg.attn_resolutions)
    m = UNetModel(image_size=cfg.resolution, in_channels=cfg.in_channels, model_channels=cfg.ch, out_channels=cfg.out_ch,
                  num_res_blocks=cfg.num_res_blocks, attention_resolutions=ds, dropout=0, channel_mult=tuple(cfg.ch_mult),
                  num_head_channels=cfg.num_head_channels, use_scale_shift_norm=True, resblock_updown=True)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    m.eval(); m.requires_grad_(False)
    return m


class RefDecoder(torch.nn.Module):
    """Latent-diffusion ``Decoder`` order, assembled from the reference's DDPM blocks (diffusion.py:816-966)."""

    def __init__(self, cfg):
        super().__init__()
        from models.ddpm.diffusion import ResnetBlock, AttnBlock, Upsample, Normalize
        nn = torch.nn
        ch, mult = cfg.ch, tuple(cfg.ch_mult)
        self.cfg = cfg
        block_in = ch * mult[-1]
        self.post_quant_conv = nn.Conv2d(cfg.in_channels, cfg.in_channels, 1)     # AutoencoderKL.decode applies it first
        self.conv_in = nn.Conv2d(cfg.in_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        rb = lambda i, o: ResnetBlock(in_channels=i, out_channels=o, dropout=0.0, temb_channels=8)
        self.mid.block_1 = rb(block_in, block_in); self.mid.attn_1 = AttnBlock(block_in); self.mid.block_2 = rb(block_in, block_in)
        self.up = nn.ModuleList()
        res = cfg.resolution
        ups = {}
        for lvl in reversed(range(len(mult))):
            up = nn.Module(); up.block = nn.ModuleList(); up.attn = nn.ModuleList()
            for b in range(cfg.num_res_blocks + 1):
                up.block.append(rb(block_in, ch * mult[lvl])); block_in = ch * mult[lvl]
                if res in cfg.attn_resolutions:
                    up.attn.append(AttnBlock(block_in))
            if lvl != 0:
                up.upsample = Upsample(block_in, True); res *= 2
            ups[lvl] = up
        for lvl in range(len(mult)):
            self.up.append(ups[lvl])
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, cfg.out_ch, 3, 1, 1)

    def load(self, params):
        sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
        for name, mod in self.named_modules():
            if name.endswith("temb_proj"):      # the decoder has no time embedding: a zero projection adds nothing
                sd[name + ".weight"] = torch.zeros_like(mod.weight); sd[name + ".bias"] = torch.zeros_like(mod.bias)
        self.load_state_dict(sd, strict=True)
        self.eval(); self.requires_grad_(False)
        return self

    def decode(self, z):
        from models.ddpm.diffusion import nonlinearity
        temb = torch.zeros(z.shape[0], 8)
        h = self.conv_in(self.post_quant_conv(z))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, temb)), temb)
        for lvl in reversed(range(len(self.up))):
            up = self.up[lvl]
            for b in range(self.cfg.num_res_blocks + 1):
                h = up.block[b](h, temb)
                if len(up.attn) > 0:
                    h = up.attn[b](h)
            if lvl != 0:
                h = up.upsample(h)
        return types.SimpleNamespace(sample=self.conv_out(nonlinearity(self.norm_out(h))))


def main():
    redit, YHS, extract, PullBackDDPM = mg.import_reference()
    from utils.utils import get_stable_diffusion_scheduler
    import tloco_oracle as tl
    import tloco_sd_oracle as tsd
    import loco_oracle as orc
    from loco_edit_amd.config import TINY_LATENT as cfg, TINY_DECODER as dcfg, synth_params
    from loco_edit_amd.tloco import cond_params
    torch.set_num_threads(8)
    tmpdir = tempfile.mkdtemp(prefix="loco_golden_tloco_sd_")
    D, NTOK = 16, 7
    params, dparams = synth_params(cfg, seed=0), synth_params(dcfg, seed=0)
    cp = cond_params(cfg, D, seed=0)
    cw, cb = torch.from_numpy(cp["cond_proj.weight"].copy()), torch.from_numpy(cp["cond_proj.bias"].copy())
    unet = mgt.ref_cond_unet(ref_latent_unet(cfg, params), cw, cb)
    vae = RefDecoder(dcfg).load(dparams)

    ed = object.__new__(redit.EditStableDiffusion)
    sched = types.SimpleNamespace()
    sched.betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2     # diffusers scaled_linear
    sched.alphas_cumprod = torch.cumprod(1.0 - sched.betas, dim=0)
    sched.scale_model_input = lambda x, t: x
    sargs = Namespace(use_yh_custom_scheduler=True, device=torch.device("cpu"), dtype=torch.float32)
    ed.scheduler = get_stable_diffusion_scheduler(sargs, sched)
    ed.unet, ed.vae = unet, vae
    ed.device, ed.dtype, ed.buffer_device, ed.memory_bound = torch.device("cpu"), torch.float32, "cpu", 50
    ed.for_steps, ed.use_yh_custom_scheduler = 100, True
    ed.guidance_scale, ed.guidance_scale_edit = 7.5, 4.0
    ed.result_folder, ed.EXP_NAME = tmpdir, "golden"
    ed.c_in, ed.image_size = cfg.in_channels, cfg.resolution
    ed.scheduler.set_timesteps(100, device="cpu")
    ed.edit_t = 0.7
    ed.edit_t_idx = (ed.scheduler.timesteps - 0.7 * 1000).abs().argmin()
    g = torch.Generator().manual_seed(31)
    for_e, edit_e, null_e = (torch.randn(1, NTOK, D, generator=g) for _ in range(3))
    ed.for_prompt_emb, ed.edit_prompt_emb, ed.null_prompt_emb = for_e, edit_e, null_e
    ed.tilda_v_score_type = "null+(for-null)+(edit-null)"

    po = orc.to_torch(params)
    po.update({k: torch.from_numpy(v.copy()) for k, v in cp.items()})
    ot = tsd.OracleTLocoSD(po, cfg, orc.to_torch(dparams), dcfg, guidance_scale=7.5, guidance_scale_edit=4.0)
    assert torch.equal(ot.sched.alphas_cumprod, sched.alphas_cumprod)
    assert torch.equal(ot.sched.timesteps, ed.scheduler.timesteps) and int(ed.edit_t_idx) == ot.edit_t_idx

    out = {"weights_seed": 0, "cond_dim": D, "for_e": for_e, "edit_e": edit_e, "null_e": null_e, "guidance_scale": 7.5,
           "guidance_scale_edit": 4.0, "alphas_cumprod": sched.alphas_cumprod.clone(),
           "timesteps": ed.scheduler.timesteps.clone(), "edit_t_idx": int(ed.edit_t_idx)}
    gx = torch.Generator().manual_seed(1)
    z = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=gx)
    t = ed.scheduler.timesteps[int(ed.edit_t_idx)]
    out["z"], out["t"] = z, t.clone()
    R = dcfg.out_resolution
    mask = torch.zeros(3, R, R, dtype=torch.bool); mask[:, 20:40, 12:44] = True
    out["mask"] = mask

    with torch.no_grad():
        # ---- 0. the decoder stand-in itself (reference blocks vs restatement)
        x_ref = vae.decode(torch.cat([z, 0.5 * z.flip(-1)])).sample
        mg.check("sd/decoder", orc.decoder_forward(orc.to_torch(dparams), dcfg, torch.cat([z, 0.5 * z.flip(-1)])), x_ref)
        out["dec_in"], out["dec_out"] = torch.cat([z, 0.5 * z.flip(-1)]), x_ref
        # ---- 1. CFG noise (edit.py:636-674) on a batch of 2
        zb = torch.cat([z, z.flip(-1)], dim=0)
        out["eps_modes"] = {}
        for mode in ("null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)"):
            e_ref = ed._classifer_free_guidance(zb, t, for_e, edit_e, null_e, mode, True)
            mg.check(f"sd/cfg[{mode}]", ot.cfg_noise(zb, t, for_e, edit_e, null_e, mode), e_ref)
            out["eps_modes"][mode] = e_ref
        # ---- 2. get_x0 = decode of the predicted clean latent (edit.py:757-781)
        x0m = ed.get_x0(z, t, ed.edit_t_idx, for_e, edit_e, null_e, mask=mask, mode="null+(for-null)+(edit-null)")
        mg.check("sd/get_x0 masked", ot.get_x0(z, t, for_e, edit_e, null_e, mask=mask), x0m)
        x0f = ed.get_x0(z, t, ed.edit_t_idx, for_e, edit_e, null_e, mask=None, mode="null+(for-null)")
        mg.check("sd/get_x0", ot.get_x0(z, t, for_e, edit_e, null_e, mode="null+(for-null)"), x0f)
        out["x0_masked"], out["x0_full"] = x0m, x0f

    # ---- 3. solver (edit.py:830-915), V0 injected
    gv = torch.Generator().manual_seed(7)
    v0 = torch.randn(cfg.n, 3, generator=gv)
    out["v0"] = v0
    real_randn = torch.randn

    def fake_randn(*size, **kw):
        if len(size) == 2 and size[0] == cfg.n:
            return v0[:, :size[1]].clone()
        return real_randn(*size, **kw)
    out["solver"] = {}
    for mode, n_iter, msk in (("null+(for-null)", 6, mask), ("null+(for-null)", 3, ~mask)):
        torch.randn = fake_randn
        try:
            with torch.no_grad():
                u, s, vT = ed.local_encoder_decoder_pullback_zt(z, t, ed.edit_t_idx, for_e, edit_e, null_e, pca_rank=3,
                                                                chunk_size=5, min_iter=n_iter, max_iter=n_iter,
                                                                convergence_threshold=1e-3, mask=msk, mode=mode)
        finally:
            torch.randn = real_randn
        ou, os_, ovT = ot.pullback(z, t, for_e, edit_e, null_e, 3, v0, min_iter=n_iter, max_iter=n_iter, mask=msk, mode=mode)
        mg.check(f"sd/solver[{mode}] s", os_, s, rtol=1e-3)
        c = mg.abs_cos_rows(ovT, vT)
        print(f"  oracle vs reference [sd/solver {mode}, L={int(msk.sum())}] |cos| min {c.min().item():.6f}")
        assert c.min() > 0.999
        key = "modify" if msk is mask else "null"
        out["solver"][key] = {"mode": mode, "n_iter": n_iter, "mask": msk, "s": s, "vT": vT, "u_norms": u.norm(dim=0)}

    # ---- 4. direction through the Jacobian of the decoded image (edit.py:784-828)
    v_grad = ed.get_delta_zt_via_grad(z, t, ed.edit_t_idx, for_e, edit_e, null_e, mask=mask,
                                      mode="null+(for-null)+(edit-null)").detach()
    mg.check("sd/delta_zt_via_grad", ot.delta_zt_via_grad(z, t, for_e, edit_e, null_e, mask), v_grad, rtol=1e-3)
    out["v_grad"] = v_grad

    # ---- 5. sampler: z_T -> z_t, then a 2-latent batch to the decoded images (edit.py:677-754)
    with torch.no_grad():
        gT = torch.Generator().manual_seed(5)
        zT = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=gT)
        zt_e, t_e, i_e = ed.DDIMforwardsteps(zT, t_start_idx=0, t_end_idx=ed.edit_t_idx, for_prompt_emb=for_e,
                                             edit_prompt_emb=edit_e, null_prompt_emb=null_e, mode="null+(for-null)")
        ozt, _, oi = ot.forwardsteps(zT, 0, ot.edit_t_idx, for_e, edit_e, null_e)
        assert int(i_e) == oi
        mg.check("sd/to_t", ozt, zt_e, rtol=1e-3, atol=1e-4)
        zb2 = torch.cat([zt_e, zt_e + 0.25 * z], dim=0)
        ed.memory_bound = 2            # batch 2 with CFG: chunk(2 // (2 // 2)), see make_golden_tloco.py
        lat, img = ed.DDIMforwardsteps(zb2, t_start_idx=ed.edit_t_idx, t_end_idx=-1, for_prompt_emb=for_e,
                                       edit_prompt_emb=edit_e, null_prompt_emb=null_e, mode="null+(for-null)")
        olat = ot.forwardsteps(zb2, ot.edit_t_idx, -1, for_e, edit_e, null_e, mode="null+(for-null)")
        olat_s, oimg, of32 = ot.decode_final(olat)
        mg.check("sd/decode latents", olat_s, lat, rtol=1e-3, atol=1e-4)
        diff = (oimg.int() - img.int()).abs()
        print(f"  oracle vs reference [sd/decode uint8] max |diff| {int(diff.max())}, differing pixels {int((diff > 0).sum())}")
        assert int(diff.max()) <= 1
        out.update(zT=zT, zt_edit=zt_e, t_edit=t_e.clone(), dec_lat_in=zb2, dec_latents=lat, dec_u8=img, dec_f32=of32)
    torch.save(out, os.path.join(GOLD, "tloco_sd_tiny.pt"))
    print("done ->", os.path.join(GOLD, "tloco_sd_tiny.pt"))


if __name__ == "__main__":
    main()
