"""Generate tests/golden/tloco_sd_inv.pt by running the REFERENCE's ``EditStableDiffusion.run_DDIMinversion``
(src/modules/edit.py:568-633, imported read-only from /root/reference with the stub modules of make_golden.py) on
stand-in networks, and pin ``tloco_sd_oracle.OracleTLocoSD.inversion`` / ``loco_oracle.encoder_forward`` against it.
Runs only in the build container; the fixture is data (inputs + expected outputs).

Stand-ins (diffusers' UNet2DConditionModel / AutoencoderKL are un-vendored): ``unet`` as in make_golden_tloco_sd.py;
``vae.encode`` = an encoder assembled from the reference's own DDPM modules (models/ddpm/diffusion.py ``ResnetBlock`` with a
zeroed ``temb_proj``, ``AttnBlock``, ``Downsample``, ``Normalize``) in the latent-diffusion ``Encoder`` order followed by the
1x1 ``quant_conv``; its ``latent_dist`` restates diffusers' DiagonalGaussianDistribution (mean | logvar chunks, logvar
clamped to [-30, 20], ``sample() = mean + std * randn``) with the normal draw taken from the seeded global generator.

    python oracle/make_golden_tloco_sd_inv.py
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
import make_golden_tloco_sd as mgs  # noqa: E402


class _Posterior:
    def __init__(self, moments):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        self.noise = None

    def sample(self):
        self.noise = torch.randn(self.mean.shape)
        return self.mean + self.std * self.noise


class RefEncoder(torch.nn.Module):
    """Latent-diffusion ``Encoder`` order + ``quant_conv``, assembled from the reference's DDPM blocks."""

    def __init__(self, cfg):
        super().__init__()
        from models.ddpm.diffusion import ResnetBlock, AttnBlock, Downsample, Normalize
        nn = torch.nn
        ch, mult = cfg.ch, tuple(cfg.ch_mult)
        self.cfg = cfg
        rb = lambda i, o: ResnetBlock(in_channels=i, out_channels=o, dropout=0.0, temb_channels=8)
        self.conv_in = nn.Conv2d(cfg.in_channels, ch, 3, 1, 1)
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(len(mult)):
            d = nn.Module(); d.block = nn.ModuleList()
            for b in range(cfg.num_res_blocks):
                d.block.append(rb(block_in, ch * mult[lvl])); block_in = ch * mult[lvl]
            if lvl != len(mult) - 1:
                d.downsample = Downsample(block_in, True)
            self.down.append(d)
        self.mid = nn.Module()
        self.mid.block_1 = rb(block_in, block_in); self.mid.attn_1 = AttnBlock(block_in); self.mid.block_2 = rb(block_in, block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, cfg.out_ch, 3, 1, 1)
        self.quant_conv = nn.Conv2d(cfg.out_ch, cfg.out_ch, 1)
        self.last = None

    def load(self, params):
        sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
        for name, mod in self.named_modules():
            if name.endswith("temb_proj"):      # no time embedding in the autoencoder: a zero projection adds nothing
                sd[name + ".weight"] = torch.zeros_like(mod.weight); sd[name + ".bias"] = torch.zeros_like(mod.bias)
        self.load_state_dict(sd, strict=True)
        self.eval(); self.requires_grad_(False)
        return self

    def moments(self, x):
        from models.ddpm.diffusion import nonlinearity
        temb = torch.zeros(x.shape[0], 8)
        h = self.conv_in(x)
        for lvl, d in enumerate(self.down):
            for blk in d.block:
                h = blk(h, temb)
            if lvl != len(self.down) - 1:
                h = d.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, temb)), temb)
        return self.quant_conv(self.conv_out(nonlinearity(self.norm_out(h))))

    def encode(self, x):
        self.last = _Posterior(self.moments(x))
        return types.SimpleNamespace(latent_dist=self.last)


def main():
    redit, YHS, extract, PullBackDDPM = mg.import_reference()
    from utils.utils import get_stable_diffusion_scheduler
    import tloco_sd_oracle as tsd
    import loco_oracle as orc
    from loco_edit_amd.config import TINY_LATENT as cfg, TINY_DECODER as dcfg, TINY_ENCODER as ecfg, synth_params
    from loco_edit_amd.tloco import cond_params
    torch.set_num_threads(8)
    tmpdir = tempfile.mkdtemp(prefix="loco_golden_tloco_sd_inv_")
    D, NTOK, INV_STEPS = 16, 7, 12
    params, dparams, eparams = synth_params(cfg, seed=0), synth_params(dcfg, seed=0), synth_params(ecfg, seed=0)
    assert ecfg.out_resolution == cfg.resolution and ecfg.out_ch == 2 * cfg.in_channels
    cp = cond_params(cfg, D, seed=0)
    cw, cb = torch.from_numpy(cp["cond_proj.weight"].copy()), torch.from_numpy(cp["cond_proj.bias"].copy())
    unet = mgt.ref_cond_unet(mgs.ref_latent_unet(cfg, params), cw, cb)
    vae = RefEncoder(ecfg).load(eparams)

    ed = object.__new__(redit.EditStableDiffusion)
    sched = types.SimpleNamespace()
    sched.betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2     # diffusers scaled_linear
    sched.alphas_cumprod = torch.cumprod(1.0 - sched.betas, dim=0)
    sched.scale_model_input = lambda x, t: x
    sargs = Namespace(use_yh_custom_scheduler=True, device=torch.device("cpu"), dtype=torch.float32)
    ed.scheduler = get_stable_diffusion_scheduler(sargs, sched)
    ed.unet, ed.vae = unet, vae
    ed.device, ed.dtype = torch.device("cpu"), torch.float32
    ed.inv_steps, ed.for_steps, ed.use_yh_custom_scheduler = INV_STEPS, 100, True
    ed.guidance_scale = 7.5
    ed.result_folder, ed.dataset_name = tmpdir, "Synthetic"
    ed.for_prompt, ed.inv_prompt = "a photo", "a photo"
    g = torch.Generator().manual_seed(31)
    for_e, edit_e, null_e = (torch.randn(1, NTOK, D, generator=g) for _ in range(3))
    inv_e = torch.randn(1, NTOK, D, generator=g)
    ed.null_prompt_emb, ed.inv_prompt_emb = null_e, inv_e
    gx = torch.Generator().manual_seed(11)
    x0 = torch.randn(1, 3, ecfg.resolution, ecfg.resolution, generator=gx).clamp(-1, 1)
    ed.dataset = [x0]

    po = orc.to_torch(params)
    po.update({k: torch.from_numpy(v.copy()) for k, v in cp.items()})
    ot = tsd.OracleTLocoSD(po, cfg, orc.to_torch(dparams), dcfg, guidance_scale=7.5, guidance_scale_edit=4.0)
    ep = orc.to_torch(eparams)

    out = {"weights_seed": 0, "cond_dim": D, "inv_steps": INV_STEPS, "guidance_scale": 7.5, "x0": x0, "inv_e": inv_e, "null_e": null_e}
    with torch.no_grad():
        mom = vae.moments(x0)
        mg.check("sd-inv/encoder moments", orc.encoder_forward(ep, ecfg, x0), mom)
        out["moments"] = mom
        for key, guidance in (("plain", None), ("cfg", True)):
            torch.manual_seed(123)
            zT = ed.run_DDIMinversion(idx=0, guidance=guidance)
            noise = vae.last.noise
            oz, oz0 = ot.inversion(x0, noise, ep, ecfg, inv_e, null_e, INV_STEPS, guidance=guidance, return_z0=True)
            mg.check(f"sd-inv/{key} z0", oz0, (vae.last.mean + vae.last.std * noise) * 0.18215)
            mg.check(f"sd-inv/{key} zT", oz, zT, rtol=1e-3, atol=1e-4)
            out[key] = {"noise": noise.clone(), "z0": oz0.clone(), "zT": zT.clone()}
        sch = tsd.SDScheduler(); sch.set_inversion_timesteps(INV_STEPS)
        ed.scheduler.set_timesteps(INV_STEPS, device="cpu", is_inversion=True)
        assert torch.equal(sch.timesteps, ed.scheduler.timesteps) and torch.equal(sch.timesteps_next, ed.scheduler.timesteps_next)
        out["timesteps"], out["timesteps_next"] = ed.scheduler.timesteps.clone(), ed.scheduler.timesteps_next.clone()
    torch.save(out, os.path.join(GOLD, "tloco_sd_inv.pt"))
    print("done ->", os.path.join(GOLD, "tloco_sd_inv.pt"), os.path.getsize(os.path.join(GOLD, "tloco_sd_inv.pt")), "bytes")


if __name__ == "__main__":
    main()
