"""Factories around the hot path: denoiser, scheduler, datasets, image writer.

Mirrors the call surface of reference ``src/utils/utils.py:52-134,472-672`` for
the unconditional models.  There is no network in this deployment, so weights
come from ``--ckpt_path`` (a state_dict with the vendored Ho-DDPM key layout,
``diffusion.py`` module tree) or from the deterministic synthesiser
(``--synthetic_weights SEED``); the hub / cluster loaders of the reference are
out of scope.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .config import CELEBA_DDPM, FFHQ_P2, UNetConfig, synth_params
from .hip import LocoEngine
from .scheduler import YHCustomScheduler

# architecture per --model_name (reference utils.py:83-100 maps names to checkpoints of these shapes)
MODEL_CONFIGS = {
    "CelebA_HQ_HF": CELEBA_DDPM,     # google/ddpm-ema-celebahq-256
    "CelebA_HQ": CELEBA_DDPM,        # SDEdit celeba_hq.ckpt (same module tree)
    "LSUN_church_HF": CELEBA_DDPM,   # google/ddpm-ema-church-256 (same architecture)
    "LSUN_bedroom_HF": CELEBA_DDPM,
    # guided-diffusion / P2 checkpoints (utils.py:112-115 -> g_DDPM with P2_DICT)
    "FFHQ_P2": FFHQ_P2, "AFHQ_P2": FFHQ_P2, "Flower_P2": FFHQ_P2, "Cub_P2": FFHQ_P2, "Metface_P2": FFHQ_P2,
}


class HipUNet:
    """``unet(x, t) -> eps`` duck type (seam 3 of SURVEY.md 8b) on the HIP engine."""

    def __init__(self, engine: LocoEngine):
        self.engine = engine
        # the engine returns the eps half only (guided_diffusion unet.py:680-684), so the scheduler never sees logvar
        self.learn_sigma = False

    def __call__(self, x: torch.Tensor, t) -> torch.Tensor:
        return self.engine.unet_forward(x.contiguous(), float(t))

    def to(self, *a, **k):
        return self


def get_custom_diffusion_model(args) -> HipUNet:
    """reference utils.py:77-133."""
    cfg: Optional[UNetConfig] = getattr(args, "unet_config", None) or MODEL_CONFIGS.get(args.model_name)
    if cfg is None:
        raise ValueError('model_name choice: ' + ", ".join(MODEL_CONFIGS))
    engine = LocoEngine(cfg, max_batch=getattr(args, "max_batch", 8), device=args.device)
    prec = getattr(args, "precision", None) or os.environ.get("LOCO_PRECISION")
    if prec:
        engine.set_precision(prec)
    # Two HIP streams (the default since round 6; LOCO_STREAMS=1: one): the two probe groups of a tangent / cotangent pass side by
    # side (statistics / apply / reduce kernels of one group beside the convolutions of the other: -3.5 ... -4.7 % per 256 x 256
    # solve on every box measured, results equal to rounding); the side stream is chosen by measurement, one stream when none runs
    # beside the current one.  Per-kernel durations are then durations under overlap: bench.py takes its per-kernel profile on one.
    if os.environ.get("LOCO_STREAMS", "2") == "2" and torch.device(args.device).type == "cuda":
        engine.set_streams_measured(2)
    ckpt = getattr(args, "ckpt_path", "")
    if ckpt:
        sd = torch.load(ckpt, map_location="cpu")
        if "state_dict" in sd:
            sd = sd["state_dict"]
        from .checkpoints import hf_unet2d_to_vendored, is_hf_unet2d
        if is_hf_unet2d(sd):          # diffusers UNet2DModel naming (google/ddpm-*-256): map to the vendored keys
            sd = hf_unet2d_to_vendored(sd, cfg)
        engine.load_state_dict(sd)
    else:
        seed = getattr(args, "synthetic_weights", None)
        if seed is None:
            raise ValueError("no checkpoint: pass --ckpt_path (vendored-DDPM state_dict) or --synthetic_weights SEED")
        engine.load_state_dict(synth_params(cfg, seed=int(seed)))
    return HipUNet(engine)


def get_custom_diffusion_scheduler(args, engine=None):
    """reference utils.py:52-75 (custom scheduler is mandatory for uncond models,
    define_argparser.py:245)."""
    if not args.use_yh_custom_scheduler:
        raise ValueError('please set use_yh_custom_scheduler = True')
    return YHCustomScheduler(args, engine=engine)


# ---------------------------------------------------------------------------
# image writer (replaces torchvision.utils.save_image at edit.py:2139,2163,2596)
def save_image(img: torch.Tensor, path: str, nrow: int = 8, padding: int = 2):
    from PIL import Image
    img = img.detach().float().cpu().clamp(0, 1)
    if img.dim() == 3:
        img = img[None]
    b, c, h, w = img.shape
    ncol = min(nrow, b)
    nr = (b + ncol - 1) // ncol
    grid = torch.zeros(c, nr * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(b):
        r, cc = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + cc * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = img[i]
    arr = (grid * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    if arr.shape[2] == 1:
        arr = arr[:, :, 0]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr).save(path)


# ---------------------------------------------------------------------------
# datasets
class SyntheticDataset:
    """Seeded stand-in for CelebAMask-HQ when no dataset is mounted: image =
    clamp(randn) in [-1,1], mask = an 'l_eye'-sized rectangle."""

    def __init__(self, image_size=256, c_in=3):
        self.res, self.c = image_size, c_in

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(int(idx))
        return torch.randn(1, self.c, self.res, self.res, generator=g).clamp(-1, 1)

    def getmask(self, idx, choose_sem, list_sem=True):
        m = torch.zeros(self.c, self.res, self.res, dtype=torch.bool)
        r = self.res
        m[:, int(r * 110 / 256):int(r * 130 / 256), int(r * 70 / 256):int(r * 110 / 256)] = True
        return m


class CelebAMaskDataset:
    """CelebAMask-HQ image + ground-truth part masks (reference
    ``src/dataset/celeba_hq_dataloader.py:9-123``): images ``CelebA-HQ-img/{idx}.jpg``,
    masks ``CelebAMask-HQ-mask-anno/{k}/{idx:05d}_{sem}.png``; both resized to
    ``res`` with PIL's default resampling; mask = ``astype(bool)`` -> [3,res,res]."""

    def __init__(self, root, res=256):
        self.root, self.res = root, res
        self.img_dir = os.path.join(root, "CelebA-HQ-img")
        self.mask_dir = os.path.join(root, "CelebAMask-HQ-mask-anno")
        if not os.path.isdir(self.img_dir):
            raise FileNotFoundError(f"CelebAMask-HQ not found under {root}")

    def _img(self, idx):
        from PIL import Image
        return Image.open(os.path.join(self.img_dir, f"{idx}.jpg")).resize((self.res, self.res))

    def __getitem__(self, idx):
        a = np.asarray(self._img(idx).convert("RGB"), dtype=np.float32) / 255.0
        t = torch.from_numpy(a).permute(2, 0, 1)
        return ((t - 0.5) / 0.5).unsqueeze(0)

    def getmask(self, idx, choose_sem, list_sem=True):
        from PIL import Image
        sub = os.path.join(self.mask_dir, str(idx // 2000))
        path = os.path.join(sub, f"{idx:05d}_{choose_sem}.png")
        if not os.path.exists(path):
            raise AssertionError(f"For the {idx}th image, semantic {choose_sem} has no annotation")
        m = np.array(Image.open(path).resize((self.res, self.res)))
        if m.ndim == 2:
            m = np.repeat(m[:, :, None], 3, axis=2)
        return torch.tensor(m.astype(bool)).permute(2, 0, 1)


class FolderDataset:
    """Image folder datasets (reference ``ImgDataset`` / ``AFHQDataset``, utils.py:588-672):
    files sorted by their integer stem (``numeric=True``, FFHQ) or lexicographically (AFHQ);
    centre-crop to a square, PIL default resize to ``res``, ToTensor, Normalize(0.5, 0.5) -> [1,3,res,res]."""

    def __init__(self, root, res=256, numeric=True):
        self.root, self.res = root, res
        names = [n for n in os.listdir(root) if "." in n and n.split(".")[1] in ("jpg", "jpeg", "png")]
        self.paths = sorted(names, key=(lambda n: int(n.split(".")[0])) if numeric else (lambda n: n.split(".")[0]))

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        from PIL import Image
        x = Image.open(os.path.join(self.root, self.paths[idx]))
        w, h = x.size
        c = min(w, h)
        x = x.crop(((w - c) / 2, (h - c) / 2, (w + c) / 2, (h + c) / 2)).resize((self.res, self.res))
        a = np.asarray(x.convert("RGB"), dtype=np.float32) / 255.0
        return ((torch.from_numpy(a).permute(2, 0, 1) - 0.5) / 0.5).unsqueeze(0)


def get_dataset(args):
    """reference utils.py:472-560 (the datasets the unconditional path uses)."""
    if args.dataset_name == "CelebA_HQ_mask":
        return CelebAMaskDataset(args.dataset_root, res=args.image_size)
    if args.dataset_name in ("FFHQ", "CelebA_HQ"):
        return FolderDataset(args.dataset_root, res=args.image_size, numeric=True)
    if args.dataset_name == "AFHQ":
        return FolderDataset(args.dataset_root, res=args.image_size, numeric=False)
    if args.dataset_name == "Synthetic":
        return SyntheticDataset(args.image_size, args.c_in)
    if args.dataset_name == "Random":
        return None
    raise ValueError('Invalid dataset name (supported here: CelebA_HQ_mask, FFHQ, AFHQ, CelebA_HQ, Synthetic, Random; '
                     'the HF-hub datasets need network access)')
