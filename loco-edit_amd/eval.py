"""Edit-quality metrics next to the throughput numbers.

The reference's ``src/eval.py`` states the intent -- SSIM, LPIPS and mask-restricted MSE between edited
and original PNGs, paired by file name (``eval.py:24-42,56-86``) -- but does not run as shipped (it calls
undefined names).  This module is a working restatement of that intent:

* ``ssim``: the Wang et al. index with the defaults of ``torchmetrics.image.StructuralSimilarityIndexMeasure``
  that ``eval.py:26-29`` instantiates (11x11 Gaussian window, sigma 1.5, k1 = 0.01, k2 = 0.03, reflect padding
  cropped from the map, data range = larger of the two dynamic ranges, mean over pixels and images);
* ``masked_mse``: mean squared error over the masked elements only (``eval.py:39-42``; inside the mask it
  measures the edit, with ``~mask`` it measures how well the null-space projection protected the rest);
* ``lpips``: the Zhang et al. perceptual distance with the AlexNet backbone that ``torchmetrics``' default
  (``eval.py:33-36``) uses, restated from the published architecture; the pretrained weights (torchvision AlexNet
  ``features.*`` + the five ``lin*`` heads of the ``lpips`` package) are not available offline, so the weights come
  from a file (``LOCO_LPIPS_WEIGHTS`` / ``--lpips_weights``) and without one it raises instead of returning a
  made-up number.

Plain torch on the host: evaluation is not on the hot path and there is no ``torchmetrics`` here to pin the
SSIM against (parity unpinned; the unit tests check the defining properties).
"""
from __future__ import annotations

import argparse
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F


def _gaussian_window(size: int, sigma: float, dtype) -> torch.Tensor:
    x = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(x / sigma) ** 2 / 2)
    g = g / g.sum()
    return g[:, None] * g[None, :]


def ssim(preds: torch.Tensor, target: torch.Tensor, data_range=None, kernel_size: int = 11, sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """Mean SSIM of a batch ``[B,C,H,W]`` (closer to 1 = more similar)."""
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError("expected preds and target of the same shape [B,C,H,W]")
    preds, target = preds.double(), target.double()
    if data_range is None:
        data_range = max(float(preds.max() - preds.min()), float(target.max() - target.min()))
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    ch = preds.shape[1]
    pad = (kernel_size - 1) // 2
    win = _gaussian_window(kernel_size, sigma, preds.dtype).expand(ch, 1, kernel_size, kernel_size)
    p = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t])
    out = F.conv2d(stack, win, groups=ch)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(preds.shape[0])
    s_pp = e_pp - mu_p * mu_p
    s_tt = e_tt - mu_t * mu_t
    s_pt = e_pt - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    full = full[..., pad:-pad, pad:-pad] if full.shape[-1] > 2 * pad and full.shape[-2] > 2 * pad else full
    return full.reshape(full.shape[0], -1).mean(dim=1).mean()


def masked_mse(preds: torch.Tensor, target: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """MSE over ``mask`` only; ``mask`` is boolean and broadcastable to the images."""
    mask = mask.to(torch.bool).expand_as(preds)
    if not bool(mask.any()):
        raise ValueError("empty mask")
    d = (preds.double() - target.double())[mask]
    return (d * d).mean()


# AlexNet feature stack of torchvision (conv index in `features`, out channels, kernel, stride, padding, max-pool before)
_ALEX = ((0, 64, 11, 4, 2, False), (3, 192, 5, 1, 2, True), (6, 384, 3, 1, 1, True), (8, 256, 3, 1, 1, False),
         (10, 256, 3, 1, 1, False))
_LPIPS_SHIFT = (-0.030, -0.088, -0.188)
_LPIPS_SCALE = (0.458, 0.448, 0.450)


def lpips_weight_names():
    """Keys a weights file must hold: torchvision AlexNet convolutions and the LPIPS 1x1 heads (``lin{i}.model.1.weight``,
    the names of the ``lpips`` package's ``alex.pth``; ``net.features.*`` / ``net.slice*`` prefixes are accepted)."""
    return [f"features.{i}.{p}" for i, *_ in _ALEX for p in ("weight", "bias")] + \
           [f"lin{j}.model.1.weight" for j in range(5)]


def _lpips_weights(weights):
    if weights is None:
        weights = os.environ.get("LOCO_LPIPS_WEIGHTS", "")
    if isinstance(weights, str):
        if not weights:
            raise NotImplementedError(
                "LPIPS needs the pretrained AlexNet + `lpips` head weights, which are not available offline: "
                "pass weights= / --lpips_weights / LOCO_LPIPS_WEIGHTS (a state dict with the keys of lpips_weight_names())")
        weights = torch.load(weights, map_location="cpu")
    w = {}
    for k, v in weights.items():
        k = k[len("net."):] if k.startswith("net.") else k
        w[k] = v
    missing = [k for k in lpips_weight_names() if k not in w]
    if missing:
        raise ValueError("LPIPS weights file lacks " + ", ".join(missing[:4]))
    return w


def lpips(preds: torch.Tensor, target: torch.Tensor, weights=None, normalize: bool = False) -> torch.Tensor:
    """Mean LPIPS (AlexNet) of a batch ``[B,3,H,W]`` in [-1, 1] (``normalize=True``: in [0, 1]); lower = more similar.
    Per tap: unit-normalise the features over channels, squared difference, non-negative 1x1 head, spatial mean;
    sum over the five ReLU taps."""
    if preds.shape != target.shape or preds.dim() != 4 or preds.shape[1] != 3:
        raise ValueError("expected preds and target of the same shape [B,3,H,W]")
    w = _lpips_weights(weights)
    x = torch.cat([preds, target]).float()
    if normalize:
        x = 2 * x - 1
    x = (x - torch.tensor(_LPIPS_SHIFT).view(1, 3, 1, 1)) / torch.tensor(_LPIPS_SCALE).view(1, 3, 1, 1)
    total = 0.0
    for j, (i, _, _, stride, pad, pool) in enumerate(_ALEX):
        if pool:
            x = F.max_pool2d(x, kernel_size=3, stride=2)
        x = F.relu(F.conv2d(x, w[f"features.{i}.weight"].float(), w[f"features.{i}.bias"].float(), stride=stride,
                            padding=pad))
        f = x / (x.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
        fp, ft = f.split(preds.shape[0])
        d = F.conv2d((fp - ft) ** 2, w[f"lin{j}.model.1.weight"].float())
        total = total + d.mean(dim=(1, 2, 3))
    return total.mean()


METRICS = {"ssim": ssim, "mmse": masked_mse, "lpips": lpips}


def _load_png(path: str) -> torch.Tensor:
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    return torch.from_numpy(a).permute(2, 0, 1).unsqueeze(0)


def evaluate_folders(folder_preds: str, folder_original: str, metric: str = "ssim", mask_folder: str = "",
                     outside_mask: bool = False, lpips_weights=None) -> dict:
    """Pair ``*.png`` by file name (``eval.py:56-72``) and average the metric; for ``mmse`` the mask of image
    ``<stem>.png`` is ``<mask_folder>/<stem>.pt`` (a bool tensor ``[3,H,W]`` or ``[H,W]``)."""
    if metric not in METRICS:
        raise ValueError("eval_metric choice: " + ", ".join(METRICS))
    pp = sorted(glob.glob(os.path.join(folder_preds, "*.png")))
    tp = sorted(glob.glob(os.path.join(folder_original, "*.png")))
    if len(pp) != len(tp) or not pp:
        raise ValueError(f"{len(pp)} predictions vs {len(tp)} originals")
    vals = []
    for a, b in zip(pp, tp):
        if os.path.basename(a) != os.path.basename(b):
            raise ValueError("pairs not match")
        x, y = _load_png(a), _load_png(b)
        if metric == "mmse":
            m = torch.load(os.path.join(mask_folder or os.path.join(folder_preds, "mask"),
                                        os.path.splitext(os.path.basename(a))[0] + ".pt"))
            m = m if m.dim() == 3 else m[None].repeat(3, 1, 1)
            vals.append(float(masked_mse(x, y, (~m if outside_mask else m)[None])))
        else:
            vals.append(float(lpips(x / 127.5 - 1, y / 127.5 - 1, weights=lpips_weights) if metric == "lpips"
                              else METRICS[metric](x, y)))
    return {"metric": metric, "n": len(vals), "mean": float(np.mean(vals)), "values": vals}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--eval_metric", type=str, default="ssim")          # eval.py:17
    ap.add_argument("--folder_preds", type=str, required=True)
    ap.add_argument("--folder_original", type=str, required=True)
    ap.add_argument("--mask_folder", type=str, default="")
    ap.add_argument("--outside_mask", action="store_true", help="mmse over ~mask (the protected region)")
    ap.add_argument("--lpips_weights", type=str, default="", help="state dict: AlexNet features + lpips heads")
    a = ap.parse_args(argv)
    r = evaluate_folders(a.folder_preds, a.folder_original, a.eval_metric, a.mask_folder, a.outside_mask,
                         a.lpips_weights or None)
    print(f"{r['metric']}: {r['mean']:.6f} over {r['n']} pairs")
    return r


if __name__ == "__main__":
    main()
