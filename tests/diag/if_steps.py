"""Diagnostic: per-step times of the config-5 solve (branch streams on), optionally after a headline-sized engine was used."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd
from argparse import Namespace
from loco_edit_amd.config import IF64_STANDIN, CELEBA_DDPM, synth_params
from loco_edit_amd.tloco import EditDeepFloydIF
from loco_edit_amd.hip import LocoEngine
pre = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0")
keep = []
from loco_edit_amd.config import TINY_ADM
if pre != "none":
    pc = {"solve_tiny": TINY_ADM, "solve_if": IF64_STANDIN}.get(pre, CELEBA_DDPM)
    eng = LocoEngine(pc, max_batch=8 if pre != "solve32" else 32, device=dev); eng.load_state_dict(synth_params(pc, 0)); keep.append(eng)
    if pre.startswith("solve") or pre in ("primal", "fwd", "jvp"):
        x = torch.randn(1, pc.in_channels, pc.resolution, pc.resolution, device=dev)
        V = torch.randn(5, eng.n, device=dev)
        ctxm = torch.cuda.stream(torch.cuda.Stream(device=dev)) if pre == "solve_side" else __import__("contextlib").nullcontext()
        with ctxm:
            if pre == "fwd":
                eng.unet_forward(x, 600.0)
            else:
                eng.pmp_primal(x, 600.0, 0.5, None, use_et=True)
                if pre != "primal":
                    for _ in range(2):
                        U = eng.pmp_jvp(V)
                        if pre != "jvp": eng.pmp_vjp(U)
        torch.cuda.synchronize()
cfg = IF64_STANDIN
args = Namespace(device=dev, dtype=torch.float32, seed=1, unet_config=cfg, synthetic_weights=0, ckpt_path="", max_batch=8,
                 precision="bf16x3", dataset_name="Random", for_steps=100, use_yh_custom_scheduler=True, guidance_scale=7.5,
                 guidance_scale_edit=7.5, prompt_emb=None, prompt_emb_seed=31, cond_dim=64, for_prompt="a", edit_prompt="b", edit_t=0.75,
                 sampling_mode=False, tilda_v_score_type="null+(for-null)+(edit-null)", ablation_method="null-space-proj", mask_type="SAM",
                 vT_path="", x_space_guidance_edit_step=1.0, x_space_guidance_scale=10.0, x_space_guidance_num_step=1, result_folder="/tmp/ifs")
ed = EditDeepFloydIF(args)
g = torch.Generator().manual_seed(1)
x = torch.randn(1, 3, 64, 64, generator=g).to(dev)
mask = torch.zeros(3, 64, 64, dtype=torch.bool); mask[:, 27:32, 17:27] = True
v0 = torch.randn(cfg.n, 5, generator=g).to(dev)
tt = ed.scheduler.timesteps[ed.edit_t_idx]
F, E, N = ed.for_prompt_emb, ed.edit_prompt_emb, ed.null_prompt_emb
ts = []
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ed.local_encoder_decoder_pullback_xt(x, tt, ed.edit_t_idx, F, E, N, pca_rank=5, min_iter=12, max_iter=12, mask=(~mask).to(dev),
                                         mode="null+(for-null)", v0=v0, verbose=False)
    torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print(pre, "steps ms:", ts, "free GB", round(torch.cuda.mem_get_info()[0] / 1e9, 1))
