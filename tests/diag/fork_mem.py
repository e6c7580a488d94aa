"""Diagnostic (by hand): device memory of the T-LOCO engine contexts with one shared parameter store (loco_fork, the default) and with
independently loaded contexts (LOCO_CFG_FORK=0): sum of loco_workspace_bytes over the branch engines + decoder, and set-up wall time.
python3 tests/diag/fork_mem.py [sd15|if_i_m]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import contextlib, io
    from argparse import Namespace
    import torch
    import loco_edit_amd  # noqa
    from loco_edit_amd import config as C
    which = sys.argv[2]
    dev = torch.device("cuda:0")
    common = dict(device=dev, dtype=torch.float32, seed=1, synthetic_weights=0, ckpt_path="", precision="bf16x3", dataset_name="Random",
                  for_steps=100, use_yh_custom_scheduler=True, guidance_scale=7.5, prompt_emb=None, prompt_emb_seed=31, cond_dim=64,
                  for_prompt="standin", edit_prompt="standin-edit", sampling_mode=False, tilda_v_score_type="null+(for-null)+(edit-null)",
                  ablation_method="null-space-proj", mask_type="SAM", vT_path="", x_space_guidance_edit_step=1.0,
                  x_space_guidance_num_step=1, result_folder=os.path.join(ROOT, "gpurun_out", "fork_mem"))
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        if which == "sd15":
            from loco_edit_amd.tloco_sd import EditStableDiffusion
            ed = EditStableDiffusion(Namespace(unet_config=C.SD15_UNET, vae_config=C.SD_VAE_DECODER, vae_ckpt_path="", max_batch=5,
                                               guidance_scale_edit=4.0, edit_t=0.7, use_sega=False, x_space_guidance_scale=8.0, **common))
        else:
            from loco_edit_amd.tloco import EditDeepFloydIF
            ed = EditDeepFloydIF(Namespace(unet_config=C.IF_I_M_UNET, max_batch=8, guidance_scale_edit=7.5, edit_t=0.75,
                                           x_space_guidance_scale=10.0, **common))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    engs = list(ed.branches.values()) + ([ed.vae_engine] if hasattr(ed, "vae_engine") else [])
    tot = sum(e.workspace_bytes() for e in engs)
    print(f"{which} LOCO_CFG_FORK={os.environ.get('LOCO_CFG_FORK', '1')}: {len(engs)} contexts, {tot / 2**30:.1f} GiB "
          f"({', '.join(f'{e.workspace_bytes() / 2**30:.1f}' for e in engs)}), set-up {dt:.1f} s, "
          f"torch reports {torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB in use", flush=True)
    sys.exit(0)
for which in (sys.argv[1:] or ["if_i_m", "sd15"]):
    for fork in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which], env=dict(os.environ, LOCO_CFG_FORK=fork),
                           capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:], flush=True)
