"""Measurement aid (by hand, GPU): wall time of the whole edit of one image through the CLI entry points
(reference flow a5 -> a11: inversion 98 steps, 40 steps to t=0.6T, top-5 basis on the mask, null-space basis on
the complement, projection, edit frames, 59-step decode of the frames), per stage."""
import os, sys, time, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
from loco_edit_amd.define_argparser import parse_args, preset
from loco_edit_amd.edit import EditUncondDiffusion

tmp = tempfile.mkdtemp()
argv = ["--model_name", "CelebA_HQ_HF", "--dataset_name", "Synthetic", "--synthetic_weights", "0",
        "--result_folder", tmp, "--run_edit_null_space_projection", "True", "--use_mask", "True",
        "--null_space_projection", "True", "--pca_rank", "5", "--pca_rank_null", "5", "--vis_num", "2",
        "--edit_t", "0.6", "--performance_boosting_t", "0.2", "--for_steps", "100", "--inv_steps", "100",
        "--use_yh_custom_scheduler", "True", "--x_space_guidance_edit_step", "1",
        "--x_space_guidance_scale", "0.5", "--x_space_guidance_num_step", "16", "--seed", "1"]
stages = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); stages[name] = stages.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
for idx in (0, 1):
    args = preset(parse_args(argv + ["--sample_idx", str(idx)]))      # the run directory is per sample (edit.py:2084-2087)
    edit = EditUncondDiffusion(args)
    edit.run_DDIMinversion = timed("inversion (98 steps, B=1)", edit.run_DDIMinversion)
    edit.DDIMforwardsteps = timed("ddim forward (40 steps to t at B=1; 59-step decode of 5 directions x 5 frames)", edit.DDIMforwardsteps)
    edit.local_encoder_decoder_pullback_xt = timed("subspace solves (top-5 on the mask + rank-5 on the complement)", edit.local_encoder_decoder_pullback_xt)
    stages.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    edit.run_edit_null_space_projection(idx=idx, op='mid', block_idx=0, vis_num=args.vis_num, vis_num_pc=args.pca_rank,
                                        pca_rank=args.pca_rank, null_space_projection=True, pca_rank_null=args.pca_rank_null,
                                        use_mask=True)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"image {idx}: {tot:.2f} s total" + ("  (first image: includes first-launch costs)" if idx == 0 else ""))
    for k, v in stages.items():
        print(f"    {k:80s} {v:6.2f} s")
    del edit
