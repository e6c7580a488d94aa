"""CLI entry: ``python -m loco_edit_amd.main <flags>`` -- dispatch of reference
``src/main.py:12-103`` for the unconditional (DDPM) models, the pixel-space DeepFloyd-IF and the latent-space
Stable Diffusion T-LOCO paths.

Multi-GPU: ``torchrun --nproc-per-node N -m loco_edit_amd.main <flags>`` runs one
process per GPU; the Jacobian probes of each subspace solve are sharded over the
ranks (``dist.ProbeSharder``), everything else is replicated, rank 0 writes the files.
"""
from . import dist as ldist
from .define_argparser import parse_args, preset


def main(argv=None):
    args = parse_args(argv)
    # one process per GPU: bind the device and create the process group before anything touches the GPU
    rank, world, device = ldist.init_from_env(args.device)
    args.device = device
    if world > 1:
        # `--seed 0` draws a seed (define_argparser.py:140-141): every rank must use rank 0's draw, or x_T and V0
        # differ across ranks
        import torch
        sh = ldist.ProbeSharder("world")
        if args.seed == 0:
            args.seed = sh.agree(int(torch.randint(1, 2**32, ())))
    args = preset(args)
    if args.is_stable_diffusion:                 # main.py:21-23
        from .tloco_sd import EditStableDiffusion
        print('is stable-diffusion')
        edit = EditStableDiffusion(args)
    elif args.is_DeepFloyd_IF_diffusion:         # main.py:24-26
        from .tloco import EditDeepFloydIF
        print('is DeepFloyd-IF')
        edit = EditDeepFloydIF(args)
    else:
        from .edit import EditUncondDiffusion
        print('is custmized diffusion model')
        edit = EditUncondDiffusion(args)
    out = None
    if args.run_edit_null_space_projection_zt:           # main.py:55-62
        out = edit.run_edit_null_space_projection_zt(
            op='mid', block_idx=0, mask_index=args.mask_index, vis_num=args.vis_num, vis_num_pc=args.pca_rank,
            pca_rank=args.pca_rank, edit_prompt=args.edit_prompt, null_space_projection=args.null_space_projection,
            pca_rank_null=args.pca_rank_null, non_semantic=args.non_semantic)
    if args.run_edit_null_space_projection_zt_semantic:  # main.py:63-69
        out = edit.run_edit_null_space_projection_zt_semantic(
            op='mid', block_idx=0, mask_index=args.mask_index, vis_num=args.vis_num, vis_num_pc=args.pca_rank,
            pca_rank=args.pca_rank, edit_prompt=args.edit_prompt, null_space_projection=args.null_space_projection,
            pca_rank_null=args.pca_rank_null)
    if args.run_edit_null_space_projection_xt:           # main.py:70-76
        out = edit.run_edit_null_space_projection_xt(
            op='mid', block_idx=0, mask_index=args.mask_index, vis_num=args.vis_num, vis_num_pc=args.pca_rank,
            pca_rank=args.pca_rank, edit_prompt=args.edit_prompt, null_space_projection=args.null_space_projection,
            pca_rank_null=args.pca_rank_null)
    if args.run_edit_null_space_projection_xt_semantic:  # main.py:77-84
        out = edit.run_edit_null_space_projection_xt_semantic(
            op='mid', block_idx=0, mask_index=args.mask_index, vis_num=args.vis_num, vis_num_pc=args.pca_rank,
            pca_rank=args.pca_rank, edit_prompt=args.edit_prompt, null_space_projection=args.null_space_projection,
            pca_rank_null=args.pca_rank_null, jacobian=args.jacobian)
    if args.run_edit_null_space_projection:      # main.py:47-52
        out = edit.run_edit_null_space_projection(
            idx=args.sample_idx, op='mid', block_idx=0,
            vis_num=args.vis_num, vis_num_pc=args.pca_rank, pca_rank=args.pca_rank, edit_prompt=args.edit_prompt,
            null_space_projection=args.null_space_projection, pca_rank_null=args.pca_rank_null,
            encoder_decoder_by_et=args.encoder_decoder_by_et, use_mask=args.use_mask, random_edit=args.random_edit)
    if args.group_edit_null_space_projection:    # main.py:87-91
        out = edit.group_edit_null_space_projection(
            idx=args.sample_idx, op='mid', block_idx=0, vis_num_pc=1, pca_rank=1, edit_prompt=args.edit_prompt,
            null_space_projection=args.null_space_projection, pca_rank_null=args.pca_rank_null,
            encoder_decoder_by_et=args.encoder_decoder_by_et)
    if args.run_ddim_forward:                    # main.py:98-99
        edit.run_DDIMforward(num_samples=5)
    if args.run_ddim_inversion:                  # main.py:102-103
        edit.run_DDIMinversion(idx=args.sample_idx)
    if world > 1:
        ldist.shutdown()
    return out


if __name__ == "__main__":
    main()
