"""CLI entry: ``python -m loco_edit_amd.main <flags>`` -- dispatch of reference
``src/main.py:12-103`` for the unconditional (DDPM) models."""
from .define_argparser import parse_args, preset
from .edit import EditUncondDiffusion


def main(argv=None):
    args = preset(parse_args(argv))
    print('is custmized diffusion model')
    edit = EditUncondDiffusion(args)
    out = None
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
    return out


if __name__ == "__main__":
    main()
