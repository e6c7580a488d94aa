"""``EditUncondDiffusion``: the LOCO-Edit pipeline for unconditional DDPMs on the
MI355X engine, with the method names, argument meaning, file layout and
``--vT_path`` format of the reference class (``src/modules/edit.py:2034-2625``).

What changed underneath: the denoiser, its Jacobian products, the scheduler
update and the subspace algebra are HIP kernels; batches stay resident on the
GPU (no ``buffer_device='cpu'`` bounce, no ``empty_cache`` per step); the solver
is ``solver.local_basis``.  Defects of the reference that cannot run
(``vis_power_spectral_density`` undefined, edit.py:2603) are not reproduced.
"""
from __future__ import annotations

import os

import torch

from . import solver
from .dist import ProbeSharder
from .utils import get_custom_diffusion_model, get_custom_diffusion_scheduler, get_dataset
from .utils import save_image as _save_image


class EditUncondDiffusion(object):
    def __init__(self, args):
        # default setting (edit.py:2037-2043)
        self.buffer_device = getattr(args, "buffer_device", "cpu")   # accepted, unused: batches stay in HBM
        self.memory_bound = getattr(args, "memory_bound", 50)
        self.device = args.device
        self.dtype = args.dtype
        self.seed = args.seed
        if self.dtype != torch.float32:
            raise ValueError("the unconditional hot path keeps fp32 tensors (scripts/main_celeba_hf_null_space_projection.sh:7); "
                             "the conv arithmetic is chosen with --precision")

        # get model (edit.py:2046-2052)
        self.unet = get_custom_diffusion_model(args)
        self.engine = self.unet.engine
        print(f'engine : {self.engine.version()}, conv arithmetic : {self.engine.get_precision()}')
        self.scheduler = get_custom_diffusion_scheduler(args, engine=self.engine)
        self.model_name = args.model_name

        self.image_size = args.image_size
        self.c_in = 3

        self.dataset = get_dataset(args)
        self.dataset_name = args.dataset_name

        self.for_steps = args.for_steps
        self.inv_steps = args.inv_steps
        self.use_yh_custom_scheduler = args.use_yh_custom_scheduler

        self.edit_t = args.edit_t
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        # edit.py:2071-2073
        self.edit_t_idx = int((self.scheduler.timesteps - self.edit_t * 1000).abs().argmin())
        self.performance_boosting_t_idx = (
            int((self.scheduler.timesteps - args.performance_boosting_t * 1000).abs().argmin())
            if args.performance_boosting_t > 0 else 1000)
        print(f'performance_boosting_t_idx: {self.performance_boosting_t_idx}')

        self.use_x_space_guidance = getattr(args, "use_x_space_guidance", False)
        self.x_space_guidance_edit_step = args.x_space_guidance_edit_step
        self.x_space_guidance_scale = args.x_space_guidance_scale
        self.x_space_guidance_num_step = args.x_space_guidance_num_step

        # path (edit.py:2084-2087)
        if self.dataset_name == "Random":
            self.result_folder = os.path.join(args.result_folder, f"sample_seed{args.seed}")
        else:
            self.result_folder = os.path.join(args.result_folder, f"sample_idx{args.sample_idx}")
        os.makedirs(self.result_folder, exist_ok=True)
        self.obs_folder = getattr(args, "obs_folder", None)
        self.vT_path = args.vT_path
        self.vT1_path = args.vT1_path
        self.sharder = ProbeSharder("world")
        # the modify-space and null-space solves of one edit share their probe batches (solver.local_basis_pair);
        # LOCO_PAIR_SOLVES=0 runs them one after the other as the reference does
        self.pair_solves = os.environ.get("LOCO_PAIR_SOLVES", "1") != "0"
        # the edited frames of all directions are decoded as one batch (LOCO_BATCH_DECODE=0: one direction after the other)
        self.batch_decode = os.environ.get("LOCO_BATCH_DECODE", "1") != "0"
        self.EXP_NAME = "exp"
        self.args = args

    # ------------------------------------------------------------------ multi-rank file discipline
    # Under torchrun every rank runs the same flow; rank 0 alone writes, and branch decisions that depend on the
    # file system are rank 0's (ranks taking different branches would hang in the solver's all-gather).
    def _save_image(self, *a, **k):
        if self.sharder.is_main:
            _save_image(*a, **k)

    def _save(self, obj, path):
        if self.sharder.is_main:
            torch.save(obj, path)

    def _exists(self, path):
        return self.sharder.agree(bool(path) and os.path.exists(path))

    def _load(self, path, **kw):
        """torch.load on rank 0, broadcast to the others (they may not see rank 0's freshly written file yet)."""
        if not self.sharder.active:
            return torch.load(path, **kw)
        obj = torch.load(path, **kw) if self.sharder.is_main else None
        return self.sharder.agree(obj if obj is None else obj.cpu())

    # ------------------------------------------------------------------ helpers
    def _step(self, xt, t, eta, noise=None):
        """unet + scheduler.step fused on the device (edit.py:2151-2160 / 2572-2581)."""
        idx = self.scheduler.index_of(t)
        t_next = self.scheduler.timesteps_next[idx]
        at, at_next = self.scheduler.alpha_at(t), self.scheduler.alpha_at(t_next)
        if eta != 0 and noise is None:
            noise = torch.randn_like(xt)
        mb = self.engine.max_batch
        if xt.shape[0] <= mb:
            return self.engine.ddim_step(xt.contiguous(), float(t), at, at_next, eta, noise)
        out = torch.empty_like(xt)
        for b0 in range(0, xt.shape[0], mb):
            sl = slice(b0, b0 + mb)
            out[sl] = self.engine.ddim_step(xt[sl].contiguous(), float(t), at, at_next, eta,
                                            None if noise is None else noise[sl].contiguous())
        return out

    # ------------------------------------------------------------------ loops
    @torch.no_grad()
    def run_DDIMforward(self, num_samples=5):
        print('start DDIMforward')
        self.EXP_NAME = 'DDIMforward'
        xT = torch.randn(num_samples, self.c_in, self.image_size, self.image_size, device=self.device, dtype=self.dtype)
        self.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=-1, vis_psd=False)

    @torch.no_grad()
    def run_DDIMinversion(self, idx, x0=None):
        """edit.py:2117-2167: x0 -> xT by DDIM with ascending t; 98 of the 99 steps run."""
        print('start DDIMinversion')
        EXP_NAME = f'DDIMinversion-{self.dataset_name}_{idx}'
        if not self.use_yh_custom_scheduler:
            raise ValueError('please set use_yh_custom_scheduler = True')
        self.scheduler.set_timesteps(self.inv_steps, device=self.device, is_inversion=True)
        timesteps = self.scheduler.timesteps
        if x0 is None:
            x0 = self.dataset[idx]
        self._save_image((x0 / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, 'original.png'))
        xt = x0.to(self.device, dtype=self.dtype).contiguous()
        for i, t in enumerate(timesteps):
            if i == len(timesteps) - 1:
                break
            xt = self._step(xt, t, eta=0)
        self._save_image((xt / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, f'xT-{EXP_NAME}.png'))
        return xt

    @torch.no_grad()
    def DDIMforwardsteps(self, xt, t_start_idx, t_end_idx, vis_psd=False, save_image=True, return_xt=True,
                         performance_boosting=False, noises=None):
        """edit.py:2508-2614.  ``noises`` (optional, {step index: tensor}) injects the
        eta=1 draws so decodes are reproducible in tests."""
        print('start DDIMforward')
        assert (t_start_idx < self.for_steps) & (t_end_idx <= self.for_steps)
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        timesteps = self.scheduler.timesteps
        xt = xt.to(device=self.device, dtype=self.dtype).contiguous()
        for i, t in enumerate(timesteps):
            if t_end_idx == i:
                print('t_end_idx : ', i)
                return xt, t, i
            elif i < t_start_idx:
                continue
            elif t_start_idx == i:
                print('t_start_idx : ', i)
            if performance_boosting & (self.performance_boosting_t_idx <= i) & \
                    (self.performance_boosting_t_idx != len(timesteps) - 1):
                eta = 1
            else:
                eta = 0
            nz = None if (noises is None or eta == 0) else noises[i].to(self.device)
            xt = self._step(xt, t, eta=eta, noise=nz)
        if save_image:
            image = (xt / 2 + 0.5).clamp(0, 1)
            self._save_image(image, os.path.join(self.result_folder, f'{self.EXP_NAME}.png'), nrow=image.size(0))
        if return_xt:
            return xt
        return

    # ------------------------------------------------------------------ x0 / eps
    def get_x0(self, t, x, mask=None):
        """edit.py:2369-2391."""
        et = self.unet(x, t)
        at = self.scheduler.alpha_at(t)
        _, P_xt = self.engine.sched_step(x.contiguous(), et, at, at, 0.0, None, want_x0=True)
        if mask is not None:
            P_xt = P_xt[:, mask.to(P_xt.device)]
        return P_xt

    def get_et(self, t, x, mask=None):
        """edit.py:2394-2403."""
        et = self.unet(x, t)
        if mask is not None:
            et = et[:, mask.to(et.device)]
        return et

    # ------------------------------------------------------------------ solver
    def local_encoder_decoder_pullback_xt(self, x, t, op=None, block_idx=None, pca_rank=50, chunk_size=25,
                                          min_iter=10, max_iter=100, convergence_threshold=1e-3, mask=None,
                                          noise=False, v0=None, verbose=True):
        """edit.py:2406-2504 -> (u [L,k], s [k], vT [k,n]).  ``op``/``block_idx``/``chunk_size``
        are accepted and ignored exactly as in the reference."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        at = self.scheduler.alpha_at(t)
        u, s, vT, self.last_n_iter = solver.local_basis(
            self.engine, x, t, at, pca_rank, mask=mask, noise=noise, min_iter=min_iter, max_iter=max_iter,
            convergence_threshold=convergence_threshold, v0=v0, sharder=self.sharder, verbose=verbose)
        return u, s, vT

    # ------------------------------------------------------------------ drivers
    @torch.no_grad()
    def group_edit_null_space_projection(self, idx, **kwargs):
        """edit.py:2171-2212: compose two saved directions."""
        if self.dataset_name == 'Random':
            xT = torch.randn(1, 3, self.image_size, self.image_size, dtype=self.dtype, device=self.device)
        else:
            xT = self.run_DDIMinversion(idx=idx)
        xt, t, t_idx = self.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=self.edit_t_idx)
        assert t_idx == self.edit_t_idx
        print('loading the basis from --vT_path')
        vT_list = [self._load(self.vT_path), self._load(self.vT1_path)]
        BASIS_NAME = "load-basis-2"
        xt_temp = xt.detach().clone()
        xt_vis_list = [xt_temp]
        n = xt[0].numel()
        for pc_idx in range(2):
            vk = vT_list[pc_idx][0, :].to(self.device, torch.float32).contiguous()
            alpha = self.x_space_guidance_scale * self.x_space_guidance_num_step
            xt_edit = self.engine.edit_axpy(xt_temp.contiguous(), vk, [alpha])
            xt_temp = xt_edit
            xt_vis_list.append(xt_edit)
        self.EXP_NAME = f'{idx}-Edit_xt-noise-{BASIS_NAME}'
        xt_vis = torch.cat(xt_vis_list, dim=0)
        self.DDIMforwardsteps(xt_vis, t_start_idx=self.edit_t_idx, t_end_idx=-1, performance_boosting=True)
        return xt

    def _get_xT_and_mask(self, idx, use_mask):
        """edit.py:2234-2267."""
        if self.dataset_name == 'Random':
            xT = torch.randn(1, self.c_in, self.image_size, self.image_size, dtype=self.dtype, device=self.device)
            mpath = os.path.join(self.result_folder, "mask/mask.pt")
            if not os.path.exists(mpath):
                raise FileNotFoundError(
                    f"{mpath} missing: SAM mask generation is outside the hot path (SURVEY.md 2.1 #9); "
                    "provide mask.pt (bool [N,res,res])")
            masks = torch.load(mpath)
            if self.args.sampling_mode:
                return None, None
            mask = masks[self.args.mask_index].squeeze(dim=0).repeat(3, 1, 1)
            return xT, mask
        if self.dataset_name in ("CelebA_HQ_mask", "Synthetic"):
            xT = self.run_DDIMinversion(idx=idx)
            mask = self.dataset.getmask(idx=self.args.sample_idx, choose_sem=self.args.choose_sem)
            return xT, (mask if use_mask or self.dataset_name == "CelebA_HQ_mask" else None)
        # FFHQ / AFHQ / ... : SAM masks cached as mask/mask.pt, bool [N,res,res] (edit.py:2252-2267)
        mpath = os.path.join(self.result_folder, "mask/mask.pt")
        if not os.path.exists(mpath):
            raise FileNotFoundError(
                f"{mpath} missing: SAM mask generation is outside the hot path (SURVEY.md 2.1 #9); "
                "provide mask.pt (bool [N,res,res]) as the reference's mask_segmentation.py writes it")
        print("loading masks")
        masks = torch.load(mpath)
        if self.args.sampling_mode:
            return None, None
        xT = self.run_DDIMinversion(idx=idx)
        mask = masks[self.args.mask_index].squeeze(dim=0).repeat(3, 1, 1) if use_mask else None
        return xT, mask

    @torch.no_grad()
    def run_edit_null_space_projection(self, idx, vis_num, vis_num_pc=5, pca_rank=50, pca_rank_null=10, op='mid',
                                       block_idx=0, null_space_projection=True, encoder_decoder_by_et=False,
                                       use_mask=True, random_edit=False, **kwargs):
        """edit.py:2216-2366."""
        xT, mask = self._get_xT_and_mask(idx, use_mask)
        if xT is None:
            return None
        # xT -> xt
        xt, t, t_idx = self.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=self.edit_t_idx)
        assert t_idx == self.edit_t_idx

        if not self._exists(self.vT_path):
            print('computing the local basis')
            tag = self.args.choose_sem if self.dataset_name in ("CelebA_HQ_mask", "Synthetic") else self.args.mask_index
            save_dir = os.path.join(self.result_folder, "basis", f'local_basis-{self.edit_t}T-select-mask-{tag}')
            os.makedirs(save_dir, exist_ok=True)
            vT_modify_path = os.path.join(save_dir, f'vT-modify-pca-rank-{pca_rank}.pt')
            vT_null_path = os.path.join(save_dir, f'vT-null-{pca_rank_null}.pt')

            vT_null = None
            have_m, have_n = self._exists(vT_modify_path), self._exists(vT_null_path)
            if null_space_projection and mask is not None and not have_m and not have_n and self.pair_solves:
                # both bases are missing: one pass carries the probes of both solves (solver.local_basis_pair)
                print('subspace solves: modify space + null space (shared probe batches)')
                at = self.scheduler.alpha_at(t)
                (u_modify, s_modify, vT_modify, self.last_n_iter), (u_null, s_null, vT_null, self.last_n_iter_null) = \
                    solver.local_basis_pair(self.engine, xt.to(self.device, torch.float32), float(t), at, pca_rank, mask,
                                            pca_rank_null, ~mask, noise=encoder_decoder_by_et, min_iter=10, max_iter=50,
                                            convergence_threshold=1e-4, sharder=self.sharder)
                self._save(vT_modify, vT_modify_path)
                self._save(vT_null, vT_null_path)
            elif have_m:
                vT_modify = self._load(vT_modify_path, map_location=self.device).to(self.device).type(self.dtype)
            else:
                print('subspace solve: modify space')
                u_modify, s_modify, vT_modify = self.local_encoder_decoder_pullback_xt(
                    x=xt, t=t, op=op, block_idx=block_idx, pca_rank=pca_rank,
                    min_iter=10, max_iter=50, convergence_threshold=1e-4, mask=mask, noise=encoder_decoder_by_et)
                self._save(vT_modify, vT_modify_path)

            if vT_null is not None:
                pass
            elif null_space_projection and self._exists(vT_null_path):
                vT_null = self._load(vT_null_path, map_location=self.device).to(self.device).type(self.dtype)
            elif null_space_projection:
                print('subspace solve: null space')
                u_null, s_null, vT_null = self.local_encoder_decoder_pullback_xt(
                    x=xt, t=t, op=op, block_idx=block_idx, pca_rank=pca_rank_null,
                    min_iter=10, max_iter=50, convergence_threshold=1e-4, mask=~mask, noise=encoder_decoder_by_et)
                self._save(vT_null, vT_null_path)

            if random_edit:
                vT_modify = torch.randn_like(vT_modify)       # same generator state on every rank (seed broadcast in main)

            # normalize vT (edit.py:2316-2323)
            if not null_space_projection:
                vT = self.engine.null_project(vT_modify.contiguous(), None)
            else:
                vT = self.engine.null_project(vT_modify.contiguous(), vT_null[:pca_rank_null, :].contiguous())
            BASIS_NAME = (f"{encoder_decoder_by_et}_{tag}-edit_{self.edit_t}T_null_proj_{null_space_projection}"
                          f"_rank{pca_rank_null}_scale_{self.x_space_guidance_scale}")
            # the reference loops range(max(vis_num_pc, vT.shape[0])) (edit.py:2329) and would raise
            # IndexError for vis_num_pc > rank; main.py always passes vis_num_pc == pca_rank
            for pc_idx in range(min(max(vis_num_pc, vT.shape[0]), vT.shape[0])):
                self.EXP_NAME = f'{idx}-Edit_xt-noise-{BASIS_NAME}-pc_{pc_idx:0=3d}'
                self._save(vT[[pc_idx], :], os.path.join(save_dir, f'{self.EXP_NAME}-vT.pt'))
        else:
            print('loading the basis from --vT_path')
            vT = self._load(self.vT_path).to(self.device, torch.float32)
            BASIS_NAME = f"edit_{self.edit_t}T-load-basis-'{os.path.basename(self.vT_path)}'"

        # edit (edit.py:2339-2364)
        original_xt = xt.detach()
        n_pc = min(vis_num_pc, vT.shape[0])
        names = [f'{idx}-Edit-random{random_edit}_xt-noise-{BASIS_NAME}-pc_{pc_idx:0=3d}' for pc_idx in range(n_pc)]
        if self.batch_decode and n_pc > 1:
            # the reference decodes one direction after the other (edit.py:2340-2364: vis_num_pc calls of
            # DDIMforwardsteps on 5 frames each); here the frames of ALL directions go through the 59 decode steps as one
            # batch (the denoiser runs 2.4x faster per frame at 25 frames than at 5), one image grid per direction
            # as before.  Only the assignment of the eta = 1 draws to frames differs (one draw per step for the batch).
            frames = [self.edit_batch(original_xt, vT[pc_idx, :], vis_num) for pc_idx in range(n_pc)]
            per = frames[0].shape[0]
            dec = self._decode_frames(torch.cat(frames, dim=0), n_pc, per)
            for pc_idx, name in enumerate(names):
                self.EXP_NAME = name
                image = (dec[pc_idx * per:(pc_idx + 1) * per] / 2 + 0.5).clamp(0, 1)
                self._save_image(image, os.path.join(self.result_folder, f'{name}.png'), nrow=image.size(0))
            return frames[-1]
        for pc_idx in range(n_pc):
            self.EXP_NAME = names[pc_idx]
            xt = self.edit_batch(original_xt, vT[pc_idx, :], vis_num)
            self.DDIMforwardsteps(xt, t_start_idx=self.edit_t_idx, t_end_idx=-1, performance_boosting=True)
        return xt

    def _decode_frames(self, batch, n_pc, per):
        """Decode the frames of all directions (``n_pc`` walks of ``per`` frames) from the edit step to x0.

        The middle frame of every walk is the unedited ``xt`` itself (zero steps along the direction), i.e. the same
        image ``n_pc`` times.  Through the deterministic part of the decode (eta = 0: up to ``performance_boosting_t_idx``,
        edit.py:2556-2559) identical inputs stay identical, so only ONE copy goes through those steps; the copies are put
        back where the stochastic part starts (eta = 1: every frame then gets its own draw, as in the reference) or at the
        end.  ``LOCO_DEDUP_DECODE=0`` decodes all copies."""
        mid = per // 2
        dup = [d * per + mid for d in range(1, n_pc)]
        dedup = (os.environ.get("LOCO_DEDUP_DECODE", "1") != "0" and per % 2 == 1 and n_pc > 1
                 and all(torch.equal(batch[i], batch[mid]) for i in dup))
        n_t = len(self.scheduler.timesteps)
        pb = self.performance_boosting_t_idx
        stochastic = pb < n_t - 1                      # DDIMforwardsteps switches to eta = 1 from index pb on
        if not dedup or (stochastic and pb <= self.edit_t_idx):
            return self.DDIMforwardsteps(batch, t_start_idx=self.edit_t_idx, t_end_idx=-1, performance_boosting=True,
                                         save_image=False)
        keep = [i for i in range(batch.shape[0]) if i not in set(dup)]
        src = [keep.index(i) if i not in dup else keep.index(mid) for i in range(batch.shape[0])]     # full index -> unique index
        uniq = batch[keep].contiguous()
        if stochastic:
            uniq, _, _ = self.DDIMforwardsteps(uniq, t_start_idx=self.edit_t_idx, t_end_idx=pb, performance_boosting=True,
                                               save_image=False)
            return self.DDIMforwardsteps(uniq[src].contiguous(), t_start_idx=pb, t_end_idx=-1, performance_boosting=True,
                                         save_image=False)
        dec = self.DDIMforwardsteps(uniq, t_start_idx=self.edit_t_idx, t_end_idx=-1, performance_boosting=True,
                                    save_image=False)
        return dec[src].contiguous()

    def edit_batch(self, original_xt, vk_row, vis_num):
        """The +/- direction walk of edit.py:2346-2363 in one kernel: the S-fold
        repeated ``xt + scale*step*vk`` equals ``xt + j*scale*step*vk`` up to fp32
        rounding of the repeated adds; the reference's order of additions is kept by
        accumulating the scalar the same way."""
        S = self.x_space_guidance_num_step
        stride = None if vis_num == 1 else (S + 1) // vis_num
        idxs = [0, S] if vis_num == 1 else list(range(0, S + 1, stride))
        step = self.x_space_guidance_scale * self.x_space_guidance_edit_step
        alphas = [-j * step for j in reversed(idxs)][:-1] + [j * step for j in idxs]
        return self.engine.edit_axpy(original_xt.contiguous(), vk_row.contiguous().view(-1), alphas)

    @torch.no_grad()
    def x_space_guidance_direct(self, xt, t_idx, vk, single_edit_step):
        """edit.py:2618-2625."""
        out = self.engine.edit_axpy(xt.contiguous(), vk.contiguous().view(-1),
                                    [self.x_space_guidance_scale * single_edit_step])
        return out
