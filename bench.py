#!/usr/bin/env python3
"""Benchmark of the LOCO-Edit hot path on MI355X.

Metric (BASELINE.json): edit-directions/sec for a top-5 PMP-Jacobian basis at
256x256, t = 0.6T, + vT cosine similarity against the reference.  One *step* =
one complete subspace solve on synthetic input:
denoiser = CelebA-HQ DDPM architecture with the deterministic synthetic
checkpoint (seed 0), x_t = randn (seed 1), t = timesteps[40] = 595.36, mask =
rows 110:130 x cols 70:110 on 3 channels (L = 2400), V0 = randn (seed 7), 12
power iterations (the reference's minimum, edit.py:2492 with min_iter=10), i.e.
per step: thin QR of V0, 1 primal pass, 12 x (k tangent passes + k cotangent
passes + Gram/eig re-orthonormalisation).

Workloads (--workload):
  celeba_top5 (default, the headline): N GPUs = N REPLICAS of the metric's unit -- every
      rank solves the top-5 basis of its own image (5 probes, no data-path collective:
      what north_star calls natural for k = 5); value = 5 N directions / step time
      (barrier on both sides, max over ranks).  The ONE-image form (k = 5 N probes of one
      image sharded over the ranks, one RCCL all-gather of the A shards per iteration)
      is timed as the extra line `celeba_top5N_one_image_sharded`: a top-5N basis is not
      N top-5 bases, so it is not the headline (LOCO_BENCH_K_TOTAL forces it: smoke test).
  p2_k64 (BASELINE config 3): FFHQ-P2 architecture, 64 probes sharded over the N
      ranks (strong scaling, 64/N per rank), keep the leading 20 rows; value = 20 /
      step time.  The default run also times one step of it and reports it under
      "extra_workloads" next to the headline line.
  tloco_sd   (BASELINE config 4): latent T-LOCO on the Stable-Diffusion-shaped stand-ins (4x64x64 latent denoiser +
             the SD autoencoder's decoder geometry, Jacobian of the decoded 3x512x512 image); explicit workload only
             (three 832 M-parameter denoiser contexts take a minute to set up)
  tloco_if_i_m (BASELINE config 5): pixel-space T-LOCO at 64x64 on the DeepFloyd IF-I-M stage-I architecture
      (config.IF_I_M_UNET, loco_edit_amd.tloco): top-5 null-space basis of the CFG-combined Jacobian ("null+(for-null)",
      guidance 7.5: two denoiser branches per product), 5 probes per GPU sharded like the headline; value = 5 N directions /
      step time.  tloco_if64: the same on the round-2 stand-in (guided-diffusion U-Net, text through the time embedding).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--precision P]
                    [--no-cpu-baseline] [--no-e2e] [--no-extra]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (python -m torch.distributed.run, one process per GPU) before anything
touches the GPU and relays rank 0's JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_PER_GPU = 5
N_ITER = 12
MIN_ITER = 10     # edit.py:2292-2310: the stop test runs from i = 11 on
# Smoke-test knobs for the two-ranks-on-one-GPU test (two processes on one device take ~4 s per power iteration): fewer
# iterations, and a probe total that does not divide over the ranks.  A line produced under them says so ("smoke") and carries
# no parity block (the reference fixture is the 12-iteration solve).
SMOKE_ITERS = int(os.environ.get("LOCO_BENCH_SMOKE_ITERS", "0"))
K_TOTAL = int(os.environ.get("LOCO_BENCH_K_TOTAL", "0"))
if SMOKE_ITERS > 0:
    N_ITER, MIN_ITER = SMOKE_ITERS, max(0, SMOKE_ITERS - 2)
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (Chip-level parameters)
PEAK_F32_MFMA_TF = 157.3      # v_mfma_f32_32x32x2_f32 (exact fp32)
PEAK_BF16_MFMA_TF = 2500.0    # v_mfma_f32_32x32x16_bf16 / _f16; the split-bf16 path issues 3 MFMA flops per algorithmic flop
DTYPE_NOTE = {
    "bf16x3": "bf16x3 (split-bf16 MFMA operands, fp32 accumulate + fp32 storage)",
    "f32": "f32",
    "f16": "f16 (single f16 MFMA per product, fp32 accumulate + fp32 storage)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["celeba_top5", "p2_k64", "tloco_if64", "tloco_if_i_m", "tloco_sd", "tloco_sd15"], default="celeba_top5")
    ap.add_argument("--streams", type=int, choices=[1, 2], default=int(os.environ.get("LOCO_STREAMS", "2")),
                    help="unconditional workloads: probe groups of a tangent / cotangent pass on 2 HIP streams (default, as the "
                         "package since round 6: -3.5 ... -4.7 %% per headline solve, same results) or on 1.  The per-kernel profile "
                         "behind `roofline` is always taken on ONE stream (isolated kernel durations, the ones rocprofv3 reports for "
                         "a one-stream run); the one-stream step is the extra line celeba_top5_one_stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event profile step")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end phase timing (inversion ... decode)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workload (p2_k64) and extra precision lines")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16"], default=os.environ.get("LOCO_PRECISION", "bf16x3"),
                    help="conv arithmetic: exact fp32 MFMA, split-bf16 (3 bf16 MFMAs per product, fp32-faithful), or "
                         "one f16 MFMA per product")
    return ap.parse_args()


def launch_ranks(n):
    """Parent of a multi-GPU run: start n fresh rank processes (this process never touches the GPU), relay their
    output, exit with their code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def synthetic_inputs(cfg, k, device, image=0):
    """`image` > 0: another image and start block (the replicas of a multi-GPU headline run; 0 = the fixture's inputs)."""
    import torch
    x = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(1 + 1000 * image)).to(device)
    mask = torch.zeros(3, cfg.resolution, cfg.resolution, dtype=torch.bool)
    r = cfg.resolution
    mask[:, r * 110 // 256:r * 130 // 256, r * 70 // 256:r * 110 // 256] = True
    v0 = torch.randn(cfg.n, k, generator=torch.Generator().manual_seed(7 + 1000 * image)).to(device)
    return x, mask.to(device), v0


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, min(n, 32))


def cpu_baseline(cfg, params, t, budget_s=30.0):
    """Reference algorithm (oracle = pinned restatement: jacfwd + autograd.functional.jacobian
    + svd) timed on the host cores on a BOUNDED sample of the same workload (<= ~30 s)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import loco_oracle as orc
    cores = usable_cores()
    torch.set_num_threads(cores)
    p = orc.to_torch(params)
    oed = orc.OracleEdit(p, cfg)
    x = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(1))
    mask = torch.zeros(3, cfg.resolution, cfg.resolution, dtype=torch.bool)
    mask[:, 110:130, 70:110] = True
    tt = torch.tensor(float(t))
    with torch.no_grad():
        t0 = time.time()
        oed.unet(x, tt)
        t_first = time.time() - t0
        t0 = time.time()
        oed.unet(x, tt)
        t_fwd = time.time() - t0
    # one k=1 power iteration costs about 4-6 forward equivalents on the CPU
    if t_first + t_fwd * 7 < budget_s:
        v0 = torch.randn(cfg.n, 1, generator=torch.Generator().manual_seed(7))
        t0 = time.time()
        oed.pullback(x, tt, 1, v0, min_iter=1, max_iter=1, mask=mask)
        t_iter1 = time.time() - t0
        scale = (3 * 5 + 1) / (3 * 1 + 1)              # reference cost model (3k+1)F, BASELINE.md section 2
        if t_iter1 * scale < budget_s:
            # the workload's own iteration, k = 5 probes, timed directly: only the iteration count is scaled
            v5 = torch.randn(cfg.n, K_PER_GPU, generator=torch.Generator().manual_seed(7))
            t0 = time.time()
            oed.pullback(x, tt, K_PER_GPU, v5, min_iter=1, max_iter=1, mask=mask)
            t_iter5 = time.time() - t0
            sample = (f"1 power iteration (jacfwd JVP + autograd VJP + svd) at k={K_PER_GPU}, 256x256, fp32, timed "
                      f"directly: {t_iter5:.1f} s (k=1: {t_iter1:.1f} s, U-Net forward {t_fwd:.2f} s); x {N_ITER} iterations")
            measured, factor = t_iter5, float(N_ITER)
        else:
            t_iter5 = t_iter1 * scale
            sample = (f"1 power iteration (jacfwd JVP + autograd VJP + svd) at k=1, 256x256, fp32: {t_iter1:.1f} s; "
                      f"U-Net forward {t_fwd:.2f} s; EXTRAPOLATED to k=5 x {N_ITER} iterations by the (3k+1)F cost model")
            measured, factor = t_iter1, scale * N_ITER
    else:
        t_iter5 = t_fwd * (3 * 5 + 1)
        sample = (f"U-Net forward 256x256 fp32: {t_fwd:.2f} s (a full power iteration would exceed the {budget_s:.0f} s "
                  f"sample budget); EXTRAPOLATED by the reference's (3k+1) forward-equivalents per iteration, k=5 x {N_ITER}")
        measured, factor = t_fwd, (3 * 5 + 1) * N_ITER
    val = K_PER_GPU / (N_ITER * t_iter5)
    return {"value": val, "unit": "edit-directions/s", "cores": cores, "kind": "port", "extrapolated": True,
            "measured_sample_s": round(measured, 3), "scale_factor_to_full_solve": round(factor, 2), "sample": sample}


def parity_vs_fixture(s, vT, name):
    """|cos| of every row of vT and the relative error of s against the fixture the REFERENCE produced on the same
    inputs (tests/golden/<name>.pt, written by oracle/make_golden.py from /root/reference; edit.py:2406-2504)."""
    import torch
    path = os.path.join(ROOT, "tests", "golden", name + ".pt")
    if not os.path.exists(path):
        return None
    g = torch.load(path)
    k = g["s_modify"].shape[0]
    ref = g["vT_modify_f16"].double()                                # fp64 dot products: n = 196608 terms
    ref = ref / ref.norm(dim=1, keepdim=True)
    v = vT[:k].detach().cpu().double()
    v = v / v.norm(dim=1, keepdim=True)
    cos = (v * ref).sum(dim=1).abs()
    ov = torch.linalg.svdvals(v @ ref.T)                             # principal-angle cosines of the two spans
    srel = ((s[:k].detach().cpu() - g["s_modify"]).abs() / g["s_modify"]).max()
    return {"fixture": f"tests/golden/{name}.pt (reference output, {g['n_iter']} iterations, same x/t/mask/V0)",
            "n_iter": int(g["n_iter"]), "cos_min": round(float(cos.min()), 6), "cos": [round(float(c), 6) for c in cos],
            "span_cos_min": round(float(ov.min()), 6), "s_relerr": float(f"{float(srel):.3e}"),
            "bar": "|cos| >= 0.99 (north_star)"}


def e2e_phases(eng, cfg, sched_cls, solver, device):
    """The reference flow a5 -> a11 on one synthetic image, seconds per phase (edit.py:2216-2366 with
    --pca_rank 5 --pca_rank_null 5, 12 power iterations per solve, scale 0.5 x 16 steps, vis_num 2)."""
    import torch
    sync = torch.cuda.synchronize
    out = {}
    inv = sched_cls(); inv.set_timesteps(100, is_inversion=True)
    fwd = sched_cls(); fwd.set_timesteps(100)
    x0 = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(0)).clamp(-1, 1).to(device)

    def chain(x, sched, i0, i1):
        for i in range(i0, i1):
            t = sched.timesteps[i]
            x = eng.ddim_step(x, float(t), sched.alpha_at(t), sched.alpha_at(sched.timesteps_next[i]), 0.0, None)
        return x
    sync(); t0 = time.perf_counter()
    xT = chain(x0, inv, 0, len(inv.timesteps) - 1)                       # 98 evaluations, B = 1 (edit.py:2147-2148)
    sync(); out["inversion_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    xt = chain(xT, fwd, 0, 40)                                           # 40 evaluations to t = 0.6T
    sync(); out["to_t_s"] = time.perf_counter() - t0
    t = float(fwd.timesteps[40]); at = fwd.alpha_at(t)
    mask = torch.zeros(3, cfg.resolution, cfg.resolution, dtype=torch.bool, device=device)
    mask[:, 110:130, 70:110] = True
    g = torch.Generator().manual_seed(7)
    v0m, v0n = torch.randn(cfg.n, 5, generator=g).to(device), torch.randn(cfg.n, 5, generator=g).to(device)
    sync(); t0 = time.perf_counter()
    _, _, vTm, _ = solver.local_basis(eng, xt, t, at, 5, mask=mask, min_iter=N_ITER, max_iter=N_ITER, v0=v0m, verbose=False)
    _, _, vTn, _ = solver.local_basis(eng, xt, t, at, 5, mask=~mask, min_iter=N_ITER, max_iter=N_ITER, v0=v0n, verbose=False)
    sync(); out["two_solves_sequential_s"] = round(time.perf_counter() - t0, 4)
    # what run_edit_null_space_projection does by default: both solves' probes in one batch per pass (one untimed call
    # first: the 10-probe launch shapes have not run yet in this process, the 5-probe ones were warmed by the headline)
    solver.local_basis_pair(eng, xt, t, at, 5, mask, 5, ~mask, min_iter=1, max_iter=1, v0_a=v0m, v0_b=v0n, verbose=False)
    sync(); t0 = time.perf_counter()
    (_, _, vTm, _), (_, _, vTn, _) = solver.local_basis_pair(eng, xt, t, at, 5, mask, 5, ~mask, min_iter=N_ITER, max_iter=N_ITER,
                                                             v0_a=v0m, v0_b=v0n, verbose=False)
    sync(); out["two_solves_s"] = time.perf_counter() - t0
    # the same pair under the reference's own arguments (edit.py:2292-2310: min_iter=10, max_iter=50) and its stop rule: with
    # five probes the reference's allclose never holds (LAPACK's sign flips, tests/golden/converge.pt), so both solves run 50
    t0 = time.perf_counter()
    (_, _, _, n50a), (_, _, _, n50b) = solver.local_basis_pair(eng, xt, t, at, 5, mask, 5, ~mask, min_iter=10, max_iter=50,
                                                               convergence_threshold=1e-4, v0_a=v0m, v0_b=v0n, verbose=False,
                                                               stop_rule="reference")
    sync(); out["two_solves_reference_stop_rule_s"] = time.perf_counter() - t0
    assert (n50a, n50b) == (50, 50)
    t0 = time.perf_counter()
    vT = eng.null_project(vTm, vTn)
    xb = eng.edit_axpy(xt, vT[0].contiguous(), [-8.0, -4.0, 0.0, 4.0, 8.0])
    sync(); out["projection_edit_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    dec = chain(xb, fwd, 40, 99)                                         # 59 steps x 5 frames of ONE direction (eta = 0)
    sync(); out["decode_one_direction_s"] = time.perf_counter() - t0
    # what run_edit_null_space_projection does (edit.py:2340-2364 decodes EVERY direction): the 5-frame walks of all five
    # directions through the 59 steps as one 25-frame batch (engine batches above max_batch are cut by the caller)
    xall = torch.cat([eng.edit_axpy(xt, vT[i].contiguous(), [-8.0, -4.0, 0.0, 4.0, 8.0]) for i in range(5)], dim=0)
    mb = eng.max_batch

    def chain_chunked(x, sched, i0, i1):
        for i in range(i0, i1):
            t = sched.timesteps[i]
            a0, a1 = sched.alpha_at(t), sched.alpha_at(sched.timesteps_next[i])
            x = torch.cat([eng.ddim_step(x[b0:b0 + mb].contiguous(), float(t), a0, a1, 0.0, None)
                           for b0 in range(0, x.shape[0], mb)], dim=0) if x.shape[0] > mb else eng.ddim_step(x, float(t), a0, a1, 0.0, None)
        return x
    # ... the way EditUncondDiffusion._decode_frames does it: the middle frame of every walk is the unedited xt, five
    # identical images; one copy goes through the deterministic steps (to index 79 = performance_boosting_t 0.2 of the
    # shipped script), the copies are put back where the reference's decode turns stochastic (eta = 1, one draw per frame)
    keep = [i for i in range(25) if not (i % 5 == 2 and i > 2)]
    src = [keep.index(i) if i in keep else keep.index(2) for i in range(25)]
    assert all(torch.equal(xall[i], xall[2]) for i in range(25) if i not in keep)
    xuniq = xall[keep].contiguous()
    chain_chunked(xall, fwd, 40, 41); chain_chunked(xuniq, fwd, 40, 41)   # warm the 25- and 21-frame launch shapes
    sync(); t0 = time.perf_counter()
    dec_all = chain_chunked(chain_chunked(xuniq, fwd, 40, 79)[src].contiguous(), fwd, 79, 99)
    sync(); out["decode_all_directions_s"] = time.perf_counter() - t0
    out = {k: round(v, 4) for k, v in out.items()}
    out["basis_plus_edit_s"] = round(out["two_solves_s"] + out["projection_edit_s"], 4)
    out["image_total_one_direction_s"] = round(sum(out[k] for k in ("inversion_s", "to_t_s", "two_solves_s",
                                                                     "projection_edit_s", "decode_one_direction_s")), 4)
    out["image_total_s"] = round(sum(out[k] for k in ("inversion_s", "to_t_s", "two_solves_s", "projection_edit_s",
                                                      "decode_all_directions_s")), 4)
    out["image_total_reference_stop_rule_s"] = round(out["image_total_s"] - out["two_solves_s"] + out["two_solves_reference_stop_rule_s"], 4)
    out["decode_batch"] = f"{xall.shape[0]} frames, engine max_batch {mb}"
    out["note"] = ("synthetic weights; image_total_s = the reference's flow for one image (inversion 98 evaluations, 40 to "
                   "t, the modify + null solves with 12 iterations each -- image_total_reference_stop_rule_s: with the 50 iterations "
                   "each that the reference's stop rule gives five-probe solves --, projection + edit walk, decode of ALL five "
                   "directions = 39 deterministic steps x 21 distinct frames (the five walks share their unedited middle frame: "
                   "one copy until the reference's decode turns stochastic at index 79) + 20 steps x 25 frames; eta = 0 "
                   "throughout in this timing); finite output: "
                   + str(bool(torch.isfinite(dec).all() and torch.isfinite(dec_all).all())))
    return out


_T0 = time.perf_counter()


def _mark(label):
    """LOCO_BENCH_TRACE=1: wall-clock marks on stderr (where a bench process spends its time outside the timed region)."""
    if os.environ.get("LOCO_BENCH_TRACE"):
        print(f"[bench trace pid {os.getpid()} rank {os.environ.get('RANK', '-')}] {time.perf_counter() - _T0:7.2f} s  {label}", file=sys.stderr, flush=True)


def main():
    a = parse()
    _mark("start")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        launch_ranks(a.gpus)            # never returns
    world = int(env_world or "1")
    if world != a.gpus and not (env_world is None and a.gpus == 1):
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import loco_edit_amd  # noqa: F401
    from loco_edit_amd.config import CELEBA_DDPM, FFHQ_P2, synth_params
    from loco_edit_amd.dist import ProbeSharder
    from loco_edit_amd.hip import LocoEngine
    from loco_edit_amd.scheduler import YHCustomScheduler
    from loco_edit_amd import solver

    _mark("imports done")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LOCO_BENCH_BACKEND=gloo lets several ranks share one GPU (a smoke test of the sharded path on a 1-GPU box);
    # the default is RCCL with one GPU per rank
    backend = os.environ.get("LOCO_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        # a mis-bound launch must not report n_gpus = N: the group has N ranks and, under RCCL, N distinct GPUs
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        ident = [None] * world
        props = torch.cuda.get_device_properties(device)
        dist.all_gather_object(ident, (os.uname().nodename, str(getattr(props, "uuid", "")) or f"idx{torch.cuda.current_device()}",
                                       torch.cuda.current_device()))
        if backend == "nccl":
            assert len(set(ident)) == world, f"ranks share a GPU: {ident}"
        n_distinct_gpus = len(set(ident))
    else:
        n_distinct_gpus = 1
    _mark("process group / device ready")
    sched = YHCustomScheduler()
    sched.set_timesteps(100)
    t = float(sched.timesteps[40])
    at = sched.alpha_at(t)
    sharder = ProbeSharder("world")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, res

    class SimulatedShard:
        """Rank 0's share of a probe batch dealt over `n` ranks, run on this one GPU (multi-GPU readiness without the hardware,
        VERDICT r04 item 8): `rows` is rank 0's block, the all-gather returns rank 0's rows on top of fixed random filler rows
        (same shapes and kernels as the sharded solve; the iterates are not meaningful, the time is)."""
        active, is_main, rank = False, True, 0

        def __init__(self, n):
            self.world = n
            self._fill = {}

        def rows(self, kk):
            from loco_edit_amd.dist import shard_bounds
            return shard_bounds(kk, self.world, 0)

        def all_gather_rows(self, local, kk):
            key = (kk,) + tuple(local.shape[1:])
            if key not in self._fill:
                g = torch.Generator(device=local.device).manual_seed(1234)
                self._fill[key] = torch.randn(key, generator=g, device=local.device, dtype=local.dtype) / (key[1] ** 0.5)
            out = self._fill[key].clone()
            out[:local.shape[0]] = local
            return out

        def barrier(self):
            pass

        def agree(self, value):
            return value

    def make_tloco(prec, real=False):
        """config 5: CFG-combined subspace solve (one engine context per prompt) on the IF-shaped stand-in or, `real`, on the
        DeepFloyd IF-I-M architecture itself (config.IF_I_M_UNET, synthetic weights, seeded 77 x 4096 text states)."""
        from argparse import Namespace
        from loco_edit_amd.config import IF64_STANDIN, IF_I_M_UNET
        from loco_edit_amd.tloco import EditDeepFloydIF
        cfg = IF_I_M_UNET if real else IF64_STANDIN
        k = K_PER_GPU * world
        args = Namespace(device=device, dtype=torch.float32, seed=1, unet_config=cfg, synthetic_weights=0, ckpt_path="", max_batch=8,
                         precision=prec, dataset_name="Random", for_steps=100, use_yh_custom_scheduler=True, guidance_scale=7.5,
                         guidance_scale_edit=7.5, prompt_emb=None, prompt_emb_seed=31, cond_dim=64, for_prompt="standin",
                         edit_prompt="standin-edit", edit_t=0.75, sampling_mode=False, tilda_v_score_type="null+(for-null)+(edit-null)",
                         ablation_method="null-space-proj", mask_type="SAM", vT_path="", x_space_guidance_edit_step=1.0,
                         x_space_guidance_scale=10.0, x_space_guidance_num_step=1,
                         result_folder=os.path.join(ROOT, "gpurun_out", "bench_tloco"))
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            ed = EditDeepFloydIF(args)
        ed.sharder = sharder
        x, mask, v0 = synthetic_inputs(cfg, k, device)
        tt = ed.scheduler.timesteps[ed.edit_t_idx]
        F_, E_, N_ = ed.for_prompt_emb, ed.edit_prompt_emb, ed.null_prompt_emb

        def step():
            u, s, vT = ed.local_encoder_decoder_pullback_xt(x, tt, ed.edit_t_idx, F_, E_, N_, pca_rank=k, min_iter=N_ITER,
                                                            max_iter=N_ITER, mask=~mask, mode="null+(for-null)", v0=v0, verbose=False)
            return u, s, vT, ed.last_n_iter
        return dict(cfg=cfg, k=k, keep=k, eng=ed.engine, params=None, step=step, x=x, mask=~mask, v0=v0, branches=2,
                    branch_streams=ed.branch_streams.enabled)

    def make_tloco_sd(prec, real=False):
        """config 4: CFG-combined subspace solve of the DECODED image's Jacobian w.r.t. the latent (denoiser engines per
        prompt + decoder engine).  `real`: the Stable Diffusion v1.x denoiser architecture itself (config.SD15_UNET: latent-
        diffusion UNetModel with SpatialTransformer blocks, 859.5 M parameters, synthetic weights) instead of the stand-in."""
        from argparse import Namespace
        from loco_edit_amd.config import SD15_UNET, SD64_XATTN_STANDIN, SD_VAE_DECODER
        from loco_edit_amd.tloco_sd import EditStableDiffusion
        cfg, vcfg = (SD15_UNET if real else SD64_XATTN_STANDIN), SD_VAE_DECODER
        k = K_PER_GPU * world
        args = Namespace(device=device, dtype=torch.float32, seed=1, unet_config=cfg, vae_config=vcfg, synthetic_weights=0,
                         ckpt_path="", vae_ckpt_path="", max_batch=(5 if real else 8), precision=prec, dataset_name="Random", for_steps=100,
                         use_yh_custom_scheduler=True, guidance_scale=7.5, guidance_scale_edit=4.0, prompt_emb=None,
                         prompt_emb_seed=31, cond_dim=64, for_prompt="standin", edit_prompt="standin-edit", edit_t=0.7,
                         sampling_mode=False, tilda_v_score_type="null+(for-null)+(edit-null)", ablation_method="null-space-proj",
                         mask_type="SAM", vT_path="", use_sega=False, x_space_guidance_edit_step=1.0, x_space_guidance_scale=8.0,
                         x_space_guidance_num_step=1, result_folder=os.path.join(ROOT, "gpurun_out", "bench_tloco_sd"))
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            ed = EditStableDiffusion(args)
        ed.sharder = sharder
        z = torch.randn(1, 4, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(1)).to(device)
        R = vcfg.out_resolution
        mask = torch.zeros(3, R, R, dtype=torch.bool)
        mask[:, R * 110 // 256:R * 130 // 256, R * 70 // 256:R * 110 // 256] = True       # the l_eye-sized rectangle at 512^2
        mask = mask.to(device)
        v0 = torch.randn(cfg.n, k, generator=torch.Generator().manual_seed(7)).to(device)
        tt = ed.scheduler.timesteps[ed.edit_t_idx]
        F_, E_, N_ = ed.for_prompt_emb, ed.edit_prompt_emb, ed.null_prompt_emb

        def step():
            u, s, vT = ed.local_encoder_decoder_pullback_zt(z, tt, ed.edit_t_idx, F_, E_, N_, pca_rank=k, min_iter=N_ITER,
                                                            max_iter=N_ITER, mask=mask, mode="null+(for-null)", v0=v0, verbose=False)
            return u, s, vT, ed.last_n_iter
        return dict(cfg=cfg, k=k, keep=k, eng=ed.engine, params=None, step=step, x=z, mask=mask, v0=v0, branches=2, dec=ed.vae_engine,
                    branch_streams=ed.branch_streams.enabled)

    def make_workload(name, prec, sharded=False):
        if name == "tloco_if64":
            return make_tloco(prec)
        if name == "tloco_if_i_m":
            return make_tloco(prec, real=True)
        if name == "tloco_sd":
            return make_tloco_sd(prec)
        if name == "tloco_sd15":
            return make_tloco_sd(prec, real=True)
        replicas = False
        if name == "celeba_top5":
            # N > 1: N independent top-5 solves, one image per rank (the metric's unit).  `sharded` (the
            # extra line) or LOCO_BENCH_K_TOTAL: ONE image, k probes dealt over the ranks, one all-gather per iteration
            replicas = world > 1 and not sharded and not K_TOTAL
            cfg, k, keep = CELEBA_DDPM, K_PER_GPU * world, K_PER_GPU * world
            if replicas:
                k = keep = K_PER_GPU
            if K_TOTAL > 0:
                k = keep = K_TOTAL
        else:
            cfg, k, keep = FFHQ_P2, 64, 20
        params = synth_params(cfg, seed=0)
        # probe batch resident per pass: the top-5 workload carries 5 (its e2e leg 5 + 5, the paired modify / null
        # solves, and its decode leg the 25 frames of all five directions: room for 32); the 64-probe workload fills the deep levels better
        # with its whole shard in one pass (measured 2.87 / 2.67 / 2.62 s per solve at 8 / 16 / 32; 0.8 GB of arena per probe)
        k_rank = (k + world - 1) // world
        mb = int(os.environ.get("LOCO_BENCH_MAX_BATCH", "0")) or (32 if name == "celeba_top5" else min(32, max(8, k_rank)))
        eng = LocoEngine(cfg, max_batch=mb, device=device)
        eng.load_state_dict(params)
        eng.set_precision(prec)
        # --streams 2 (LOCO_STREAMS=2 in the package): the two probe groups of a pass on two HIP streams, the side stream
        # chosen by measurement; 1 when no stream of this process runs beside the current one
        n_streams = eng.set_streams_measured(2) if a.streams == 2 else 1
        x, mask, v0 = synthetic_inputs(cfg, k, device, image=rank if replicas else 0)
        own = ProbeSharder(None) if replicas else None       # a replica's solve never enters a collective

        def step(shard=None):
            # the reference's flow (edit.py:2292-2310 min_iter=10) cut at its 12th iteration: the stop test is evaluated
            # where the reference evaluates it (i = 11: one 2-float readback) inside the timed region; with k >= 2 probes
            # it cannot end the loop (solver.default_stop_rule: LAPACK's sign flips, tests/golden/converge.pt)
            return solver.local_basis(eng, x, t, at, k, mask=mask, min_iter=MIN_ITER, max_iter=N_ITER,
                                      convergence_threshold=1e-4, v0=v0, sharder=shard or own or sharder, verbose=False,
                                      stop_rule="reference")
        return dict(cfg=cfg, k=k, keep=keep, eng=eng, params=params, step=step, x=x, mask=mask, v0=v0, streams=n_streams,
                    replicas=replicas)

    w = make_workload(a.workload, a.precision)
    _mark("workload built")
    eng, cfg, k, keep = w["eng"], w["cfg"], w["k"], w["keep"]
    for _ in range(a.warmup):            # warm-up outside the clock stamps (timed() below then runs 0 more)
        w["step"]()
    ck0 = eng.clock_stamp()
    elapsed, (u, s, vT, n_iter) = timed(w["step"], a.steps, 0)
    ck1 = eng.clock_stamp()
    torch.cuda.synchronize()
    sclk = LocoEngine.sclk_mhz(ck0, ck1)
    _mark("warm-up + timed steps done")
    ms_per_step = elapsed / a.steps * 1e3
    replicas = bool(w.get("replicas"))
    # replicas: every rank kept `keep` directions of its own image in that (max-over-ranks) time
    value = keep * (world if replicas else 1) / (elapsed / a.steps)
    k_local = k if replicas else sharder.rows(k)[1] - sharder.rows(k)[0]

    # ---- roofline leg: per-kernel HIP-event profile of one more identical step
    roofline = None
    F = eng.unet_flops()
    if not a.no_profile:
        # every rank runs the extra step (it contains the all-gather); only rank 0 records the per-kernel events
        if rank == 0:
            eng.profile_enable(True)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        w["step"]()
        torch.cuda.synchronize()
        t_prof = time.perf_counter() - tp0
    if rank == 0 and not a.no_profile:
        rep = eng.profile_report()
        eng.profile_enable(False)
        if os.environ.get("LOCO_BENCH_SHAPES"):      # diagnostics: one more step profiled per LAYER SHAPE, dumped for tests/diag/shape_excess.py
            eng.profile_enable(2)
            w["step"]()
            torch.cuda.synchronize()
            json.dump(eng.profile_report(), open(os.environ["LOCO_BENCH_SHAPES"], "w"))
            eng.profile_enable(False)
        dom = max(rep.items(), key=lambda kv: kv[1]["ms"])
        name, r = dom
        achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
        if a.precision == "bf16x3":
            peak, issued = PEAK_BF16_MFMA_TF, 3.0 * achieved
        elif a.precision == "f16":
            peak, issued = PEAK_BF16_MFMA_TF, achieved
        else:
            peak, issued = PEAK_F32_MFMA_TF, achieved
        tot_ms = sum(v["ms"] for v in rep.values())
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and a.workload == "celeba_top5":
            try:
                import hashlib
                from loco_edit_amd.hip import library_path
                tj = json.load(open(tpath))
                sha = hashlib.sha256(open(library_path(), "rb").read()).hexdigest()
                if tj.get("_lib_sha256") == sha:
                    traffic = tj.get(name)
                    traffic_src = tj.get("_source", "") + (" -- stored by profiles/collect.sh for exactly this library build "
                                                            "(sha256 match), not re-measured by this run")
                else:
                    traffic_src = ("profiles/traffic.json was measured on another build of libloco_hip.so (sha256 mismatch): not "
                                   "reported; re-run profiles/collect.sh")
            except Exception:
                traffic = None
        executed = w.get("branches", 1) * (1 + 2 * k_local * N_ITER) * F      # the primal runs once per solve and branch
        roofline = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "mfma_flops_issued_TFLOPs": round(issued, 2), "mfma_issue_frac": round(issued / peak, 4),
            "note": ("achieved = algorithmic 2*MAC / time, frac = achieved / dense bf16 MFMA peak; split-bf16 issues 3 "
                     "MFMA flops per algorithmic flop, so the matrix pipe itself runs at mfma_issue_frac of peak "
                     "(the exact-fp32 MFMA peak is 157.3 TF/s)" if a.precision == "bf16x3" else
                     "achieved = algorithmic 2*MAC / time on " + ("the exact-fp32 MFMA" if a.precision == "f32"
                                                                  else "one f16 MFMA per product")),
            "launches": r["launches"], "avg_launch_ms": round(r["ms"] / r["launches"], 4),
            "measured_on": ("one more step of the same solve with a HIP event pair around every conv launch on the stream it is "
                            "launched on; that step runs the probe groups on ONE stream (isolated kernel durations, the ones "
                            "rocprofv3 --kernel-trace reports), the timed steps behind `value` run them on "
                            f"{w.get('streams', 1)} stream(s)"),
            "flops_per_launch": r["flops"] / r["launches"],
            "conv_share_of_step": round(tot_ms / (t_prof * 1e3), 3),
            "whole_step_TFLOPs_executed_per_gpu": round(executed / (ms_per_step * 1e-3) / 1e12, 2),
            "all_conv_kernels": {n: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                     "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                 for n, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])},
        }

    _mark("profile step done")
    # ---- parity leg (outside the timed region): the metric's second half, vT cosine vs the reference
    parity = None
    if a.workload == "celeba_top5" and not SMOKE_ITERS and not K_TOTAL:
        if world == 1 or replicas:      # (rank 0's replica runs the fixture's inputs)
            ps, pvT = s, vT
        else:   # the k = 5 solve of the fixture, replicated on every rank (no collective), reported by rank 0
            x5, m5, v05 = synthetic_inputs(cfg, K_PER_GPU, device)
            _, ps, pvT, _ = solver.local_basis(eng, x5, t, at, K_PER_GPU, mask=m5, min_iter=N_ITER, max_iter=N_ITER,
                                               v0=v05, sharder=ProbeSharder(None), verbose=False)
        if rank == 0:
            parity = parity_vs_fixture(ps, pvT, "celeba256")

    _mark("parity leg done")
    extra = {}
    e2e = None
    if rank == 0 and world == 1 and a.workload == "celeba_top5" and not a.no_e2e:
        try:
            e2e = e2e_phases(eng, cfg, YHCustomScheduler, solver, device)
        except Exception as ex:            # never lose the headline line to an optional leg
            e2e = {"error": repr(ex)[:200]}
    if a.workload == "celeba_top5" and not a.no_extra:
        # other conv arithmetic modes on the same workload, one timed step each, with their measured cosine
        if world == 1:
            for prec in ("f16", "f32"):
                if prec == a.precision:
                    continue
                try:
                    eng.set_precision(prec)
                except Exception:
                    continue
                el, (_, s2, vT2, _) = timed(w["step"], 1, 1)
                extra[f"celeba_top5_{prec}"] = {"value": round(keep / el, 4), "unit": "edit-directions/s",
                                                "ms_per_step": round(el * 1e3, 3), "dtype": DTYPE_NOTE[prec],
                                                "parity": parity_vs_fixture(s2, vT2, "celeba256")}
            eng.set_precision(a.precision)
            # the same solve in the other stream mode (loco_set_streams).  Two streams: the probe groups of a pass side by side,
            # statistics / apply kernels of one group beside the convolutions of the other.  An extra line, not the headline:
            # the pass then consists of 2- and 3-probe launches (isolated, each is a worse kernel than the 5-probe launch:
            # 82.9 us for 2.5 probes on average against 101.7 us for 5 under rocprofv3 on one box) whose overlap is what wins,
            # so per-kernel durations and the roofline fraction are only meaningful on one stream
            if w["streams"] == 2:
                eng.set_streams(1)
                other, n_st = "celeba_top5_one_stream", 1
            else:
                n_st = eng.set_streams_measured(2)      # side stream picked by measurement (loco_set_side_stream): not queue luck
                other = "celeba_top5_two_streams"
            el, (_, s2, vT2, _) = timed(w["step"], 2, 1)
            eng.set_streams(w["streams"])
            extra[other] = {"value": round(keep / (el / 2), 4), "unit": "edit-directions/s", "streams": n_st,
                            "ms_per_step": round(el / 2 * 1e3, 3), "dtype": DTYPE_NOTE[a.precision],
                            "parity": parity_vs_fixture(s2, vT2, "celeba256")}
        if world > 1 and replicas:
            # ONE image, 5 N probes dealt over the ranks (one all-gather of the A shards per iteration, replicated k x k algebra):
            # the form the sharded solver exists for, as an extra line under its own name -- its unit is a row of a top-5N basis
            try:
                ws = make_workload("celeba_top5", a.precision, sharded=True)
                el, (_, ss, vTs, _) = timed(ws["step"], 2, 1)
                if rank == 0:
                    kk = ws["k"]
                    extra["celeba_top5N_one_image_sharded"] = {
                        "value": round(kk / (el / 2), 4), "unit": f"rows/s of ONE top-{kk} basis (not {world} top-5 bases)",
                        "ms_per_step": round(el / 2 * 1e3, 3), "scaling": "weak", "n_gpus": world, "probes_total": kk,
                        "probes_per_gpu": sharder.rows(kk)[1] - sharder.rows(kk)[0], "n_iter": N_ITER,
                        "collective": "one all-gather of the A shards per iteration (torch.distributed, backend " + backend + ")",
                        "orthonormality_err": float(f"{float((vTs.double() @ vTs.double().T - torch.eye(kk, device=device, dtype=torch.float64)).abs().max()):.2e}")}
                del ws
            except Exception as ex:
                if rank == 0:
                    extra["celeba_top5N_one_image_sharded"] = {"error": repr(ex)[:200]}
            torch.cuda.empty_cache()
        # BASELINE config 5 next to the headline: T-LOCO null-space basis on the DeepFloyd IF-I-M architecture, 2 CFG branches
        try:
            w3 = make_workload("tloco_if_i_m", a.precision)
            el3, (_, s5, vT5, _) = timed(w3["step"], 3, 1)       # three timed steps, as `--workload tloco_if_i_m` runs by default
            el = el3 / 3
            if rank == 0:
                F3 = w3["eng"].unet_flops()
                kl = sharder.rows(w3["k"])[1] - sharder.rows(w3["k"])[0]
                extra["tloco_if_i_m"] = {
                    "value": round(w3["k"] / el, 4), "unit": "edit-directions/s (top-5 null-space basis per GPU, CFG-combined Jacobian)",
                    "ms_per_step": round(el * 1e3, 3), "scaling": "weak", "n_gpus": world, "probes_per_gpu": kl, "n_iter": N_ITER,
                    "cfg_branches": 2, "unet_GFLOP": round(F3 / 1e9, 2),
                    "whole_step_TFLOPs_executed_per_gpu": round(2 * (1 + 2 * kl * N_ITER) * F3 / el / 1e12, 2),
                    "singular_values_head": [round(float(v), 4) for v in s5.tolist()[:5]],
                    "denoiser": "the DeepFloyd IF-I-M stage-I architecture (64x64, 192 x (1,2,3,4), 3 ResBlocks per level, GELU, "
                                "(skip + h)/sqrt 2, attention at 32/16/8 over [77 text ; image] keys, 315 M U-Net parameters + host-side "
                                "text conditioning of the 77x4096 states; synthetic weights) -- parity of the network is unpinned (no "
                                "diffusers / deepfloyd_if, no weights); r03 and before timed a guided-diffusion stand-in under 'tloco_if64'"}
            del w3
        except Exception as ex:            # never lose the headline line to an optional workload
            if rank == 0:
                extra["tloco_if_i_m"] = {"error": repr(ex)[:200]}
        torch.cuda.empty_cache()
        # BASELINE config 4 next to the headline (single GPU only: four engine contexts): latent T-LOCO on the Stable
        # Diffusion v1 denoiser architecture itself, Jacobian of the decoded 512^2 image
        if world == 1:
            try:
                w4 = make_workload("tloco_sd15", a.precision)
                el2, (_, s4, vT4, _) = timed(w4["step"], 2, 1)      # two timed steps: single steps of this two-context workload scatter
                el = el2 / 2
                Fu, Fd = w4["eng"].unet_flops(), w4["dec"].unet_flops()
                extra["tloco_sd15"] = {
                    "value": round(w4["k"] / el, 4), "unit": "edit-directions/s (top-5 basis, decoded-image Jacobian w.r.t. the latent)",
                    "ms_per_step": round(el * 1e3, 3), "n_iter": N_ITER, "cfg_branches": 2, "mask_L": int(w4["mask"].sum().item()),
                    "denoiser_GFLOP": round(Fu / 1e9, 2), "decoder_GFLOP": round(Fd / 1e9, 2),
                    "whole_step_TFLOPs_executed": round((1 + 2 * w4["k"] * N_ITER) * (2 * Fu + Fd) / el / 1e12, 2),
                    "singular_values_head": [round(float(v), 4) for v in s4.tolist()[:5]],
                    "networks": "the Stable Diffusion v1.x denoiser architecture (latent-diffusion UNetModel 320x(1,2,4,4), "
                                "SpatialTransformer blocks, 77x768 prompt states, 859.5 M parameters, synthetic weights) + the SD "
                                "autoencoder decoder geometry (49.5 M parameters); parity of these two networks is unpinned "
                                "(diffusers is not vendored: checked against autodiff of a restatement)"}
                del w4
            except Exception as ex:        # never lose the headline line to an optional workload
                extra["tloco_sd15"] = {"error": repr(ex)[:200]}
            torch.cuda.empty_cache()
        # BASELINE config 3 next to the headline: FFHQ-P2, 64 probes over the ranks, keep 20
        w2 = make_workload("p2_k64", a.precision)
        el, (_, s3, vT3, _) = timed(w2["step"], 1, 1)
        if rank == 0:
            F2 = w2["eng"].unet_flops()
            kl = sharder.rows(64)[1] - sharder.rows(64)[0]
            extra["p2_k64"] = {
                "value": round(20 / el, 4), "unit": "edit-directions/s (20 kept of 64 probes)", "ms_per_step": round(el * 1e3, 3),
                "scaling": "strong", "n_gpus": world, "probes_per_gpu": kl, "n_iter": N_ITER,
                # every rank evaluates the shared primal once per solve and its own share of the 2 x 64 x n_iter probe passes
                "ideal_speedup_vs_1gpu": round((1 + 2 * 64 * N_ITER) / (1 + 2 * kl * N_ITER), 3),
                "whole_step_TFLOPs_executed_per_gpu": round((1 + 2 * kl * N_ITER) * F2 / el / 1e12, 2),
                "singular_values_head": [round(float(v), 4) for v in s3.tolist()[:5]],
                "orthonormality_err": float(f"{float((vT3[:20].double() @ vT3[:20].double().T - torch.eye(20, device=device, dtype=torch.float64)).abs().max()):.2e}"),
            }
        if world == 1:
            # What the strong-scaling run should show, predicted on the one GPU there is: rank 0's share of the 64 probes for
            # N = 2, 4, 8 (32 / 16 / 8 probes per pass, the shared primal, the replicated k x k algebra on all 64 rows) timed
            # here, + the per-iteration all-gather of the A rows priced at 7 xGMI links x 153 GB/s.  Below 64^2 an 8-probe pass
            # leaves most launches narrower than the chip, so the prediction falls short of `ideal` -- by how much is the point.
            try:
                pred = {}
                n_in = w2["v0"].shape[0]
                for nr in (2, 4, 8):
                    sh = SimulatedShard(nr)
                    el_n, _ = timed(lambda: w2["step"](sh), 1, 1)
                    gather_s = N_ITER * 64 * n_in * 4 * (nr - 1) / nr / (7 * 153e9)
                    kl_n = sh.rows(64)[1]
                    pred[str(nr)] = {"probes_per_gpu": kl_n, "ms_rank_shard_on_one_gpu": round(el_n * 1e3, 2),
                                     "ms_all_gather_at_7x153GBps": round(gather_s * 1e3, 3),
                                     "predicted_speedup": round(el / (el_n + gather_s), 3),
                                     "ideal_speedup": round((1 + 2 * 64 * N_ITER) / (1 + 2 * kl_n * N_ITER), 3)}
                extra["p2_k64"]["predicted_strong_scaling"] = pred
            except Exception as ex:
                extra["p2_k64"]["predicted_strong_scaling"] = {"error": repr(ex)[:200]}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == "celeba_top5":
        cpu = cpu_baseline(cfg, w["params"], t)

    if rank == 0:
        if a.workload == "tloco_sd15":
            metric = "edit-directions/sec (top-5 CFG-combined PMP-Jacobian basis of the decoded 512^2 image w.r.t. the 4x64^2 latent, T-LOCO, Stable Diffusion v1.x architecture)"
            wl = ("T-LOCO latent space on the Stable Diffusion v1.x denoiser ARCHITECTURE (latent-diffusion UNetModel 320x(1,2,4,4), "
                  "SpatialTransformer blocks with 8 heads at 64/32/16 and in the middle block, 77x768 prompt states, 859.5 M "
                  "parameters, synthetic weights), mode null+(for-null) guidance 7.5 = 2 branches, + the SD autoencoder's decoder "
                  "geometry (64 -> 512), l_eye-sized mask on the decoded image, t=0.7T, 12 power iterations, probes sharded 5 per "
                  f"GPU; per probe-pass {2 * eng.unet_flops() / 1e12:.2f} TFLOP of denoiser + {w['dec'].unet_flops() / 1e12:.2f} TFLOP of decoder")
            scaling = "weak"
        elif a.workload == "tloco_sd":
            metric = "edit-directions/sec (top-5 CFG-combined PMP-Jacobian basis of the decoded 512^2 image w.r.t. the 4x64^2 latent, T-LOCO)"
            wl = ("T-LOCO latent space: SD-shaped stand-in denoiser (4x64x64, 320x(1,2,4,4), text cross-attention over 77x768 prompt states behind every attention block, mode null+(for-null) "
                  "guidance 7.5 = 2 branches) + the SD autoencoder's decoder geometry (49.5 M parameters, 64 -> 512), l_eye-sized "
                  "mask on the decoded image, t=0.7T, 12 power iterations, probes sharded 5 per GPU; per probe-pass "
                  f"{2 * eng.unet_flops() / 1e12:.2f} TFLOP of denoiser + {w['dec'].unet_flops() / 1e12:.2f} TFLOP of decoder")
            scaling = "weak"
        elif a.workload in ("tloco_if64", "tloco_if_i_m"):
            metric = "edit-directions/sec (top-5 CFG-combined PMP-Jacobian null-space basis @64^2, T-LOCO)"
            wl = ("T-LOCO pixel space 64x64, " + ("DeepFloyd IF-I-M stage-I architecture (synthetic weights)" if a.workload == "tloco_if_i_m"
                                                  else "IF-shaped stand-in conditional denoiser") + ", mode null+(for-null) guidance 7.5 (2 branches), "
                  "complement of an l_eye-sized mask, t=0.75T, 12 power iterations, probes sharded 5 per GPU")
            scaling = "weak"
        elif a.workload == "celeba_top5":
            metric = "edit-directions/sec (top-5 PMP-Jacobian SVD @256^2) + vT cos-sim vs ref"
            wl = ("CelebA-HQ DDPM 256x256 top-5 local basis (l_eye-sized mask, L=2400), t=0.6T, 12 power iterations, " +
                  ("one GPU" if world == 1 else
                   f"{world} REPLICAS: every rank solves the top-5 basis of its own image (5 probes per GPU, no data-path collective; "
                   "the one-image 5N-probe sharded solve is extra_workloads.celeba_top5N_one_image_sharded)" if replicas else
                   f"ONE image, {k} probes SHARDED over the {world} ranks (one all-gather per iteration): a top-{k} basis, smoke knob"))
            scaling = "weak"
        else:
            metric = "edit-directions/sec (rank-20 of 64 probes, FFHQ-P2 @256^2)"
            wl = ("FFHQ-P2 256x256 rank-20 basis from 64 probes (l_eye-sized mask, L=2400), t=0.6T, 12 power iterations, "
                  "64 probes sharded over the GPUs")
            scaling = "strong"
        out = {
            "metric": metric,
            "value": round(value, 4), "unit": "edit-directions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE_NOTE[a.precision], "data": "synthetic",
            "smoke": ({"iters": SMOKE_ITERS, "k_total": K_TOTAL} if (SMOKE_ITERS or K_TOTAL) else None),
            "config": {"workload": wl, "multi_gpu_form": ("single" if world == 1 else "replicas" if replicas else "sharded"),
                       "probes_total": k * (world if replicas else 1), "probes_per_gpu": k_local, "kept": keep * (world if replicas else 1),
                       "n_iter": int(n_iter),
                       "mask_L": int(w["mask"].sum().item()), "weights": "synthetic seed 0",
                       **({"streams": w["streams"]} if "streams" in w else {}),
                       "convergence_check": ("executed inside the timed region as in the reference flow (min_iter=10, max_iter=12: "
                                             "one row-wise allclose + 2-float readback at i = 11)" if a.workload in ("celeba_top5", "p2_k64")
                                             else "min_iter == max_iter == 12")},
            "singular_values": [round(float(v), 4) for v in s.tolist()[:5]],
            "parity": parity, "roofline": roofline,
            # whole image (a5 -> a11) at the headline's 12 iterations per solve and under the CLI's default stop rule
            # (LOCO_STOP_RULE=reference: five-probe solves run max_iter = 50, as the reference's own loop does)
            "image_total_s": (e2e or {}).get("image_total_s"),
            "image_total_reference_stop_rule_s": (e2e or {}).get("image_total_reference_stop_rule_s"),
            **({"cfg_branch_streams": "the CFG branches (one engine context per prompt) run side by side on two HIP streams "
                "(LOCO_CFG_STREAMS=0: one after the other); kernels of the two branches overlap, so this workload's per-kernel "
                "averages are durations under overlap, not isolated kernel times"} if w.get("branch_streams") else {}), "cpu_baseline": cpu, "e2e": e2e, "extra_workloads": extra or None,
            "clock": {"sclk_mhz_avg_over_timed_region": round(sclk, 1),
                      "method": "(d s_memtime / d s_memrealtime) x 100 MHz between two one-lane stamps around the timed steps "
                                "(loco_clock_stamp); rank 0's GPU; the chip's maximum is 2400 MHz"},
            "distinct_gpus": n_distinct_gpus,
            # the headline scales WEAKLY (N replicas: one top-5 solve per rank, each on its own image); the line the
            # north-star ">= 6x at 8 GPUs" is about is the STRONG-scaling 64-probe workload of the same run:
            "strong_scaling": ({"workload": "p2_k64 (FFHQ-P2 256^2, 64 probes sharded, keep 20)", "value": extra["p2_k64"]["value"],
                                "unit": "edit-directions/s", "n_gpus": world, "ms_per_step": extra["p2_k64"]["ms_per_step"],
                                "ideal_speedup_vs_1gpu": extra["p2_k64"]["ideal_speedup_vs_1gpu"],
                                "predicted_on_one_gpu": extra["p2_k64"].get("predicted_strong_scaling")}
                               if "p2_k64" in extra and "value" in extra.get("p2_k64", {}) else None),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
