"""Diagnostic (by hand): the dual-probe conv tile against the 128 x 256 tile, bit for bit.  Runs one forward batch, one J V and
one U^T J of B samples / probes at 256 x 256 in a child process per setting of a 0 / 1 environment switch (default
LOCO_CONV_DUAL; also LOCO_TSTATS_PB) and compares the outputs (same products in the same order: expected difference exactly 0).
      python3 tests/diag/dual_check.py [B] [cfg] [ENV_NAME]"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import loco_edit_amd  # noqa
    import loco_edit_amd.hip as H
    from loco_edit_amd import config as C
    B = int(sys.argv[2]); cfg = getattr(C, sys.argv[3]); out = sys.argv[4]
    eng = H.LocoEngine(cfg, max_batch=max(B, 8))
    eng.load_state_dict(C.synth_params(cfg, 0))
    eng.set_precision("bf16x3")
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    n = cfg.n; R = int(round((n // 3) ** 0.5))
    x = torch.randn(B, 3, R, R, generator=g).to(dev)
    fwd = eng.unet_forward(x, 600.0).clone()
    mask = torch.zeros(3, R, R, dtype=torch.bool)
    mask[:, 100:140, 90:150] = True
    eng.pmp_primal(x[:1], 600.0, 0.05, mask=mask.to(dev))
    V = torch.randn(B, n, generator=g).to(dev)
    JV = eng.pmp_jvp(V).clone()
    U = torch.randn(B, JV.shape[1], generator=g).to(dev)
    UJ = eng.pmp_vjp(U).clone()
    torch.cuda.synchronize()
    torch.save({"fwd": fwd.cpu(), "JV": JV.cpu(), "UJ": UJ.cpu()}, out)
    sys.exit(0)
import torch
B = sys.argv[1] if len(sys.argv) > 1 else "5"
cfg = sys.argv[2] if len(sys.argv) > 2 else "CELEBA_DDPM"
ENVN = sys.argv[3] if len(sys.argv) > 3 else "LOCO_CONV_DUAL"      # the 0 / 1 switch under test
res = {}
SETTINGS = ("0", "1") if os.environ.get("DUAL_CHECK_SKIP_REPEAT") else ("0", "0b", "1")      # "0b": the default twice (run-to-run)
TMP = tempfile.mkdtemp(prefix="dual_check_")      # private to this run: concurrent runs must not read each other's files
for v in SETTINGS:
    out = os.path.join(TMP, f"{v}.pt")
    env = dict(os.environ, **{ENVN: (os.environ.get("DUAL_CHECK_ON", "1") if v[0] == "1" else "0")})      # DUAL_CHECK_ON: the value that means `on`
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B, cfg, out], check=True, env=env)
    res[v] = torch.load(out)
shutil.rmtree(TMP, ignore_errors=True)
for k in (res["0"] if "0b" in res else ()):
    print(f"{k}: run-to-run difference of the 128 x 256 tile itself: {(res['0'][k] - res['0b'][k]).abs().max().item():.3e}")
ok = True
RTOL = float(os.environ.get("DUAL_CHECK_RTOL", "0"))      # > 0: a kernel with another summation order (rel-L2 per output <= RTOL)
for k in res["0"]:
    a, b = res["0"][k], res["1"][k]
    d = (a - b).abs().max().item()
    rel = d / a.abs().max().item()
    l2 = ((a - b).norm() / a.norm()).item()
    nan = bool(torch.isnan(b).any())
    print(f"{k}: max|diff| {d:.3e} (rel {rel:.3e}, rel-L2 {l2:.3e}) nan={nan} |ref|max {a.abs().max().item():.3e}")
    if d != 0.0 and RTOL == 0:
        ne = (a != b).reshape(a.shape[0], -1)
        print(f"   differing elements per sample / probe row: {ne.sum(dim=1).tolist()} of {ne.shape[1]}")
    ok = ok and not nan and (d == 0.0 if RTOL == 0 else l2 <= RTOL)
print(ENVN, "0 vs 1:", ("PASS (bit-identical)" if RTOL == 0 else f"PASS (rel-L2 <= {RTOL:g})") if ok else "FAIL")
