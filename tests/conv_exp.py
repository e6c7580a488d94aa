"""Diagnostic (by hand): time what-if variants of the dominant conv (libloco_exp<N>.so, see LOCO_EXP in conv_bf16.hip)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import loco_edit_amd  # noqa
    import loco_edit_amd.hip as H
    H._LIB_PATH = sys.argv[2]
    from loco_edit_amd.config import CELEBA_DDPM, synth_params
    eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
    eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
    eng.set_precision("bf16x3")
    for mode in (3, 4, 0):
        row = []
        for shape in ((128, 128, 256, 256), (256, 128, 256, 256), (128, 128, 128, 128), (256, 256, 64, 64), (512, 512, 16, 16)):
            ms = eng.bench_conv(shape[0], shape[1], shape[2], shape[3], 5, mode, 9, 5, 5)
            row.append(f"{shape[0]}->{shape[1]}@{shape[2]}: {ms*1e3:6.1f}")
        print(f"  mode {mode}: " + "  ".join(row) + "  (us)")
else:
    import glob
    for lib in sorted(glob.glob(os.path.join(ROOT, "loco-edit_amd", "libloco_exp*.so"))):
        print(os.path.basename(lib), flush=True)
        subprocess.run([sys.executable, __file__, "child", lib])
