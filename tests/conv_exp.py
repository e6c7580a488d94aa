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
    for shape in ((128, 128, 256, 256), (256, 256, 64, 64)):
        ms = eng.bench_conv(shape[0], shape[1], shape[2], shape[3], 5, 3, 9, 5, 5)
        print(f"  {shape}: {ms*1e3:.1f} us")
else:
    import glob
    for lib in sorted(glob.glob(os.path.join(ROOT, "loco-edit_amd", "libloco_exp*.so"))):
        print(os.path.basename(lib), flush=True)
        subprocess.run([sys.executable, __file__, "child", lib])
