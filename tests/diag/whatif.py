"""Diagnostic (by hand): build what-if variants of the tangent conv (MODE 3) that each DROP one ingredient of the stage
loop -- results are wrong by construction, only the launch time is meaningful -- to see what the matrix pipe waits for.
The variants are textual patches of a scratch copy of csrc/ (nothing of this is in the shipped sources); the libraries
land in tests/diag/lib/ (git-ignored, they travel to the GPU box with gpurun).

    python tests/diag/whatif.py build        # here (cross-compile)
    python tests/diag/whatif.py run          # on the GPU box: times 128->128 @256^2, 5 probes, both arithmetics
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "loco-edit_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "diag", "lib")

CONV = "                if (st_part >= 0) { Hs = Hnxt; stage_h(cclamp(chunk + 1), st_part); }\n"
LOAD = "                if (MIDLOAD && ld_part >= 0) prefetch_h(cclamp(ld_chunk), ld_part);\n"
WAIT = '                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");\n'
DMA = "                    dma_w(c2, r2, Wnx2);               // weights of the stage after next\n"
BAR = "        __builtin_amdgcn_s_barrier();\n"
EPI = "    // epilogue.  D[row = cout][col = pixel]"
MMA = "    auto mma_frag_head = [&](const Frag& f) { mma_one(f, 0, 0); };"

VARIANTS = {
    "v0_base": [],
    "v1_noconv": [(CONV, "")],
    "v2_nohalo": [(CONV, ""), (LOAD, ""), (WAIT, '                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n')],
    "v3_nohalo_nodma": [(CONV, ""), (LOAD, ""), (WAIT, '                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n'), (DMA, "")],
    "v4_nohalo_nodma_nobar": [(CONV, ""), (LOAD, ""), (WAIT, '                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n'),
                              (DMA, ""), (BAR, "")],
    "v5_noepi": [(EPI, "    if (acc[0][0][0] != 12345.f) return;\n" + EPI)],
    "v6_nomma": [(MMA, "    auto mma_one_off = [&](const Frag&, int, int) {};\n" + MMA.replace("mma_one(f, 0, 0)", "mma_one_off(f, 0, 0)")),
                 ("                if (i + j > 0) mma_one(f, i, j);", "                if (i + j > 0) mma_one_off(f, i, j);")],
    "v7_nodma_only": [(DMA, "")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [f for f in os.listdir(os.path.join(CSRC, "build")) if f.endswith(".o")]
    for name, patches in VARIANTS.items():
        work = f"/tmp/whatif_{name}"
        shutil.rmtree(work, ignore_errors=True)
        shutil.copytree(CSRC, work, ignore=shutil.ignore_patterns("build_*"))
        src = open(os.path.join(work, "conv_bf16_kernel.h")).read()
        for old, new in patches:
            assert src.count(old) >= 1, (name, old)
            src = src.replace(old, new)
        open(os.path.join(work, "conv_bf16_kernel.h"), "w").write(src)
        flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-mllvm",
                 "-amdgpu-sched-strategy=max-ilp"]
        procs = []
        for f in ("conv_bf16_inst_c", "conv_bf16_inst_c_f16"):
            # fix the relative include of the public header inside the scratch copy
            procs.append(subprocess.Popen(["hipcc"] + flags + ["-I", CSRC, "-c", os.path.join(work, f + ".hip"), "-o",
                                                                os.path.join(work, "build", f + ".o")], cwd=work))
        for p in procs:
            assert p.wait() == 0, name
        link = [os.path.join(work, "build", o) for o in objs]
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950"] + link + ["-o", os.path.join(OUT, f"libloco_{name}.so")])
        print("built", name, flush=True)


def run():
    for name in VARIANTS:
        lib = os.path.join(OUT, f"libloco_{name}.so")
        code = ("import os,sys; sys.path.insert(0, %r); import loco_edit_amd; from loco_edit_amd.config import CELEBA_DDPM, synth_params; "
                "from loco_edit_amd.hip import LocoEngine; e = LocoEngine(CELEBA_DDPM, max_batch=8); e.load_state_dict(synth_params(CELEBA_DDPM, 0));\n"
                "for p in ('bf16x3', 'f16'):\n"
                "    e.set_precision(p)\n"
                "    for (ci, co) in ((128, 128), (256, 128)):\n"
                "        ms = e.bench_conv(ci, co, 256, 256, 5, 3, 9, -1, 20)\n"
                "        print(%r, p, ci, co, f'{ms*1e3:.1f} us', flush=True)\n") % (ROOT, name)
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LOCO_HIP_LIB=lib))


if __name__ == "__main__":
    if len(sys.argv) > 2:      # subset of variants: python tests/diag/whatif.py build v0_base,v1_noconv
        VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[2].split(",")}
    {"build": build, "run": run}[sys.argv[1]]()
