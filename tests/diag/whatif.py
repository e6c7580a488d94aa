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

CONV = "                if (cv1 >= 0) { Hs = Hnxt; stage_h(cclamp(chunk + 1), cv1); }\n"
CONV2 = "                if (cv2 >= 0) { Hs = Hnxt; stage_h(cclamp(chunk + 1), cv2); }\n"
LOAD = "                if (ld2 >= 0) prefetch_h(cclamp(chunk + 1), ld2);\n"
LOAD0 = "                if (ld0 >= 0) prefetch_h(cclamp(chunk + 2), ld0);\n"
WAIT = '                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");\n'
DMA = "                    dma_w(c2, r2, Wnx2);               // weights of the stage after next\n"
BAR = "        __builtin_amdgcn_s_barrier();\n"
EPI = "    // epilogue.  D[row = cout][col = pixel]"
MMA = "    auto mma_frag_head = [&](const Frag& f) { mma_one(f, 0, 0); };"

# cycle stamps (s_memtime) of one workgroup: per kernel row, cycles from the stage start to (1) the end of tap 0,
# (2) the end of tap 1, (3) the end of tap 2, (4) the end of the vmcnt wait, (5) the end of the barrier
ST_DECL = ("    const int nch = cend - cbeg;\n",
           "    const int nch = cend - cbeg;\n    unsigned long long tacc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long tprev = 0;\n"
           "#define STAMP0 { tprev = __builtin_readcyclecounter(); }\n"
           "#define STAMP(i) { unsigned long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tprev; tprev = tn; }\n")
ST_OUT = ("    // epilogue.  D[row = cout][col = pixel]",
          "    if (blockIdx.x == 300 && (tid & 63) == 0) { for (int i = 0; i < 16; ++i) a.partial[wave * 16 + i] = (float)tacc[i]; }\n"
          "    unsigned long long tep = __builtin_readcyclecounter();\n"
          "    // epilogue.  D[row = cout][col = pixel]")
ST_EPI = ("\n// ---------------------------------------------------------------------------\n// Kernel entry points",
          "\n// Kernel entry points")
STAMPS = [
    ST_DECL, ST_OUT,
    ("                Ws = Wcur; Hs = Hcur;\n                mma_frag_head(fa);", "                STAMP0\n                Ws = Wcur; Hs = Hcur;\n                mma_frag_head(fa);"),
    ("                mma_frag_tail(fa);\n                __builtin_amdgcn_sched_barrier(0);\n                mma_frag_head(fb);",
     "                mma_frag_tail(fa);\n                __builtin_amdgcn_sched_barrier(0);\n                STAMP(row * 5 + 0)\n                mma_frag_head(fb);"),
    ("                mma_frag_tail(fb);\n                __builtin_amdgcn_sched_barrier(0);\n                mma_frag_head(fa);",
     "                mma_frag_tail(fb);\n                __builtin_amdgcn_sched_barrier(0);\n                STAMP(row * 5 + 1)\n                mma_frag_head(fa);"),
    ("                // the LDS-DMA of this stage (older than the part loads issued in it) must have landed before the barrier\n",
     "                STAMP(row * 5 + 2)\n"),
    ("                stage_end();\n            }\n        };\n        for (int ci = 0; ci < nch; ci += 2) {",
     "                STAMP(row * 5 + 3)\n                stage_end();\n                STAMP(row * 5 + 4)\n            }\n        };\n        for (int ci = 0; ci < nch; ci += 2) {"),
]

NT = "__builtin_nontemporal_store(v, &ob[(long)co * out_plane + pix]);"
VARIANTS = {
    "v8_stamps": STAMPS,
    "v9_store_sc1": [(NT, "__hip_atomic_store(&ob[(long)co * out_plane + pix], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")],
    "v9_store_plain": [(NT, "ob[(long)co * out_plane + pix] = v;")],
    "v0_base": [],
    "v1_noconv": [(CONV, ""), (CONV2, "")],
    "v2_nohalo": [(CONV, ""), (CONV2, ""), (LOAD, ""), (LOAD0, ""), (WAIT, '                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n')],
    "v3_nohalo_nodma": [(CONV, ""), (CONV2, ""), (LOAD, ""), (LOAD0, ""), (WAIT, '                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n'), (DMA, "")],
    "v4_nohalo_nodma_nobar": [(CONV, ""), (CONV2, ""), (LOAD, ""), (LOAD0, ""), (WAIT, '                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n'),
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


def rev(revs):
    """Whole-library builds of earlier commits (same-box A/B of kernel revisions): libloco_rev_<name>.so"""
    os.makedirs(OUT, exist_ok=True)
    for r in revs:
        work = f"/tmp/whatif_rev_{r.replace('~', '_').replace('^', '_')}"
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        subprocess.check_call(f"git -C {ROOT} archive {r} loco-edit_amd/csrc include | tar -x -C {work}", shell=True)
        subprocess.check_call(["make", "-C", os.path.join(work, "loco-edit_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(work, "loco-edit_amd", "libloco_hip.so"), os.path.join(OUT, f"libloco_rev_{r.replace('~', '_').replace('^', '_')}.so"))
        print("built rev", r, flush=True)


def stamps():
    """Run the stamped build on the 128->128 tangent conv and print the per-row phase cycles of waves 0 and 4."""
    lib = os.path.join(OUT, "libloco_v8_stamps.so")
    code = ("import os,sys; sys.path.insert(0, %r); import loco_edit_amd, torch; from loco_edit_amd.config import CELEBA_DDPM, synth_params; "
            "from loco_edit_amd.hip import LocoEngine; e = LocoEngine(CELEBA_DDPM, max_batch=8); e.load_state_dict(synth_params(CELEBA_DDPM, 0));\n"
            "for p in ('bf16x3', 'f16'):\n"
            "    e.set_precision(p)\n"
            "    ms = e.bench_conv(128, 128, 256, 256, 5, 3, 9, -1, 3)\n"
            "    torch.cuda.synchronize(); w = e.debug_tensor('workspace', 128).cpu().view(8, 16)\n"
            "    print(p, f'{ms*1e3:.1f} us; cycles per chunk-row phase (8 chunks summed): rows x [tap0, tap1, tap2, vmcnt, barrier]')\n"
            "    for wv in (0, 4):\n"
            "        print('  wave', wv, [[int(w[wv, r*5+i]) for i in range(5)] for r in range(3)], 'stage-loop total', int(w[wv,:15].sum()))\n") % (ROOT,)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LOCO_HIP_LIB=lib))


def run():
    names = sorted(f[len("libloco_"):-3] for f in os.listdir(OUT) if f.endswith(".so"))
    for name in names:
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
    if len(sys.argv) > 2 and sys.argv[1] != "rev":      # subset of variants: python tests/diag/whatif.py build v0_base,v1_noconv
        VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[2].split(",")}
    if sys.argv[1] == "rev":
        rev(sys.argv[2].split(","))
    elif sys.argv[1] == "stamps":
        stamps()
    else:
        {"build": build, "run": run}[sys.argv[1]]()
