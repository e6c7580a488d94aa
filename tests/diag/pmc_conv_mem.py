"""Diagnostic (by hand): memory-side PMC counters of the dominant conv shape (tangent 128 -> 128 @256^2, 5 probes) through
`loco_bench_conv` (diag build).  Lists what `rocprofv3 -L` offers on the box, keeps the candidates that exist, and runs them in
passes of a few counters per hardware block (each pass = its own rocprofv3 run, program directly after `--`, no trace domains
beside --kernel-trace).  A pass that fails is re-run one counter at a time.

    python3 tests/diag/pmc_conv_mem.py <tag> [prec] [mode] [B]     ->  gpurun_out/<tag>/conv3x3_tan_pmc_mem_<prec>.csv
"""
import csv, glob, os, re, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_mem"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
mode = sys.argv[3] if len(sys.argv) > 3 else "3"
B = sys.argv[4] if len(sys.argv) > 4 else "5"
kern = sys.argv[5] if len(sys.argv) > 5 else "conv_"
out = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp", LOCO_HIP_LIB=os.path.join(ROOT, "loco-edit_amd", "libloco_hip_diag.so"))
if not os.path.exists(env["LOCO_HIP_LIB"]):      # (make -C loco-edit_amd/csrc diag; the library must travel with the snapshot)
    sys.exit("pmc_conv_mem.py: " + env["LOCO_HIP_LIB"] + " is missing: nothing would be collected")

lst = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp", env=env)
open(os.path.join(out, "rocprofv3_L.txt"), "w").write(lst.stdout + lst.stderr)
avail = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", lst.stdout + lst.stderr))

# candidates per block, in the order of interest
CAND = {
    "SQ": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM", "SQ_WAIT_INST_VMEM",
           "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_FLAT",
           "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INST_LEVEL_VMEM",
           "SQ_INST_LEVEL_LDS", "SQ_LEVEL_WAVES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
           "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_ANY",
           "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_IFETCH",
           "SQ_INSTS_LDS_DMA", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_LDS_DATA_FIFO_FULL",
           "SQ_LDS_CMD_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL"],
    "TA": ["TA_TA_BUSY_sum", "TA_BUSY_avr", "TA_BUSY_max", "TA_BUFFER_WAVEFRONTS_sum", "TA_FLAT_READ_WAVEFRONTS_sum",
           "TA_FLAT_WRITE_WAVEFRONTS_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum",
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum", "TA_FLAT_WAVEFRONTS_sum", "TA_TOTAL_WAVEFRONTS_sum", "TA_BUFFER_READ_WAVEFRONTS_sum",
           "TA_BUFFER_TOTAL_CYCLES_sum"],
    "TD": ["TD_TD_BUSY_sum", "TD_TC_STALL_sum", "TD_LOAD_WAVEFRONT_sum", "TD_STORE_WAVEFRONT_sum", "TD_SPI_STALL_sum",
           "TD_COALESCABLE_WAVEFRONT_sum"],
    "TCP": ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum",
            "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TOTAL_ACCESSES_sum",
            "TCP_TOTAL_READ_sum", "TCP_TOTAL_WRITE_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TD_TCP_STALL_CYCLES_sum",
            "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum",
            "TCP_TCC_NC_READ_REQ_sum", "TCP_TCC_UC_READ_REQ_sum", "TCP_TCC_CC_READ_REQ_sum", "TCP_TCC_RW_READ_REQ_sum",
            "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum"],
    "TCC": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum",
            "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RD_UNCACHED_32B_sum",
            "TCC_TAG_STALL_sum", "TCC_BUSY_sum", "TCC_BUSY_avr", "TCC_EA0_WRREQ_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum",
            "TCC_NC_REQ_sum", "TCC_STREAMING_REQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_CYCLE_sum"],
    "GRBM": ["GRBM_GUI_ACTIVE", "GRBM_COUNT", "GRBM_TA_BUSY", "GRBM_TC_BUSY", "GRBM_SPI_BUSY"],
}
PER_PASS = {"SQ": 8, "TA": 2, "TD": 2, "TCP": 4, "TCC": 4, "GRBM": 2}

def run_pass(name, counters):
    d = os.path.join(out, name)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--",
           "python3", os.path.join(ROOT, "tests", "diag", "conv_pmc.py"), mode, prec, B]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"] and "splitk" not in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        os.remove(f)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)
    if not acc:
        open(os.path.join(out, name + ".err"), "w").write(r.stdout[-3000:] + r.stderr[-3000:])
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

res = {}
missing = []
ONLY = os.environ.get('PMC_BLOCKS', '')
for blk, cands in CAND.items():
    if ONLY and blk not in ONLY.split(','):
        continue
    have = [c for c in cands if c in avail]
    missing += [c for c in cands if c not in avail]
    n = PER_PASS[blk]
    for i in range(0, len(have), n):
        grp = have[i:i + n]
        got = run_pass(f"{blk}_{i // n}", grp)
        if not got and len(grp) > 1:
            for c in grp:
                got.update(run_pass(f"{blk}_{c}", [c]))
        res.update(got)
        print(blk, i // n, {k: f"{v[0]:.6g}" for k, v in got.items()}, flush=True)

with open(os.path.join(out, f"conv3x3_tan_pmc_mem_{prec}.csv"), "w") as f:
    f.write("counter,mean_per_launch,launches\n")
    for k in sorted(res):
        f.write(f"{k},{res[k][0]:.6g},{res[k][1]}\n")
    f.write("# not offered by rocprofv3 -L on this box: " + " ".join(missing) + "\n")
print(open(os.path.join(out, f"conv3x3_tan_pmc_mem_{prec}.csv")).read())
if not res:
    sys.exit("pmc_conv_mem.py: no counter was collected (see the *.err files under " + out + ")")
