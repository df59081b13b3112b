#!/bin/bash
# Which layer of ed_cnn_mfma_kernel owns the LDS bank conflicts, the LDS cycles, the vector instructions? Ablation builds of the kernel
# (tools/lab/mkvariant.py "cs1=-DEDM_SKIP=1@cnn_mfma_kernels.hip" ... : a layer's block is skipped, results wrong by design) under the
# SQ counters, each pass its own rocprofv3 run. A layer's share = the full kernel's count minus the count of the build without it.
# usage (box): tools/lab/cnn_lds_by_phase.sh <outdir under gpurun_out> prod cs1 cs2 cs4 cs8 cs31
set -e
OUT=gpurun_out/$1; shift; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/${v}_a -- python3 tools/lab/ab_cnn.py --rounds 1 --reps 2 $v > $OUT/${v}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${v}_b -- python3 tools/lab/ab_cnn.py --rounds 1 --reps 2 $v > $OUT/${v}_b.log 2>&1
done
python3 tools/lab/ab_cnn.py --rounds 6 "$@" 2>&1 | grep -v amdgpu.ids > $OUT/timing.txt
python3 - $OUT "$@" <<'PY'
import csv, glob, sys, collections
out, names = sys.argv[1], sys.argv[2:]
G = 65536.0  # groups of 4 utterances per launch
keys = ["SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES",
        "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES"]
tab = {}
for v in names:
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/%s_[ab]/*/*counter_collection.csv" % (out, v)):
        for r in csv.DictReader(open(f)):
            if "ed_cnn_mfma" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    tab[v] = {k: sum(x) / len(x) / G for k, x in agg.items()}
print("per group of 4 utterances (counter / 65 536); quad-cycle counters (ACTIVE_INST_*, WAIT_*, WAVE_CYCLES) as reported")
print("%-6s " % "build" + " ".join("%12s" % k.replace("SQ_", "")[:12] for k in keys))
for v in names:
    print("%-6s " % v + " ".join("%12.1f" % tab[v].get(k, float("nan")) for k in keys))
if names and names[0] in tab:
    full = tab[names[0]]
    print("share of the full kernel (%s) that disappears with the layer:" % names[0])
    for v in names[1:]:
        print("%-6s " % v + " ".join("%12.1f" % (full.get(k, 0) - tab[v].get(k, 0)) for k in keys))
print(open(out + "/timing.txt").read())
PY
