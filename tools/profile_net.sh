#!/bin/bash
# Counter passes over the network kernels on the kws_conv graph (kept off its hand-written kernel), 262 144 inputs per launch:
# the general matrix-core kernel (ed_net_mfma_kernel) and the graph's own (ed_net_mfma_spec, edison_net_specialize) in the same
# runs. Each pass is its own rocprofv3 run (--pmc with --kernel-trace only; the program directly after --).
# usage (box): tools/profile_net.sh <outdir-name under gpurun_out>
set -e
OUT=gpurun_out/$1
mkdir -p "$OUT"
export TMPDIR=/tmp EDISON_NET_FORCE_GENERAL=1 EDISON_JIT_CACHE=$PWD/$OUT/jit
i=0
for set in "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -- python3 tools/bench_net.py --specialize --reps 3 > "$OUT/pmc$i.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/bench_net.py --specialize --reps 10 > "$OUT/stats.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
for kern, label in (("ed_net_mfma_kernel", "general kernel"), ("ed_net_mfma_spec", "the graph's own kernel")):
    agg = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + "/pmc*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(sys.argv[1] + "/stats/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not agg: continue
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    n = 262144.0
    print("%s (%s), kws_conv graph, 262144 inputs per launch; per launch, mean of %d launches" % (kern, label, len(next(iter(agg.values())))))
    for k in sorted(c): print("   %-30s %16.1f   %10.1f per input" % (k, c[k], c[k] / n))
    if dur:
        d = sorted(dur)[len(dur) // 2]
        print("   kernel duration (median of %d, trace without counters): %.1f us = %.1f M inputs/s" % (len(dur), d / 1e3, n / d * 1e3))
        print("   matrix pipe busy: %.1f %% of the kernel at 2.4 GHz; LDS bank conflicts: %.1f %% of LDS active cycles" % (
            100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (d * 2.4), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
