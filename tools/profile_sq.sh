#!/bin/bash
# SQ counter passes (each its own rocprofv3 run, --pmc with --kernel-trace only) over a kernel-timing tool.
# usage (box): tools/profile_sq.sh <outdir-name> <python tool and args...>
set -e
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/sq$i" -- python3 "$@" > "$OUT/sq$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/sq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ed_mfcc" in r["Kernel_Name"] or "ed_cnn" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-24s %16.1f  (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
