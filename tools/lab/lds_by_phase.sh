# LDS counters of the own kernel of kws_conv (ahead-of-time builds u0 = full, u1 = no epilogue, u2 = no k-loop, u3 = neither, u8 = no input
# stage, u16 = no softmax / outputs: timing-only ablations, results wrong by design): which phase owns the bank conflicts
set -e
OUT=gpurun_out/$1; mkdir -p $OUT; export TMPDIR=/tmp
for v in u0 u1 u2 u3 u8 u16; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/$v -- python3 tools/lab/ab_net.py --rounds 1 --reps 2 $v > $OUT/$v.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for v in ("u0", "u1", "u2", "u3", "u8", "u16"):
    agg = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + "/" + v + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ed_net_mfma" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(x) / len(x) / 262144 for k, x in agg.items()}
    print("%-4s per input: LDS instructions %6.1f  LDS active cycles %7.1f  bank-conflict cycles %7.1f  CU busy cycles %7.1f" % (v, c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_BUSY_CU_CYCLES", 0)))
PY
