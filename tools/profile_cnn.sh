#!/bin/bash
# Counter passes over ed_cnn_mfma_kernel at BASELINE's 262 144 utterances per launch: matrix-core busy cycles and MFMA
# instruction counts (north_star: int8 MFMA utilisation against gfx950 peak), then the SQ wait / active split.
# Each pass is its own rocprofv3 run (--pmc with --kernel-trace only; the program directly after --).
# usage (box): tools/profile_cnn.sh <outdir-name under gpurun_out>
set -e
OUT=gpurun_out/$1
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -- python3 tools/bench_mfcc.py --frames 4096 --reps 2 --utts 262144 > "$OUT/pmc$i.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/bench_mfcc.py --frames 4096 --reps 2 --utts 262144 > "$OUT/stats.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ed_cnn_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(sys.argv[1] + "/stats/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "ed_cnn_mfma" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
print("ed_cnn_mfma_kernel, 262144 utterances per launch; counters are per launch (mean of %d launches)" % len(next(iter(agg.values()))))
for k in sorted(c): print("   %-30s %18.1f" % (k, c[k]))
if dur:
    d = sorted(dur)[len(dur) // 2]
    print("kernel duration (median of %d, kernel trace without counters): %.1f us" % (len(dur), d / 1e3))
    n_mfma = c.get("SQ_INSTS_VALU_MFMA_I8", 0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    # per group of 4 utterances: 146 v_mfma_i32_32x32x32_i8 (32 768 MAC, 32 cycles) + 20 v_mfma_i32_16x16x64_i8 (16 384 MAC, 16 cycles)
    issued_macs = n_mfma / 166.0 * (146 * 32768 + 20 * 16384)
    print("MFMA instructions per utterance: %.2f (per group of 4: 146 x 32x32x32 + 20 x 16x16x64 = 166)" % (n_mfma / 262144))
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (guide: = 32 x N for a 32-cycle MFMA), summed over the SIMDs
    print("matrix pipe busy cycles per SIMD: %.0f; kernel = %.0f cycles at 2.4 GHz -> pipe busy %.1f %% of the kernel (upper bound of the clock: the busier the lower)" % (
        busy / 1024, d * 2.4, 100 * busy / 1024 / (d * 2.4)))
    print("issued int8 MAC rate: %.2f POP/s (2 op per MAC) = %.1f %% of the 5.03 POP/s dense int8 peak (1024 MAC/clk/SIMD x 1024 SIMDs x 2.4 GHz)" % (
        issued_macs * 2 / (d * 1e-9) / 1e15, 100 * issued_macs * 2 / (d * 1e-9) / 5.03e15))
    print("useful int8 MAC rate (784 752 MAC per utterance): %.2f POP/s = %.1f %% of peak" % (
        262144 * 784752 * 2 / (d * 1e-9) / 1e15, 100 * 262144 * 784752 * 2 / (d * 1e-9) / 5.03e15))
    import json
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (d * 1e-9) / 1e9 if d else 0
    json.dump(dict(utterances_per_launch=262144, kernel_us_unprofiled=d / 1e3, mfma_instructions=n_mfma, mfma_per_utterance=n_mfma / 262144,
                   SQ_VALU_MFMA_BUSY_CYCLES=busy, mfma_busy_cycles_per_simd=busy / 1024,
                   mfma_pipe_busy_frac_at_2p4GHz=busy / 1024 / (d * 2.4), SQ_VALU_MFMA_COEXEC_CYCLES=c.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
                   SQ_ACTIVE_INST_VALU_quadcycles=c.get("SQ_ACTIVE_INST_VALU"), SQ_LDS_IDX_ACTIVE=c.get("SQ_LDS_IDX_ACTIVE"),
                   SQ_LDS_BANK_CONFLICT=c.get("SQ_LDS_BANK_CONFLICT"), clock_GHz_from_GRBM_GUI_ACTIVE_profiled_pass=clk,
                   note="counters: mean per launch over separate rocprofv3 --pmc passes; duration: kernel trace pass without counters"),
              open(sys.argv[1] + "/cnn_counters.json", "w"), indent=1)
PY
