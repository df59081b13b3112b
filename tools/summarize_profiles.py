#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small, committed files under profiles/.

usage: summarize_profiles.py ROUND STATS_DIR FETCH_DIR WRITE_DIR [SQ_DIR ...]
  STATS_DIR  : rocprofv3 --kernel-trace --stats of `python3 bench.py ...`
  FETCH_DIR  : rocprofv3 --pmc FETCH_SIZE --kernel-trace of the same command   (separate pass)
  WRITE_DIR  : rocprofv3 --pmc WRITE_SIZE --kernel-trace of the same command   (separate pass)
HBM bytes are corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) is doubled on gfx950 for wide
coalesced streaming reads; WRITE_SIZE (KB) is taken as is.
"""
import collections, csv, glob, json, os, sys

def one(d, pat):
    f = glob.glob(os.path.join(d, "*", pat)) + glob.glob(os.path.join(d, pat))
    return f[0] if f else None

def main():
    rnd, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    sq_dirs = [a for a in sys.argv[5:] if not a.startswith("steady=")]
    steady = [a[7:] for a in sys.argv[5:] if a.startswith("steady=")]
    out = {"round": rnd, "kernels": {}, "hbm": {}, "sq_per_launch": {}}
    # steady=DIR: kernel trace of the default bench command (thousands of launches): mean duration of every 100
    # consecutive dispatches in launch order -- shows the power-management transient and the settled value
    for d in steady:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(one(d, "*kernel_trace.csv"))):
            if "ed_mfcc" in r["Kernel_Name"] or "ed_cnn" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        out["steady_blocks_of_100_ns"] = {}
        for k, v in per.items():
            v.sort()
            dur = [x[1] for x in v]
            out["steady_blocks_of_100_ns"][k] = [round(sum(dur[i:i + 100]) / len(dur[i:i + 100]), 1) for i in range(0, len(dur), 100)]
    for r in csv.DictReader(open(one(stats_dir, "*kernel_stats.csv"))):
        if "ed_mfcc" in r["Name"] or "ed_cnn" in r["Name"]:
            out["kernels"][r["Name"].split("(")[0].replace("void ", "")] = dict(
                calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=int(r["MinNs"]), max_ns=int(r["MaxNs"]), pct=float(r["Percentage"]))
    # per-dispatch durations of our kernels, grouped by launch size class (short = MFCC-only, long = KWS)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(one(stats_dir, "*kernel_trace.csv"))):
        if ("ed_mfcc" in r["Kernel_Name"] or "ed_cnn" in r["Kernel_Name"]):
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out["dispatch_ns"] = {k: sorted(v) for k, v in per.items()}
    for name, d, mult in (("FETCH_SIZE", fetch_dir, 2.0), ("WRITE_SIZE", write_dir, 1.0)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(one(d, "*counter_collection.csv"))):
            if ("ed_mfcc" in r["Kernel_Name"] or "ed_cnn" in r["Kernel_Name"]) and r["Counter_Name"] == name:
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                k = r["Kernel_Name"].split("(")[0].replace("void ", "") + (":long" if dur > 1e6 else ":short")
                agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out["hbm"].setdefault(k, {})[name + "_KB_raw"] = sum(v) / len(v)
            out["hbm"][k][name + "_bytes_corrected"] = sum(v) / len(v) * 1024 * mult
    for d in sq_dirs:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(one(d, "*counter_collection.csv"))):
            if ("ed_mfcc" in r["Kernel_Name"] or "ed_cnn" in r["Kernel_Name"]):
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out["sq_per_launch"].setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
    os.makedirs("profiles", exist_ok=True)
    path = os.path.join("profiles", "r%s_rocprof_summary.json" % rnd)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)

if __name__ == "__main__":
    main()
