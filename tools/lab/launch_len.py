#!/usr/bin/env python3
"""Where does the 65 536-frame launch lose its 10 % against the 8.1 M-frame KWS launch? (round 3 review, item 3a)

Two candidate causes, separated here in ONE process, interleaved (guide rule 24):
  * a per-launch cost (prologue, CU drain) that a long launch amortises, or
  * the data: the KWS mix has 20 % quiet / silent utterances -> less power -> more clock.
Matrix: {config-2 data (noise + two-tone, bench.synth_frames), KWS mix (bench.synth_utterances)} x {65 536-frame PLAIN launches,
one PLAIN launch over all 8 126 464 frames}, plus the grouped launch (31 frames per utterance: the KWS step's instantiation)
on both data sets. All through the product library (edison_mfcc_batch_dev / edison_kws... are not used: only the MFCC kernel).

    tools/lab/launch_len.py [--rounds R] [--lib NAME]      ('prod' or a tools/lab/mkvariant.py name)
"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from edison_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--utts", type=int, default=262144)
ap.add_argument("--lib", default="prod")
ap.add_argument("--legs", default="cfg2:s,cfg2:l,kwsmix:s,kwsmix:l", help="subset, e.g. under rocprofv3 --kernel-trace --stats one leg per run")
a = ap.parse_args()
if a.lib != "prod":
    os.environ["EDISON_LIB"] = os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % a.lib)
    _lib.LIB_PATH = os.environ["EDISON_LIB"]
from edison_amd.context import Context

dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.use_torch_stream(st)
NU, NF_S = a.utts, 65536
NF_L = NU * 31
data = {}
if "cfg2" in a.legs:
    data["cfg2"] = bench.synth_frames(NF_L, 20, dev)                          # [NF_L, 1024] noise + two-tone everywhere
if "kwsmix" in a.legs:
    data["kwsmix"] = bench.synth_utterances(NU, 21, dev).reshape(NF_L, 1024)  # 80 % speech-level, 15 % 1 %-FS noise, 5 % silence
out_s = torch.empty((NF_S, 13), dtype=torch.float32, device=dev)
out_l = torch.empty((NF_L, 13), dtype=torch.float32, device=dev)
n_slices = NF_L // NF_S


def short(d):
    x = data[d]
    def f(i):
        s = (i * 37) % n_slices   # walks the whole 16.6 GB: every launch reads from HBM
        ctx.mfcc_t(x[s * NF_S:(s + 1) * NF_S], NF_S, 1024, _lib.MFCC_B, 13, out=out_s)
    return f, NF_S, 400


def long_plain(d):
    x = data[d]
    def f(i):
        ctx.mfcc_t(x, NF_L, 1024, _lib.MFCC_B, 13, out=out_l)
    return f, NF_L, 4


legs = []
for spec in a.legs.split(","):
    d, _, k = spec.partition(":")
    legs.append(("%s short(65536)" % d,) + short(d) if k == "s" else ("%s long(%d)" % (d, NF_L),) + long_plain(d))
times = {n: [] for n, *_ in legs}
# settle
import time
f0, t_end, i = legs[0][1], time.perf_counter() + 0.3, 0
while time.perf_counter() < t_end:
    f0(i); i += 1
    if i % 64 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
for r in range(a.rounds):
    for name, f, nfr, reps in (legs if r % 2 == 0 else legs[::-1]):
        for i in range(max(2, reps // 8)): f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): f(i)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / reps * 1e6 / nfr)   # ns per frame
print("lib = %s, %d rounds, interleaved; ns per frame incl. launch gaps (event time / launches / frames)" % (a.lib, a.rounds))
for name, f, nfr, reps in legs:
    med = statistics.median(times[name])
    print("%-28s median %.4f ns/frame  min %.4f  -> %7.2f us per 65 536 frames, %.3f of 8 TB/s" % (
        name, med, min(times[name]), med * 65536 / 1e3, 2100 / med / 8000))
