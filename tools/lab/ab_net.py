#!/usr/bin/env python3
"""Interleaved A/B timing of variants of the GENERAL network kernel on the kws_conv graph in one process (see ab_mfcc.py).
usage (GPU box): tools/lab/ab_net.py [--utts N] name1 name2 ...   ('prod' = the product library)"""
import argparse, ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EDISON_NET_SPECIALIZE", "0")  # a model load must not take a cached own kernel by itself: this tool times the general one
import torch
os.environ["EDISON_NET_FORCE_GENERAL"] = "1"
from edison_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+"); ap.add_argument("--utts", type=int, default=262144)
ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
feat = torch.randint(-128, 128, (a.utts, 403), generator=g, device=dev, dtype=torch.int32).to(torch.int8)

class V:
    def __init__(self, name):
        p = _lib.LIB_PATH if name == "prod" else os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % name)
        self.name, self.L = name, ctypes.CDLL(p)
        for n in ("edison_init", "edison_set_stream", "edison_net_batch_dev", "edison_model_load", "edison_last_error"):
            fn = getattr(self.L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
        self.h = ctypes.c_void_p()
        assert self.L.edison_init(0, ctypes.byref(self.h)) == 0
        assert self.L.edison_model_load(self.h, _lib.DEFAULT_MODEL.encode()) == 0
        assert self.L.edison_set_stream(self.h, ctypes.c_void_p(st.cuda_stream)) == 0
        self.lo = torch.zeros((a.utts, 10), dtype=torch.int8, device=dev); self.so = torch.zeros_like(self.lo)
        self.am = torch.zeros((a.utts,), dtype=torch.int32, device=dev); self.t = []
    def launch(self):
        r = self.L.edison_net_batch_dev(self.h, feat.data_ptr(), a.utts, self.lo.data_ptr(), self.so.data_ptr(), self.am.data_ptr())
        assert r == 0, (self.name, r, self.L.edison_last_error(self.h))

_lib._share_torch_hip_runtime()
vs = [V(n) for n in a.names]
for v in vs: v.launch()
torch.cuda.synchronize()
for v in vs[1:]:
    print("%-12s outputs equal to %s: logits %s softmax %s argmax %s" % (v.name, vs[0].name, torch.equal(v.lo, vs[0].lo), torch.equal(v.so, vs[0].so), torch.equal(v.am, vs[0].am)))
for i in range(100): vs[0].launch()
for r in range(a.rounds):
    for v in (vs if r % 2 == 0 else vs[::-1]):
        for i in range(3): v.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.reps): v.launch()
        e1.record(); torch.cuda.synchronize()
        v.t.append(e0.elapsed_time(e1) / a.reps * 1e3)
base = statistics.median(vs[0].t)
for v in vs:
    med = statistics.median(v.t)
    print("%-12s median %8.1f us  min %8.1f us  %7.1f Mutt/s  %5.2f int8 POP/s of real MACs   %+5.1f%% vs %s" % (
        v.name, med, min(v.t), a.utts / med, a.utts * 784752 * 2 / med / 1e9, (base / med - 1) * 100, vs[0].name))
