#!/usr/bin/env python3
"""Interleaved A/B of a graph's OWN kernel compiled with different EDISON_JIT_DEFINE values, in one process.
usage (GPU box): tools/lab/ab_net_own.py [--utts N] [--model path.ednn] none EMM_PRIO=1 ...   ('none' = no extra define)"""
import argparse, ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["EDISON_NET_SPECIALIZE"] = "0"
os.environ["EDISON_NET_FORCE_GENERAL"] = "1"   # kws_conv would take its hand-written kernel otherwise
os.environ["EDISON_JIT_CACHE"] = "off"
import torch
from edison_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("defs", nargs="+"); ap.add_argument("--utts", type=int, default=262144); ap.add_argument("--model", default=_lib.DEFAULT_MODEL)
ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
L = _lib.lib()
_lib._share_torch_hip_runtime()

class V:
    def __init__(self, d):
        self.name = d
        self.h = ctypes.c_void_p()
        assert L.edison_init(0, ctypes.byref(self.h)) == 0
        assert L.edison_model_load(self.h, a.model.encode()) == 0
        assert L.edison_set_stream(self.h, ctypes.c_void_p(st.cuda_stream)) == 0
        if d == "general":
            pass
        else:
            if d == "none": os.environ.pop("EDISON_JIT_DEFINE", None)
            else: os.environ["EDISON_JIT_DEFINE"] = d
            r = L.edison_net_specialize(self.h)
            assert r == 0, (d, r, L.edison_last_error(self.h))
        self.t = []
    def alloc(self, n_in, n_out):
        self.lo = torch.zeros((a.utts, n_out), dtype=torch.int8, device=dev); self.so = torch.zeros_like(self.lo)
        self.am = torch.zeros((a.utts,), dtype=torch.int32, device=dev)
    def launch(self):
        r = L.edison_net_batch_dev(self.h, feat.data_ptr(), a.utts, self.lo.data_ptr(), self.so.data_ptr(), self.am.data_ptr())
        assert r == 0, (self.name, r, L.edison_last_error(self.h))

vs = [V(d) for d in a.defs]
from edison_amd.context import Context
c = Context(0); c.load_model(a.model); i_ = c.net_info(); n_in, n_out = i_["in_h"] * i_["in_w"] * i_["in_c"], i_["n_out"]; c.close()
feat = torch.randint(-128, 128, (a.utts, n_in), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
for v in vs: v.alloc(n_in, n_out); v.launch()
torch.cuda.synchronize()
for v in vs[1:]:
    print("%-16s outputs equal to %s: %s" % (v.name, vs[0].name, torch.equal(v.lo, vs[0].lo) and torch.equal(v.so, vs[0].so) and torch.equal(v.am, vs[0].am)))
for i in range(50): vs[0].launch()
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
    m = statistics.median(v.t)
    print("%-16s median %8.1f us  min %8.1f us  %7.1f M inputs/s  %+5.1f%% vs %s" % (v.name, m, min(v.t), a.utts / m, (base / m - 1) * 100, vs[0].name))
