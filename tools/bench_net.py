#!/usr/bin/env python3
"""Kernel timing of the general network path (cnn_net_kernels.hip) on the GPU box.
usage: bench_net.py [--header tests/golden/alt_models/kws_small.h] [--n 262144] [--reps 10]
Without --header the shipped kws_conv blob is loaded and BOTH its kernels are timed: the matrix-core one
(edison_net_batch_dev) and the general one (edison_net_layers_dev, which also writes every layer's output)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("EDISON_NET_SPECIALIZE", "0")  # a model load must not take a cached own kernel by itself: this tool times the general one
import torch
from edison_amd.context import Context, _t_ptr

ap = argparse.ArgumentParser()
ap.add_argument("--header", default=None)
ap.add_argument("--n", type=int, default=262144)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--specialize", action="store_true", help="time the general kernel, then the graph's own (edison_net_specialize)")
a = ap.parse_args()

dev = torch.device("cuda", 0)
ctx = Context(0) if a.header is None else Context(0, model_path=None)
if a.header:
    ctx.load_weights_h(a.header)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx.use_torch_stream(st)
info = ctx.net_info()
n_in = info["in_h"] * info["in_w"] * info["in_c"]
x = torch.randint(-128, 128, (a.n, n_in), dtype=torch.int8, device=dev)
logits = torch.empty((a.n, info["n_out"]), dtype=torch.int8, device=dev)
am = torch.empty((a.n,), dtype=torch.int32, device=dev)
acts = torch.empty((a.n, info["acts_bytes"]), dtype=torch.int8, device=dev)
macs = 0
h, w, c = info["in_h"], info["in_w"], info["in_c"]


def timed(fn, tag):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("%-34s %.3f ms / %d inputs = %.2f M inputs/s" % (tag, ms, a.n, a.n / ms / 1e3))


L = ctx._L
timed(lambda: ctx._check(L.edison_net_batch_dev(ctx._h, _t_ptr(x), a.n, _t_ptr(logits), None, _t_ptr(am))),
      "edison_net_batch_dev (%s)" % {0: "layer-by-layer VALU kernel", 1: "kws_conv matrix-core kernel", 2: "general matrix-core kernel"}[2 if os.environ.get("EDISON_NET_FORCE_GENERAL") == "1" and info["accelerated"] == 1 else info["accelerated"]])
if a.specialize:
    import time
    t0 = time.time()
    st_ = ctx.net_specialize()
    print("edison_net_specialize: %s in %.2f s" % ({1: "compiled by hipcc", 2: "from the cache", 3: "compiled by hipRTC"}[st_], time.time() - t0))
    l2, a2 = torch.empty_like(logits), torch.empty_like(am)
    timed(lambda: ctx._check(L.edison_net_batch_dev(ctx._h, _t_ptr(x), a.n, _t_ptr(l2), None, _t_ptr(a2))), "edison_net_batch_dev (the graph's own kernel)")
    print("own kernel == general kernel: logits %s argmax %s" % (torch.equal(l2, logits), torch.equal(a2, am)))
timed(lambda: ctx._check(L.edison_net_layers_dev(ctx._h, _t_ptr(x), a.n, _t_ptr(acts))), "edison_net_layers_dev (general + dumps)")
