#!/usr/bin/env python3
"""Kernel-only timing of the variant-D (float32 ML-KWS) MFCC kernel: 512-sample frames at hop 256, 12 q7 features."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from edison_amd.context import Context
from edison_amd.mfcc.mfcc_f32 import MfccF32

dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.use_torch_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
n_samp = 64 * 1024 * 1024
audio = (torch.randn((n_samp,), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16)
m = MfccF32(ctx=ctx)
n = (n_samp - 512) // 256 + 1
out = torch.empty((n, 12), dtype=torch.int8, device=dev)
for _ in range(3): m.compute_t(audio, n, 256, out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.compute_t(audio, n, 256, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("f32 (variant D): %.3f ms / %d frames of 512 @ hop 256 = %.1f Mframes/s" % (ms, n, n / ms / 1e3))
