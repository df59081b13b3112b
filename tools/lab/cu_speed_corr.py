#!/usr/bin/env python3
"""Are the per-CU speeds of the MFCC kernel stable from launch to launch? Runs the stamped build (ED2_STAMP=2) several times,
reads every workgroup's (= CU's) busy span (first wave into the loop .. last wave out) and prints the correlation of those
spans between launches, their spread, and what a static per-CU rebalancing could save (max vs mean of the mean spans)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from edison_amd import _lib
_lib._share_torch_hip_runtime()
L = ctypes.CDLL(os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_stamp.so"))
for n in ("edison_init", "edison_set_stream", "edison_mfcc_batch_dev"):
    fn = getattr(L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
L.ed_set_debug_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
h = ctypes.c_void_p(); assert L.edison_init(0, ctypes.byref(h)) == 0
st = torch.cuda.current_stream(); L.edison_set_stream(h, ctypes.c_void_p(st.cuda_stream))
NPH, WPB, frames = 19, 12, 65536
dbg = torch.zeros((256 * WPB, NPH), dtype=torch.int64, device=dev)
L.ed_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr())); torch.cuda.synchronize()
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
out = torch.empty((frames, 13), dtype=torch.float32, device=dev)
def launch(i): L.edison_mfcc_batch_dev(h, bufs[i % 3].data_ptr(), frames, 1024, _lib.MFCC_B, 13, out.data_ptr(), None, 1.0)
for i in range(3000): launch(i)
spans, ends = [], []
for rep in range(12):
    for i in range(50): launch(i)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    t0 = d[:, 14].min()
    e = (d[:, 16].reshape(256, WPB).max(axis=1) - t0) / 100
    s = (d[:, 15].reshape(256, WPB).min(axis=1) - t0) / 100
    spans.append(e - s); ends.append(e)
spans, ends = np.array(spans), np.array(ends)
c = np.corrcoef(spans)
print("busy span per CU: mean %.2f us, std over CUs (mean over launches) %.2f, std over launches (mean over CUs) %.2f" % (spans.mean(), spans.mean(axis=0).std(), spans.std(axis=0).mean()))
print("correlation of the per-CU spans between launches: mean off-diagonal %.2f (min %.2f)" % ((c.sum() - len(c)) / (len(c) ** 2 - len(c)), c.min()))
m = spans.mean(axis=0)
print("mean span by CU: min %.2f  median %.2f  max %.2f us;  launch end (max over CUs) mean %.2f us, mean over CUs of the end %.2f us" % (m.min(), np.median(m), m.max(), ends.max(axis=1).mean(), ends.mean()))
print("by blockIdx %% 8: " + " ".join("%.2f" % m[np.arange(256) % 8 == x].mean() for x in range(8)))
print("a static rebalancing by these means could take the launch end from %.2f to about %.2f us (mean of ends + residual noise %.2f)" % (ends.max(axis=1).mean(), ends.mean(), (spans - m).std()))
