#!/usr/bin/env python3
"""Run the stamped diagnostic build of ed_mfcc2_kernel (tools/lab/mkvariant.py stamp=-DED2_STAMP=1) and print where a
frame-pair iteration spends its cycles (shares, not lengths: the stamps forbid overlaps the real kernel has), the
cycles per frame per SIMD and the in-kernel clock (s_memtime / s_memrealtime)."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from edison_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--name", default="stamp")
ap.add_argument("--wpb", type=int, default=12)
a = ap.parse_args()
_lib._share_torch_hip_runtime()
L = ctypes.CDLL(os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % a.name))
for n in ("edison_init", "edison_set_stream", "edison_mfcc_batch_dev"):
    fn = getattr(L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
L.ed_set_debug_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
h = ctypes.c_void_p(); assert L.edison_init(0, ctypes.byref(h)) == 0
st = torch.cuda.current_stream(); L.edison_set_stream(h, ctypes.c_void_p(st.cuda_stream))
NPH = 19
dbg = torch.zeros((256 * 8 * 16, NPH), dtype=torch.int64, device=dev)
L.ed_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr())); torch.cuda.synchronize()
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
out = torch.empty((a.frames, 13), dtype=torch.float32, device=dev)
for i in range(3000 if a.frames <= 131072 else max(6, 200000000 // a.frames)): L.edison_mfcc_batch_dev(h, bufs[i % 3].data_ptr(), a.frames, 1024, _lib.MFCC_B, 13, out.data_ptr(), None, 1.0)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
nw = int((d[:, 12] > 0).sum())
d = d[:nw]
widx = np.arange(nw)
npairs_w = ((a.frames + 1) // 2 - widx + nw - 1) // nw
blk = widx // a.wpb
ll = (d[:, 16] - d[:, 15]) / 100
print("loop length us percentiles 5/25/50/75/95/100: %s" % " ".join("%.1f" % np.percentile(ll, q) for q in (5, 25, 50, 75, 95, 100)))
print("loop end us percentiles 5/25/50/75/95/100: %s" % " ".join("%.1f" % np.percentile((d[:, 16] - d[:, 14].min()) / 100, q) for q in (5, 25, 50, 75, 95, 100)))
print("busy fraction of the launch window (sum of loop lengths / (waves * (last end - first entry))): %.3f" % (ll.sum() / (nw * (d[:, 16].max() - d[:, 14].min()) / 100)))
for np_ in sorted(set(npairs_w)):
    m = npairs_w == np_
    print("  waves with %d pairs: %d, loop length median %.2f us (%.2f us per pair)" % (np_, m.sum(), np.median(ll[m]), np.median(ll[m]) / np_))
print("  by blockIdx %% 8 (XCD group): " + " ".join("%.1f" % np.median(ll[blk % 8 == x]) for x in range(8)))
print("  by wave in block: " + " ".join("%.1f" % np.median(ll[widx % a.wpb == x]) for x in range(a.wpb)))
names = ["unpack + next loads issued", "pass 1 + twiddles", "transpose 1 (VALU swaps)", "pass 2 + twiddles", "transpose 2 (LDS)", "pass 3", "split + |X| (bpermute, sqrt)",
         "spectrum -> LDS", "mel (LDS reads + fma)", "fold + log", "DCT", "store"]
tot = d[:, :12].sum(axis=1)
t_first = d[:, 14].min()
print("launch timeline (us after the first wave's entry): last entry %.2f | loop start median %.2f max %.2f | loop end median %.2f max %.2f | loop length median %.2f max %.2f" % (
    (d[:, 14].max() - t_first) / 100, (np.median(d[:, 15]) - t_first) / 100, (d[:, 15].max() - t_first) / 100, (np.median(d[:, 16]) - t_first) / 100,
    (d[:, 16].max() - t_first) / 100, np.median(d[:, 16] - d[:, 15]) / 100, (d[:, 16] - d[:, 15]).max() / 100))
# per workgroup (= per CU): when its last wave left the loop, when its first wave entered it
nb = nw // a.wpb
bend = (d[:nb * a.wpb, 16].reshape(nb, a.wpb).max(axis=1) - t_first) / 100
bstart = (d[:nb * a.wpb, 15].reshape(nb, a.wpb).min(axis=1) - t_first) / 100
print("per workgroup: last wave out of the loop, us, percentiles 0/5/25/50/75/95/100: %s" % " ".join("%.1f" % np.percentile(bend, q) for q in (0, 5, 25, 50, 75, 95, 100)))
print("per workgroup: first wave into the loop, us, percentiles 0/5/50/95/100: %s" % " ".join("%.1f" % np.percentile(bstart, q) for q in (0, 5, 50, 95, 100)))
print("per workgroup: busy span (last out - first in), us, percentiles 0/5/50/95/100: %s" % " ".join("%.1f" % np.percentile(bend - bstart, q) for q in (0, 5, 50, 95, 100)))
print("  corr(first in, last out) = %.2f; by blockIdx %% 8 median last-out: %s" % (np.corrcoef(bstart, bend)[0, 1], " ".join("%.1f" % np.median(bend[np.arange(nb) % 8 == x]) for x in range(8))))
print("  by (blockIdx // 8) %% 4 median last-out: %s" % " ".join("%.1f" % np.median(bend[(np.arange(nb) // 8) % 4 == x]) for x in range(4)))
pairs = a.frames / 2 / len(d)
print("waves %d, pairs per wave %.2f, stamped loop cycles per wave (median) %.0f, in-kernel clock %.3f GHz" % (
    len(d), pairs, np.median(d[:, 12]), np.median(d[:, 12] / d[:, 13]) * 0.1))
for i, n in enumerate(names):
    if tot.sum() == 0: break
    print("  %-32s %5.1f %%   %7.0f cycles per pair" % (n, 100 * d[:, i].sum() / tot.sum(), np.median(d[:, i]) / pairs))
print("  loop: %.0f cycles per pair per wave" % (np.median(d[:, 12]) / pairs))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
NREP = 300 if a.frames <= 131072 else 6
for i in range(NREP): L.edison_mfcc_batch_dev(h, bufs[i % 3].data_ptr(), a.frames, 1024, _lib.MFCC_B, 13, out.data_ptr(), None, 1.0)
e1.record(); torch.cuda.synchronize()
print("  launch-to-launch time of this build: %.2f us" % (e0.elapsed_time(e1) / NREP * 1e3))
