#!/usr/bin/env python3
"""Per-CU timeline of 65 536-frame MFCC launches that are in flight TOGETHER (tools/lab/mkvariant.py stamp2=-DED2_STAMP=2).

Launches rotate over 4 output slots and over q HIP streams (different priorities = different hardware queues); every wave stamps its
entry, loop start and loop end in real time (s_memrealtime, 100 MHz) together with the CU it ran on (HW_ID, XCC_ID) into the slot of
its launch. After N launches the last four launches are read: per CU, how long a workgroup's waves were in their loops, and how long
the CU waited between the last wave of one launch's workgroup leaving and the first wave of the next launch's workgroup entering.
usage (GPU box): tools/lab/stamp_overlap.py [--queues 1,2] [--launches 400]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from edison_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--name", default="stamp2")
ap.add_argument("--wpb", type=int, default=12); ap.add_argument("--queues", default="1,2"); ap.add_argument("--launches", type=int, default=400)
a = ap.parse_args()
_lib._share_torch_hip_runtime()
L = ctypes.CDLL(os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % a.name))
for n in ("edison_init", "edison_set_stream", "edison_mfcc_batch_dev", "edison_queues_calibrate", "edison_queues_fork", "edison_queues_join", "edison_mfcc_batch_queue_dev"):
    fn = getattr(L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
L.ed_set_debug_buffer.argtypes = [ctypes.c_void_p]
L.ed_set_debug_slots.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
dev = torch.device("cuda", 0)
h = ctypes.c_void_p(); assert L.edison_init(0, ctypes.byref(h)) == 0
main = torch.cuda.Stream(); torch.cuda.set_stream(main)
side = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
NPH, NSLOT, SLOTW = 19, 4, 256 * 16
dbg = torch.zeros((NSLOT * SLOTW, NPH), dtype=torch.int64, device=dev)
outs = torch.zeros((NSLOT, a.frames, 13), dtype=torch.float32, device=dev)
L.ed_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
L.ed_set_debug_slots(ctypes.c_void_p(outs.data_ptr()), a.frames * 13 * 4, NSLOT, SLOTW); torch.cuda.synchronize()
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]

# prepared calls: the host must stay far below the GPU's ~45 us per batch (the first version of this tool converted its arguments per
# call, was host-bound, and "found" that the second queue's launch starts late)
L.edison_set_stream(h, ctypes.c_void_p(main.cuda_stream))
C1 = [(h, ctypes.c_void_p(bufs[i % 3].data_ptr()), ctypes.c_int64(a.frames), ctypes.c_int64(1024), ctypes.c_int(_lib.MFCC_B), ctypes.c_int(13),
       ctypes.c_void_p(outs[i % NSLOT].data_ptr()), None, ctypes.c_float(1.0)) for i in range(12)]
C2 = [(h, ctypes.c_int(i & 1)) + C1[i][1:] for i in range(12)]
su, bu, pk = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
assert L.edison_queues_calibrate(h, bufs[0].data_ptr(), a.frames, 1024, _lib.MFCC_B, ctypes.byref(su), ctypes.byref(bu), ctypes.byref(pk)) == 0
print("calibration (stamped build): serial %.2f us, kept %.2f us, pair %d" % (su.value, bu.value, pk.value), flush=True)

def cu_key(d):
    hw, xcc = d[:, 17].astype(np.int64), d[:, 18].astype(np.int64)
    return ((xcc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)

for q in [int(x) for x in a.queues.split(",")]:
    for warm in (True, False):
        dbg.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if q == 1:
            for i in range(a.launches): L.edison_mfcc_batch_dev(*C1[i % 12])
        else:
            L.edison_queues_fork(h)
            for i in range(a.launches): L.edison_mfcc_batch_queue_dev(*C2[i % 12])
            L.edison_queues_join(h)
        e1.record(main); torch.cuda.synchronize()
    print("==== %d queue(s): %.2f us per launch over %d launches (stamped build)" % (q, e0.elapsed_time(e1) / a.launches * 1e3, a.launches))
    d = dbg.cpu().numpy().astype(np.float64).reshape(NSLOT, SLOTW, NPH)
    order = sorted(range(NSLOT), key=lambda s: d[s][d[s][:, 14] > 0][:, 14].min())   # slots by first entry = launch order
    recs = []   # per workgroup: (launch rank, cu, first entry, first loop start, last loop end, sum of loop lengths, cycles)
    for rank, s in enumerate(order):
        x = d[s]; nw = int((x[:, 12] > 0).sum()); x = x[:nw]; nb = nw // a.wpb
        x = x[:nb * a.wpb].reshape(nb, a.wpb, NPH)
        keys = cu_key(x[:, 0, :])
        for b in range(nb):
            recs.append((rank, int(keys[b]), x[b, :, 14].min(), x[b, :, 15].min(), x[b, :, 16].max(), (x[b, :, 16] - x[b, :, 15]).sum(), np.median(x[b, :, 12] / x[b, :, 13])))
    r = np.array(recs)
    t0 = r[:, 2].min()
    for rank in range(NSLOT):
        m = r[:, 0] == rank
        print("  launch %d: first entry %.1f us, workgroup entries p5/p50/p95 %.1f/%.1f/%.1f, last-wave-out p5/p50/p95/max %.1f/%.1f/%.1f/%.1f, distinct CUs %d, clock %.3f GHz" % (
            rank, (r[m, 2].min() - t0) / 100, *[(np.percentile(r[m, 2], p) - t0) / 100 for p in (5, 50, 95)],
            *[(np.percentile(r[m, 4], p) - t0) / 100 for p in (5, 50, 95, 100)], len(set(r[m, 1])), np.median(r[m, 6]) * 0.1))
    span = (r[:, 4] - r[:, 2]) / 100
    busy = r[:, 5] / 100 / a.wpb
    print("  workgroup: entry -> last wave out p5/p50/p95 %.1f/%.1f/%.1f us; mean loop time per wave p50 %.1f us; entry -> first loop start p50 %.2f us" % (
        *[np.percentile(span, p) for p in (5, 50, 95)], np.median(busy), np.median((r[:, 3] - r[:, 2]) / 100)))
    gaps = []
    for cu in set(r[:, 1]):
        w = r[r[:, 1] == cu]; w = w[np.argsort(w[:, 2])]
        for i in range(len(w) - 1):
            gaps.append((w[i + 1, 2] - w[i, 4]) / 100)
    gaps = np.array(gaps)
    print("  per CU, next workgroup's first entry - this workgroup's last wave out: n %d, p5/p25/p50/p75/p95 %.2f/%.2f/%.2f/%.2f/%.2f us (negative: two workgroups on one CU at once)" % (
        len(gaps), *[np.percentile(gaps, p) for p in (5, 25, 50, 75, 95)]))
    window = (r[:, 4].max() - t0) / 100
    print("  window of the four launches %.1f us; CU-time in loops / (CUs x window) = %.3f" % (window, r[:, 5].sum() / 100 / a.wpb / (len(set(r[:, 1])) * window)))
