#!/usr/bin/env python3
"""Two queues from a Python host that is fast enough: prepared ctypes calls (arguments converted once) for the serial sequence and for
the library's own two queues (edison_queues_fork / edison_mfcc_batch_queue_dev / edison_queues_join), interleaved, medians; plus the host's
own time per call (issue time of 64 calls, nothing waited for). usage (GPU box): tools/lab/ab_queues_fast.py [--reps N] [--rounds R]"""
import argparse, ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--reps", type=int, default=400); ap.add_argument("--rounds", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(); torch.cuda.set_stream(main)
ctx = Context(0); ctx.use_torch_stream(main)
g = torch.Generator(device=dev); g.manual_seed(1)
bufs = [(torch.randn((a.frames, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
outs = [torch.zeros((a.frames, 13), dtype=torch.float32, device=dev) for _ in range(2)]
L = ctx._L
import time as _t
_t0 = _t.perf_counter(); cal = ctx.queues_calibrate(bufs[0], a.frames); print('calibration: %s in %.0f ms' % (cal, (_t.perf_counter() - _t0) * 1e3), flush=True)
def serial_call(b, o):
    args = (ctx._h, ctypes.c_void_p(b.data_ptr()), ctypes.c_int64(a.frames), ctypes.c_int64(1024), ctypes.c_int(_lib.MFCC_B), ctypes.c_int(13),
            ctypes.c_void_p(o.data_ptr()), None, ctypes.c_float(1.0))
    fn = L.edison_mfcc_batch_dev
    return lambda: fn(*args)
S = [serial_call(bufs[i % 3], outs[0]) for i in range(6)]
Q = [ctx.mfcc_queue_call(i % 2, bufs[i % 3], a.frames, 1024, _lib.MFCC_B, 13, out=outs[i % 2]) for i in range(6)]
side = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
def set_call(st):
    args = (ctx._h, ctypes.c_void_p(st.cuda_stream)); fn = L.edison_set_stream
    return lambda: fn(*args)
SET = [set_call(side[0]), set_call(side[1])]; SETMAIN = set_call(main)
T = [serial_call(bufs[i % 3], outs[i % 2]) for i in range(6)]
# raw HIP streams made here (not torch's pool): same priorities
hip = ctypes.CDLL("libamdhip64.so")
raw = [ctypes.c_void_p(), ctypes.c_void_p()]
assert hip.hipStreamCreateWithPriority(ctypes.byref(raw[0]), 1, 0) == 0 and hip.hipStreamCreateWithPriority(ctypes.byref(raw[1]), 1, -1) == 0
SETRAW = [(lambda a_=(ctx._h, raw[k]): L.edison_set_stream(*a_)) for k in range(2)]
class RawStream:
    def __init__(self, h): self.cuda_stream = h.value
rawt = [torch.cuda.ExternalStream(raw[k].value) for k in range(2)]
def run(mode, reps, timed=True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    if mode == 1:
        for i in range(reps): S[i % 6]()
    elif mode in (3, 4):   # 3: torch pool streams of priority 0 / -1 through edison_set_stream; 4: streams created here with the same priorities
        ss, st = (SET, side) if mode == 3 else (SETRAW, rawt)
        for s_ in st: s_.wait_event(e0)
        for i in range(reps): ss[i % 2](); T[i % 6]()
        for s_ in st:
            d = torch.cuda.Event(); d.record(s_); main.wait_event(d)
        SETMAIN()
    else:
        ctx.queues_fork()
        for i in range(reps): Q[i % 6]()
        ctx.queues_join()
    e1.record(main)
    if timed:
        torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
# correctness: queue results == serial results
S[2](); torch.cuda.synchronize(); r2 = outs[0].clone(); S[3](); torch.cuda.synchronize(); r3 = outs[0].clone()
ctx.queues_fork(); [Q[i]() for i in range(4)]; ctx.queues_join(); torch.cuda.synchronize()
print("two queues bit-identical to serial:", torch.equal(outs[0], r2) and torch.equal(outs[1], r3))
# the host's own time per call
torch.cuda.synchronize(); t0 = time.perf_counter(); [S[i % 6]() for i in range(64)]; t1 = time.perf_counter(); torch.cuda.synchronize()
ctx.queues_fork(); t2 = time.perf_counter(); [Q[i % 6]() for i in range(64)]; t3 = time.perf_counter(); ctx.queues_join(); torch.cuda.synchronize()
print("host time per call: serial %.1f us, queue call %.1f us" % ((t1 - t0) / 64 * 1e6, (t3 - t2) / 64 * 1e6))
for i in range(8): run(1, 400)
t = {1: [], 2: [], 3: [], 4: []}
for r in range(a.rounds):
    for m in ((1, 2, 3, 4) if r % 2 == 0 else (4, 3, 2, 1)):
        run(m, 60, False); t[m].append(run(m, a.reps))
b = statistics.median(t[1])
for m in (1, 2, 3, 4):
    med = statistics.median(t[m])
    print("mode %d (1 serial, 2 library queues, 3 torch pool streams prio 0/-1, 4 streams created here prio 0/-1): median %.2f us  min %.2f us  %.4f of 8 TB/s  %+.1f %% vs serial" % (m, med, min(t[m]), 2100 * a.frames / med / 1e6 / 8, (b / med - 1) * 100))
