#!/usr/bin/env python3
"""edison_queues_calibrate under hostile conditions: prints what it kept and the two-queue / serial ratio measured afterwards.
usage: [GPU_MAX_HW_QUEUES=1] calibrate_env.py [n_extra_streams]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
dev = torch.device("cuda", 0)
extra = [torch.cuda.Stream(priority=(-1 if i % 3 == 0 else 0)) for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
for s in extra:
    with torch.cuda.stream(s): torch.zeros(1, device=dev).add_(1)     # every stream has really been used
main = torch.cuda.Stream(); torch.cuda.set_stream(main)
ctx = Context(0); ctx.use_torch_stream(main)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 65536
bufs = [(torch.randn((N, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
outs = [torch.zeros((N, 13), dtype=torch.float32, device=dev) for _ in range(2)]
cal = ctx.queues_calibrate(bufs[0], N)
Q = [ctx.mfcc_queue_call(i & 1, bufs[i % 3], N, 1024, _lib.MFCC_B, 13, out=outs[i & 1]) for i in range(6)]
def run(two, reps=300):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    if two:
        ctx.queues_fork()
        for i in range(reps): Q[i % 6]()
        ctx.queues_join()
    else:
        for i in range(reps): ctx.mfcc_t(bufs[i % 3], N, 1024, _lib.MFCC_B, 13, out=outs[0])
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for _ in range(5): run(False)
t1, t2 = [], []
for r in range(6):
    t1.append(run(False)); t2.append(run(True))
print("GPU_MAX_HW_QUEUES=%s, %d other streams in the process: calibration kept %s (serial %.2f us, kept %.2f us); afterwards serial %.2f us, queue calls %.2f us (%+.1f %%)" % (
    os.environ.get("GPU_MAX_HW_QUEUES", "default"), len(extra), cal["pair"], cal["serial_us"], cal["best_us"], statistics.median(t1), statistics.median(t2),
    (statistics.median(t1) / statistics.median(t2) - 1) * 100))
