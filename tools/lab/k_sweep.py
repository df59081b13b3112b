"""How long must a timed region be for the two queues to pay? W 5 + K steps of the 65 536-frame MFCC step, one queue against the context's
calibrated two queues (fork ... join inside the region, as bench.py times it), K = 10 ... 800, interleaved, medians of 15 regions each:
wall clock per step (what bench.py's `value` is made of) and HIP-event time per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from edison_amd import _lib  # noqa: E402
from edison_amd.context import Context  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.use_torch_stream(stream)
    nf = 65536
    g = torch.Generator(device="cpu").manual_seed(3)
    bufs = [(torch.randn((nf, 1024), generator=g) * 3000).clamp(-32768, 32767).to(torch.int16).to(dev) for _ in range(3)]
    outs = [torch.empty((nf, 13), dtype=torch.float32, device=dev) for _ in range(2)]
    cal = ctx.queues_calibrate(bufs[0], nf)
    print("calibration:", cal, flush=True)
    s_calls = [ctx.mfcc_call(bufs[i % 3], nf, 1024, _lib.MFCC_B, 13, out=outs[0]) for i in range(3)] if hasattr(ctx, "mfcc_call") else None
    q_calls = [ctx.mfcc_queue_call(i & 1, bufs[i % 3], nf, 1024, _lib.MFCC_B, 13, out=outs[i & 1]) for i in range(6)]

    def serial_step(i):
        if s_calls:
            s_calls[i % 3]()
        else:
            ctx.mfcc_t(bufs[i % 3], nf, 1024, _lib.MFCC_B, 13, out=outs[0])

    def region(two, k):
        step = (lambda i: q_calls[i % 6]()) if two else serial_step
        if two:
            ctx.queues_fork()
        for i in range(5):
            step(i)
        if two:
            ctx.queues_join()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if two:
            ctx.queues_fork()
        for i in range(k):
            step(5 + i)
        if two:
            ctx.queues_join()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6 / k
        return wall, e0.elapsed_time(e1) * 1e3 / k

    # settle
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for i in range(256):
            serial_step(i)
        torch.cuda.synchronize()
    print("%6s | %22s | %22s | two queues vs one (wall)" % ("K", "one queue wall / event", "two queues wall / event"))
    for k in (10, 20, 40, 80, 160, 400, 800):
        r1, r2 = [], []
        for rep in range(15):
            for two in ((0, 1) if rep % 2 == 0 else (1, 0)):
                (r2 if two else r1).append(region(bool(two), k))
        m = lambda v, j: sorted(x[j] for x in v)[len(v) // 2]
        print("%6d | %9.2f / %9.2f | %9.2f / %9.2f | %+.1f %%" % (k, m(r1, 0), m(r1, 1), m(r2, 0), m(r2, 1), (m(r1, 0) / m(r2, 0) - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
