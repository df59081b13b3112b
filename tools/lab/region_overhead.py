"""Where do the ~20-70 us go that a 20-step timed region's wall clock has over its HIP-event time? Stamps inside the region (host perf_counter):
t0 | e0.record | 20 launches issued | e1.record | synchronize returned -- for the sequence bench.py runs: first region of the process, then the
queue calibration, then three more regions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from edison_amd import _lib  # noqa: E402
from edison_amd.context import Context  # noqa: E402


def main():
    spin = "spin" in sys.argv[1:]
    pre = "pre" in sys.argv[1:]      # the events recorded once before the clock starts (what bench.py does since round 5)
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.use_torch_stream(stream)
    nf = 65536
    g = torch.Generator(device="cpu").manual_seed(3)
    bufs = [(torch.randn((nf, 1024), generator=g) * 3000).clamp(-32768, 32767).to(torch.int16).to(dev) for _ in range(3)]
    out = torch.empty((nf, 13), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step(i):
        ctx.mfcc_t(bufs[i % 3], nf, 1024, _lib.MFCC_B, 13, out=out)

    def region(name, k=20, w=5):
        for i in range(w):
            step(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if pre:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        t1 = time.perf_counter()
        for i in range(k):
            step(w + i)
        t2 = time.perf_counter()
        e1.record()
        t3 = time.perf_counter()
        if spin:
            while not e1.query():
                pass
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        ev = e0.elapsed_time(e1) * 1e3
        print("%-22s wall %7.1f us  events %7.1f us  | e0.record %5.1f  %d launches %6.1f (%.1f each)  e1.record %5.1f  wait %6.1f | wall - events %5.1f" % (
            name, (t4 - t0) * 1e6, ev, (t1 - t0) * 1e6, k, (t2 - t1) * 1e6, (t2 - t1) * 1e6 / k, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6 - ev), flush=True)

    region("first of the process")
    region("second")
    ctx.queues_calibrate(bufs[0], nf)
    torch.cuda.synchronize()
    region("after the calibration")
    region("next")
    region("next")
    time.sleep(0.2)
    region("after 0.2 s of sleep")
    region("next")


if __name__ == "__main__":
    main()
