"""Does replaying the serial W + K launch sequence as ONE hipGraph shorten the gaps between the launches? 20 and 200 launches of the 65 536-frame MFCC step,
captured on the bench's stream (torch.cuda.CUDAGraph around the library's launches), replayed, against the same launches issued one by one; interleaved, medians."""
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
    out = torch.empty((nf, 13), dtype=torch.float32, device=dev)

    def step(i):
        ctx.mfcc_t(bufs[i % 3], nf, 1024, _lib.MFCC_B, 13, out=out)
    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    for k in (20, 200):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for i in range(k):
                step(i)
        torch.cuda.synchronize()
        ref = out.clone()
        gr.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(ref, out))

        def region(graph):
            for i in range(5):
                step(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            torch.cuda.synchronize()
            e0.record()
            t0 = time.perf_counter()
            if graph:
                gr.replay()
            else:
                for i in range(k):
                    step(i)
            e1.record()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / k, e0.elapsed_time(e1) * 1e3 / k
        ra, rb = [], []
        for rep in range(21):
            for gph in ((0, 1) if rep % 2 == 0 else (1, 0)):
                (rb if gph else ra).append(region(bool(gph)))
        m = lambda v, j: sorted(x[j] for x in v)[len(v) // 2]
        print("K = %3d: one by one %.2f us wall / %.2f events per step; one hipGraph of K kernel nodes %.2f / %.2f (%+.1f %% wall); outputs equal: %s" % (
            k, m(ra, 0), m(ra, 1), m(rb, 0), m(rb, 1), (m(ra, 0) / m(rb, 0) - 1) * 100, same), flush=True)


if __name__ == "__main__":
    main()
