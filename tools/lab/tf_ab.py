"""Variant TF (windowed frames) at 65 536 frames: the two-frame kernel's WINDOW instance against the one-frame kernel it ran on until round 5
(ED_MFCC_ONE_FRAME=1 selects the latter; read once per process, hence the child). Prints us per launch for both and whether the outputs agree bit for bit."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    import torch
    from edison_amd import _lib
    from edison_amd.context import Context
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)     # one stream for torch's events and the library's launches
    torch.cuda.set_stream(stream)
    ctx.use_torch_stream(stream)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.randn((65536, 1024), generator=g) * 3000).clamp(-32768, 32767).to(torch.int16).to(dev)
    out = torch.empty((65536, 13), dtype=torch.float32, device=dev)
    res = {}
    for name, var in (("TF", _lib.MFCC_TF), ("A", _lib.MFCC_A)):
        for _ in range(300):
            ctx.mfcc_t(x, 65536, 1024, var, 13, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                ctx.mfcc_t(x, 65536, 1024, var, 13, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 200 * 1e3)
        res[name] = sorted(ts)[2]
    ctx.mfcc_t(x, 65536, 1024, _lib.MFCC_TF, 13, out=out)
    torch.cuda.synchronize()
    import hashlib
    res["sha"] = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(run())
        raise SystemExit(0)
    env = dict(os.environ, ED_MFCC_ONE_FRAME="1")
    one = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
    print("one frame per wave :", one.stdout.strip() or one.stderr[-500:])
    two = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, timeout=600)
    print("two frames per wave:", two.stdout.strip() or two.stderr[-500:])
