#!/usr/bin/env python3
"""Is the MFCC launch deterministic, and do two libraries agree bit for bit? usage: det_check.py libA libB"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 65536
x = (torch.randn((N, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16)
_lib._share_torch_hip_runtime()
def load(name):
    p = _lib.LIB_PATH if name == "prod" else os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % name)
    L = ctypes.CDLL(p)
    for n in ("edison_init", "edison_set_stream", "edison_mfcc_batch_dev"):
        fn = getattr(L, n); fn.restype, fn.argtypes = _lib.SIGNATURES[n]
    h = ctypes.c_void_p(); assert L.edison_init(0, ctypes.byref(h)) == 0
    L.edison_set_stream(h, ctypes.c_void_p(st.cuda_stream))
    return L, h
def run(L, h, n=N, variant=1):
    o = torch.zeros((n, 13), dtype=torch.float32, device=dev)
    assert L.edison_mfcc_batch_dev(h, x.data_ptr(), n, 1024, variant, 13, o.data_ptr(), None, 1.0) == 0
    torch.cuda.synchronize(); return o
libs = [load(n) for n in sys.argv[1:]]
for name, (L, h) in zip(sys.argv[1:], libs):
    a = run(L, h); same = all(torch.equal(a, run(L, h)) for _ in range(5))
    print("%s: six launches bit-identical: %s" % (name, same))
a, b = run(*libs[0]), run(*libs[1])
d = (a != b)
print("differing values %d of %d, rows %d; max |d| %g" % (int(d.sum()), d.numel(), int(d.any(dim=1).sum()), float((a - b).abs().max())))
if d.any():
    r = torch.nonzero(d.any(dim=1))[:8, 0].tolist(); print("first differing rows", r, "cols", [torch.nonzero(d[i])[:, 0].tolist() for i in r[:4]])
    small = run(*libs[0], n=64), run(*libs[1], n=64)
    print("64-frame launch equal:", torch.equal(*small), "; lib0 big[:64] == lib0 small:", torch.equal(a[:64], small[0]), "; lib1:", torch.equal(b[:64], small[1]))
