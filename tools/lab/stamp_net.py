#!/usr/bin/env python3
"""Phase stamps of the general matrix-core network kernel (variant built with -DEMM_STAMP=1@cnn_net_mfma_kernels.hip):
cycles of workgroup 0 per phase, summed over its batches. Shares, not lengths (the stamps add barriers)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["EDISON_NET_FORCE_GENERAL"] = "1"
os.environ["EDISON_LIB"] = os.path.join(ROOT, "edison_amd/csrc/abl/libedison_hip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "netstamp"))
os.environ.setdefault("EDISON_NET_SPECIALIZE", "0")  # a model load must not take a cached own kernel by itself: this tool times the general one
import torch
from edison_amd import _lib
from edison_amd.context import Context, _t_ptr
ctx = Context(0)
L = ctx._L
L.ed_set_net_debug_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
dbg = torch.zeros(48, dtype=torch.int64, device=dev)
L.ed_set_net_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
ctx.use_torch_stream()
n = 262144
x = torch.randint(-128, 128, (n, 403), dtype=torch.int32, device=dev).to(torch.int8)
lo = torch.empty((n, 10), dtype=torch.int8, device=dev); am = torch.empty((n,), dtype=torch.int32, device=dev)
for _ in range(3): ctx._check(L.edison_net_batch_dev(ctx._h, _t_ptr(x), n, _t_ptr(lo), None, _t_ptr(am)))
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(float)
tot = d.sum()
info = ctx.net_info()
print("input load %5.1f %%" % (100 * d[0] / tot))
for li, l in enumerate(info["layers"]):
    print("layer %d type %d: outputs of the layer in front %5.1f %%   run record (scalar loads) %5.1f %%   setup (zero, expand, tables) %5.1f %%   body (tiles / pool / softmax) %5.1f %%" % (li, l["type"], 100 * d[4 + 5 * li] / tot, 100 * d[3 + 5 * li] / tot, 100 * d[1 + 5 * li] / tot, 100 * d[2 + 5 * li] / tot))
print("inside the tile groups of all layers: group setup %5.1f %%, seeds + k-loop %5.1f %%, epilogue %5.1f %%, before the first group %5.1f %%" % tuple(100 * d[i] / tot for i in (40, 41, 42, 43)))
print("outputs + loop %5.1f %%" % (100 * (d[46] + d[47]) / tot))
waves = 256 * info.get("mm_waves", 12)
print("cycles per input of wave 0 of workgroup 0: %.0f (3 such waves share a SIMD)" % (tot / (n / waves)))
