#!/usr/bin/env python3
"""Run the stamped diagnostic MFCC kernel (built from tools/ablate/mfcc_kernels_stamp.hip) and print the share of
wave time per phase. Never quote this build's run time: the stamps forbid overlaps the real kernel has."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from edison_amd import _lib
from edison_amd.context import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = Context(0); ctx.use_torch_stream()
dev = torch.device("cuda", 0)
a = (torch.randn((n, 1024), device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16)
out = torch.empty((n, 13), dtype=torch.float32, device=dev)
dbg = torch.zeros((8192, 12), dtype=torch.int64, device=dev)
L = _lib.lib()
# the stamp build dumps through the `fft` pointer of the non-stages kernel: call the internal launcher via mfcc_stages? no:
# edison_mfcc_batch_dev passes fft = NULL, so use a tiny shim: edison_mfcc_stages_dev would select STAGES=true. Instead the
# stamp build is compiled with -DED_STAMP_DBG and reads the buffer address from the environment-provided global.
L.ed_set_debug_buffer.argtypes = [ctypes.c_void_p]
L.ed_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(3):
    ctx.mfcc_t(a, n, 1024, _lib.MFCC_B, 13, out=out)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(float)
d = d[d.sum(1) > 0]
tot = d.sum(1).mean()
names = ["unpack+prefetch", "pass1+tw", "T1 lds", "pass2+tw", "T2 lds", "pass3", "Pz+split+sqrt", "S write", "mel", "log+DCT", "store", "loop/wait-data"]
fpw = n / d.shape[0]
print("waves %d, frames/wave %.1f, cycles/frame/wave %.0f (memtime ticks)" % (d.shape[0], fpw, tot / fpw))
for i, nm in enumerate(names):
    print("  %-16s %7.0f ticks/frame  %5.1f%%" % (nm, d[:, i].mean() / fpw, 100 * d[:, i].mean() / tot))
