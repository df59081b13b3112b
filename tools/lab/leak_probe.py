#!/usr/bin/env python3
"""Device-memory growth per object type (hipMemGetInfo around create / use / destroy loops): a leak shows as MB per iteration."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["EDISON_NET_SPECIALIZE"] = "0"
os.environ["EDISON_JIT_CACHE"] = "/tmp/edison_leak_jit"
from edison_amd.context import Context
from edison_amd.stream import Stream
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 1e6
c0 = Context(0); c0.close()
base = used(); print("base %.1f MB" % base)
for i in range(6):
    c = Context(0, model_path=None); c.load_weights_h("tests/golden/alt_models/kws_small.h"); c.net_specialize(); c.net(np.zeros((10, 403), np.int8)); c.close()
print("6 x (context + load + specialize + net + close): +%.1f MB" % (used() - base))
big = Context(0)
for label, kw, push in (("chunk 16, direct", dict(hop=512, chunk_frames=16), True), ("chunk 16, direct, never pushed", dict(hop=512, chunk_frames=16), False),
                        ("chunk 16, graph mode", dict(hop=512, chunk_frames=16, graph=True), True), ("chunk 1 (mapped path)", dict(hop=512, chunk_frames=1), True),
                        ("chunk 4096 (plain path)", dict(hop=512, chunk_frames=4096), True), ("chunk 16 with the output filter", dict(hop=512, chunk_frames=16, output_filter=True), True)):
    m = used()
    for i in range(20):
        st = Stream(big, **kw)
        if push:
            st.push(np.zeros(kw["chunk_frames"] * kw["hop"], np.int16))
        st.close()
    print("20 x stream (%s): +%.2f MB" % (label, used() - m))
big.close()
print("after closing the context: %+.1f MB vs base" % (used() - base))
