#!/usr/bin/env python3
"""One-off: sub-graphs of a failing fuzz graph through the batch (matrix-core) path against oracle/net_ref.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from edison_amd import nnom_import
from edison_amd.context import Context
from oracle import net_ref
rng = np.random.default_rng(5)
def conv(oc, kh, kw, sh, sw, c, rs, bl, relu, same):
    return dict(type=1, out_ch=oc, kh=kh, kw=kw, sh=sh, sw=sw, w=rng.integers(-100, 101, oc * kh * kw * c).astype(np.int8),
                b=rng.integers(-100, 101, oc).astype(np.int8), out_rshift=rs, bias_lshift=bl, relu=relu, same=same)
c1 = conv(5, 1, 3, 1, 2, 1, 10, 5, 0, 1)
c2 = conv(1, 4, 2, 2, 2, 5, 6, 4, 1, 0)
pool = dict(type=2, kh=1, kw=2, sh=1, sw=2, same=1)
sm = dict(type=4)
ctx = Context(0, model_path=None)
shape = (18, 10, 1)
x = rng.integers(-128, 128, (64, 180)).astype(np.int8)
for name, layers in (("conv1", [c1]), ("conv1+conv2", [c1, c2]), ("conv1+conv2+pool", [c1, c2, pool]), ("all", [c1, c2, pool, sm]),
                     ("conv1 VALID", [dict(c1, same=0)]), ("conv1 sw1", [dict(c1, sw=1)]), ("conv1 oc8", [conv(8, 1, 3, 1, 2, 1, 10, 5, 0, 1)]),
                     ("conv1 kh2", [conv(5, 2, 3, 1, 2, 1, 10, 5, 0, 1)])):
    blob = nnom_import.build_blob(shape, [dict(L) for L in layers])
    ctx.load_model_bytes(blob)
    ref = net_ref.run(blob, x)
    out = ctx.net(x)
    bad = np.argwhere(out["logits"] != ref["logits"])
    print("%-20s accelerated %s: %d of %d logits differ%s" % (name, ctx.net_info().get("accelerated"), len(bad), out["logits"].size,
          "" if not len(bad) else "  first %s got %d want %d" % (bad[0].tolist(), out["logits"][tuple(bad[0])], ref["logits"][tuple(bad[0])])))
    if len(bad):
        cols = sorted(set(bad[:, 1].tolist()))
        print("     differing output elements (of input 0):", [int(c) for c in bad[bad[:, 0] == 0][:, 1]][:40], " distinct columns:", len(cols))
