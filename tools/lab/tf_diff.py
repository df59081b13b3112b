"""Where do variant TF's batch outputs (two-frame WINDOW kernel) differ from the stage dump (one-frame kernel)?"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from edison_amd import _lib
from edison_amd.context import Context
ctx = Context(0)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "mfcc_golden.npz"))
for name in ("edison", "two_tone", "noise", "extremes", "quiet"):
    x = g["in_" + name]
    st = ctx.mfcc_stages(x, variant=_lib.MFCC_TF)["mfcc"]
    b = ctx.mfcc(x, variant=_lib.MFCC_TF, n_coef=32)
    d = np.abs(st - b)
    bad = np.argwhere(d > 0)
    print(name, st.shape, "max |d|", d.max(), "n differing", len(bad), "frames", sorted(set(bad[:, 0].tolist()))[:12], "max |ref|", np.abs(st).max())
    a_st = ctx.mfcc_stages(x, variant=_lib.MFCC_A)["mfcc"]
    a_b = ctx.mfcc(x, variant=_lib.MFCC_A, n_coef=32)
    print("   variant A same:", np.array_equal(a_st, a_b))
