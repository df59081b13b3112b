#!/usr/bin/env python3
"""Random sequential int8 graphs through the general network kernel against oracle/net_ref.py (which is pinned to
the reference's NNoM on six graphs): shapes, kernels, strides, paddings and channel counts the fixtures do not hold.
Graphs the planner refuses (reference quirks, LDS budget) are counted and skipped.   usage (box): tools/fuzz_net.py [n_graphs [seed [own]]]
With a third argument every graph that has a matrix-core plan is ALSO run on its own kernel (edison_net_specialize: ~1 s of
compilation per graph) and compared again. 67 inputs per graph: ragged against every per-wave batch the planner picks."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (loads torch's HIP runtime first)
from edison_amd import _lib, nnom_import  # noqa: E402
from edison_amd.context import Context  # noqa: E402
from oracle import net_ref  # noqa: E402

T_CONV, T_POOL, T_DENSE, T_SOFTMAX = 1, 2, 3, 4


def out_dim(n, k, s, same):
    return -(-n // s) if same else -(-(n - k + 1) // s)


def random_graph(rng):
    h, w, c = int(rng.integers(3, 25)), int(rng.integers(3, 25)), int(rng.choice([1, 1, 2, 3, 4, 8]))
    shape, layers = (h, w, c), []
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.choice(["conv", "conv", "pool"])
        kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        sh, sw = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        same = int(rng.integers(0, 2))
        oh, ow = out_dim(h, kh, sh, same), out_dim(w, kw, sw, same)
        if oh < 1 or ow < 1:
            continue
        if kind == "conv":
            oc = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16]))
            layers.append(dict(type=T_CONV, out_ch=oc, kh=kh, kw=kw, sh=sh, sw=sw, w=rng.integers(-100, 101, oc * kh * kw * c).astype(np.int8),
                               b=rng.integers(-100, 101, oc).astype(np.int8), out_rshift=int(rng.integers(5, 11)),
                               bias_lshift=int(rng.integers(0, 7)), relu=int(rng.integers(0, 2)), same=same))
            h, w, c = oh, ow, oc
        else:
            layers.append(dict(type=T_POOL, kh=kh, kw=kw, sh=sh, sw=sw, same=same))
            h, w = oh, ow
    for _ in range(int(rng.integers(0, 3))):
        no = int(rng.integers(1, 40))
        layers.append(dict(type=T_DENSE, out=no, w=rng.integers(-100, 101, no * h * w * c).astype(np.int8), b=rng.integers(-100, 101, no).astype(np.int8),
                           out_rshift=int(rng.integers(5, 12)), bias_lshift=int(rng.integers(0, 7)), relu=int(rng.integers(0, 2))))
        h, w, c = 1, 1, no
    if not layers:
        return None
    if rng.integers(0, 2):
        layers.append(dict(type=T_SOFTMAX))
    return shape, layers


def main():
    n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
    own = len(sys.argv) > 3
    os.environ["EDISON_NET_SPECIALIZE"] = "0"   # loads leave the graph on the general kernel; `own` asks for the own kernel explicitly
    if own:
        os.environ["EDISON_JIT_CACHE"] = "off"
    ctx = Context(0, model_path=None)
    ran = refused = on_mfma = on_own = 0
    while ran + refused < n_graphs:
        g = random_graph(rng)
        if g is None:
            continue
        shape, layers = g
        plain = [dict(L) for L in layers]
        try:
            # Dense weights are random anyway: whatever matrix the importer's de-interleave makes of the stream is the model
            blob = nnom_import.build_blob(shape, [dict(L) for L in layers])
            ctx.load_model_bytes(blob)
        except _lib.EdisonError as e:
            assert e.code in (_lib.E_SIZE, _lib.E_NO_IMPL), str(e)
            refused += 1
            continue
        x = rng.integers(-128, 128, (67, shape[0] * shape[1] * shape[2])).astype(np.int8)
        x[:8] = rng.integers(-10, 11, (8, x.shape[1]))
        ref = net_ref.run(blob, x)
        got = ctx.net_layers(x)
        want = np.concatenate(ref["acts"], axis=1)
        if not np.array_equal(got, want):
            first = int(np.argwhere(got != want)[0][1])
            raise SystemExit("MISMATCH on graph %s %s at activation byte %d" % (shape, [(L["type"], {k: v for k, v in L.items() if k not in ("w", "b")}) for L in plain], first))
        out = ctx.net(x)
        if not (np.array_equal(out["argmax"], ref["argmax"]) and np.array_equal(out["logits"], ref["logits"])):
            bad = np.argwhere(out["logits"] != ref["logits"])
            raise SystemExit("MISMATCH (batch path, accelerated=%s) on graph %s %s: %d logits differ, first at %s; argmax equal: %s" % (
                ctx.net_info().get("accelerated"), shape, [(L["type"], {k: v for k, v in L.items() if k not in ("w", "b")}) for L in plain],
                len(bad), bad[:1].tolist(), np.array_equal(out["argmax"], ref["argmax"])))
        ran += 1
        on_mfma += 1 if ctx.net_info().get("accelerated") == 2 else 0
        if own and ctx.net_info().get("accelerated") == 2:
            ctx.net_specialize()
            out2 = ctx.net(x)
            if not (np.array_equal(out2["argmax"], ref["argmax"]) and np.array_equal(out2["logits"], ref["logits"])):
                raise SystemExit("MISMATCH (the graph's OWN kernel) on graph %s %s" % (shape, [(L["type"], {k: v for k, v in L.items() if k not in ("w", "b")}) for L in plain]))
            on_own += 1
    print("general network kernel: %d random graphs bit-exact against oracle/net_ref.py (%d of them on the matrix-core kernel, the rest layer by layer%s), %d refused by the planner" % (
        ran, on_mfma, "; %d of them again on their own run-time-compiled kernel" % on_own if own else "", refused))


if __name__ == "__main__":
    main()
