"""oracle/net_ref.py -- TEST INFRASTRUCTURE (the parity checker), not product code.

numpy restatement of the int8 arithmetic NNoM 0.3.0 + CMSIS-NN (portable branches) perform for ANY sequential
graph stored in an .ednn blob (tools/import_weights_h.py). Pinned by tests/golden/net_golden.npz: layer outputs
of the reference's own NNoM compiled around other generated model headers (tests/golden/gen_fixtures_net.py).

  Conv2D   conv_out = (bias << bias_shift) + NN_ROUND(out_shift) + sum over the taps inside the image;
           out = sat8(conv_out >> out_shift)      arm_convolve_HWC_q7_basic_nonsquare.c:188-221 (and the fast,
           1x1, RGB and square variants: same formula on their portable branches); padding = (k-1)/2 for
           PADDING_SAME, output = ceil(in/stride) or ceil((in-k+1)/stride)   nnom_conv2d.c:66-71,92-104
  ReLU     max(x, 0) in place, tail activation                               arm_relu_q7.c, nnom.c:986-989
  MaxPool  max over the part of the window inside the image, from -129       nnom_local.c:117-159
  Dense    the conv formula over the flattened HWC input                     arm_fully_connected_q7_opt.c:374-473
  Softmax  arm_softmax_q7.c:215-260, portable branch
  argmax   first maximum of the last layer's output                          nnom_utils.c:275-284
"""
import struct

import numpy as np

T_CONV, T_POOL, T_DENSE, T_SOFTMAX = 1, 2, 3, 4


def parse_blob(blob):
    if blob[:8] != b"EDNNOM1\0":
        raise ValueError("not an .ednn blob")
    in_h, in_w, in_c, n_layers, payload_bytes = struct.unpack_from("<5i", blob, 8)
    recs = [struct.unpack_from("<12i", blob, 40 + 48 * i) for i in range(n_layers)]
    payload = np.frombuffer(blob, dtype=np.int8, count=payload_bytes, offset=40 + 48 * n_layers)
    return (in_h, in_w, in_c), recs, payload


def _ceil_div(a, b):
    return -(-a // b)


def _sat8(a):
    return np.clip(a, -128, 127).astype(np.int8)


def _windows(h, w, kh, kw, sh, sw, same):
    ph, pw = ((kh - 1) // 2, (kw - 1) // 2) if same else (0, 0)
    oh = _ceil_div(h, sh) if same else _ceil_div(h - kh + 1, sh)
    ow = _ceil_div(w, sw) if same else _ceil_div(w - kw + 1, sw)
    return ph, pw, oh, ow


def run(blob, x):
    """x: [n][in_h*in_w*in_c] int8 -> dict(acts=[per-layer (n, out_n) int8], logits, softmax (or None), argmax)."""
    (h, w, c), recs, payload = parse_blob(blob)
    x = np.ascontiguousarray(x, dtype=np.int8).reshape(-1, h, w, c)
    n = x.shape[0]
    cur = x.astype(np.int32)
    acts = []
    has_softmax = False
    for v in recs:
        t = v[0]
        if t in (T_CONV, T_POOL):
            kh, kw, sh, sw, same = v[2], v[3], v[4], v[5], (v[8] >> 1) & 1
            ph, pw, oh, ow = _windows(h, w, kh, kw, sh, sw, same)
            if t == T_CONV:
                oc, bl, rs, relu = v[1], v[6], v[7], v[8] & 1
                wt = payload[v[9]:v[9] + oc * kh * kw * c].astype(np.int32).reshape(oc, kh, kw, c)
                bias = payload[v[10]:v[10] + oc].astype(np.int32)
                acc = np.broadcast_to((bias << bl) + ((1 << rs) >> 1), (n, oh, ow, oc)).astype(np.int64).copy()
            else:
                acc = np.full((n, oh, ow, c), -129, dtype=np.int32)
            for ky in range(kh):
                for kx in range(kw):
                    # output rows/cols whose tap (ky, kx) falls inside the image
                    oy = [y for y in range(oh) if 0 <= y * sh - ph + ky < h]
                    ox = [q for q in range(ow) if 0 <= q * sw - pw + kx < w]
                    if not oy or not ox:
                        continue
                    iy = [y * sh - ph + ky for y in oy]
                    ix = [q * sw - pw + kx for q in ox]
                    patch = cur[:, iy][:, :, ix]                              # (n, len(oy), len(ox), c)
                    if t == T_CONV:
                        acc[np.ix_(range(n), oy, ox)] += np.einsum("nyxc,oc->nyxo", patch, wt[:, ky, kx, :])
                    else:
                        sub = acc[np.ix_(range(n), oy, ox)]
                        acc[np.ix_(range(n), oy, ox)] = np.maximum(sub, patch)
            if t == T_CONV:
                out = _sat8(acc >> rs).astype(np.int32)
                if relu:
                    out = np.maximum(out, 0)
                h, w, c = oh, ow, oc
            else:
                out = acc.astype(np.int8).astype(np.int32)                       # -129 would wrap like the C store
                h, w = oh, ow
            cur = out
        elif t == T_DENSE:
            no, bl, rs, relu, ni = v[1], v[6], v[7], v[8] & 1, v[11]
            wt = payload[v[9]:v[9] + no * ni].astype(np.int64).reshape(no, ni)
            bias = payload[v[10]:v[10] + no].astype(np.int64)
            acc = cur.reshape(n, ni).astype(np.int64) @ wt.T + (bias << bl) + ((1 << rs) >> 1)
            out = _sat8(acc >> rs).astype(np.int32)
            if relu:
                out = np.maximum(out, 0)
            cur = out.reshape(n, 1, 1, no)
            h, w, c = 1, 1, no
        elif t == T_SOFTMAX:
            vin = cur.reshape(n, -1)
            base = np.maximum(vin.max(axis=1), -128) - 8
            shift = np.clip(vin - base[:, None], 0, 7)
            s = (1 << shift).sum(axis=1)
            ob = (1 << 20) // s
            out = _sat8(ob[:, None] >> np.clip(13 + base[:, None] - vin, 0, 31)).astype(np.int32)
            cur = out.reshape(n, 1, 1, -1)
            has_softmax = True
        else:
            raise ValueError("unknown layer type %d" % t)
        acts.append(cur.reshape(n, -1).astype(np.int8))
    last = acts[-1]
    return dict(acts=acts, logits=acts[-2] if has_softmax else last, softmax=last if has_softmax else None,
                argmax=np.argmax(last, axis=1).astype(np.int32))
