#!/usr/bin/env python3
"""Convert an NNoM-generated ``weights.h`` into this repo's binary model blob.

The reference ships its trained int8 keyword-spotting network as a generated C
header (``firmware/src/ai/nnom/kws_nnom/weights.h``: ``#define X_KERNEL_0 {…}``
arrays, ``*_SHIFT`` macros and the ``nnom_model_create()`` graph, lines 3-161).
The product on the MI355X consumes *parameters*, not C source, so this tool
parses the header and emits a small self-describing binary (``.ednn``) that the
C-ABI library (``edison_model_load``), the Python host side and the oracle all
read.  Only numbers leave the header: tensors, shifts and the layer list.

Layout decisions taken here (and nowhere else):
  * conv kernels stay OHWI (``w[o][ky][kx][ci]``), exactly as CMSIS-NN indexes
    them (arm_convolve_HWC_q7_basic_nonsquare.c:209-211);
  * the dense matrix is DE-INTERLEAVED from ``arm_fully_connected_q7_opt``'s
    storage order (arm_fully_connected_q7_opt.c:374-473, enabled by
    ``DENSE_WEIGHT_OPT 1`` in nnom_port.h:34) into plain row-major
    ``[out][in]``; the blob's flag word records that this was done.

Blob format (little endian):
  char     magic[8]   = b"EDNNOM1\\0"
  int32    in_h, in_w, in_c
  int32    n_layers
  int32    payload_bytes
  int32    reserved[3]
  n_layers x int32[12] records:
     [0] type  1=conv2d  2=maxpool  3=dense  4=softmax
     conv2d : [1]=out_ch [2]=kh [3]=kw [4]=sh [5]=sw [6]=bias_lshift [7]=out_rshift
              [8]=flags (bit 0 ReLU tail activation, bit 1 PADDING_SAME)   [9]=weight_off [10]=bias_off [11]=in_ch
     maxpool: [2]=kh [3]=kw [4]=sh [5]=sw [8]=flags (bit 1 PADDING_SAME)
     dense  : [1]=out [6]=bias_lshift [7]=out_rshift [8]=flags (bit 0 ReLU) [9]=weight_off [10]=bias_off [11]=in
              (Flatten() is a no-op on HWC memory and leaves no record)
     softmax: -
  int8     payload[payload_bytes]   (offsets above index into it)

Usage:  tools/import_weights_h.py /path/to/weights.h edison_amd/data/kws_nnom.ednn
"""
import re
import struct
import sys

import numpy as np

MAGIC = b"EDNNOM1\0"
T_CONV, T_POOL, T_DENSE, T_SOFTMAX = 1, 2, 3, 4


def _eval_int(expr, sym):
    """Evaluate a +/- expression of integer literals and already known macros."""
    def repl(m):
        name = m.group(0)
        if name not in sym:
            raise KeyError("unknown macro %s in %r" % (name, expr))
        return str(sym[name])
    flat = re.sub(r"[A-Za-z_]\w*", repl, expr)
    if not re.fullmatch(r"[\d\s()+\-]+", flat):
        raise ValueError("unsupported macro expression %r" % expr)
    return int(eval(flat, {"__builtins__": {}}))  # digits, + - ( ) only (checked above)


def parse_weights_h(text):
    """Return (input_shape, layers, arrays) parsed from the text of a weights.h."""
    arrays, sym, pending = {}, {}, []
    for m in re.finditer(r"#define\s+(\w+)\s+\{([^}]*)\}", text):
        arrays[m.group(1)] = np.array([int(v) for v in m.group(2).replace("\n", " ").split(",") if v.strip()],
                                      dtype=np.int64)
    for m in re.finditer(r"^#define\s+(\w+)\s+(\(?[^{}\n]+?\)?)\s*$", text, flags=re.M):
        name, expr = m.group(1), m.group(2).strip()
        if name in arrays:
            continue
        pending.append((name, expr))
    # macros may reference later ones; iterate to a fixed point
    for _ in range(8):
        rest = []
        for name, expr in pending:
            try:
                sym[name] = _eval_int(expr, sym)
            except (KeyError, ValueError):
                rest.append((name, expr))
        if not rest or len(rest) == len(pending):
            break
        pending = rest

    # static const int8_t conv2d_1_weights[] = CONV2D_1_KERNEL_0;
    c_arrays = {m.group(1): m.group(2) for m in
                re.finditer(r"static const int8_t\s+(\w+)\[\]\s*=\s*(\w+);", text)}
    # static const nnom_weight_t conv2d_1_w = { (const void*)conv2d_1_weights, CONV2D_1_OUTPUT_RSHIFT};
    structs = {m.group(1): (m.group(2), m.group(3)) for m in
               re.finditer(r"static const nnom_(?:weight|bias)_t\s+(\w+)\s*=\s*\{\s*\(const void\*\)\s*(\w+)\s*,\s*(\w+)\s*\}",
                           text)}

    def tensor(struct_name):
        arr_name, shift_name = structs[struct_name]
        vals = arrays[c_arrays[arr_name]]
        if vals.min() < -128 or vals.max() > 127:
            raise ValueError("%s does not fit int8" % arr_name)
        return vals.astype(np.int8), int(sym[shift_name])

    m = re.search(r"Input\(shape\((\d+),\s*(\d+),\s*(\d+)\)", text)
    if not m:
        raise ValueError("no Input(shape(h,w,c)) in weights.h")
    in_shape = tuple(int(g) for g in m.groups())

    layers = []
    for line in re.findall(r"layer\[\d+\]\s*=\s*(.*);", text):
        mc = re.search(r"Conv2D\((\d+),\s*kernel\((\d+),\s*(\d+)\),\s*stride\((\d+),\s*(\d+)\),\s*(\w+),\s*&(\w+),\s*&(\w+)\)", line)
        mp = re.search(r"MaxPool\(kernel\((\d+),\s*(\d+)\),\s*stride\((\d+),\s*(\d+)\),\s*(\w+)\)", line)
        md = re.search(r"Dense\((\d+),\s*&(\w+),\s*&(\w+)\)", line)
        if mc:
            if mc.group(6) not in ("PADDING_VALID", "PADDING_SAME"):
                raise ValueError("unknown padding %s" % mc.group(6))
            w, rs = tensor(mc.group(7))
            b, bl = tensor(mc.group(8))
            layers.append(dict(type=T_CONV, out_ch=int(mc.group(1)), kh=int(mc.group(2)), kw=int(mc.group(3)),
                               sh=int(mc.group(4)), sw=int(mc.group(5)), w=w, b=b, out_rshift=rs, bias_lshift=bl,
                               relu=0, same=int(mc.group(6) == "PADDING_SAME")))
        elif "act_relu()" in line:
            if not layers or layers[-1]["type"] not in (T_CONV, T_DENSE):
                raise ValueError("ReLU tail activation is only supported after Conv2D or Dense")
            layers[-1]["relu"] = 1
        elif mp:
            if mp.group(5) not in ("PADDING_VALID", "PADDING_SAME"):
                raise ValueError("unknown padding %s" % mp.group(5))
            layers.append(dict(type=T_POOL, kh=int(mp.group(1)), kw=int(mp.group(2)),
                               sh=int(mp.group(3)), sw=int(mp.group(4)), same=int(mp.group(5) == "PADDING_SAME")))
        elif md:
            w, rs = tensor(md.group(2))
            b, bl = tensor(md.group(3))
            layers.append(dict(type=T_DENSE, out=int(md.group(1)), w=w, b=b, out_rshift=rs, bias_lshift=bl, relu=0))
        elif "Softmax()" in line:
            layers.append(dict(type=T_SOFTMAX))
        elif "Input(" in line or "Output(" in line or "Flatten()" in line:
            continue
        else:
            raise ValueError("unsupported layer in weights.h: %s" % line)
    return in_shape, layers


def deinterleave_dense_opt(stream, rows, cols):
    """Invert arm_fully_connected_q7_opt's weight order (portable branch,
    arm_fully_connected_q7_opt.c:374-473): rows in groups of 4, columns in groups
    of 4, 16 bytes per (row-group, column-block); leftover columns of a row group
    follow as 4 bytes (one per row); leftover rows follow plain row-major."""
    w = np.zeros((rows, cols), dtype=np.int8)
    p = 0
    # byte i of a 16-byte block -> (row offset, column offset)
    blk = [(0, 0), (1, 0), (0, 2), (1, 2), (2, 0), (3, 0), (2, 2), (3, 2),
           (0, 1), (1, 1), (0, 3), (1, 3), (2, 1), (3, 1), (2, 3), (3, 3)]
    for r in range(0, rows - rows % 4, 4):
        for c in range(0, cols - cols % 4, 4):
            for i, (dr, dc) in enumerate(blk):
                w[r + dr, c + dc] = stream[p + i]
            p += 16
        for c in range(cols - cols % 4, cols):
            for dr in range(4):
                w[r + dr, c] = stream[p]
                p += 1
    for r in range(rows - rows % 4, rows):
        w[r, :] = stream[p:p + cols]
        p += cols
    assert p == rows * cols
    return w


def out_dim(n, k, s, same):
    """NN_CEILIF(n, s) for PADDING_SAME, NN_CEILIF(n - k + 1, s) otherwise (nnom_conv2d.c:92-104)."""
    return -(-n // s) if same else -(-(n - k + 1) // s)


def build_blob(in_shape, layers):
    h, w_, c = in_shape
    payload = bytearray()
    records = []

    def put(arr):
        # keep every tensor 16-byte aligned inside the payload
        while len(payload) % 16:
            payload.append(0)
        off = len(payload)
        payload.extend(np.ascontiguousarray(arr, dtype=np.int8).tobytes())
        return off

    for L in layers:
        rec = [0] * 12
        rec[0] = L["type"]
        if L["type"] == T_CONV:
            k = L["kh"] * L["kw"] * c
            assert L["w"].size == L["out_ch"] * k, "conv weight size mismatch"
            rec[1:9] = [L["out_ch"], L["kh"], L["kw"], L["sh"], L["sw"], L["bias_lshift"], L["out_rshift"],
                        L["relu"] | (L["same"] << 1)]
            rec[9], rec[10], rec[11] = put(L["w"]), put(L["b"]), c
            h, w_, c = out_dim(h, L["kh"], L["sh"], L["same"]), out_dim(w_, L["kw"], L["sw"], L["same"]), L["out_ch"]
        elif L["type"] == T_POOL:
            rec[2:6] = [L["kh"], L["kw"], L["sh"], L["sw"]]
            rec[8] = L["same"] << 1
            h, w_ = out_dim(h, L["kh"], L["sh"], L["same"]), out_dim(w_, L["kw"], L["sw"], L["same"])
        elif L["type"] == T_DENSE:
            n_in = h * w_ * c
            assert L["w"].size == L["out"] * n_in, "dense weight size mismatch"
            plain = deinterleave_dense_opt(L["w"], L["out"], n_in)
            rec[1], rec[6], rec[7], rec[8] = L["out"], L["bias_lshift"], L["out_rshift"], L["relu"]
            rec[9], rec[10], rec[11] = put(plain), put(L["b"]), n_in
            h, w_, c = 1, 1, L["out"]
        records.append(rec)
    while len(payload) % 16:
        payload.append(0)
    head = MAGIC + struct.pack("<8i", in_shape[0], in_shape[1], in_shape[2], len(records), len(payload), 1, 0, 0)
    body = b"".join(struct.pack("<12i", *r) for r in records)
    return head + body + bytes(payload)


def main(argv):
    if len(argv) != 3:
        print(__doc__)
        return 2
    with open(argv[1], "r") as f:
        text = f.read()
    in_shape, layers = parse_weights_h(text)
    blob = build_blob(in_shape, layers)
    with open(argv[2], "wb") as f:
        f.write(blob)
    print("wrote %s: input %s, %d layers, %d bytes" % (argv[2], in_shape, len(layers), len(blob)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
