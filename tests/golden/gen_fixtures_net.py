#!/usr/bin/env python3
"""Golden vectors for the GENERAL network path (any sequential NNoM graph), produced by the reference's own code.

The reference ships one generated model header (firmware/src/ai/nnom/kws_nnom/weights.h). To pin other topologies
this script writes model headers of its own in the same generated format (random int8 weights: the DATA is ours, the
format is NNoM's), compiles the REFERENCE NNoM 0.3.0 + CMSIS-NN sources around each of them (`make -C oracle alt`,
sources compiled where they lie under /root/reference, nothing copied) and records what model_run() produces for
seeded inputs, layer by layer. Committed results:

    tests/golden/alt_models/<name>.h                      the generated headers (input of tools/import_weights_h.py)
    tests/golden/net_golden.npz                            in_<name>, acts_<name> (all compute-layer outputs back to back)

The graphs are chosen to reach every CMSIS-NN kernel NNoM's dispatch can pick (nnom_conv2d.c:128-200,
nnom_maxpool.c:118-150): basic / fast / 1x1-fast / RGB, square and non-square, SAME and VALID padding, strides,
ReLU after Conv2D and Dense, Flatten, Dense with row and column tails of the weight interleave, with and without Softmax.

Run here only (needs /root/reference and gcc):   python3 tests/golden/gen_fixtures_net.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from edison_amd import nnom_import as imp         # noqa: E402
from oracle import net_ref                        # noqa: E402

# layer: ("conv", out_ch, (kh, kw), (sh, sw), "SAME"|"VALID", relu) | ("pool", (kh, kw), (sh, sw), pad) |
#        ("flatten",) | ("dense", out, relu) | ("softmax",)
MODELS = {
    # non-square, fast_nonsquare + 1x1 fast kernels, Flatten, Dense tails (17 rows, 17 columns), ReLU after Dense
    "same_stride": ((20, 12, 1), [("conv", 8, (3, 3), (1, 1), "SAME", 1), ("pool", (2, 2), (2, 2), "SAME"),
                                  ("conv", 12, (3, 5), (2, 1), "VALID", 1), ("conv", 6, (1, 1), (1, 1), "VALID", 0),
                                  ("flatten",), ("dense", 17, 1), ("dense", 5, 0), ("softmax",)]),
    # 3 input channels on a non-square image, odd channel counts (basic_nonsquare), tall pooling window, no Softmax
    "odd_no_softmax": ((9, 7, 3), [("conv", 5, (2, 3), (1, 2), "SAME", 1), ("pool", (3, 1), (1, 1), "VALID"),
                                   ("dense", 7, 0)]),
    # square images: arm_convolve_HWC_q7_fast, arm_maxpool_q7_HWC, arm_convolve_HWC_q7_basic, arm_convolve_HWC_q7_RGB
    "square": ((16, 16, 4), [("conv", 8, (3, 3), (1, 1), "SAME", 1), ("pool", (2, 2), (2, 2), "VALID"),
                             ("conv", 3, (3, 3), (2, 2), "VALID", 1), ("conv", 4, (1, 1), (1, 1), "VALID", 0),
                             ("dense", 10, 0), ("softmax",)]),
    # another classifier with the keyword-spotting geometry (31x13x1 -> 10): goes through edison_kws_* / edison_stream_*
    "kws_small": ((31, 13, 1), [("conv", 8, (3, 3), (2, 1), "SAME", 1), ("pool", (2, 2), (2, 2), "VALID"),
                                ("dense", 10, 0), ("softmax",)]),
    # the reference's own other architectures (audio/edison/train/kws_keras.py:170-352) with random parameters:
    # 'tiny_conv' as written there (:176-198) ...
    "tiny_conv": ((31, 13, 1), [("conv", 8, (8, 10), (2, 2), "SAME", 1), ("flatten",), ("dense", 10, 0), ("softmax",)]),
    # ... and 'low_latency_conv' (:322-352: a kernel wider than the image, stride (1,4), three Dense layers) with 10
    # filters and 32-wide Dense layers instead of 186 / 128, to keep the committed header small
    "low_latency_small": ((31, 13, 1), [("conv", 10, (8, 31), (1, 4), "SAME", 1), ("flatten",), ("dense", 32, 0),
                                        ("dense", 32, 0), ("dense", 10, 0), ("softmax",)]),
    # SAME padding whose pad is 0 (kernel extent 1 or 2) but whose last windows still overhang the right / bottom edge,
    # because the output is ceil(in/stride) wide: a graph tools/fuzz_net.py found the kernel wrong on
    "even_same": ((4, 20, 4), [("conv", 3, (2, 3), (2, 1), "SAME", 0), ("conv", 1, (1, 2), (1, 1), "SAME", 1),
                               ("pool", (2, 2), (1, 1), "VALID"), ("softmax",)]),
}


def out_dim(n, k, s, same):
    return -(-n // s) if same else -(-(n - k + 1) // s)


def write_header(name, in_shape, layers, rng):
    """A model header in the format NNoM's generator emits (same macros and statements as the reference's weights.h)."""
    h, w, c = in_shape
    defs, decls, body = [], [], []
    n_conv = n_dense = n_pool = 0
    idx = 0
    body.append("\tlayer[0] = Input(shape(%d, %d, %d), nnom_input_data);" % in_shape)

    def arr(v):
        return "{" + ", ".join(str(int(t)) for t in v) + "}"

    def tensors(tag, wn, bn):
        wv = rng.integers(-90, 91, wn)
        bv = rng.integers(-100, 101, bn)
        rs, bl = int(rng.integers(6, 10)), int(rng.integers(0, 6))
        up = tag.upper()
        defs.append("#define %s_KERNEL_0 %s\n\n#define %s_BIAS_0 %s\n" % (up, arr(wv), up, arr(bv)))
        defs.append("#define %s_OUTPUT_RSHIFT (%d)\n#define %s_BIAS_LSHIFT (%d)\n" % (up, rs, up, bl))
        decls.append("static const int8_t %s_weights[] = %s_KERNEL_0;" % (tag, up))
        decls.append("static const nnom_weight_t %s_w = { (const void*)%s_weights, %s_OUTPUT_RSHIFT};" % (tag, tag, up))
        decls.append("static const int8_t %s_bias[] = %s_BIAS_0;" % (tag, up))
        decls.append("static const nnom_bias_t %s_b = { (const void*)%s_bias, %s_BIAS_LSHIFT};" % (tag, tag, up))

    for L in layers:
        if L[0] == "conv":
            n_conv += 1
            tag = "conv2d_%d" % n_conv
            _, oc, (kh, kw), (sh, sw), pad, relu = L
            tensors(tag, oc * kh * kw * c, oc)
            idx += 1
            body.append("\tlayer[%d] = model.hook(Conv2D(%d, kernel(%d, %d), stride(%d, %d), PADDING_%s, &%s_w, &%s_b), layer[%d]);"
                        % (idx, oc, kh, kw, sh, sw, pad, tag, tag, idx - 1))
            if relu:
                idx += 1
                body.append("\tlayer[%d] = model.active(act_relu(), layer[%d]);" % (idx, idx - 1))
            h, w, c = out_dim(h, kh, sh, pad == "SAME"), out_dim(w, kw, sw, pad == "SAME"), oc
        elif L[0] == "pool":
            n_pool += 1
            _, (kh, kw), (sh, sw), pad = L
            idx += 1
            body.append("\tlayer[%d] = model.hook(MaxPool(kernel(%d, %d), stride(%d, %d), PADDING_%s), layer[%d]);"
                        % (idx, kh, kw, sh, sw, pad, idx - 1))
            h, w = out_dim(h, kh, sh, pad == "SAME"), out_dim(w, kw, sw, pad == "SAME")
        elif L[0] == "flatten":
            idx += 1
            body.append("\tlayer[%d] = model.hook(Flatten(), layer[%d]);" % (idx, idx - 1))
        elif L[0] == "dense":
            n_dense += 1
            tag = "dense_%d" % n_dense
            _, no, relu = L
            tensors(tag, no * h * w * c, no)
            idx += 1
            body.append("\tlayer[%d] = model.hook(Dense(%d, &%s_w, &%s_b), layer[%d]);" % (idx, no, tag, tag, idx - 1))
            if relu:
                idx += 1
                body.append("\tlayer[%d] = model.active(act_relu(), layer[%d]);" % (idx, idx - 1))
            h, w, c = 1, 1, no
        elif L[0] == "softmax":
            idx += 1
            body.append("\tlayer[%d] = model.hook(Softmax(), layer[%d]);" % (idx, idx - 1))
    n_out = h * w * c
    idx += 1
    body.append("\tlayer[%d] = model.hook(Output(shape(%d,1,1), nnom_output_data), layer[%d]);" % (idx, n_out, idx - 1))
    text = ("/* generated by tests/golden/gen_fixtures_net.py: model '%s', random int8 parameters (seeded) */\n"
            "#include \"nnom.h\"\n\n" % name + "\n".join(defs) + "\n/* weights for each layer */\n" + "\n".join(decls) +
            "\n\n/* nnom model */\nstatic int8_t nnom_input_data[%d];\nstatic int8_t nnom_output_data[%d];\n"
            "static nnom_model_t* nnom_model_create(void)\n{\n\tstatic nnom_model_t model;\n\tnnom_layer_t* layer[%d];\n\n"
            "\tnew_model(&model);\n\n" % (in_shape[0] * in_shape[1] * in_shape[2], n_out, idx + 1) + "\n".join(body) +
            "\n\tmodel_compile(&model, layer[0], layer[%d]);\n\treturn &model;\n}\n" % idx)
    os.makedirs(os.path.join(HERE, "alt_models"), exist_ok=True)
    with open(os.path.join(HERE, "alt_models", name + ".h"), "w") as f:
        f.write(text)
    # the reference shim includes "kws_nnom/weights.h": give the compiler a scratch include directory of that shape
    d = os.path.join(ROOT, "oracle", "_ref", "alt_src", name)
    os.makedirs(os.path.join(d, "kws_nnom"), exist_ok=True)
    with open(os.path.join(d, "kws_nnom", "weights.h"), "w") as f:
        f.write(text)
    return d


# nnom_layer_type_t values of the layers that leave no record in the blob (nnom.h:49-80)
NNOM_INPUT, NNOM_OUTPUT, NNOM_FLATTEN = 2, 3, 23


def reference_layers(so_path, x):
    L = ctypes.CDLL(so_path)
    L.nnom_ref_run_layers.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int32]
    assert L.nnom_ref_init() == 0
    assert L.nnom_ref_in_bytes() == x.shape[1]
    rows = []
    for u in range(x.shape[0]):
        dump = np.zeros(1 << 16, dtype=np.int8)
        sizes = np.zeros(16, dtype=np.int32)
        types = np.zeros(16, dtype=np.int32)
        inp = np.ascontiguousarray(x[u])
        n = L.nnom_ref_run_layers(inp.ctypes.data, dump.ctypes.data, dump.size, sizes.ctypes.data, types.ctypes.data, 16)
        assert n > 0, "model_run failed: %d" % n
        off, keep = 0, []
        for i in range(n):
            if types[i] not in (NNOM_INPUT, NNOM_OUTPUT, NNOM_FLATTEN):
                keep.append(dump[off:off + sizes[i]].copy())
            off += sizes[i]
        rows.append(np.concatenate(keep))
    return np.stack(rows)


def main():
    out = {}
    for k, (name, (in_shape, layers)) in enumerate(MODELS.items()):
        rng = np.random.default_rng(700 + k)
        alt_dir = write_header(name, in_shape, layers, rng)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "alt", "ALT_DIR=" + alt_dir, "ALT_NAME=" + name])
        n_in = in_shape[0] * in_shape[1] * in_shape[2]
        x = rng.integers(-128, 128, (48, n_in)).astype(np.int8)
        x[0] = 0
        x[1] = 127
        x[2] = -128
        x[3:12] = rng.integers(-20, 21, (9, n_in))                                  # quiet inputs: unsaturated layers
        acts = reference_layers(os.path.join(ROOT, "oracle", "_ref", "alt_%s.so" % name), x)
        # the importer + the numpy restatement must reproduce the reference before the vectors are worth committing
        with open(os.path.join(HERE, "alt_models", name + ".h")) as f:
            shape, parsed = imp.parse_weights_h(f.read())
        blob = imp.build_blob(shape, parsed)
        mine = np.concatenate(net_ref.run(blob, x)["acts"], axis=1)
        assert mine.shape == acts.shape, (name, mine.shape, acts.shape)
        assert np.array_equal(mine, acts), "%s: oracle/net_ref.py differs from the reference NNoM build" % name
        out["in_" + name] = x
        out["acts_" + name] = acts
        print("%-16s input %s, %d bytes of layer outputs per input, numpy restatement == reference" % (name, in_shape, acts.shape[1]))
    np.savez_compressed(os.path.join(HERE, "net_golden.npz"), **out)
    print("wrote tests/golden/net_golden.npz")


if __name__ == "__main__":
    main()
