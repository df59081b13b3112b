"""GPU parity tests of the GENERAL network path (edison_net_*, csrc/cnn_net_kernels.hip + model_net.c): any sequential
NNoM int8 graph, not only the shipped kws_conv one. Integer arithmetic => every comparison is bit for bit.

The expected values in tests/golden/net_golden.npz come from the REFERENCE's own NNoM 0.3.0 + CMSIS-NN compiled around
six other generated model headers (tests/golden/gen_fixtures_net.py); larger seeded batches are compared with the numpy
restatement oracle/net_ref.py, which that script checked against the same reference build.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"]


@pytest.fixture(scope="module")
def net_golden():
    return np.load(os.path.join(GOLDEN, "net_golden.npz"))


def _header(name):
    return os.path.join(GOLDEN, "alt_models", name + ".h")


def _blob(name):
    from edison_amd import nnom_import
    with open(_header(name)) as f:
        shape, layers = nnom_import.parse_weights_h(f.read())
    return nnom_import.build_blob(shape, layers)


def _context(built_lib, name):
    from edison_amd.context import Context
    c = Context(0, model_path=None)
    c.load_weights_h(_header(name))
    return c


@pytest.mark.parametrize("name", NAMES)
def test_net_matches_reference_nnom(built_lib, net_golden, name):
    c = _context(built_lib, name)
    info = c.net_info()
    x, ref = net_golden["in_" + name], net_golden["acts_" + name]
    assert info["acts_bytes"] == ref.shape[1]
    # every fixture graph runs its Conv2D / Dense layers on the matrix cores (2); edison_net_layers (per-layer dumps)
    # always comes from the layer-by-layer kernel
    assert info["accelerated"] == 2
    assert np.array_equal(c.net_layers(x), ref)
    out = c.net(x)
    last = info["layers"][-1]
    n_out = info["n_out"]
    final = ref[:, last["acts_offset"]:last["acts_offset"] + n_out]
    if info["has_softmax"]:
        pre = info["layers"][-2]
        assert np.array_equal(out["softmax"], final)
        assert np.array_equal(out["logits"], ref[:, pre["acts_offset"]:pre["acts_offset"] + n_out])
    else:
        assert out["softmax"] is None and np.array_equal(out["logits"], final)
    assert np.array_equal(out["argmax"], np.argmax(final, axis=1))            # first maximum (nnom_utils.c:275-284)
    c.close()


@pytest.mark.parametrize("name", NAMES)
def test_net_vs_numpy_restatement_seeded(built_lib, name):
    """3000 inputs per graph: full-range noise, quiet inputs, constant extremes; more than one pass of the persistent grid."""
    from oracle import net_ref
    c = _context(built_lib, name)
    info = c.net_info()
    n_in = info["in_h"] * info["in_w"] * info["in_c"]
    rng = np.random.default_rng(41)
    x = rng.integers(-128, 128, (3000, n_in)).astype(np.int8)
    x[:300] = rng.integers(-12, 13, (300, n_in))
    x[300], x[301], x[302] = 0, 127, -128
    ref = net_ref.run(_blob(name), x)
    assert np.array_equal(c.net_layers(x), np.concatenate(ref["acts"], axis=1))
    out = c.net(x)
    assert np.array_equal(out["logits"], ref["logits"]) and np.array_equal(out["argmax"], ref["argmax"])
    if info["has_softmax"]:
        assert np.array_equal(out["softmax"], ref["softmax"])
    c.close()


def test_general_kernel_on_the_shipped_graph(ctx, cnn_golden):
    """The shipped kws_conv model also has a general plan: its layer outputs equal the specialised kernels' and the
    reference vectors of tests/golden/cnn_golden.npz; edison_net_batch keeps the matrix-core kernel for it."""
    info = ctx.net_info()
    assert info["accelerated"] == 1 and (info["in_h"], info["in_w"], info["in_c"], info["n_out"]) == (31, 13, 1, 10)
    assert [L["type"] for L in info["layers"]] == [1, 2, 1, 2, 1, 1, 3, 4] and info["acts_bytes"] == 10420
    rng = np.random.default_rng(42)
    x = rng.integers(-128, 128, (700, 403)).astype(np.int8)
    x[:100] = rng.integers(-30, 31, (100, 403))
    general = ctx.net_layers(x)
    special = ctx.cnn_layers(x)
    assert np.array_equal(general, np.concatenate([special[k] for k in ("conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "dense", "softmax")], axis=1))
    a, b = ctx.net(x), ctx.cnn(x)
    for k in ("logits", "softmax", "argmax"):
        assert np.array_equal(a[k], b[k])
    # the reference NNoM build's own layer outputs (tests/golden/gen_fixtures.py)
    g = ctx.net_layers(cnn_golden["feats"])
    want = np.concatenate([cnn_golden[k].reshape(g.shape[0], -1) for k in ("conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "dense", "softmax")], axis=1)
    assert np.array_equal(g, want)


def test_other_classifier_through_kws_and_stream(built_lib, net_golden):
    """A different graph with the keyword-spotting geometry (31x13x1 -> 10) serves edison_kws_batch, edison_cnn_batch
    and the stream through the general kernel."""
    from edison_amd import _lib
    from edison_amd.stream import Stream
    from oracle import net_ref
    c = _context(built_lib, "kws_small")
    blob = _blob("kws_small")
    rng = np.random.default_rng(43)
    audio = np.clip(rng.normal(0, 2500, 40 * 32000), -32768, 32767).astype(np.int16)
    out = c.kws(audio, n_utt=40, utt_stride=32000)
    ref = net_ref.run(blob, out["feat"].reshape(40, 403))
    assert np.array_equal(out["logits"], ref["logits"]) and np.array_equal(out["softmax"], ref["softmax"])
    assert np.array_equal(out["argmax"], ref["argmax"])
    assert np.array_equal(c.cnn(net_golden["in_kws_small"])["softmax"], net_golden["acts_kws_small"][:, -10:])
    with pytest.raises(_lib.EdisonError) as e:                                   # the kws_conv dump layout is not this graph's
        c.cnn_layers(net_golden["in_kws_small"])
    assert e.value.code == _lib.E_SIZE
    # stream: window i of a push == the batch call on the same 31 frames
    st = Stream(c, hop=1024, chunk_frames=4)
    a = audio[:44 * 1024]
    soft = np.concatenate([st.push(a[i * 4096:(i + 1) * 4096])["softmax"] for i in range(11)])
    for i in range(30, 44):
        w = c.kws(a[(i - 30) * 1024:(i + 1) * 1024], n_utt=1, utt_stride=31 * 1024)
        assert np.array_equal(soft[i], w["softmax"][0])
    # a reload invalidates the device addresses the stream's graphs captured
    c.load_weights_h(_header("kws_small"))
    with pytest.raises(_lib.EdisonError):
        st.push(a[:4096])
    st.close()
    c.close()


def test_planner_refusals(built_lib):
    """Graphs the reference itself would run differently from the plain formula, or reject, are refused at load time."""
    from edison_amd import _lib, nnom_import
    from edison_amd.context import Context
    T_CONV, T_POOL, T_DENSE, T_SOFTMAX = 1, 2, 3, 4
    rng = np.random.default_rng(44)

    def conv(oc, kh, kw, sh, sw, cin, same=0):
        return dict(type=T_CONV, out_ch=oc, kh=kh, kw=kw, sh=sh, sw=sw, w=rng.integers(-9, 9, oc * kh * kw * cin).astype(np.int8),
                    b=np.zeros(oc, np.int8), out_rshift=7, bias_lshift=0, relu=1, same=same)

    def dense(no, ni):
        return dict(type=T_DENSE, out=no, w=rng.integers(-9, 9, no * ni).astype(np.int8), b=np.zeros(no, np.int8), out_rshift=7,
                    bias_lshift=0, relu=0)

    c = Context(0, model_path=None)
    cases = [
        # square image, non-square kernel: the reference's square kernels would use kernel.w for both axes
        ((12, 12, 1), [conv(4, 3, 5, 1, 1, 1)], _lib.E_NO_IMPL),
        # 1x1, C_in % 4 == 0, C_out % 2 == 0, stride 2: arm_convolve_1x1_HWC_q7_fast_nonsquare returns SIZE_MISMATCH
        ((10, 6, 4), [conv(2, 1, 1, 2, 2, 4)], _lib.E_SIZE),
        # Softmax in the middle
        ((10, 6, 1), [dict(type=T_SOFTMAX), dense(3, 60)], _lib.E_NO_IMPL),
        # activations beyond the two LDS buffers
        ((100, 60, 1), [conv(16, 3, 3, 1, 1, 1, same=1)], _lib.E_NO_IMPL),
        # kernel larger than the image
        ((4, 6, 1), [conv(2, 5, 3, 1, 1, 1)], _lib.E_SIZE),
    ]
    for shape, layers, code in cases:
        with pytest.raises(_lib.EdisonError) as e:
            c.load_model_bytes(nnom_import.build_blob(shape, layers))
        assert e.value.code == code, (shape, str(e.value))
    with pytest.raises(_lib.EdisonError):
        c.net_info()                                                              # nothing loaded after the refusals
    # a model that loads but is not 31x13x1 -> 10 cannot serve the fixed-shape entry points
    c.load_model_bytes(nnom_import.build_blob((10, 6, 1), [conv(4, 3, 3, 1, 1, 1), dense(3, 8 * 4 * 4)]))
    assert c.net(np.zeros((2, 60), np.int8))["logits"].shape == (2, 3)
    with pytest.raises(_lib.EdisonError) as e:
        c.cnn(np.zeros((1, 403), np.int8))
    assert e.value.code == _lib.E_SIZE
    # truncated blob
    blob = nnom_import.build_blob((10, 6, 1), [conv(4, 3, 3, 1, 1, 1)])
    with pytest.raises(_lib.EdisonError):
        c.load_model_bytes(blob[:-20])
    c.close()


def test_fused_pool_as_the_last_layer_and_multi_unit_groups(built_lib):
    """Shapes the random fuzz found or that exercise every (windows, units) instantiation of the matrix-core kernel:
    a MaxPool fused into the convolution in front of it that is ALSO the graph's last (logits) layer (round 2: the
    outputs were never written), with and without a Softmax behind it; row-tile pairs (C_out 64), four column tiles at
    once, a 2x2 window. Batch path (edison_net_batch_dev) against the numpy restatement, bit for bit."""
    from edison_amd import nnom_import
    from edison_amd.context import Context
    from oracle import net_ref
    rng = np.random.default_rng(5)

    def conv(oc, kh, kw, sh, sw, c, rs=8, bl=3, relu=1, same=0):
        return dict(type=1, out_ch=oc, kh=kh, kw=kw, sh=sh, sw=sw, w=rng.integers(-100, 101, oc * kh * kw * c).astype(np.int8),
                    b=rng.integers(-100, 101, oc).astype(np.int8), out_rshift=rs, bias_lshift=bl, relu=relu, same=same)

    def pool(kh, kw):
        return dict(type=2, kh=kh, kw=kw, sh=kh, sw=kw, same=0)

    cases = {
        "pool last": ((18, 10, 1), [conv(5, 1, 3, 1, 2, 1, 10, 5, 0, 1), conv(1, 4, 2, 2, 2, 5, 6, 4), pool(1, 2)]),
        "pool + softmax last": ((18, 10, 1), [conv(5, 1, 3, 1, 2, 1, 10, 5, 0, 1), conv(4, 4, 2, 2, 2, 5, 6, 4), pool(1, 2), dict(type=4)]),
        "row-tile pairs": ((9, 7, 16), [conv(64, 3, 3, 1, 1, 16, 9), conv(8, 3, 3, 1, 1, 64, 10)]),
        "four column tiles": ((16, 12, 3), [conv(8, 3, 3, 1, 1, 3, 9, same=1), conv(4, 3, 3, 1, 1, 8, 9)]),
        "2x2 window": ((14, 14, 2), [conv(16, 3, 3, 1, 1, 2, 9), pool(2, 2), conv(8, 3, 3, 1, 1, 16, 10)]),
    }
    c = Context(0, model_path=None)
    for name, (shape, layers) in cases.items():
        blob = nnom_import.build_blob(shape, [dict(L) for L in layers])
        c.load_model_bytes(blob)
        assert c.net_info()["accelerated"] == 2, name
        x = rng.integers(-128, 128, (77, shape[0] * shape[1] * shape[2])).astype(np.int8)
        ref, out = net_ref.run(blob, x), c.net(x)
        assert np.array_equal(out["logits"], ref["logits"]), name
        assert np.array_equal(out["argmax"], ref["argmax"]), name
        if layers[-1]["type"] == 4:
            assert np.array_equal(out["softmax"], ref["softmax"]), name
    c.close()


def test_every_requantisation_mode_of_the_matrix_core_kernel(built_lib):
    """The matrix-core kernel requantises through the high byte of sat16(v >> (rs - 8)) where the planner proved that exact
    (model_net_mm.c: ED_RUN_RS_HI) -- with no shift (rs = 8), a right shift (rs > 8) or a left one (rs < 8, small weights) -- and with a
    shift and two clamps per value otherwise (rs < 8 under large weights; a C_out that is no multiple of 4 stores bytes); with and
    without a ReLU, on 32 x 32 and on 16 x 16 tiles, with accumulators driven to both rails. All bit-equal to oracle/net_ref.py."""
    from edison_amd import nnom_import
    from edison_amd.context import Context
    from oracle import net_ref
    rng = np.random.default_rng(12)

    def conv(oc, k, c, rs, wmax, relu, bl=0):
        return dict(type=1, out_ch=oc, kh=k, kw=k, sh=1, sw=1, w=rng.integers(-wmax, wmax + 1, oc * k * k * c).astype(np.int8),
                    b=rng.integers(-100, 101, oc).astype(np.int8), out_rshift=rs, bias_lshift=bl, relu=relu, same=0)

    c = Context(0, model_path=None)
    n_cases = 0
    for rs in (0, 2, 5, 7, 8, 9, 12, 15):
        for wmax in (2, 127):
            for relu in (0, 1):
                for oc in (8, 6):                      # 6: byte stores
                    # layer 1 on 32 x 32 tiles (5 x 3 pixels x 2 inputs per wave), layer 2 on 16 x 16 tiles (3 x 1 pixels)
                    layers = [conv(64, 3, 16, 9, 60, 1, 2), conv(32, 3, 64, rs, wmax, relu, 3), conv(oc, 3, 32, rs, wmax, relu, 1)]
                    blob = nnom_import.build_blob((9, 7, 16), [dict(L) for L in layers])
                    c.load_model_bytes(blob)
                    assert c.net_info()["accelerated"] == 2, (rs, wmax, relu, oc)
                    x = rng.integers(-128, 128, (45, 9 * 7 * 16)).astype(np.int8)
                    x[:4] = np.array([127, -128, 127, -128], np.int8)[:, None]     # rails: saturation on both sides
                    ref, out = net_ref.run(blob, x), c.net(x)
                    assert np.array_equal(out["logits"], ref["logits"]), (rs, wmax, relu, oc)
                    assert np.array_equal(out["argmax"], ref["argmax"]), (rs, wmax, relu, oc)
                    n_cases += 1
    assert n_cases == 64
    c.close()
