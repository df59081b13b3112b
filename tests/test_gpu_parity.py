"""GPU parity tests (-m gpu): the HIP hot path, called through the C-ABI, against the committed golden vectors
(generated from the reference) and against the oracle on seeded inputs.

Tolerances (fp32 kernel vs the reference's float64): SURVEY.md A.1
    variant A  |d| <= 1e-3 + 1e-4*|ref|      variant B  |d| <= 1e-2 + 1e-5*|ref|
The int8 CNN (every layer, softmax, argmax) is compared bit for bit.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STREAMS = ["edison", "hey", "two_tone", "noise", "quiet", "extremes"]
TOL = {"A": (1e-3, 1e-4), "B": (1e-2, 1e-5)}


def _close(got, ref, variant):
    atol, rtol = TOL[variant]
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert not bad.any(), "max |d| %.3e at %s" % (np.abs(got - ref).max(), np.argwhere(bad)[:4].tolist())


def _variant(v):
    from edison_amd import _lib
    return _lib.MFCC_A if v == "A" else _lib.MFCC_B


# ---------------------------------------------------------------------------------------------- MFCC

@pytest.mark.parametrize("name", STREAMS)
@pytest.mark.parametrize("variant", ["A", "B"])
def test_mfcc_golden(ctx, mfcc_golden, name, variant):
    x = mfcc_golden["in_" + name]
    got = ctx.mfcc(x, variant=_variant(variant), n_coef=32)
    _close(got, mfcc_golden["%s_mfcc_%s" % (variant, name)], variant)


@pytest.mark.parametrize("variant", ["A", "B"])
def test_mfcc_stages_golden(ctx, mfcc_golden, variant):
    for name in ("edison", "two_tone", "extremes"):
        x = mfcc_golden["in_" + name]
        st = ctx.mfcc_stages(x, variant=_variant(variant))
        spec = mfcc_golden["%s_spec_%s" % (variant, name)]
        nb = 512 if variant == "A" else 513
        scale = np.abs(spec).max()
        assert np.abs(st["spectrogram"][:, :nb] - spec[:, :nb]).max() <= 2e-6 * scale
        mel = mfcc_golden["%s_mel_%s" % (variant, name)]
        assert np.abs(st["mel_spectrogram"] - mel).max() <= 2e-6 * np.abs(mel).max() + 1e-30
        n = x.shape[0] // 1024
        fft = np.fft.fft(x[:n * 1024].reshape(n, 1024).astype(np.float64))[:, :513]
        assert np.abs(st["fft"] - fft).max() <= 2e-6 * np.abs(fft).max() + 1e-30
        if variant == "A":
            lm = mfcc_golden["A_logmel_" + name]
            assert np.abs(st["log_mel_spectrogram"] - lm).max() <= 1e-4


def test_two_frame_kernel_agrees_with_one_frame_kernel(ctx, mfcc_golden):
    """The batched path runs ed_mfcc2_kernel (two frames per wavefront, packed fp32), the stage-dump path the one-frame
    ed_mfcc_kernel: same algorithm and operation order, different instruction selection (FMA contraction), so they
    agree to a few ulp -- and a frame's result must not depend on whether it rides in the .x or the .y half."""
    x = np.concatenate([mfcc_golden["in_noise"], mfcc_golden["in_edison"]])
    for variant in ("A", "B"):
        two = ctx.mfcc(x, variant=_variant(variant), n_coef=32)
        one = ctx.mfcc_stages(x, variant=_variant(variant))["mfcc"]
        assert np.abs(two - one).max() <= 2e-6 * np.abs(one).max()
        shifted = ctx.mfcc(x[1024:], variant=_variant(variant), n_coef=32)      # every frame changes its half
        assert np.array_equal(shifted, two[1:])
        odd = ctx.mfcc(x[:5 * 1024], variant=_variant(variant), n_coef=32)      # odd batch: the last pair repeats A
        assert np.array_equal(odd, two[:5])


def test_mfcc_vs_oracle_seeded(ctx, oracle_mod):
    rng = np.random.default_rng(20)
    t = np.arange(1024) / 16000.0
    frames = []
    for i in range(256):
        ph = rng.uniform(0, 2 * np.pi, 2)
        f = rng.normal(0, 3000, 1024) + 1000 * np.cos(2 * np.pi * 1000 * t + ph[0]) + 500 * np.cos(2 * np.pi * 125 * t + ph[1])
        frames.append(np.clip(f, -32768, 32767))
    x = np.concatenate(frames).astype(np.int16)
    for variant, ov in (("A", 0), ("B", 1)):
        got = ctx.mfcc(x, variant=_variant(variant), n_coef=32)
        _close(got, oracle_mod.mfcc(x, ov, n_threads=4), variant)


def test_mfcc_tf_variant_against_the_float64_restatement(ctx, built_lib, oracle_mod, mfcc_golden):
    """Variant TF = mfcc_utils.mfcc_tf (mfcc_utils.py:201-253), the TensorFlow curve of `main.py mfcc host`. PARITY UNPINNED:
    TensorFlow is not installed and the reference holds no output of it, so the GPU is held to the oracle's float64
    restatement of tf.signal's definitions (oracle.h), with variant A's bar (same logarithm and DCT scale). What IS pinned:
    the 513-bin mel matrix both use is the reference's own (golden mel_W513, test_host_cpu.py)."""
    from edison_amd import _lib
    from edison_amd.mfcc import mfcc_utils as mfu
    for name in ("edison", "two_tone", "noise", "extremes", "quiet"):
        x = mfcc_golden["in_" + name]
        ref, rst = oracle_mod.mfcc(x, oracle_mod.VARIANT_TF, stages=True)
        st = ctx.mfcc_stages(x, variant=_lib.MFCC_TF)
        # a band whose energy lies below the float32 FFT's rounding floor (a windowed constant frame has exact zeros outside
        # three bins in float64, ~1e-7 of the line in float32 -- TensorFlow's float32 FFT has such a floor of its own) has a
        # logarithm that is all rounding: those frames are held to the linear stages only
        ok = rst["mel_spectrogram"].min(axis=1) > 1e-5 * rst["spectrogram"].max(axis=1)
        assert ok.sum() >= (1 if name == "extremes" else len(ok) - 1), (name, ok)
        full = st
        st = {k: v[ok] for k, v in full.items()}
        rst, ref, lin, rlin = {k: v[ok] for k, v in rst.items()}, ref[ok], full, rst
        assert np.abs(lin["spectrogram"] - rlin["spectrogram"]).max() <= 2e-6 * np.abs(rlin["spectrogram"]).max() + 1e-30
        assert np.abs(lin["mel_spectrogram"] - rlin["mel_spectrogram"]).max() <= 2e-6 * np.abs(rlin["mel_spectrogram"]).max() + 1e-30
        _close(st["mfcc"], ref, "A")
        assert np.abs(st["spectrogram"] - rst["spectrogram"]).max() <= 2e-6 * np.abs(rst["spectrogram"]).max() + 1e-30
        assert np.abs(st["mel_spectrogram"] - rst["mel_spectrogram"]).max() <= 2e-6 * np.abs(rst["mel_spectrogram"]).max() + 1e-30
        # the window takes the leakage away, so quiet bands sit further below the strongest line than in variant A and carry
        # more of the float32 FFT's rounding floor (~1e-7 of that line) into their logarithm: 1.1e-4 measured on two_tone
        assert np.abs(st["log_mel_spectrogram"] - rst["log_mel_spectrogram"]).max() <= 5e-4
        # the batch entry points run the two-frame kernel's WINDOW instances (round 5; the stage dump is the one-frame kernel's): the same
        # operations in the same order per frame, but the compiler pairs multiplies and additions into fused operations differently in the two
        # texts, and behind a window the quiet bands' logarithms carry the float32 transform's rounding floor (above): the two kernels are held to
        # each other, and the batch to the oracle, with variant A's bar (1.3e-4 between the kernels measured on two_tone, 8e-6 on speech)
        b32 = ctx.mfcc(x, variant=_lib.MFCC_TF, n_coef=32)
        _close(b32[ok], full["mfcc"][ok], "A")
        _close(b32[ok], ref, "A")
        assert np.array_equal(ctx.mfcc(x, variant=_lib.MFCC_TF, n_coef=13), b32[:, :13])
    x = mfcc_golden["in_noise"]
    _close(ctx.mfcc(x, frame_step=333, variant=_lib.MFCC_TF, n_coef=32), oracle_mod.mfcc(x, oracle_mod.VARIANT_TF, frame_step=333), "A")
    _close(ctx.mfcc(x[1:], variant=_lib.MFCC_TF, n_coef=32), oracle_mod.mfcc(x[1:], oracle_mod.VARIANT_TF), "A")
    rows = np.stack([x[:4096], x[4096:8192]])
    got = ctx.mfcc_rows(rows, 4, variant=_lib.MFCC_TF, n_coef=32)
    assert np.array_equal(got[1], ctx.mfcc(x[4096:8192], variant=_lib.MFCC_TF, n_coef=32))
    # a window is not a scale: TF differs from A on the same frames, and a constant frame has no energy outside DC's neighbours
    assert np.abs(ctx.mfcc(x, variant=_lib.MFCC_TF, n_coef=32) - ctx.mfcc(x, variant=_lib.MFCC_A, n_coef=32)).max() > 0.1
    # the reference's structure (:239-252): DC cut from fft / spectrogram / matrix, float32 like TensorFlow's tensors
    xe = mfcc_golden["in_edison"]
    o = mfu.mfcc_tf(xe, 16000, len(xe), 1024, 1024, 0, 1024, 32, 80.0, 7600.0)
    assert len(o) == 10 and set(o[3]) == {"t_start", "t_end", "fft", "spectrogram", "mel_weight_matrix", "mel_spectrogram",
                                          "log_mel_spectrogram", "mfcc"}
    assert o[3]["fft"].shape == (512,) and o[3]["spectrogram"].shape == (512,) and o[3]["mel_weight_matrix"].shape == (512, 32)
    assert o[3]["mfcc"].shape == (32,) and o[3]["t_start"] == 3 * 1024 / 16000
    np.testing.assert_allclose(o[3]["mel_weight_matrix"], mfcc_golden["mel_W513"][1:], rtol=0, atol=1e-7)
    _close(np.array([f["mfcc"] for f in o]), oracle_mod.mfcc(xe, oracle_mod.VARIANT_TF), "A")
    with pytest.raises(NotImplementedError):
        mfu.mfcc_tf(xe, 16000, len(xe), 1024, 1024, 0, 2048, 32, 80.0, 7600.0)
    # always a logarithm (no use_log flag), no stream stage, no one-launch microphone push
    with pytest.raises(_lib.EdisonError):
        ctx.mfcc(xe, variant=_lib.MFCC_TF, use_log=True)
    L, o_, h = _lib.lib(), _lib.StreamOpts(), ctypes.c_void_p()
    L.edison_stream_default_opts(ctypes.byref(o_))
    o_.mfcc_variant = _lib.MFCC_TF
    assert L.edison_stream_create_ex(ctx._h, ctypes.byref(o_), ctypes.byref(h)) == _lib.E_ARGUMENT


def test_mfcc_overlap_unaligned_ncoef_log(ctx, oracle_mod, mfcc_golden):
    x = mfcc_golden["in_noise"]
    from edison_amd import _lib
    got = ctx.mfcc(x[:4096], frame_step=512, variant=_lib.MFCC_B, n_coef=32)        # 50 % overlap
    _close(got, mfcc_golden["B_mfcc_overlap512"], "B")
    got = ctx.mfcc(x, frame_step=333, variant=_lib.MFCC_B, n_coef=32)               # odd hop -> 2-byte aligned frames
    _close(got, oracle_mod.mfcc(x, 1, frame_step=333), "B")
    got = ctx.mfcc(x[1:], frame_step=1024, variant=_lib.MFCC_A, n_coef=32)          # odd base offset
    _close(got, oracle_mod.mfcc(x[1:], 0), "A")
    full = ctx.mfcc(x, variant=_lib.MFCC_B, n_coef=32)
    for nc in (1, 13, 31):
        assert np.array_equal(ctx.mfcc(x, variant=_lib.MFCC_B, n_coef=nc), full[:, :nc])
    got = ctx.mfcc(mfcc_golden["in_edison"], variant=_lib.MFCC_B, n_coef=32, use_log=True)
    _close(got, mfcc_golden["Blog_mfcc_edison"], "A")


def test_mfcc_other_filterbanks_and_wide_tables(built_lib, oracle_mod, mfcc_golden, monkeypatch):
    """gen_mel_weight_matrix takes the band edges and the sample rate as arguments (mfcc_utils.py:36); the device
    tap tables are rebuilt by edison_mfcc_configure. Also runs the shipped filterbank through the kernel instance
    compiled for the larger (3+6 quads per lane) table shape, which the shipped edges never select on their own."""
    from edison_amd import _lib
    from edison_amd.context import Context
    x = np.concatenate([mfcc_golden["in_edison"], mfcc_golden["in_noise"]])
    c = Context(0)
    try:
        for fs, lo, hi in ((16000, 20.0, 8000.0), (16000, 300.0, 3400.0), (8000, 80.0, 3800.0), (44100, 80.0, 7600.0)):
            c.configure_mfcc(fs, lo, hi)
            for variant, ov in (("A", 0), ("B", 1)):
                got = c.mfcc(x, variant=_variant(variant), n_coef=32)
                ref = oracle_mod.mfcc(x, ov, sample_rate=fs, lower_edge_hertz=lo, upper_edge_hertz=hi, n_threads=4)
                _close(got, ref, variant)
    finally:
        c.close()
    monkeypatch.setenv("EDISON_FORCE_WIDE_MEL", "1")
    c = Context(0)
    try:
        for variant in ("A", "B"):
            got = c.mfcc(mfcc_golden["in_edison"], variant=_variant(variant), n_coef=32)
            _close(got, mfcc_golden["%s_mfcc_edison" % variant], variant)
    finally:
        c.close()


def test_mfcc_empty_and_ragged(ctx, built_lib):
    from edison_amd import _lib
    assert ctx.mfcc(np.zeros(0, np.int16)).shape == (0, 32)
    assert ctx.mfcc(np.zeros(1023, np.int16)).shape == (0, 32)          # shorter than one frame: zero frames, like frames()
    assert ctx.mfcc(np.zeros(2047, np.int16)).shape == (1, 32)          # ragged tail ignored
    with pytest.raises(ValueError):
        ctx.mfcc(np.zeros(1024, np.int16), n_frames=2)
    with pytest.raises(_lib.EdisonError):
        ctx.mfcc(np.zeros(1024, np.int16), n_coef=33)
    with pytest.raises(_lib.EdisonError):
        ctx.mfcc(np.zeros(1024, np.int16), variant=7)
    z = ctx.mfcc(np.zeros(1024, np.int16), variant=_lib.MFCC_A, n_coef=32)   # silence is legal: ln(0 + 1e-6)
    assert np.isfinite(z).all() and abs(z[0, 0] - 32 * 2 * np.log(1e-6) / 8) < 1e-3


def test_net_input_features(ctx, oracle_mod, kws_golden):
    """int8 net input = round_half_even(clip(mfcc[:13])) (kws_nnom.py:359-361)."""
    from edison_amd import _lib
    for mode in ("zero", "edge"):
        a = kws_golden["kws_%s_audio" % mode]
        _, feat = ctx.mfcc(a, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
        assert np.array_equal(feat, kws_golden["kws_%s_feat" % mode])
    rng = np.random.default_rng(5)
    x = np.clip(rng.normal(0, 2500, 64 * 31 * 1024), -32768, 32767).astype(np.int16)
    m, feat = ctx.mfcc(x, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
    # the int8 written by the kernel is exactly the rounding of the fp32 it wrote
    assert np.array_equal(feat, np.clip(m, -128, 127).round().astype(np.int8))
    ref = oracle_mod.net_input(oracle_mod.mfcc(x, 1, n_threads=4)[:, :13])
    diff = np.abs(ref.astype(int) - feat.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3     # fp32-vs-fp64 may flip a value sitting on x.5


# ---------------------------------------------------------------------------------------------- CNN

def test_cnn_golden_layers(ctx, cnn_golden):
    lay = ctx.cnn_layers(cnn_golden["feats"])
    for k in ("conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "dense", "softmax"):
        assert np.array_equal(lay[k], cnn_golden[k]), k
    r = ctx.cnn(cnn_golden["feats"])
    assert np.array_equal(r["logits"], cnn_golden["dense"])
    assert np.array_equal(r["softmax"], cnn_golden["output"])
    assert np.array_equal(r["argmax"], cnn_golden["argmax"])


def test_cnn_vs_oracle_random(ctx, oracle_mod, oracle_model):
    rng = np.random.default_rng(123)
    f = np.concatenate([rng.integers(-128, 128, (12000, 403)),
                        np.clip(rng.normal(0, 25, (6000, 403)), -128, 127).round(),
                        np.clip(rng.normal(60, 80, (2000, 403)), -128, 127).round()]).astype(np.int8)
    g = ctx.cnn(f)
    o = oracle_mod.cnn(oracle_model, f, n_threads=8)
    for k in ("logits", "softmax", "argmax"):
        assert np.array_equal(g[k], o[k]), k
    # ties and saturated softmax rows are in the sample (dense random inputs drive class 8 hard)
    assert (g["softmax"] == 127).any() and (g["softmax"] == 0).any()


def test_cnn_vs_reference_build(ctx, oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref not present")
    rng = np.random.default_rng(321)
    f = rng.integers(-128, 128, (1500, 403)).astype(np.int8)
    g, r = ctx.cnn(f), oracle_mod.nnom_ref_batch(f)
    for k in ("logits", "softmax", "argmax"):
        assert np.array_equal(g[k], r[k]), k


def test_cnn_edge_sizes(ctx, cnn_golden):
    assert ctx.cnn(np.zeros((0, 403), np.int8))["argmax"].shape == (0,)
    r = ctx.cnn(cnn_golden["feats"][:1])
    assert r["logits"][0].tolist() == [-6, -5, -17, -15, -18, -11, -4, 2, -4, 8] and r["argmax"][0] == 9
    # ties in the softmax output resolve to the FIRST maximum (nnom_utils.c:275-284)
    r = ctx.cnn(cnn_golden["feats"])
    for sm, am in zip(r["softmax"], r["argmax"]):
        assert am == int(np.argmax(sm))


# ---------------------------------------------------------------------------------------------- KWS

def test_kws_golden_wav(ctx, kws_golden):
    for mode in ("zero", "edge"):
        r = ctx.kws(kws_golden["kws_%s_audio" % mode], n_utt=1)
        assert np.array_equal(r["feat"].reshape(31, 13), kws_golden["kws_%s_feat" % mode])
        assert np.array_equal(r["logits"][0], kws_golden["kws_%s_logits" % mode])
        assert np.array_equal(r["softmax"][0], kws_golden["kws_%s_softmax" % mode])
        assert r["argmax"][0] == kws_golden["kws_%s_argmax" % mode] == 0      # "edison"


def test_kws_batch_vs_oracle(ctx, oracle_mod, oracle_model):
    rng = np.random.default_rng(21)
    n = 96
    parts = [np.clip(rng.normal(0, 3000, (n // 3, 32000)), -32768, 32767),
             np.clip(rng.normal(0, 0.01 * 32767, (n // 3, 32000)), -32768, 32767),
             np.zeros((n // 3, 32000))]
    audio = np.concatenate(parts).astype(np.int16)
    for stride, a in ((32000, audio), (31744, np.ascontiguousarray(audio[:, :31744]))):
        r = ctx.kws(a.reshape(-1), n_utt=n, utt_stride=stride)
        o = oracle_mod.cnn(oracle_model, r["feat"], n_threads=4)     # CNN exactness on the features the GPU made
        for k in ("logits", "softmax", "argmax"):
            assert np.array_equal(r[k], o[k]), k
        ref_feat = np.stack([oracle_mod.net_input(oracle_mod.mfcc(a[u], 1)[:, :13]).reshape(-1) for u in range(n)])
        d = np.abs(ref_feat.astype(int) - r["feat"].astype(int))
        assert d.max() <= 1 and (d != 0).sum() <= 4
        ro = oracle_mod.cnn(oracle_model, ref_feat, n_threads=4)
        assert (ro["argmax"] != r["argmax"]).sum() <= 1          # argmax flips only via a flipped feature


# ---------------------------------------------------------------------------------------------- legacy C surface

def test_legacy_ai_surface(built_lib, ctx, cnn_golden):
    L = built_lib
    assert L.aiInitialize() == 0
    out = np.zeros(10, np.int8)
    for i in (0, 1, 3, 20, 57):
        f = np.ascontiguousarray(cnn_golden["feats"][i])
        assert L.aiRunInference(f.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(out, cnn_golden["output"][i])
        label, prob = ctypes.c_uint32(), ctypes.c_float()
        assert L.aiNnomPredict(ctypes.byref(label), ctypes.byref(prob)) == 0    # runs on the static input buffer
        assert label.value == cnn_golden["argmax"][i]
        s = int(cnn_golden["output"][i].astype(int).sum())
        assert abs(prob.value - cnn_golden["output"][i].max() / s) < 1e-6
    ob = np.frombuffer((ctypes.c_int8 * 10).from_address(L.aiNnomGetOutputBuffer()), dtype=np.int8)
    assert np.array_equal(ob, cnn_golden["output"][57])


def test_legacy_frame_and_push_flow(built_lib, ctx, kws_golden):
    """app.c's streaming flow on the host: per frame MFCC -> mfccToNetInputPush -> aiRunInference."""
    L = built_lib
    assert L.aiInitialize() == 0
    a = kws_golden["kws_zero_audio"]
    out32 = np.zeros(32, np.float32)
    for i in range(31):
        fr = np.ascontiguousarray(a[i * 1024:(i + 1) * 1024])
        assert L.edison_mfcc_frame(fr.ctypes.data_as(ctypes.c_void_p), 1, out32.ctypes.data_as(ctypes.c_void_p)) == 0
        m16 = np.clip(np.round(out32[:13]), -32768, 32767).astype(np.int16)
        L.mfccToNetInputPush(m16.ctypes.data_as(ctypes.c_void_p), 13, 31)
    buf = np.frombuffer((ctypes.c_int8 * 403).from_address(L.aiNnomGetInputBuffer()), dtype=np.int8).copy()
    assert np.array_equal(buf.reshape(31, 13), kws_golden["kws_zero_feat"])
    out = np.zeros(10, np.int8)
    assert L.aiRunInference(buf.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(out, kws_golden["kws_zero_softmax"])


def test_legacy_audio_calc_mfccs_flow(built_lib, ctx, kws_golden, oracle_mod, oracle_model):
    """app.c:190-203 with the firmware's own names: audioCalcMFCCs -> mfccToNetInput -> aiRunInference. The int16
    coefficients are the firmware's Q15 arithmetic (variant C), bit for bit against the oracle."""
    L = built_lib
    L.audioInit()
    a = kws_golden["kws_zero_audio"]
    out = ctypes.c_void_p()
    rows = []
    for i in range(31):
        fr = np.ascontiguousarray(a[i * 1024:(i + 1) * 1024])
        L.audioCalcMFCCs(fr.ctypes.data_as(ctypes.c_void_p), ctypes.byref(out))
        m16 = np.frombuffer((ctypes.c_int16 * 32).from_address(out.value), dtype=np.int16).copy()
        rows.append(m16)
        L.mfccToNetInput(m16.ctypes.data_as(ctypes.c_void_p), 13, 31, i)
    rows = np.stack(rows)
    ref = oracle_mod.mfcc_q15(a[:31 * 1024])
    assert np.array_equal(rows, ref)
    net_in = np.frombuffer((ctypes.c_int8 * 403).from_address(L.aiNnomGetInputBuffer()), dtype=np.int8).copy()
    assert np.array_equal(net_in.reshape(31, 13), oracle_mod.net_input_q15(ref))
    res = np.zeros(10, np.int8)
    assert L.aiRunInference(net_in.ctypes.data_as(ctypes.c_void_p), res.ctypes.data_as(ctypes.c_void_p)) == 0
    o = oracle_mod.cnn(oracle_model, net_in.reshape(1, 403))
    assert np.array_equal(res, o["softmax"][0])
    assert int(np.argmax(res)) == 0                          # "edison", as the board answers (README.md:124-125)


# ---------------------------------------------------------------------------------------------- Python mirror / CLI

def test_python_mirror_mfcc_utils(ctx, mfcc_golden):
    from edison_amd.mfcc import mfcc_utils as mfu
    x = mfcc_golden["in_edison"]
    o = mfu.mfcc(x, 16000, len(x), 1024, 1024, 0, 1024, 32, 80.0, 7600.0)
    assert len(o) == 10 and set(o[0]) == {"t_start", "t_end", "fft", "spectrogram", "mel_weight_matrix",
                                          "mel_spectrogram", "log_mel_spectrogram", "mfcc"}
    assert o[3]["t_start"] == 3 * 1024 / 16000 and o[3]["fft"].shape == (512,) and o[3]["mel_weight_matrix"].shape == (512, 32)
    _close(np.array([f["mfcc"] for f in o]), mfcc_golden["A_mfcc_edison"], "A")
    ob = mfu.mfcc_mcu(x, 16000, len(x), 1024, 1024, 0, 1024, 32, 80.0, 7600.0, 128)
    assert ob[0]["fft"].shape == (1024,) and ob[0]["spectrogram"].shape == (1024,) and ob[0]["mel_weight_matrix"].shape == (513, 32)
    _close(np.array([f["mfcc"] for f in ob]), mfcc_golden["B_mfcc_edison"], "B")
    sp = np.array([f["spectrogram"] for f in ob])
    assert np.abs(sp - mfcc_golden["B_spec_edison"]).max() <= 2e-6 * mfcc_golden["B_spec_edison"].max()
    assert np.array_equal(ob[2]["log_mel_spectrogram"], ob[2]["mel_spectrogram"])     # use_log=False: same array content
    b = mfu.batch_mfcc(mfcc_golden["batch_in"], 16000, 2048, 1024, 1024, 0, 1024, 32, 80.0, 7600.0)
    assert b.shape == (4, 2, 32)
    _close(b, mfcc_golden["batch_out"], "A")
    # reconfiguring the filterbank goes through edison_mfcc_configure and back
    o2 = mfu.mfcc_mcu(x, 16000, len(x), 1024, 1024, 0, 1024, 32, 300.0, 6000.0, 128)
    assert np.abs(np.array([f["mfcc"] for f in o2]) - mfcc_golden["B_mfcc_edison"]).max() > 1.0
    o3 = mfu.mfcc_mcu(x, 16000, len(x), 1024, 1024, 0, 1024, 32, 80.0, 7600.0, 128)
    _close(np.array([f["mfcc"] for f in o3]), mfcc_golden["B_mfcc_edison"], "B")


def test_cli_entry_points(ctx, kws_golden, mfcc_golden, tmp_path, capsys):
    import scipy.io.wavfile as wavfile
    from edison_amd import main as cli
    wav = str(tmp_path / "edison_16k_16b.wav")
    wavfile.write(wav, 16000, mfcc_golden["in_edison"])
    assert cli.main(["main.py", "mfcc", "host", wav]) == 0
    out = capsys.readouterr().out
    assert "Number of input samples = 11243" in out and "(4, 10, 13)" in out   # own, tf, mcu, mcu log (mfcc.py:209-216)
    assert cli.main(["main.py", "kws", "mcu", "fileinf", wav]) == 0
    out = capsys.readouterr().out
    assert "edison" in out.splitlines()[-2]
    assert cli.main(["main.py", "kws", "mcu", "frame", wav]) == 0
    capsys.readouterr()
    # `kws mcu file` = frameInference: host leg (variant B) vs the board's leg (variant C on the GPU). The reference
    # prints this block for the same wav in README.md:121-139; rmse / scale / correlation of the MFCC block must agree
    # (the deviation extremes depend on the last bits of the float32 host values and are pinned in test_gpu_q15.py).
    assert cli.main(["main.py", "kws", "mcu", "file", wav]) == 0
    out = capsys.readouterr().out
    assert out.count("edison") >= 3 and "Comparing: predictions" in out
    blk = out.split("Comparing: MFCC=net input")[1]
    assert "rmse 2.036" in blk and "scale 0.959=1/1.043" in blk and "correlation coeff 0.997" in blk
    assert cli.main(["main.py", "kws", "mcu", "single"]) == 0
    assert "_noise" in capsys.readouterr().out               # all-zero input: class 9 (SURVEY 8c known answer)
    assert cli.main(["main.py", "mfcc", "mcu", "single"]) == 0
    out = capsys.readouterr().out
    assert all(k in out for k in ("host/mcu fft scale", "host/mcu spectrum scale", "host/mcu mel spectrum scale", "host/mcu dct scale"))
    assert cli.main(["main.py", "mfcc", "mcu", "file", wav]) == 0
    assert "correlation coeff 0.99" in capsys.readouterr().out


def test_c_program_links_against_the_abi(built_lib, ctx, kws_golden, tmp_path):
    """A plain C translation unit written with the reference's own function names builds against include/edison_hip.h,
    links libedison_hip.so and classifies the reference wav (examples/host_kws.c)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler on this box")
    exe = str(tmp_path / "host_kws")
    libdir = os.path.join(root, "edison_amd", "csrc")
    subprocess.check_call([cc, "-O1", "-Wall", os.path.join(root, "examples", "host_kws.c"), "-I", os.path.join(root, "include"),
                           "-L", libdir, "-ledison_hip", "-Wl,-rpath," + libdir, "-o", exe])
    pcm = str(tmp_path / "edison.pcm")
    kws_golden["kws_zero_audio"].astype("<i2").tofile(pcm)
    out = subprocess.run([exe, pcm], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "firmware-style: edison" in out.stdout and "batched:        edison" in out.stdout
    assert "AI net information" in out.stdout and "last inference time:" in out.stdout and "#8 Softmax" in out.stdout


def test_c_host_runs_independent_batches_three_ways(built_lib, ctx):
    """examples/host_mfcc_pipeline.c: independent batches from a plain C host -- one call per batch, ONE
    edison_mfcc_batches_dev call for the list, one call per batch alternating over two HIP streams of different priority -- must
    agree bit for bit (the program exits 3 otherwise) and print one JSON line with the three rates."""
    import json
    import os
    import subprocess
    from edison_amd import build as edbuild
    edbuild.build_examples()
    exe = os.path.join(edbuild.EXAMPLES_BIN, "host_mfcc_pipeline")
    out = subprocess.run([exe, "5", "8191", "3"], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["list_equals_serial"] is True and d["two_queues_equals_serial"] is True and d["batches"] == 5 and d["frames_per_batch"] == 8191


def test_batch_mfcc_rows_in_one_launch(ctx, oracle_mod):
    """edison_mfcc_rows (the mirror of batch_mfcc, mfcc_utils.py:75-131) handles all rows with ONE launch through the
    kernel's grouped addressing: the result must equal the per-row calls bit for bit -- on the reference's config-2
    shape (rows of exactly one frame: every frame pair straddles two rows) and on rows of several overlapping frames
    with slack behind the last frame -- and the oracle within the variant-A tolerance."""
    from edison_amd import _lib
    rng = np.random.default_rng(75)
    data = np.clip(rng.normal(0, 3000, (4096, 1024)), -32768, 32767).astype(np.int16)
    got = ctx.mfcc_rows(data, 1, frame_step=1024, variant=_lib.MFCC_A, n_coef=32)
    assert got.shape == (4096, 1, 32)
    flat = ctx.mfcc(data.reshape(-1), n_frames=4096, frame_step=1024, variant=_lib.MFCC_A, n_coef=32)
    assert np.array_equal(got[:, 0, :], flat)              # rows of one frame = the plain batch
    for r in (0, 1, 2047, 4095):
        assert np.array_equal(got[r, 0], ctx.mfcc(data[r], n_frames=1, variant=_lib.MFCC_A, n_coef=32)[0])
    ref = oracle_mod.mfcc(data[:64].reshape(-1), oracle_mod.VARIANT_A)
    assert np.all(np.abs(got[:64, 0, :] - ref) <= 1e-3 + 1e-4 * np.abs(ref))
    # odd number of rows, 5 frames per row at hop 512 inside rows of 3100 samples (28 unused samples per row)
    data = np.clip(rng.normal(0, 3000, (37, 3100)), -32768, 32767).astype(np.int16)
    got = ctx.mfcc_rows(data, 5, frame_step=512, variant=_lib.MFCC_B, n_coef=13)
    for r in range(37):
        assert np.array_equal(got[r], ctx.mfcc(data[r], n_frames=5, frame_step=512, variant=_lib.MFCC_B, n_coef=13)), r
    # the Python mirror goes through it
    from edison_amd.mfcc import mfcc_utils as mfu
    out = mfu.batch_mfcc(data[:3, :2048], 16000, 2048, 1024, 1024, 0, 1024, 32, 80, 7600)
    assert out.shape == (3, 2, 32) and out.dtype == np.float64
    assert np.array_equal(out[1].astype(np.float32), ctx.mfcc(data[1, :2048], n_frames=2, variant=_lib.MFCC_A, n_coef=32))


@pytest.mark.parametrize("shifts", [(8, 8, 7, 9), (7, 9, 8, 8), (9, 8, 6, 10), (8, 7, 7, 9), (10, 10, 9, 7)])
def test_retrained_kws_conv_with_other_output_shifts_on_the_matrix_core_kernel(shifts):
    """The hand-written CNN kernel serves any retrained model of the kws_conv shape. Round 4 gave conv1 / conv2 a shorter
    requantisation for an output shift of exactly 8 (the shipped model) and kept the general form for every other shift: both
    instantiations, every layer's shift changed in turn, ragged batches (partial groups of 1-3 utterances take the natural column
    order, full groups the conflict-free one) -- logits, softmax and argmax bit for bit against the numpy oracle of the same blob
    (oracle/net_ref.py, itself pinned on the reference NNoM for every fixture graph). Same weights, other shifts: what a retraining
    that re-quantises the activations differently would ship."""
    import struct
    from edison_amd import _lib
    from edison_amd.context import Context
    from oracle import net_ref
    blob = bytearray(open(_lib.DEFAULT_MODEL, "rb").read())
    (h, w, c), recs, _ = net_ref.parse_blob(bytes(blob))
    conv = [i for i, r in enumerate(recs) if r[0] == net_ref.T_CONV]
    assert len(conv) == 4
    for i, rs in zip(conv, shifts):
        struct.pack_into("<i", blob, 40 + 48 * i + 4 * 7, rs)            # out_rshift of the layer
    blob = bytes(blob)
    c2 = Context(0, model_path=None)
    c2.load_model_bytes(blob)
    assert c2.net_info()["accelerated"] == 1                           # the hand-written kernel, not the general one
    rng = np.random.default_rng(sum(shifts))
    for n in (1, 2, 3, 4, 5, 7, 33, 4099):
        x = rng.integers(-128, 128, (n, 403)).astype(np.int8)
        x[0] = 127
        x[-1] = -128
        got = c2.cnn(x)
        ref = net_ref.run(blob, x)
        assert np.array_equal(got["logits"], ref["logits"]), (shifts, n)
        assert np.array_equal(got["softmax"], ref["softmax"]) and np.array_equal(got["argmax"], ref["argmax"]), (shifts, n)
    c2.close()


def test_dataset_features_equal_the_per_utterance_loop(ctx, oracle_mod, kws_golden):
    """edison_amd.kws.features.dataset_features = the MFCC leg of the reference's load_data (kws_keras.py:443-468: mfcc_mcu per
    utterance, [first_mfcc : first_mfcc + num_mfcc], * net_input_scale, clip -- floats, not rounded --, channel axis) for a whole data
    set in ONE launch: against the oracle's float64 variant B utterance by utterance (variant B's fp32 bar, where nothing is clipped;
    clipped entries equal the bound), and rounded it is the int8 net input of the KWS path."""
    from edison_amd.kws.features import dataset_features
    rng = np.random.default_rng(31)
    x = np.clip(rng.normal(0, 2500, (7, 32000)), -32768, 32767).astype(np.int16)
    x[3] = kws_golden["kws_zero_audio"][:32000]
    x[5] = 0
    f = dataset_features(x, ctx=ctx)
    assert f.shape == (7, 31, 13, 1) and f.dtype == np.float64
    for u in range(7):
        ref = oracle_mod.mfcc(x[u, :31 * 1024], oracle_mod.VARIANT_B)[:, :13]
        want = np.clip(ref * 1.0, -128, 127)
        got = f[u, :, :, 0]
        inside = (ref > -128) & (ref < 127)
        assert np.all(np.abs(got - want)[inside] <= 1e-2 + 1e-5 * np.abs(ref)[inside]) and np.all(got[~inside] == want[~inside]), u
    r = ctx.kws(x.reshape(-1), n_utt=7, utt_stride=32000)
    assert np.abs(np.round(f[..., 0]).astype(int) - r["feat"].reshape(7, 31, 13).astype(int)).max() <= 1   # x.5 cases may round the other way
    # a sub-range of coefficients and the logarithm
    g = dataset_features(x[:2], first_mfcc=2, num_mfcc=5, use_mfcc_log=True, ctx=ctx)
    ref = oracle_mod.mfcc(x[1, :31 * 1024], oracle_mod.VARIANT_B, use_log=True)[:, 2:7]
    assert g.shape == (2, 31, 5, 1) and np.abs(g[1, :, :, 0] - np.clip(ref, -128, 127)).max() <= 2e-3
