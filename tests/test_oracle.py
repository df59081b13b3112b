"""CPU tests: the oracle (oracle/) against the golden vectors generated from the reference itself."""
import numpy as np
import pytest

STREAMS = ["edison", "hey", "two_tone", "noise", "quiet", "extremes"]


@pytest.mark.parametrize("name", STREAMS)
@pytest.mark.parametrize("variant", ["A", "B"])
def test_oracle_mfcc_matches_reference(oracle_mod, mfcc_golden, name, variant):
    x = mfcc_golden["in_" + name]
    got = oracle_mod.mfcc(x, 0 if variant == "A" else 1)
    ref = mfcc_golden["%s_mfcc_%s" % (variant, name)]
    assert got.shape == ref.shape
    # float64 restatement vs float64 numpy/scipy: agreement to rounding (ln() amplifies near-zero mel energies)
    np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-9)


def test_oracle_mfcc_stages(oracle_mod, mfcc_golden):
    for name in ("edison", "two_tone", "extremes"):
        x = mfcc_golden["in_" + name]
        _, st = oracle_mod.mfcc(x, 0, stages=True)
        np.testing.assert_allclose(st["spectrogram"], mfcc_golden["A_spec_" + name], rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(st["mel_spectrogram"], mfcc_golden["A_mel_" + name], rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(st["log_mel_spectrogram"], mfcc_golden["A_logmel_" + name], rtol=1e-7, atol=1e-9)
        _, st = oracle_mod.mfcc(x, 1, stages=True)
        np.testing.assert_allclose(st["spectrogram"], mfcc_golden["B_spec_" + name], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(st["mel_spectrogram"], mfcc_golden["B_mel_" + name], rtol=1e-9, atol=1e-9)


def test_oracle_mfcc_log_overlap_batch(oracle_mod, mfcc_golden):
    got = oracle_mod.mfcc(mfcc_golden["in_edison"], 1, use_log=True)
    np.testing.assert_allclose(got, mfcc_golden["Blog_mfcc_edison"], rtol=1e-9, atol=1e-9)
    got = oracle_mod.mfcc(mfcc_golden["in_noise"][:4096], 1, frame_step=512)
    np.testing.assert_allclose(got, mfcc_golden["B_mfcc_overlap512"], rtol=1e-9, atol=1e-9)
    b = mfcc_golden["batch_in"]
    got = np.stack([oracle_mod.mfcc(row, 0) for row in b])
    np.testing.assert_allclose(got, mfcc_golden["batch_out"], rtol=1e-7, atol=1e-9)


def test_oracle_mel_matrix(oracle_mod, mfcc_golden):
    for key, args in (("mel_W512", (32, 512, 16000, 80.0, 7600.0)), ("mel_W513", (32, 513, 16000, 80.0, 7600.0)),
                      ("mel_W129_20", (20, 129, 8000, 125.0, 3800.0))):
        W = oracle_mod.mel_weight_matrix(*args)
        np.testing.assert_allclose(W, mfcc_golden[key], rtol=0, atol=1e-13)
    # structure the GPU kernel relies on: 920 / 922 non-zeros, DC row zero
    assert (mfcc_golden["mel_W512"] != 0).sum() == 920 and (mfcc_golden["mel_W513"] != 0).sum() == 922
    assert not mfcc_golden["mel_W513"][0].any()


def test_mfcc_tf_restatement_against_numpy_and_scipy(oracle_mod, mfcc_golden):
    """oracle VARIANT_TF (mfcc_utils.mfcc_tf, mfcc_utils.py:201-253) is PARITY UNPINNED: TensorFlow is not installed here and
    the reference holds no output of it. This test only shows that the C restatement computes what tf.signal documents --
    periodic Hann window, rfft, |.|, the 513-bin mel matrix (pinned: the reference's own, golden mel_W513), ln(x + 1e-6),
    DCT-II * rsqrt(2 * 32) -- by recomputing it with numpy.fft and scipy.fft."""
    import scipy.fft
    x = np.concatenate([mfcc_golden["in_edison"], mfcc_golden["in_noise"][:3072], mfcc_golden["in_extremes"][:2048]])
    n = x.shape[0] // 1024
    fr = x[:n * 1024].reshape(n, 1024).astype(np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(1024) / 1024.0)
    spec = np.abs(np.fft.rfft(fr * win, axis=1))
    mel = spec @ mfcc_golden["mel_W513"]
    lm = np.log(mel + 1e-6)
    want = scipy.fft.dct(lm, type=2, axis=1) / np.sqrt(2.0 * 32)
    got, st = oracle_mod.mfcc(x, oracle_mod.VARIANT_TF, stages=True)
    assert st["spectrogram"].shape == (n, 513)
    np.testing.assert_allclose(st["spectrogram"], spec, rtol=0, atol=1e-9 * spec.max())
    np.testing.assert_allclose(st["mel_spectrogram"], mel, rtol=0, atol=1e-9 * mel.max())
    np.testing.assert_allclose(st["log_mel_spectrogram"], lm, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-6)
    assert np.abs(got - oracle_mod.mfcc(x, oracle_mod.VARIANT_A)).max() > 0.1   # a windowed frame is not variant A's frame


def test_reference_known_answers(mfcc_golden, cnn_golden, kws_golden):
    """The vectors SURVEY.md 8(c) recorded from the reference, re-checked on the committed fixtures."""
    np.testing.assert_allclose(mfcc_golden["A_mfcc_edison"][3][:3], [97.5289613, 2.47939590, 0.516462789], rtol=1e-8)
    np.testing.assert_allclose(mfcc_golden["B_mfcc_edison"][3][:3], [184.2785055, 65.51893126, 34.81352503], rtol=1e-8)
    assert cnn_golden["dense"][0].tolist() == [-6, -5, -17, -15, -18, -11, -4, 2, -4, 8]
    assert cnn_golden["softmax"][0].tolist() == [0, 0, 0, 0, 0, 0, 0, 3, 0, 127] and cnn_golden["argmax"][0] == 9
    assert cnn_golden["dense"][1].tolist() == [-47, -46, -26, -13, -42, -44, -55, -44, 89, -53] and cnn_golden["argmax"][1] == 8
    assert cnn_golden["dense"][2].tolist() == [-81, -42, -54, -36, -36, -1, -78, -25, 81, -58] and cnn_golden["argmax"][2] == 8
    assert cnn_golden["dense"][3].tolist() == [-46, -53, -33, -24, -50, -15, -57, -31, 76, -45]
    assert kws_golden["kws_zero_logits"].tolist() == [28, -13, -11, -19, -26, -21, -1, -7, -4, -12]
    assert kws_golden["kws_zero_softmax"].tolist() == [127, 0, 0, 0, 0, 0, 0, 0, 0, 0] and kws_golden["kws_zero_argmax"] == 0
    assert kws_golden["kws_edge_argmax"] == 0


def test_oracle_cnn_matches_reference_layers(oracle_mod, oracle_model, cnn_golden):
    r = oracle_mod.cnn(oracle_model, cnn_golden["feats"], want_acts=True)
    names = ["conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "dense", "softmax"]
    for n, a in zip(names, r["acts"]):
        assert np.array_equal(a, cnn_golden[n]), n
    assert np.array_equal(r["logits"], cnn_golden["dense"])
    assert np.array_equal(r["softmax"], cnn_golden["softmax"]) and np.array_equal(r["softmax"], cnn_golden["output"])
    assert np.array_equal(r["argmax"], cnn_golden["argmax"])


def test_oracle_cnn_matches_reference_build(oracle_mod, oracle_model):
    """Restatement vs the reference NNoM/CMSIS-NN build itself (oracle/_ref) on fresh random inputs."""
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(11)
    f = np.concatenate([rng.integers(-128, 128, (300, 403)), np.clip(rng.normal(0, 30, (200, 403)), -128, 127).round()]).astype(np.int8)
    a, b = oracle_mod.nnom_ref_batch(f), oracle_mod.cnn(oracle_model, f, n_threads=4)
    for k in ("logits", "softmax", "argmax"):
        assert np.array_equal(a[k], b[k]), k


def test_oracle_net_input_rounding(oracle_mod, kws_golden):
    for mode in ("zero", "edge"):
        got = oracle_mod.net_input(kws_golden["kws_%s_mfcc" % mode])
        assert np.array_equal(got, kws_golden["kws_%s_feat" % mode])
    m = np.array([[0.5, 1.5, 2.5, -0.5, -1.5, 126.5, 127.5, 300.0, -128.5, -129.0, -500.0, 0.49999, -0.49999]])
    assert oracle_mod.net_input(m).tolist() == [[0, 2, 2, 0, -2, 126, 127, 127, -128, -128, -128, 0, 0]]


def _readme_compare(a, b):
    """compare() of the reference's kws_on_mcu.py:159-168 on float32 arrays, formatted the way it prints."""
    dev = 100.0 * (1.0 - (b.ravel() + 1e-9) / (a.ravel() + 1e-9))
    return ["%.3f" % dev.max(), "%.3f" % dev.min(), "%.3f" % np.mean(dev),
            "%.3f" % np.sqrt(np.mean((b.ravel() - a.ravel()) ** 2)), "%.3f" % (b.max() / a.max()),
            "%.3f" % np.corrcoef(a.ravel(), b.ravel())[0, 1]]


def test_oracle_q15_reproduces_published_board_comparison(oracle_mod, q15_golden):
    """Variant C's pin: README.md:121-139 prints six statistics of (host variant B) vs (board variant C) on
    edison_16k_16b.wav. host32 in the fixture is the reference's own mfcc_mcu output; the Q15 oracle must give
    every printed digit, and only with the frozen table conversion."""
    host32 = q15_golden["host32_edison_edge"]
    x = q15_golden["in_edison_edge"]
    want = q15_golden["readme"].tolist()
    assert want == ["5873.424", "-51588.922", "-105.840", "2.036", "0.959", "0.997"]
    got = oracle_mod.mfcc_q15(x)[:, :13].astype(np.float32)
    assert _readme_compare(host32, got) == want
    for tw, rc in ((0, 0), (1, 0), (1, 1)):
        other = oracle_mod.mfcc_q15(x, tables=oracle_mod.Q15Tables(tw_mode=tw, rc_mode=rc))[:, :13].astype(np.float32)
        assert _readme_compare(host32, other) != want


def test_oracle_q15_tables_and_golden(oracle_mod, q15_golden, mfcc_golden):
    arr = oracle_mod.Q15Tables().arrays()
    for k in ("mel_coef", "mel_start", "mel_count", "tw1024", "tw16", "rfa", "rfb"):
        assert np.array_equal(arr[k], q15_golden["tbl_" + k]), k
    # spot values of the regenerated CMSIS tables: cos/sin(pi/8) as the top halfword of their Q31 value
    assert arr["tw16"][:4].tolist() == [0x7FFF, 0, 0x7641, 0x30FB]
    assert arr["rfa"][:2].tolist() == [0x4000, -0x4000] and arr["rfb"][:2].tolist() == [0x4000, 0x4000]
    for name in ("edison", "hey", "two_tone", "noise", "quiet", "extremes"):
        m, st = oracle_mod.mfcc_q15(mfcc_golden["in_" + name], stages=True, n_threads=2)
        assert np.array_equal(m, q15_golden["C_mfcc_" + name])
        if name in ("edison", "two_tone", "extremes"):
            assert np.array_equal(st["spectrogram"], q15_golden["C_spec_" + name])
            assert np.array_equal(st["mel_spectrogram"], q15_golden["C_mel_" + name])
    assert np.array_equal(oracle_mod.mfcc_q15(mfcc_golden["in_noise"][:4096], frame_step=512),
                          q15_golden["C_mfcc_overlap512"])
    assert np.array_equal(oracle_mod.net_input_q15(q15_golden["C_mfcc_edison_edge"]), q15_golden["C_feat_edison_edge"])


def test_oracle_q15_is_a_scaled_fft(oracle_mod, mfcc_golden):
    """Independent sanity of the restated radix-4 stages: the five stages scale by 1/8, 1/4, 1/4, 1/4, 1/2, so the
    Q15 spectrum must sit within a few LSB of FFT(x)/1024, and the magnitude within a few LSB of |.|."""
    x = mfcc_golden["in_noise"][:8 * 1024]
    _, st = oracle_mod.mfcc_q15(x, stages=True)
    f = np.fft.fft(x.reshape(8, 1024).astype(np.float64)) / 1024.0
    got = st["fft"][..., 0].astype(np.float64) + 1j * st["fft"][..., 1]
    assert np.abs(got - f).max() < 8.0          # five truncating stages: a few LSB of downward bias
    # sqrt of a Q31 fraction, top halfword: sqrt(s / 2^31) * 2^15 = sqrt(s) / sqrt(2) -- the 1/sqrt2 of variant B
    assert np.abs(st["spectrogram"] - np.abs(got[:, :513]) / np.sqrt(2.0)).max() < 1.001
    assert oracle_mod.mfcc_q15(np.zeros(2048, np.int16)).tolist() == [[0] * 32] * 2


def test_softmax_saturation_cases(oracle_mod, oracle_model, cnn_golden):
    """arm_softmax_q7 portable branch: the class more than 136 below the maximum comes out 0."""
    i = 1  # all +127 input: logits 89 and -55
    d = cnn_golden["dense"][i].astype(int)
    assert d.max() - d.min() > 136
    assert cnn_golden["softmax"][i][d.argmin()] == 0


@pytest.mark.parametrize("name", ["same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"])
def test_net_restatement_matches_reference_nnom(name):
    """oracle/net_ref.py (any sequential NNoM graph) against layer outputs of the reference's own NNoM 0.3.0 + CMSIS-NN
    compiled around six other generated model headers (tests/golden/gen_fixtures_net.py), through the importer."""
    import os
    from edison_amd import nnom_import
    from oracle import net_ref
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gold = np.load(os.path.join(g, "net_golden.npz"))
    with open(os.path.join(g, "alt_models", name + ".h")) as f:
        shape, layers = nnom_import.parse_weights_h(f.read())
    out = net_ref.run(nnom_import.build_blob(shape, layers), gold["in_" + name])
    assert np.array_equal(np.concatenate(out["acts"], axis=1), gold["acts_" + name])


def test_net_restatement_on_the_shipped_graph(cnn_golden):
    """The same numpy restatement on the committed kws_conv blob reproduces the reference build's layer outputs."""
    import os
    from oracle import net_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "edison_amd", "data", "kws_nnom.ednn"), "rb") as f:
        out = net_ref.run(f.read(), cnn_golden["feats"])
    n = cnn_golden["feats"].shape[0]
    for got, key in zip(out["acts"], ("conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "dense", "softmax")):
        assert np.array_equal(got, cnn_golden[key].reshape(n, -1)), key
    assert np.array_equal(out["argmax"], cnn_golden["argmax"].ravel())


def test_variant_d_restatement_against_the_reference_made_fixture(oracle_mod):
    """tests/golden/mfccf32_golden.npz: variant D as the reference's own mfcc_compute + CMSIS float transform compute it
    (tests/golden/gen_fixtures_f32.py, reference object code). The restatement that checks the GPU must reproduce it without the
    reference at hand: int8 equal but for rounding-boundary values, log-mel within 1e-3 clear of the float32 rounding floor, within that floor elsewhere."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mfccf32_golden.npz"))
    x = g["audio"]
    for i, bar in ((0, 0.999), (1, 0.98), (2, 0.999)):
        nf, off, flen, bits, pre, hop = g["cfg%d" % i]
        m = oracle_mod.MfccF32(int(nf), int(off), int(flen), int(bits), float(pre))
        pi, _, plm = m(x, frame_step=int(hop))
        ri, rlm = g["mfcc%d" % i], g["logmel%d" % i]
        assert pi.shape == ri.shape
        d = np.abs(ri.astype(int) - pi.astype(int))
        assert d.max() <= 1 and (d == 0).mean() >= bar, (i, d.max(), (d == 0).mean())
        from test_oracle_refpins import _variant_d_logmel_close
        _variant_d_logmel_close(x, int(hop), int(flen), float(pre), rlm, plm)


def test_numpy_restatement_for_other_geometries_equals_the_reference(oracle_mod):
    """oracle.mfcc_numpy (variants A / B at any frame length, numpy's FFT as in the reference) against the vectors the REFERENCE's own
    functions produced for six other geometries (tests/golden/gen_fixtures_geom.py): 33 ... 2048 samples, powers of two and not,
    5 ... 64 mel bins, other edges and scales. This is the checker the GPU fuzzers use for the generality kernel."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mfcc_geom_golden.npz"))
    for idx, name in enumerate(g["names"]):
        N, step, nm, lo, hi, scale = g["configs"][idx]
        N, step, nm = int(N), int(step), int(nm)
        for sname in ("edison", "noise"):
            x = g["in_" + sname]
            k = "%s_%s_" % (name, sname)
            for tag, variant, use_log in (("A", oracle_mod.VARIANT_A, False), ("B", oracle_mod.VARIANT_B, False), ("Blog", oracle_mod.VARIANT_B, True)):
                got = oracle_mod.mfcc_numpy(x, variant, N, step, num_mel_bins=nm, lower_edge_hertz=lo, upper_edge_hertz=hi, mel_mtx_scale=scale, use_log=use_log)
                ref = g[k + tag + "_mfcc"]
                assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (name, sname, tag)


def test_mfcc_numpy_variant_tf_against_the_c_restatement(oracle_mod, mfcc_golden):
    """The two restatements of variant TF (both unpinned: no TensorFlow here) agree at the reference geometry: oracle.mfcc_numpy (numpy's rfft, the
    window product in float32 as tf.signal does it) and oracle/mfcc_ref.c (its own FFT, the window product in float64) -- to what one float32
    rounding of the windowed samples moves a coefficient."""
    for name in ("edison", "noise"):
        x = mfcc_golden["in_" + name]
        a = oracle_mod.mfcc_numpy(x, oracle_mod.VARIANT_TF, 1024, 1024)
        b = oracle_mod.mfcc(x, oracle_mod.VARIANT_TF)
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 1e-4 + 1e-5 * np.abs(b).max(), (name, float(np.abs(a - b).max()))
