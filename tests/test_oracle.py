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


def test_softmax_saturation_cases(oracle_mod, oracle_model, cnn_golden):
    """arm_softmax_q7 portable branch: the class more than 136 below the maximum comes out 0."""
    i = 1  # all +127 input: logits 89 and -55
    d = cnn_golden["dense"][i].astype(int)
    assert d.max() - d.min() > 136
    assert cnn_golden["softmax"][i][d.argmin()] == 0
