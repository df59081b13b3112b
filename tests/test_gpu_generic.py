"""GPU tests (-m gpu) of the generality path: MFCC variants A / B for geometries other than audio/config.py's 1024 / 32, through the
Python mirror of the reference's functions (edison_amd/mfcc/mfcc_utils.py -> edison_mfcc_generic through the C-ABI), against golden
vectors made by the REFERENCE's own functions with those geometries (tests/golden/gen_fixtures_geom.py: frame lengths 33 ... 2048,
powers of two and not, 5 ... 64 mel bins, other edges and matrix scales). The kernel computes in float64 like the reference, so the bar
is 1e-9 of each array's largest value (+ 1e-9 absolute for the logarithms)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mfcc_geom_golden.npz")


def _near(got, ref, what):
    ref = np.asarray(ref)
    got = np.asarray(got)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    tol = 1e-9 * max(1.0, float(np.abs(ref).max())) + 1e-9
    d = np.abs(got - ref).max() if ref.size else 0.0
    assert d <= tol, "%s: max |d| %.3e > %.3e" % (what, d, tol)


@pytest.fixture(scope="module")
def geom():
    return np.load(GOLDEN)


@pytest.mark.parametrize("idx", range(6))
def test_other_geometries_equal_the_reference(ctx, geom, idx):
    from edison_amd.mfcc import mfcc_utils as mfu
    name = str(geom["names"][idx])
    N, step, nm, lo, hi, scale = geom["configs"][idx]
    N, step, nm = int(N), int(step), int(nm)
    fs = 16000
    for sname in ("edison", "noise"):
        x = geom["in_" + sname]
        k = "%s_%s_" % (name, sname)
        a = mfu.mfcc(x, fs, len(x), N, step, 0, N, nm, lo, hi)
        b = mfu.mfcc_mcu(x, fs, len(x), N, step, 0, N, nm, lo, hi, scale)
        bl = mfu.mfcc_mcu(x, fs, len(x), N, step, 0, N, nm, lo, hi, scale, True)
        assert len(a) == len(b) == geom[k + "A_mfcc"].shape[0] == 1 + (len(x) - N) // step
        assert set(a[0]) == {"t_start", "t_end", "fft", "spectrogram", "mel_weight_matrix", "mel_spectrogram", "log_mel_spectrogram", "mfcc"}
        _near(np.array([f["mfcc"] for f in a]), geom[k + "A_mfcc"], k + "A_mfcc")
        _near(np.array([f["mfcc"] for f in b]), geom[k + "B_mfcc"], k + "B_mfcc")
        _near(np.array([f["mfcc"] for f in bl]), geom[k + "Blog_mfcc"], k + "Blog_mfcc")
        for f in (0, len(a) - 1):
            assert a[f]["t_start"] == f * step / fs and a[f]["t_end"] == (f * step + N) / fs
            _near(a[f]["fft"], geom[k + "A_fft_%d" % f], k + "A_fft")
            _near(a[f]["spectrogram"], geom[k + "A_spec_%d" % f], k + "A_spec")
            _near(a[f]["mel_spectrogram"], geom[k + "A_mel_%d" % f], k + "A_mel")
            _near(a[f]["log_mel_spectrogram"], geom[k + "A_logmel_%d" % f], k + "A_logmel")
            _near(b[f]["fft"], geom[k + "B_fft_%d" % f], k + "B_fft")
            _near(b[f]["spectrogram"], geom[k + "B_spec_%d" % f], k + "B_spec")
            _near(b[f]["mel_spectrogram"], geom[k + "B_mel_%d" % f], k + "B_mel")
            assert a[f]["mel_weight_matrix"].shape == (N // 2, nm) and b[f]["mel_weight_matrix"].shape == (N // 2 + 1, nm)
            assert a[f]["fft"].shape == (N // 2,) and b[f]["fft"].shape == (N,) and b[f]["spectrogram"].shape == (N,)
    rows = np.stack([geom["in_noise"][:3000], geom["in_noise"][3000:6000]])
    _near(mfu.batch_mfcc(rows, fs, 3000, N, step, 0, N, nm, lo, hi), geom[name + "_batch"], name + "_batch")


def test_generic_entry_point_limits_and_net_input(ctx, geom):
    """The C-ABI call itself: device-pointer form equals the host form; feat = the net-input rounding of the first n_coef
    coefficients; geometries beyond the limits and other variants are refused with EDISON_E_NO_IMPL, bad edges with E_ARGUMENT;
    the fast geometry (1024 / 32) through the generality kernel agrees with the fast kernels within their fp32 tolerance."""
    import ctypes
    import torch
    from edison_amd import _lib
    L = _lib.lib()
    x = np.ascontiguousarray(geom["in_noise"][:4096])
    N, step, nm = 512, 256, 20
    n = 1 + (len(x) - N) // step
    mf = np.zeros((n, nm)); feat = np.zeros((n, 7), np.int8)
    r = L.edison_mfcc_generic(ctx._h, x.ctypes.data, n, N, step, _lib.MFCC_B, nm, 16000.0, 125.0, 3800.0, 128.0, None, None, None, None, mf.ctypes.data, 7,
                              feat.ctypes.data, 0.5)
    assert r == 0, L.edison_last_error(ctx._h)
    want = np.clip((mf[:, :7].astype(np.float32) * np.float32(0.5)), -128, 127).round().astype(np.int8)
    assert np.array_equal(feat, want)
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).to(dev)
    md = torch.zeros((n, nm), dtype=torch.float64, device=dev)
    r = L.edison_mfcc_generic_dev(ctx._h, xd.data_ptr(), n, N, step, _lib.MFCC_B, nm, 16000.0, 125.0, 3800.0, 128.0, None, None, None, None, md.data_ptr(), 0, None, 1.0)
    assert r == 0 and np.array_equal(md.cpu().numpy(), mf)
    for bad, code in (((8192, 32), _lib.E_NO_IMPL), ((2, 32), _lib.E_NO_IMPL), ((512, 257), _lib.E_NO_IMPL), ((512, 0), _lib.E_NO_IMPL)):
        r = L.edison_mfcc_generic(ctx._h, x.ctypes.data, 1, bad[0], step, _lib.MFCC_B, bad[1], 16000.0, 125.0, 3800.0, 128.0, None, None, None, None, mf.ctypes.data, 0, None, 1.0)
        assert r == code, (bad, r)
    assert L.edison_mfcc_generic(ctx._h, x.ctypes.data, 1, N, step, _lib.MFCC_C, nm, 16000.0, 125.0, 3800.0, 128.0, None, None, None, None, mf.ctypes.data, 0, None, 1.0) == _lib.E_NO_IMPL
    assert L.edison_mfcc_generic(ctx._h, x.ctypes.data, 1, N, step, _lib.MFCC_B, nm, 16000.0, 3800.0, 125.0, 128.0, None, None, None, None, mf.ctypes.data, 0, None, 1.0) == _lib.E_ARGUMENT
    assert L.edison_mfcc_generic(ctx._h, x.ctypes.data, 0, N, step, _lib.MFCC_B, nm, 16000.0, 125.0, 3800.0, 128.0, None, None, None, None, None, 0, None, 1.0) == 0
    # the shipped geometry both ways
    y = np.ascontiguousarray(geom["in_edison"])
    m64 = np.zeros((10, 32))
    assert L.edison_mfcc_generic(ctx._h, y.ctypes.data, 10, 1024, 1024, _lib.MFCC_B, 32, 16000.0, 80.0, 7600.0, 128.0, None, None, None, None, m64.ctypes.data, 0, None, 1.0) == 0
    fast = ctx.mfcc(y, variant=_lib.MFCC_B, n_coef=32)
    assert np.all(np.abs(fast - m64) <= 1e-2 + 1e-5 * np.abs(m64))


@pytest.mark.parametrize("idx", [0, 2, 5])
def test_variant_tf_at_other_geometries(ctx, geom, oracle_mod, idx):
    """mfcc_tf (mfcc_utils.py:201-253) for geometries other than 1024 / 32 through the generality kernel. PARITY UNPINNED -- TensorFlow is not in this
    image and the reference holds no output of it --: held to the oracle's numpy restatement of tf.signal's definitions (float32 window product,
    then float64), 1e-9 of the largest value like variants A / B above. At the reference geometry the generality kernel and the fast variant-TF
    kernel (float32 throughout) are held to each other with variant A's bar."""
    from edison_amd import _lib
    from edison_amd.mfcc import mfcc_utils as mfu
    N, step, nm, lo, hi, _ = geom["configs"][idx]
    N, step, nm = int(N), int(step), int(nm)
    fs = 16000
    for sname in ("edison", "noise"):
        x = geom["in_" + sname]
        o = mfu.mfcc_tf(x, fs, len(x), N, step, 0, N, nm, lo, hi)
        ref = oracle_mod.mfcc_numpy(x, oracle_mod.VARIANT_TF, N, step, num_mel_bins=nm, lower_edge_hertz=lo, upper_edge_hertz=hi)
        assert len(o) == ref.shape[0] == 1 + (len(x) - N) // step
        got = np.array([f["mfcc"] for f in o], np.float64)
        tol = 2e-6 * max(1.0, float(np.abs(ref).max()))     # the mirror hands float32 arrays back, like TensorFlow's tensors
        assert np.abs(got - ref).max() <= tol, (sname, float(np.abs(got - ref).max()), tol)
        assert o[0]["fft"].shape == (N // 2,) and o[0]["spectrogram"].shape == (N // 2,) and o[0]["mel_weight_matrix"].shape == (N // 2, nm)
        assert o[0]["mfcc"].shape == (nm,) and o[0]["mfcc"].dtype == np.float32
        # the float64 arrays of the kernel itself, against the same restatement: 1e-9
        st = mfu._generic(ctx, x, ref.shape[0], N, step, _lib.MFCC_TF, nm, fs, lo, hi)
        _near(st["mfcc"], ref, "TF mfcc %d %s" % (idx, sname))
        xw = (x[:N].astype(np.float32) * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N) / N)).astype(np.float32)).astype(np.float64)
        _near(st["fft"][0], np.fft.rfft(xw), "TF fft")
        _near(st["spectrogram"][0], np.abs(np.fft.rfft(xw)), "TF spectrogram")
    with pytest.raises(NotImplementedError):
        mfu.mfcc_tf(x, fs, len(x), N, step, 0, 2 * N, nm, lo, hi)     # fft_len != frame_len
    with pytest.raises(_lib.EdisonError):
        mfu._generic(ctx, x, 2, N, step, _lib.MFCC_TF, nm, fs, lo, hi, use_log=True)


def test_variant_tf_generic_against_the_fast_kernel_at_the_reference_geometry(ctx, geom, oracle_mod):
    from edison_amd import _lib
    from edison_amd.mfcc import mfcc_utils as mfu
    x = geom["in_noise"]
    n = 1 + (len(x) - 1024) // 1024
    g = mfu._generic(ctx, x, n, 1024, 1024, _lib.MFCC_TF, 32, 16000, 80.0, 7600.0)["mfcc"]
    fast = ctx.mfcc(x, variant=_lib.MFCC_TF, n_coef=32)
    ref = oracle_mod.mfcc(x, oracle_mod.VARIANT_TF)              # the C restatement (float64 window product)
    assert np.abs(g - fast).max() <= 1e-3 + 1e-4 * np.abs(g).max()
    assert np.abs(g - ref).max() <= 1e-3 + 1e-4 * np.abs(ref).max()
