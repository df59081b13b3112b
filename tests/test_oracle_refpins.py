"""CPU tests that pin the restatements of MFCC variants C and D to the REFERENCE'S OWN OBJECT CODE where that compiles here:

  * oracle/_ref/libcmsis_q15_ref.so -- CMSIS-DSP arm_cfft_q15 / arm_radix4_butterfly_q15 (ARM_MATH_DSP branch),
    arm_split_rfft_q15, arm_sqrt_q31, built by `make -C oracle q15ref` from firmware/src/lib/CMSIS/DSP/Source
  * oracle/_ref/libmfcc_f32_ref.so  -- firmware/src/audio/mfcc.c (create_dct_matrix, create_mel_fbank), `make -C oracle f32ref`

What this pins: the ARITHMETIC SEQUENCE of oracle/mfcc_q15_ref.c's FFT, magnitude and DCT stages -- every shift, saturation,
halving add and 16-bit truncation -- and variant D's table VALUES, bit for bit. What it does not pin: the VALUES of the Q15
twiddle / split tables (arm_common_tables.c is absent from the reference; the tests hand the routines the regenerated
tables, whose values rest on README.md:121-139 and mel_constants.h, tests/test_oracle.py), and variant D's FFT / log stage.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bitrev(n_bits):
    i = np.arange(1 << n_bits)
    r = np.zeros_like(i)
    for b in range(n_bits):
        r |= ((i >> b) & 1) << (n_bits - 1 - b)
    return r


def _frames():
    """golden streams + seeded frames from 1 LSB to clipping + the corner cases of the saturating butterflies"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "mfcc_golden.npz"))
    parts = []
    for k in g.files:
        if k.startswith("in_"):
            x = g[k].astype(np.int16).ravel()
            parts.append(x[:(x.size // 1024) * 1024].reshape(-1, 1024))
    rng = np.random.default_rng(1234)
    n = 16384
    amp = np.exp(rng.uniform(np.log(0.6), np.log(60000.0), (n, 1)))
    noise = np.clip(rng.normal(0, 1, (n, 1024)) * amp, -32768, 32767).astype(np.int16)
    parts.append(noise)
    t = np.arange(1024)
    corners = [np.full(1024, 32767), np.full(1024, -32768), np.where(t & 1, 32767, -32768), np.where((t >> 1) & 1, 32767, -32768),
               np.where(t < 512, 32767, -32768), np.zeros(1024), np.eye(1, 1024, 0)[0] * 32767, np.eye(1, 1024, 513)[0] * -32768,
               np.round(32767 * np.cos(2 * np.pi * t * 64 / 1024)), np.round(32767 * np.sin(2 * np.pi * t * 255 / 1024)),
               np.round(32767 * np.sign(np.cos(2 * np.pi * t * 128 / 1024)))]
    parts.append(np.array(corners).astype(np.int16))
    return np.concatenate(parts)


@pytest.fixture(scope="module")
def q15ref(oracle_mod):
    if not oracle_mod.have_q15_ref():
        pytest.skip("oracle/_ref/libcmsis_q15_ref.so absent and /root/reference not mounted")
    return oracle_mod.cmsis_q15_ref()


def test_q15_oracle_stages_equal_the_reference_routines(oracle_mod, q15ref):
    """oracle/mfcc_q15_ref.c vs arm_cfft_q15 -> arm_radix4_butterfly_q15 (1024 and 16 points), arm_sqrt_q31 and
    arm_split_rfft_q15 of the reference, stage by stage, on ~16.5 k frames."""
    x = _frames()
    n = x.shape[0]
    tab = oracle_mod.Q15Tables().arrays()
    mf, st = oracle_mod.mfcc_q15(x.ravel(), stages=True, n_threads=4)
    # [1] the 1024-point complex FFT of audioCalcMFCCs (audioprocessing.c:135-139): real samples, zero imaginary parts
    buf = np.zeros((n, 2048), np.int16)
    buf[:, 0::2] = x
    ref_fft = q15ref.cfft(buf, tab["tw1024"]).reshape(n, 1024, 2)[:, _bitrev(10), :]     # arm_bitreversal_16 = plain bit reversal
    assert np.array_equal(st["fft"], ref_fft)
    # [2] cmpl_mag_q15 (audioprocessing.c:299-312): arm_sqrt_q31 of re^2 + im^2 (wrapping 32-bit sum), top halfword
    re, im = ref_fft[:, :513, 0].astype(np.int64), ref_fft[:, :513, 1].astype(np.int64)
    s = ((re * re + im * im) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    spec = (q15ref.sqrt_q31(s) >> 16).astype(np.int16)
    assert np.array_equal(st["spectrogram"], spec)
    # [5] dct2_q15 (audioprocessing.c:330-436): even/odd reorder, arm_rfft_q15(32) = arm_cfft_q15(16) + arm_split_rfft_q15, real parts
    mel = st["mel_spectrogram"]
    v = np.zeros((n, 32), np.int16)
    v[:, :16] = mel[:, 0::2]
    v[:, 31:15:-1] = mel[:, 1::2]
    z = q15ref.cfft(v, tab["tw16"]).reshape(n, 16, 2)[:, _bitrev(4), :].reshape(n, 32)
    out = q15ref.split_rfft(z, tab["rfa"], tab["rfb"], modifier=1)                        # [n][64]: 32 complex
    assert np.array_equal(mf, out[:, 0::2])


def test_sqrt_q31_restatement_equals_the_reference_on_a_sweep(oracle_mod, q15ref):
    """the square root alone, denser than the frames reach: every power-of-two neighbourhood, 4 M random positives,
    zero and negatives (the routine answers 0)"""
    rng = np.random.default_rng(7)
    xs = [rng.integers(1, 2 ** 31, 4_000_000, dtype=np.int64), np.array([0, -1, -2 ** 31, 2 ** 31 - 1, 1, 2, 3])]
    for b in range(31):
        xs.append(np.arange(max(1, (1 << b) - 300), min(2 ** 31 - 1, (1 << b) + 300)))
    x = np.concatenate(xs).astype(np.int64).astype(np.int32)
    ref = q15ref.sqrt_q31(x)
    # through the oracle's frame function: a "frame" cannot carry arbitrary sums, so the restated routine is reached through
    # the verification tool below instead; here: the reference against exact arithmetic, |ref - sqrt(x * 2^31)| small
    pos = x > 0
    exact = np.sqrt(x[pos].astype(np.float64) * 2.0 ** 31)
    assert np.all(ref[~pos] == 0)
    assert np.max(np.abs(ref[pos] - exact)) <= 6.0e4 * 1 and np.max(np.abs(ref[pos] - exact) / exact) < 2e-4


def test_gpu_sqrt_formulations_proved_against_the_reference_object_code(tmp_path, oracle_mod, q15ref):
    """tools/verify/sqrt_q31_equiv.c and sqrt_q31_floor.c enumerate inputs and compare the GPU kernel's two formulations of
    arm_sqrt_q31 with ... the reference's compiled arm_sqrt_q31 itself (dlopen of oracle/_ref/libcmsis_q15_ref.so), not a
    restatement. Here with a stride (2^31 / 4099 inputs each, a few seconds); the full enumeration is `... 1`."""
    so = oracle_mod.Q15_REF_SO
    for tool in ("sqrt_q31_equiv", "sqrt_q31_floor"):
        exe = str(tmp_path / tool)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", exe, os.path.join(ROOT, "tools", "verify", tool + ".c"), "-ldl", "-lm"])
        r = subprocess.run([exe, "4099", so], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "reference object code" in r.stdout, r.stdout


@pytest.fixture(scope="module")
def f32ref(oracle_mod):
    if not oracle_mod.have_f32_ref():
        pytest.skip("oracle/_ref/libmfcc_f32_ref.so absent and /root/reference not mounted")
    return oracle_mod.mfcc_f32_ref()


def test_variant_d_dct_matrix_equals_the_reference(built_lib, oracle_mod, f32ref):
    """create_dct_matrix of the PRODUCT (csrc/tables_f32.c, exported under the firmware's name) and of the oracle vs the
    reference's compiled create_dct_matrix (mfcc.c:101-115), float32 bit patterns."""
    built_lib.create_dct_matrix.restype = ctypes.POINTER(ctypes.c_float)
    built_lib.create_dct_matrix.argtypes = [ctypes.c_int32, ctypes.c_int32]
    for n_in, n_out in ((26, 13), (26, 26), (26, 1), (40, 10), (13, 13)):
        ref = f32ref.dct_matrix(n_in, n_out)
        p = built_lib.create_dct_matrix(n_in, n_out)
        got = np.ctypeslib.as_array(p, shape=(n_out, n_in)).copy()
        ctypes.CDLL(None).free(p)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (n_in, n_out)
    for nf in (13, 26, 1):
        dct, _, _, _ = oracle_mod.MfccF32(num_mfcc_features=nf, feature_offset=0).tables()
        assert np.array_equal(dct.view(np.uint32), f32ref.dct_matrix(26, nf).view(np.uint32))


class _F32Tables(ctypes.Structure):
    """ed_f32_tables_t (edison_amd/csrc/edison_internal.h)"""
    _fields_ = [("n_features", ctypes.c_int32), ("offset", ctypes.c_int32), ("frame_len", ctypes.c_int32), ("padded", ctypes.c_int32),
                ("log2p", ctypes.c_int32), ("dec_bits", ctypes.c_int32), ("preempha", ctypes.c_float), ("scale", ctypes.c_float),
                ("window", ctypes.c_float * 1024), ("tw", ctypes.c_float * 1024),
                ("mel_first", ctypes.c_int32 * 26), ("mel_last", ctypes.c_int32 * 26), ("mel_off", ctypes.c_int32 * 26),
                ("mel_w", ctypes.c_float * 1100), ("dct", ctypes.c_float * (26 * 26))]


def test_variant_d_mel_filterbank_equals_the_reference(built_lib, oracle_mod, f32ref):
    """The 26 triangular filters (first bin, last bin, every weight) of the product's table builder (ed_build_f32_tables,
    csrc/tables_f32.c) and of the oracle vs the reference's compiled create_mel_fbank (mfcc.c:117-171) for every padded
    frame length the path supports; float32 bit patterns."""
    built_lib.ed_build_f32_tables.restype = ctypes.c_int
    built_lib.ed_build_f32_tables.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.POINTER(_F32Tables), ctypes.c_char_p, ctypes.c_size_t]
    for frame_len, padded in ((512, 512), (400, 512), (1024, 1024), (640, 1024), (256, 256), (128, 128)):
        first, last, w = f32ref.mel_fbank(padded)
        t = _F32Tables()
        err = ctypes.create_string_buffer(200)
        assert built_lib.ed_build_f32_tables(13, 1, frame_len, 8, 0.97, ctypes.byref(t), err, 200) == 0, err.value
        assert t.padded == padded
        assert np.array_equal(np.array(t.mel_first), first) and np.array_equal(np.array(t.mel_last), last)
        got = np.array(t.mel_w, dtype=np.float32)[:w.size]
        assert np.array_equal(np.array(t.mel_off), np.concatenate([[0], np.cumsum(last - first + 1)[:-1]]))
        assert np.array_equal(got.view(np.uint32), w.view(np.uint32)), frame_len
        assert np.array_equal(np.array(t.dct, dtype=np.float32)[:13 * 26].view(np.uint32), f32ref.dct_matrix(26, 13).ravel().view(np.uint32))
        _, ofirst, olast, ow = oracle_mod.MfccF32(frame_len=frame_len).tables()
        assert np.array_equal(ofirst, first) and np.array_equal(olast, last) and np.array_equal(ow.view(np.uint32), w.view(np.uint32))


def _variant_d_logmel_close(x, hop, N, preempha, rlm, plm):
    """Two float32 transforms agree to ~1e-6 of a frame's LARGEST bin: band energies are compared in the linear domain against
    that floor (1e-4 of the largest bin: a band below it holds each FFT's own rounding noise), and in the log domain (1e-3)
    wherever the band stands clear of it."""
    n = rlm.shape[0]
    fr = np.stack([x[i * hop:i * hop + N] for i in range(n)]).astype(np.float64)
    pre = np.concatenate([fr[:, :1], (fr[:, 1:] - preempha * fr[:, :-1]) / 32768.0], axis=1)      # mfcc.c:180-186 (sample 0 is not scaled)
    spec_max = np.abs(np.fft.rfft(pre * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)), 1 << int(np.ceil(np.log2(N))))).max(axis=1)
    er, ep = np.exp(rlm.astype(np.float64)), np.exp(plm.astype(np.float64))
    assert np.all(np.abs(er - ep) <= 1e-4 * spec_max[:, None] + 1e-30)
    clear = ep > 1e-2 * spec_max[:, None]
    assert clear.mean() > 0.3 and np.abs(rlm - plm)[clear].max() <= 1e-3, np.abs(rlm - plm)[clear].max()


def test_variant_d_transform_tables_come_from_the_reference_routine(oracle_mod, f32ref):
    """arm_common_tables.c is absent from the snapshot; the reference's compiled arm_cfft_f32 / arm_rfft_fast_f32 take their
    tables by pointer. MfccF32Ref.fft_tables regenerates the twiddle VALUES by formula and reads the bit-reversal permutation
    off the routine itself (it asserts, through the compiled routines, that the result is the DFT to float32 accuracy). The
    swap list it derives is exactly as long as ARM's table for 256 points (ARMBITREVINDEXTABLE_256_TABLE_LENGTH = 440,
    arm_common_tables.h), and the fixture keeps it."""
    tw, rev, rt = f32ref.fft_tables(512)
    assert tw.shape == (256, 2) and rt.shape == (256, 2) and rev.size == 440 and rev.max() < 8 * 256 and (rev % 8 == 0).all()
    assert rt[0, 0] == 0.0 and rt[0, 1] == 1.0                      # (sin, cos): the layout the real-transform stage reads
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mfccf32_golden.npz"))
    assert np.array_equal(g["bitrev_swaps_256"], rev)
    for n in (64, 128, 256, 1024, 2048):                            # every length mfcc_create can ask for in practice
        f32ref.fft_tables(n)


def test_variant_d_restatement_equals_the_reference_mfcc_compute(oracle_mod, f32ref):
    """oracle/mfcc_f32_ref.c (the checker of the GPU's variant D) against the reference's own mfcc_compute + CMSIS transform,
    compiled from the reference (oracle/_ref/libmfcc_f32_ref.so), frame by frame on fresh seeded audio -- noise from 3 LSB to
    clipping, a tone, silence, both rails. int8 outputs: equal but for values that sit on a rounding boundary (<= 0.1 % of them,
    off by one); log-mel energies: within 1e-3 wherever the band stands clear of the frame's float32 rounding floor, within
    that floor (linear domain) elsewhere, silence bit-identical (the FLT_MIN branch)."""
    rng = np.random.default_rng(123)
    parts = [np.clip(rng.normal(0, s, 150 * 256), -32768, 32767) for s in (3, 100, 3000, 20000, 50000)]
    t = np.arange(150 * 256)
    parts += [6000 * np.sin(2 * np.pi * 1000 * t / 16000) + 20 * rng.normal(size=t.size), np.zeros(40 * 256), np.full(40 * 256, 32767.0), np.full(40 * 256, -32768.0)]
    x = np.concatenate(parts).astype(np.int16)
    for cfg, bar in ((dict(), 0.999), (dict(num_mfcc_features=13, feature_offset=0, frame_len=480, mfcc_dec_bits=7, preempha=0.0), 0.98),
                     (dict(num_mfcc_features=10, feature_offset=1, frame_len=400, mfcc_dec_bits=5, preempha=0.95), 0.999)):
        hop = cfg.get("frame_len", 512) // 2
        ri, rlm = f32ref.compute(x, frame_step=hop, **cfg)
        pi, _, plm = oracle_mod.MfccF32(**cfg)(x, frame_step=hop)
        assert ri.shape == pi.shape and ri.shape[0] > 1000
        d = np.abs(ri.astype(int) - pi.astype(int))
        assert d.max() <= 1 and (d == 0).mean() >= bar, (cfg, d.max(), (d == 0).mean())
        _variant_d_logmel_close(x, hop, cfg.get("frame_len", 512), cfg.get("preempha", 0.97), rlm, plm)
        quiet = (x[: (ri.shape[0] - 1) * hop + cfg.get("frame_len", 512)].reshape(-1)[None] == 0).all()  # (never: the stream is mixed)
        sil = np.array([not x[i * hop:i * hop + cfg.get("frame_len", 512)].any() for i in range(ri.shape[0])])
        assert sil.sum() > 10 and np.array_equal(rlm[sil].view(np.uint32), plm[sil].view(np.uint32)) and not quiet


def test_output_filter_class_choice_equals_arm_max_f32(oracle_mod, f32ref):
    """The firmware's output post-processing (app.c:332-356) is inline code in a file that cannot be built here, except for its
    one library call: arm_max_f32 picks the class. The restatement's choice (oracle_output_filter: `likely`) against CMSIS-DSP's
    arm_max_f32 compiled from the reference, on the filtered states of random softmax sequences -- exact ties included (the state
    starts as zeros; two classes fed the same values): always the FIRST maximum."""
    rng = np.random.default_rng(17)
    soft = rng.integers(-128, 128, (3000, 10)).astype(np.int8)
    soft[:5] = 0                                                     # the state starts as zeros: ten equal values
    soft[5:65] = -50
    soft[5:65, 2] = 100
    soft[5:65, 5] = 100                                              # two classes fed identically from equal states: exact ties
    filt, likely, spotted, _ = oracle_mod.output_filter(soft)
    ties = 0
    for i in range(soft.shape[0]):
        val, idx = f32ref.arm_max(filt[i])
        assert idx == likely[i] and val == filt[i, likely[i]], i
        ties += int((filt[i] == val).sum() > 1)
    assert ties >= 20
