"""GPU tests of MFCC variant D (the firmware's float32 ML-KWS extractor, firmware/src/audio/mfcc.c).

Parity unpinned: the reference holds no vectors for this path and it cannot be linked here (CMSIS tables missing), so
the checker is the oracle's restatement (oracle/mfcc_f32_ref.c, FFT in double). Bars: the scaled sums before rounding
within |d| <= 0.02 + 2e-5 |ref| (float32 FFT + logf + 26-term DCT, times 2^dec_bits = 256), the int8 output equal
wherever the reference value is not within that distance of a rounding boundary.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(got_i8, got_f32, ref_i8, ref_f32, atol=0.02, rtol=2e-5):
    tol = atol + rtol * np.abs(ref_f32)
    assert np.all(np.abs(got_f32 - ref_f32) <= tol), "max |d| %.4g" % np.abs(got_f32 - ref_f32).max()
    frac = np.abs(ref_f32 - np.trunc(ref_f32))                     # distance to the nearest .5 boundary (half away)
    near = (np.abs(frac - 0.5) <= tol) | (np.abs(np.abs(ref_f32) - 127.0) <= tol) | (np.abs(np.abs(ref_f32) - 128.0) <= tol)
    assert np.array_equal(got_i8[~near], ref_i8[~near])
    assert np.abs(got_i8.astype(int) - ref_i8.astype(int))[near].max(initial=0) <= 1


@pytest.mark.parametrize("params", [dict(), dict(mfcc_dec_bits=2), dict(num_mfcc_features=26, feature_offset=0, frame_len=400, mfcc_dec_bits=0),
                                    dict(frame_len=1024, mfcc_dec_bits=3, preemph=0.0), dict(frame_len=256, feature_offset=3)])
def test_f32_vs_oracle(ctx, oracle_mod, mfcc_golden, params):
    from edison_amd.mfcc.mfcc_f32 import MfccF32
    op = dict(params)
    if "preemph" in op:
        op["preempha"] = op.pop("preemph")
    ref = oracle_mod.MfccF32(**op)
    m = MfccF32(ctx=ctx, **params)
    assert m.n_out == ref.n_out
    rng = np.random.default_rng(41)
    x = np.concatenate([mfcc_golden["in_edison"], mfcc_golden["in_noise"][:8192], mfcc_golden["in_quiet"],
                        (rng.integers(-3, 4, 4096)).astype(np.int16)])
    hop = m.frame_len // 2                                          # the NNoM example's 50 % overlap (app.c:583)
    got, gf, glm = m.compute(x, frame_step=hop, want_float=True)
    ri, rf, rlm = ref(x, frame_step=hop, n_threads=4)
    assert got.shape == ri.shape and got.shape[0] > 50
    assert np.all(np.abs(glm - rlm) <= 1e-3)
    _check(got, gf, ri, rf, atol=0.02 * max(1.0, (1 << params.get("mfcc_dec_bits", 8)) / 16.0))
    assert np.array_equal(m.compute(x, frame_step=hop), got)
    # Ill-conditioned frames (silence, DC, full-scale square waves at fs/2: nothing but leakage inside 20..4000 Hz).
    # A float32 FFT is accurate to ~1e-6 of the frame's largest bin, so those band energies are rounding noise in ANY
    # float32 implementation, the MCU's included; they are compared in the linear domain against that floor.
    e = mfcc_golden["in_extremes"]
    _, _, glm = m.compute(e, frame_step=hop, want_float=True)
    _, _, rlm = ref(e, frame_step=hop)
    n = glm.shape[0]
    N = m.frame_len
    fr = np.stack([e[i * hop:i * hop + N] for i in range(n)]).astype(np.float64)
    pre = np.concatenate([fr[:, :1], (fr[:, 1:] - params.get("preemph", 0.97) * fr[:, :-1]) / 32768.0], axis=1)
    spec_max = np.abs(np.fft.rfft(pre * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)), 1 << int(np.ceil(np.log2(N))))).max(axis=1)
    lin = np.abs(np.exp(glm.astype(np.float64)) - np.exp(rlm.astype(np.float64)))
    assert np.all(lin <= 1e-4 * spec_max[:, None] + 1e-30)
    m.close()


def test_f32_silence_and_errors(ctx, oracle_mod):
    """All-zero audio: every band is exactly 0 -> FLT_MIN -> logf (mfcc.c:225-231)."""
    from edison_amd import _lib
    from edison_amd.mfcc.mfcc_f32 import MfccF32
    m = MfccF32(ctx=ctx, mfcc_dec_bits=0)
    got, gf, glm = m.compute(np.zeros(2048, np.int16), want_float=True)
    ri, rf, rlm = oracle_mod.MfccF32(mfcc_dec_bits=0)(np.zeros(2048, np.int16))
    assert np.allclose(glm, np.log(np.float32(1.17549435e-38)), rtol=1e-6) and np.allclose(glm, rlm, rtol=1e-6)
    assert np.allclose(gf, rf, atol=1e-3) and np.array_equal(got, ri)
    assert m.compute(np.zeros(100, np.int16)).shape == (0, 12)
    with pytest.raises(ValueError):
        m.compute(np.zeros(600, np.int16), n_frames=2)
    m.close()
    for bad in (dict(frame_len=2000), dict(frame_len=64), dict(num_mfcc_features=27), dict(feature_offset=13)):
        with pytest.raises(_lib.EdisonError):
            MfccF32(ctx=ctx, **bad)


def test_f32_firmware_names(built_lib, ctx, oracle_mod, mfcc_golden):
    """mfcc_create / mfcc_compute / mfcc_delete as app.c:540,583 call them (process-global context)."""
    L = built_lib
    h = L.mfcc_create(13, 1, 512, 8, 0.97)
    assert h
    x = mfcc_golden["in_edison"]
    ref = oracle_mod.MfccF32()
    ri, rf, _ = ref(x, frame_step=256)
    out = np.zeros(12, np.int8)
    for f in range(6):
        fr = np.ascontiguousarray(x[f * 256:f * 256 + 512])
        L.mfcc_compute(h, fr.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        d = np.abs(out.astype(int) - ri[f].astype(int))
        assert d.max() <= 1 and (d != 0).sum() <= 1
    L.mfcc_delete(h)


def test_nnom_example_front_end(ctx, oracle_mod):
    """edison_f32_stream_* against a literal replay of appNnomKwsRun's buffers (app.c:545-623) on the host: 768-sample
    audio buffer (256 old + 512 new), two mfcc_compute calls per event, a 63-row ring unrolled oldest row first. The
    features themselves come from the batch path (pinned against the oracle above), so the windows must be equal bit for
    bit -- for one event per push and for many."""
    from edison_amd.mfcc.mfcc_f32 import MfccF32, NnomKwsFrontEnd
    rng = np.random.default_rng(63)
    n_ev = 70                                                        # more than 63 / 2 events: the ring wraps
    raw = (rng.normal(0, 900000, n_ev * 512)).astype(np.int32)       # 32-bit DMA words, volume as in app.c:572-575
    x = NnomKwsFrontEnd.dma_to_int16(raw)
    assert x.dtype == np.int16 and np.array_equal(x, np.clip(raw >> 8, -32768, 32767))
    m = MfccF32(ctx=ctx)
    audio = np.zeros(768, np.int16)
    ring = np.zeros((63, 12), np.int8)
    idx, want = 0, []
    for e in range(n_ev):
        audio[:256] = audio[512:768]                                 # memcpy(audio_buffer_16bit, &audio_buffer_16bit[512], 256 * 2)
        audio[256:] = x[e * 512:(e + 1) * 512]
        for i in range(2):
            ring[idx] = m.compute(audio[i * 256:i * 256 + 512], n_frames=1)[0]
            idx = (idx + 1) % 63
        want.append(np.concatenate([ring[idx:], ring[:idx]]))        # mfcc_features_seq
    want = np.stack(want)
    fe = NnomKwsFrontEnd(ctx=ctx, max_events=16)
    got = np.concatenate([fe.push(x[:512]), fe.push(x[512:512 * 40]), fe.push(x[512 * 40:])])
    assert fe.events_seen == n_ev and got.shape == (n_ev, 63, 12)
    assert np.array_equal(got, want)
    fe.reset()
    assert np.array_equal(fe.push(x[:1024]), want[:2])
    fe.close(); m.close()
