"""CPU tests of the product's host side: the C-ABI library loads and exports what the header declares, the
host-only C functions (mel matrix, model loader, legacy glue) behave like the reference, and nothing computes
without a GPU (the product has no CPU fallback)."""
import ctypes
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol(built_lib):
    from edison_amd import _lib
    header = open(os.path.join(ROOT, "include", "edison_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{}]*\)\s*;", header))
    declared = {d for d in declared if not d.startswith("EDISON_")}
    assert len(declared) >= 35
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert declared <= exported, declared - exported


def test_library_contains_gfx950_code_object(built_lib):
    from edison_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"ed_mfcc_kernel" in blob and b"ed_cnn_kernel" in blob


def test_no_device_fails_loudly(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from edison_amd import _lib
    from edison_amd.context import Context
    h = ctypes.c_void_p()
    assert built_lib.edison_init(0, ctypes.byref(h)) == _lib.E_NO_DEVICE
    assert b"no CPU path" in built_lib.edison_last_error(None)
    with pytest.raises(_lib.EdisonError):
        Context(0)
    # legacy surface: initialisation reports the error code instead of computing on the host
    assert built_lib.aiInitialize() == _lib.E_NO_DEVICE
    out = np.zeros(10, np.int8)
    assert built_lib.aiRunInference(np.zeros(403, np.int8).ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) < 0


def test_missing_library_raises(monkeypatch):
    from edison_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libedison_hip.so")
    with pytest.raises(FileNotFoundError, match="no CPU implementation"):
        _lib.lib()


def test_product_does_not_import_oracle():
    """The product path must never route through the checker (oracle/) or the reference."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "edison_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip")):
                txt = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt or "libnnom_ref" in txt \
                        or "/root/reference" in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_gen_mel_weight_matrix_host_c(built_lib, mfcc_golden):
    from edison_amd.mfcc import mfcc_utils as mfu
    np.testing.assert_allclose(mfu.gen_mel_weight_matrix(32, 512, 16000, 80.0, 7600.0), mfcc_golden["mel_W512"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(mfu.gen_mel_weight_matrix(32, 513, 16000, 80.0, 7600.0), mfcc_golden["mel_W513"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(mfu.gen_mel_weight_matrix(), mfcc_golden["mel_W129_20"], rtol=0, atol=1e-13)
    W = mfu.gen_mel_weight_matrix(32, 513, 16000, 80.0, 7600.0)
    assert (W != 0).sum() == 922 and (W[1:] != 0).sum(axis=1).max() == 2  # each bin feeds <= 2 adjacent bands


def test_frames_and_mel_helpers():
    from edison_amd.mfcc import mfcc_utils as mfu
    f = mfu.frames(np.arange(10), frame_length=4, frame_step=3)
    assert f.shape == (3, 4) and f[2].tolist() == [6, 7, 8, 9]
    assert mfu.frames(np.arange(3), 3, 1).shape == (1, 3)
    assert abs(mfu.hertz_to_mel(700.0) - 1127.0 * np.log(2.0)) < 1e-12
    # geometries beyond the generality kernel's limits are refused before anything touches the device; inside them (512 / 32,
    # 1024 / 26 ...) the call goes to edison_mfcc_generic (tests/test_gpu_generic.py)
    with pytest.raises(NotImplementedError):
        mfu._prepare(16000, 8192, 32, 80.0, 7600.0)
    with pytest.raises(NotImplementedError):
        mfu._prepare(16000, 1024, 300, 20.0, 4000.0)
    with pytest.raises(NotImplementedError):
        mfu._prepare(16000, 2, 32, 20.0, 4000.0)


def test_config_matches_reference_values():
    from edison_amd import config as c
    assert (c.fs, c.frame_length, c.num_mel_bins, c.num_mfcc, c.n_frames, c.nSamples) == (16000, 1024, 32, 13, 31, 32000)
    assert (c.lower_edge_hertz, c.upper_edge_hertz, c.mel_mtx_scale) == (80.0, 7600.0, 128)
    assert len(c.keywords) == 10 and c.keywords[0] == "edison" and c.keywords[9] == "_noise"


def test_legacy_glue_host_side(built_lib):
    """mfccToNetInput / mfccToNetInputPush (app.c:675-719), aiGetInputShape, keyword table: no GPU involved."""
    L = built_lib
    x, y = ctypes.c_uint16(), ctypes.c_uint16()
    L.aiGetInputShape(ctypes.byref(x), ctypes.byref(y))
    assert (x.value, y.value) == (13, 31)
    assert L.aiGetKeywordCount() == 10
    assert [L.aiGetKeywordFromIndex(i).decode() for i in range(10)] == ["edison", "cinema", "bedroom", "office",
                                                                       "livingroom", "kitchen", "on", "off", "_cold", "_noise"]
    buf = (ctypes.c_int8 * 403).from_address(L.aiNnomGetInputBuffer())
    ctypes.memset(L.aiNnomGetInputBuffer(), 0, 403)
    rows = []
    for i in range(33):  # more pushes than rows: the window keeps the newest 31
        m = (np.arange(13) * 40 - 200 + i).astype(np.int16)  # spans beyond [-128, 127] -> clipped
        rows.append(np.clip(m, -128, 127).astype(np.int8))
        L.mfccToNetInputPush(m.ctypes.data_as(ctypes.c_void_p), 13, 31)
    got = np.frombuffer(buf, dtype=np.int8).reshape(31, 13).copy()
    assert np.array_equal(got, np.stack(rows[-31:]))   # firmware appends the newest row at the bottom
    m = np.full(13, 77, np.int16)
    L.mfccToNetInput(m.ctypes.data_as(ctypes.c_void_p), 13, 31, 4)
    assert (np.frombuffer(buf, dtype=np.int8).reshape(31, 13)[4] == 77).all()


def _py_fsm(events, thr=0.5):
    """Independent restatement of edisonFSM's transitions (app.c:756-872) for the scenario test below.
    events: (pred_max, pred_idx, dt_us); returns the state after each call and the executed commands."""
    LOC, VAL, WAKE, TIMEOUT = {1, 2, 3, 4, 5}, {6, 7}, 0, 5000
    state, timeout, loc, val, states, cmds = "RESET", 0, None, None, [], []
    for mx, idx, dt in events:
        hit = mx > thr
        nxt = state
        if state == "RESET":
            nxt = "IDLE"
        elif state == "IDLE":
            if hit and idx == WAKE:
                timeout, nxt = 0, "HOT"
        elif state == "HOT":
            timeout += dt // 1000
            if hit and idx in LOC:
                loc, timeout, nxt = idx, 0, "LOC"
            if timeout > TIMEOUT:
                nxt = "IDLE"
        elif state == "LOC":
            timeout += dt // 1000
            if hit and idx in VAL:
                val, nxt = idx, "SET"
            if timeout > TIMEOUT:
                nxt = "IDLE"
        elif state == "SET":
            cmds.append((loc, val))
            nxt = "IDLE"
        state = nxt
        states.append(state)
    return states, cmds


def test_fsm_scenarios(built_lib):
    """edisonFSM without the LEDs (host C, no GPU): wake word -> location -> value, the 5 s timeouts, the truncating
    `hotTimeout += dt/1000`, a value that arrives on the very call that times out, non-keywords ignored."""
    from edison_amd.stream import Fsm
    from edison_amd.context import KEYWORDS
    frame = 64000                                                 # one 1024-sample hop at 16 kHz, in us
    happy = [(0.0, 9, frame), (90.0, 0, frame), (3.0, 8, frame), (80.0, 5, frame), (0.2, 9, frame), (70.0, 6, frame),
             (0.0, 9, frame), (0.0, 9, frame)]
    wrong_order = [(0, 0, frame), (90.0, 6, frame), (90.0, 0, frame), (90.0, 7, frame), (90.0, 2, frame), (90.0, 0, frame),
                   (90.0, 7, frame), (1.0, 0, frame)]
    timeouts = [(0, 0, frame), (90.0, 0, frame)] + [(0.1, 9, frame)] * 80 + [(90.0, 3, frame)]
    late_value = [(0, 0, frame), (90.0, 0, frame), (90.0, 1, frame)] + [(0.1, 9, 999)] * 5 + [(0.1, 9, 2500000), (0.1, 9, 2500000),
                                                                                              (90.0, 6, 1000)]
    rng = np.random.default_rng(5)
    fuzz = [(float(rng.choice([0.0, 0.4, 0.6, 100.0])), int(rng.integers(0, 10)), int(rng.choice([999, 64000, 1700000])))
            for _ in range(3000)]
    for ev in (happy, wrong_order, timeouts, late_value, fuzz):
        f = Fsm()
        got = [f.step(*e) for e in ev]
        want, cmds = _py_fsm(ev)
        assert got == want
        assert f.commands == len(cmds)
        if cmds:
            assert f.last_command == (KEYWORDS[cmds[-1][0]], KEYWORDS[cmds[-1][1]])
    f = Fsm()
    assert [f.step(*e) for e in happy] == ["IDLE", "HOT", "HOT", "LOC", "LOC", "SET", "IDLE", "IDLE"]
    assert f.last_command == ("kitchen", "on")
    f = Fsm()
    assert [f.step(*e) for e in late_value][-1] == "IDLE" and f.commands == 0   # the value came with the timeout: dropped


def _fsm_view(f):
    """What of the product's `edison_fsm` struct has a counterpart in the firmware's statics whatever the history: state, counter, wake
    word, executed commands; the pending location while it can still be consumed (LOC, SET), the pending value in SET."""
    return (f.state, f.hot_timeout_ms, f.wake_idx, f.loc_idx if f.state in (3, 4) else None, f.val_idx if f.state == 4 else None,
            f.last_loc, f.last_val, f.commands)


def _ref_view(m):
    last = m.last_command_idx()
    return (m.ediState, m.hotTimeout, m.wakeWordIdx, m.pending_location_idx() if m.ediState in (3, 4) else None,
            m.pending_value_idx() if m.ediState == 4 else None, last[0], last[1], len(m.executed))


def test_fsm_equals_the_independent_restatement_of_edisonFSM(built_lib):
    """`edison_fsm_step` (legacy.c over csrc/edison_fsm_core.h, the function the GPU stages compile too) against oracle/fsm_ref.py,
    which is written from /root/reference/firmware/src/app.c:727-928 with the firmware's own tables and pointer walks and shares no
    code with the product. (1) scenario streams: whole commands, both 5 s time-outs, a value on the call that times out, dt below
    1 ms, thresholds 0.5 .. 80. (2) an exhaustive single-step walk: every state x hit / miss x every class x dt in {999, 1000,
    64 000, 5 001 000} us x a counter around the 5000 ms limit. Parity unpinned (the reference holds no vectors for the machine)."""
    from edison_amd import _lib
    from oracle import fsm_ref
    L = built_lib
    rng = np.random.default_rng(11)
    for thr, dt in [(0.5, 64000), (30.0, 64000), (50.0, 32000), (80.0, 1_000_000), (0.5, 999), (0.5, 1_700_000)]:
        n = 6000
        idx = rng.integers(0, 10, n)
        # long holds of one class (the moving average of a real stream) mixed with jitter
        for k in range(0, n, 40):
            if rng.random() < 0.7:
                idx[k:k + int(rng.integers(1, 120))] = rng.choice([0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
        mx = rng.choice([0.0, 0.4, 0.5, 0.6, 29.9, 30.1, 79.0, 81.0, 127.0], n).astype(np.float32)
        f = _lib.Fsm()
        L.edison_fsm_init(ctypes.byref(f))
        m = fsm_ref.EdisonFsmRef(true_threshold=thr)
        for i in range(n):
            got = L.edison_fsm_step(ctypes.byref(f), float(mx[i]), int(idx[i]), dt, float(thr))
            want = m.step(mx[i], idx[i], dt)
            assert got == want and _fsm_view(f) == _ref_view(m), (thr, dt, i, _fsm_view(f), _ref_view(m))
        if thr <= 50.0 and dt >= 1000:
            assert len(m.executed) > 0 and f.commands == len(m.executed)     # the stream really drives commands through SET
    # (2) exhaustive single steps
    locs, vals = [1, 2, 3, 4, 5], [6, 7]
    n_cases = 0
    for state in range(5):
        for hit in (0.0, 127.0):
            for cls in range(10):
                for dt in (999, 1000, 64000, 5_001_000):
                    for t0 in (0, 4935, 4936, 4937, 4999, 5000, 5001, 0xFFFFFFFF - 10):
                        for li in (locs if state in (3, 4) else [locs[0]]):
                            for vi in (vals if state == 4 else [vals[0]]):
                                f = _lib.Fsm()
                                L.edison_fsm_init(ctypes.byref(f))
                                m = fsm_ref.EdisonFsmRef()
                                if state != 0:
                                    L.edison_fsm_step(ctypes.byref(f), 0.0, 9, 1000, 0.5)     # RESET -> IDLE resolves the roles
                                    m.step(0.0, 9, 1000)
                                    f.state, f.hot_timeout_ms, f.loc_idx, f.val_idx = state, t0, li, vi
                                    m.ediState, m.hotTimeout = state, t0
                                    m.loc = [e["keywordIdx"] if e["name"] else None for e in m.ediLocations].index(li)
                                    m.val = [e["keywordIdx"] if e["name"] else None for e in m.ediValues].index(vi)
                                got = L.edison_fsm_step(ctypes.byref(f), hit, cls, dt, 0.5)
                                want = m.step(hit, cls, dt)
                                assert got == want and _fsm_view(f) == _ref_view(m), (state, hit, cls, dt, t0, li, vi, _fsm_view(f), _ref_view(m))
                                n_cases += 1
    assert n_cases > 4000
    # a state that does not exist: an argument error on the host, an exception in the restatement (Error_Handler, app.c:875)
    f = _lib.Fsm(); L.edison_fsm_init(ctypes.byref(f)); f.state = 7
    assert L.edison_fsm_step(ctypes.byref(f), 1.0, 0, 1000, 0.5) == _lib.E_ARGUMENT
    m = fsm_ref.EdisonFsmRef(); m.ediState = 7
    with pytest.raises(ValueError):
        m.step(1.0, 0, 1000)


def test_fsm_roles_of_the_ten_classes(built_lib):
    """What the GPU stage of the state machine gets instead of strings: the wake word's class and the location / value masks, resolved
    from the firmware's tables (EDI_WAKEWORD app.c:50, ediLocations / ediValues app.c:135-147) against keywords.txt."""
    from edison_amd.context import KEYWORDS
    w, lm, vm = ctypes.c_int32(), ctypes.c_uint32(), ctypes.c_uint32()
    built_lib.edison_fsm_roles(ctypes.byref(w), ctypes.byref(lm), ctypes.byref(vm))
    assert KEYWORDS[w.value] == "edison"
    assert {KEYWORDS[i] for i in range(10) if lm.value >> i & 1} == {"cinema", "bedroom", "office", "livingroom", "kitchen"}
    assert {KEYWORDS[i] for i in range(10) if vm.value >> i & 1} == {"on", "off"}


def test_mel_constants_generator_reproduces_the_firmware_header(built_lib, q15_golden):
    """calcCConstants (mirror of mfcc_on_mcu.py:68-145) must emit firmware/src/audio/mel_constants.h byte for byte:
    same length and SHA-256 as the reference's committed file (fingerprint taken by gen_fixtures_q15.py)."""
    import hashlib
    from edison_amd.mfcc import mfcc_on_mcu
    text = mfcc_on_mcu.calcCConstants()
    assert len(text.encode()) == int(q15_golden["mel_constants_bytes"])
    assert hashlib.sha256(text.encode()).hexdigest() == str(q15_golden["mel_constants_sha256"])
    compact, starts, counts = mfcc_on_mcu.melMtxToUnspares(np.array([[0, 0], [3, 0], [4, 7], [0, 9], [0, 0]]))
    assert compact.tolist() == [3, 4, 7, 9] and starts.tolist() == [1, 2] and counts.tolist() == [2, 2]


def test_oracle_output_filter(oracle_mod):
    """The filter oracle against numpy float64 -> float32 arithmetic written out step by step."""
    rng = np.random.default_rng(6)
    soft = rng.integers(0, 128, (200, 10)).astype(np.int8)
    soft[50:60] = 0
    soft[60:70, 3] = 127
    filt, likely, spotted, state = oracle_mod.output_filter(soft)
    y = np.zeros(10, np.float32)
    for i in range(200):
        y = (np.float64(0.9) * y.astype(np.float64) + (np.float64(1.0) - np.float64(0.9)) * soft[i].astype(np.float32).astype(np.float64)).astype(np.float32)
        assert np.array_equal(filt[i], y)
        assert likely[i] == int(np.argmax(y))                     # np.argmax = first maximum, like arm_max_f32
        assert spotted[i] == (likely[i] if y.max() > 0.5 else -1)
    assert np.array_equal(state, y)
    # state carried over: two calls == one call
    f1, _, _, s1 = oracle_mod.output_filter(soft[:77])
    f2, _, _, _ = oracle_mod.output_filter(soft[77:], state=s1)
    assert np.array_equal(np.concatenate([f1, f2]), filt)


def test_weights_importer_roundtrip(tmp_path, oracle_mod):
    """tools/import_weights_h.py: parse a synthetic weights.h, check graph, shifts and the dense de-interleave."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import import_weights_h as imp
    rng = np.random.default_rng(0)
    W = rng.integers(-128, 128, (10, 96)).astype(np.int8)
    # forward interleave = what NNoM's generator emits for arm_fully_connected_q7_opt
    stream = []
    blk = [(0, 0), (1, 0), (0, 2), (1, 2), (2, 0), (3, 0), (2, 2), (3, 2), (0, 1), (1, 1), (0, 3), (1, 3), (2, 1), (3, 1), (2, 3), (3, 3)]
    for r in range(0, 8, 4):
        for c in range(0, 96, 4):
            stream += [W[r + dr, c + dc] for dr, dc in blk]
    stream += W[8].tolist() + W[9].tolist()
    assert np.array_equal(imp.deinterleave_dense_opt(np.array(stream, np.int8), 10, 96), W)
    # odd sizes: leftover columns and rows
    W2 = rng.integers(-128, 128, (6, 7)).astype(np.int8)
    s2 = []
    for c in range(0, 4, 4):
        s2 += [W2[dr, c + dc] for dr, dc in blk]
    for c in range(4, 7):
        s2 += [W2[dr, c] for dr in range(4)]
    s2 += W2[4].tolist() + W2[5].tolist()
    assert np.array_equal(imp.deinterleave_dense_opt(np.array(s2, np.int8), 6, 7), W2)
    # the committed blob parses with the independent oracle reader and has the SURVEY A.2 shifts
    M = oracle_mod.Model()
    shifts = [(L["bias_lshift"], L["out_rshift"]) for L in M.layers if "out_rshift" in L]
    assert shifts == [(4, 8), (5, 8), (7, 7), (9, 9), (2, 10)]
    assert [L["w"].size for L in M.layers if "w" in L] == [400, 4608, 18432, 18432, 960]


def test_model_loader_rejects_bad_blobs(built_lib):
    """ed_parse_model is reached through edison_model_load_mem only with a context; validate the blob format
    checks host-side through a tiny C driver-free path: the header magic and the topology table."""
    blob = open(os.path.join(ROOT, "edison_amd", "data", "kws_nnom.ednn"), "rb").read()
    assert blob[:8] == b"EDNNOM1\0"
    in_h, in_w, in_c, n_layers, payload, flags = struct.unpack("<6i", blob[8:32])
    assert (in_h, in_w, in_c, n_layers, flags) == (31, 13, 1, 8, 1)
    assert len(blob) == 40 + 48 * n_layers + payload
    types = [struct.unpack("<12i", blob[40 + 48 * i:88 + 48 * i])[0] for i in range(n_layers)]
    assert types == [1, 2, 1, 2, 1, 1, 3, 4]


def test_pad_or_cut_and_cli_dispatch(tmp_path, capsys):
    from edison_amd.kws import kws_host
    from edison_amd import main as cli
    d = np.arange(10, dtype=np.int16)
    assert kws_host.pad_or_cut(d, 16, "zero").tolist() == list(range(10)) + [0] * 6
    assert kws_host.pad_or_cut(d, 16, "edge").tolist() == list(range(10)) + [9] * 6
    assert kws_host.pad_or_cut(d, 4).tolist() == [0, 1, 2, 3]
    assert cli.main(["main.py", "nosuch"]) == 1
    assert cli.main(["main.py", "mfcc", "bogus"]) == 1
    assert cli.main(["main.py", "kws", "bogus"]) == 1
    assert cli.main(["main.py", "kws", "live", "host"]) == 0   # reference quirk (exit 1 after running) not kept
    assert cli.main(["main.py", "mfcc", "host", str(tmp_path / "missing.wav")]) == 1
    assert cli.main(["main.py", "mfcc", "mcu"]) == 1
    assert cli.main(["main.py", "mfcc", "mcu", "nosuch"]) == 1
    assert cli.main(["main.py", "kws", "mcu", "mic"]) == 1
    hdr = tmp_path / "mel_constants.h"
    assert cli.main(["main.py", "mfcc", "mcu", "calc", str(hdr)]) == 0     # host-only: no GPU needed
    assert hdr.read_text().startswith("#define MEL_SAMPLE_SIZE          1024\n") and hdr.stat().st_size > 200000


def test_shard_range():
    from edison_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 262144, 2097152, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2097152, 3, 8) == (786432, 1048576)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


@pytest.mark.parametrize("variant,nbins,scale", [(0, 512, 0.5), (1, 513, 0.5 / 1024.0 / np.sqrt(2.0))])
def test_mfcc_lane_tables_hold_the_whole_filterbank(built_lib, mfcc_golden, variant, nbins, scale):
    """tables.c spreads the mel matrix over the wavefront (quarter r of band pair (b, 31-b) per lane) and is free to
    choose the column order and the half order (bank-conflict-aware assignment): whatever it chooses, putting the
    lane tables back together must give the reference's matrix (x the variant's spectrum scale), each weight once."""
    from edison_amd import _lib
    built_lib.ed_build_mfcc_tables.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                               ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    buf = (ctypes.c_char * 65536)()
    err = ctypes.create_string_buffer(256)
    assert built_lib.ed_build_mfcc_tables(variant, 16000.0, 80.0, 7600.0, 128.0, buf, err, 256) == _lib.OK, err.value
    i32 = np.frombuffer(buf, dtype=np.int32)
    f32 = np.frombuffer(buf, dtype=np.float32)
    o = 2 * 8 * 64 * 2                                            # tw1, tw2 (floats)
    slo, shi, band, half = (i32[o + 64 * k:o + 64 * (k + 1)] for k in range(4))
    o += 4 * 64 + 2 * 64 * 4 + 4 * 64 * 2                         # + dct4, twp
    w4 = f32[o:o + 9 * 64 * 4].reshape(9, 64, 4)
    nlo, nhi = int(i32[o + 9 * 64 * 4]), int(i32[o + 9 * 64 * 4 + 1])
    assert (nlo, nhi) == (2, 5)
    assert sorted(band[:16].tolist()) == list(range(16)) and all((band[16 * r:16 * r + 16] == band[:16]).all() for r in range(4))
    assert set(half.tolist()) <= {0, 1}
    W = np.zeros((516, 32))
    for lane in range(64):
        for part, (first, n, base) in enumerate(((slo[lane], nlo, 0), (shi[lane], nhi, nlo))):
            j = band[lane] if part == 0 else 31 - band[lane]
            for t in range(n):
                W[4 * (first + t):4 * (first + t) + 4, j] += w4[base + t, lane]
    ref = mfcc_golden["mel_W512" if nbins == 512 else "mel_W513"] * scale
    np.testing.assert_allclose(W[:nbins], ref, rtol=2e-7, atol=1e-12)
    assert not W[nbins:].any()


def test_dct2_makhoul_helper_and_mfcc_tf_signature():
    """mfcc_utils.dct2Makhoul (mfcc_utils.py:324-343) = scipy's DCT-II, with the reordered input and its FFT."""
    from scipy.fftpack import dct
    from edison_amd.mfcc import mfcc_utils as mfu
    rng = np.random.default_rng(5)
    for n in (32, 31, 2, 1):
        x = rng.normal(size=n)
        d, v, V = mfu.dct2Makhoul(x)
        np.testing.assert_allclose(d, dct(x, 2), rtol=1e-12, atol=1e-12)
        assert sorted(v.tolist()) == sorted(x.tolist()) and np.allclose(V, np.fft.fft(v))
    # mfcc_tf takes the reference's positional arguments (mfcc_utils.py:201-204) and, like its siblings, refuses a geometry
    # the GPU path is not built for before it touches the device
    import inspect
    assert list(inspect.signature(mfu.mfcc_tf).parameters) == ["data", "fs", "nSamples", "frame_len", "frame_step", "frame_count",
                                                              "fft_len", "mel_nbins", "mel_lower_hz", "mel_upper_hz", "unused"]
    with pytest.raises(NotImplementedError):
        mfu.mfcc_tf(np.zeros(1024), 16000, 1024, 1024, 1024, 0, 2048, 32, 80.0, 7600.0)
    with pytest.raises(NotImplementedError):     # other geometries go to the generality kernel (round 5) -- within its limits, checked before the device is touched
        mfu.mfcc_tf(np.zeros(16384), 16000, 16384, 8192, 8192, 0, 8192, 32, 80.0, 7600.0)


def test_net_out_filt_mirror():
    """kws_live.netOutFilt (kws_live.py:139-152): same recurrence, same leading zero row, float64."""
    from edison_amd.kws import kws_live
    rng = np.random.default_rng(6)
    outs = rng.uniform(0, 1, (40, 10)).tolist()
    got = kws_live.netOutFilt(outs, 0.9)
    ref = [10 * [0]]
    for o in outs:                                        # the reference's loop, element by element
        ref.append([0.9 * ref[-1][i] + (1.0 - 0.9) * o[i] for i in range(10)])
    assert got.shape == (41, 10) and np.array_equal(got, np.array(ref))


def test_create_dct_matrix_host(built_lib):
    """create_dct_matrix (firmware/src/audio/mfcc.h:61, mfcc.c:101-115): sqrt(2/N) cos(pi/N (n + 1/2) k) in float32."""
    built_lib.create_dct_matrix.restype = ctypes.POINTER(ctypes.c_float)
    built_lib.create_dct_matrix.argtypes = [ctypes.c_int32, ctypes.c_int32]
    p = built_lib.create_dct_matrix(26, 13)
    m = np.ctypeslib.as_array(p, shape=(13, 26)).copy()
    ctypes.CDLL(None).free(p)
    k, n = np.arange(13, dtype=np.float32)[:, None], np.arange(26, dtype=np.float32)[None, :]
    f = np.float32
    ang = f(np.pi) / f(26) * (n + f(0.5)) * k                                 # the firmware's float32 operation order
    np.testing.assert_allclose(m, np.sqrt(f(2.0) / f(26)) * np.cos(ang), rtol=0, atol=2e-7)
    np.testing.assert_allclose(m, np.sqrt(2.0 / 26) * np.cos(np.pi / 26 * (n.astype(float) + 0.5) * k), rtol=0, atol=2e-6)
    assert not built_lib.create_dct_matrix(0, 3)


def test_planner_refuses_overlapping_tensors_under_asan(tmp_path):
    """model_net.c sizes its weight / seed buffers from the payload: records whose tensors OVERLAP in the payload add
    up to more than that. The planner must refuse such a blob (EDISON_E_SIZE), not write past the buffers -- run under
    gcc's AddressSanitizer (CPU build only), on the two shapes the advisor's report names."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "edison_internal.h"
int main(int argc, char **argv)
{
	FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	void *blob = malloc(n); if (fread(blob, 1, n, f) != (size_t)n) return 3; fclose(f);
	ed_net_plan_t plan; int8_t *w; int32_t *s; char err[256] = "";
	int rc = ed_plan_net(blob, n, &plan, &w, &s, err, sizeof(err));
	printf("%d %s\n", rc, err); free(w); free(s); free(blob); return 0;
}
''')
    exe = tmp_path / "drv"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-I", os.path.join(root, "edison_amd", "csrc"),
                           "-o", str(exe), str(drv), os.path.join(root, "edison_amd", "csrc", "model_net.c")])

    def blob(h, w, c, layers, payload_bytes):
        recs = b"".join(struct.pack("<12i", *r) for r in layers)
        return b"EDNNOM1\0" + struct.pack("<8i", h, w, c, len(layers), payload_bytes, 0, 0, 0) + recs + bytes(payload_bytes)

    conv = lambda oc, cin, woff: (1, oc, 1, 1, 1, 1, 0, 0, 0, woff, 0, cin)
    cases = {
        # two 1x1 convolutions with 64 output channels, both reading the weight range at offset 0 (input 1x2x64)
        "weights": blob(1, 2, 64, [conv(64, 64, 0), conv(64, 64, 0)], 4096),
        # 32 one-channel layers: every layer adds a seed although the payload holds only 16 bytes
        "seeds": blob(1, 2, 1, [conv(1, 1, 0)] * 32, 16),
    }
    for name, b in cases.items():
        f = tmp_path / (name + ".ednn")
        f.write_bytes(b)
        r = subprocess.run([str(exe), str(f)], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr[-2000:])      # ASan aborts with a non-zero code on a bad write
        assert "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
        code = int(r.stdout.split()[0])
        assert code in (0, -3), (name, r.stdout)               # accepted within capacity, or EDISON_E_SIZE
    # the advisor's first shape exceeds the capacity by construction: it must be the refusal
    r = subprocess.run([str(exe), str(tmp_path / "weights.ednn")], capture_output=True, text=True)
    assert r.stdout.startswith("-3 "), r.stdout


# ---- the graph's own kernel (edison_net_specialize): what can be checked without a GPU
def _alt_blob(name):
    from edison_amd import nnom_import
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alt_models", name + ".h")) as f:
        shape, layers = nnom_import.parse_weights_h(f.read())
    return nnom_import.build_blob(shape, layers)


def _hiprtc_compile(src, headers, opts):
    """hipRTC through ctypes, the calls csrc/edison_net_jit.hip makes; returns (status, log, code-object bytes)."""
    import ctypes
    try:
        R = ctypes.CDLL("libhiprtc.so")
    except OSError:
        try:
            R = ctypes.CDLL("/opt/rocm/lib/libhiprtc.so")
        except OSError:
            pytest.skip("libhiprtc.so is not installed")
    prog = ctypes.c_void_p()
    hs = (ctypes.c_char_p * len(headers))(*[h[1] for h in headers])
    hn = (ctypes.c_char_p * len(headers))(*[h[0] for h in headers])
    R.hiprtcCreateProgram.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p)]
    assert R.hiprtcCreateProgram(ctypes.byref(prog), src, b"cnn_net_mfma_kernels.hip", len(headers), hs, hn) == 0
    arr = (ctypes.c_char_p * len(opts))(*opts)
    R.hiprtcCompileProgram.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    r = R.hiprtcCompileProgram(prog, len(opts), arr)
    n = ctypes.c_size_t()
    R.hiprtcGetProgramLogSize.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    R.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
    log = ctypes.create_string_buffer(n.value + 1)
    R.hiprtcGetProgramLog.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    R.hiprtcGetProgramLog(prog, log)
    R.hiprtcGetCodeSize.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    R.hiprtcGetCodeSize(prog, ctypes.byref(n))
    code = ctypes.create_string_buffer(max(n.value, 1))
    R.hiprtcGetCode.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    if r == 0:
        R.hiprtcGetCode(prog, code)
    return r, log.value.decode(errors="replace"), code.raw[:n.value]


def test_net_spec_source_and_the_embedded_kernel_text(built_lib):
    """edison_net_spec_source: the plan of a graph as C++ constants; the library carries the kernel's text and headers."""
    import ctypes
    from edison_amd import _lib
    from edison_amd.context import net_spec_source
    text = net_spec_source(open(_lib.DEFAULT_MODEL, "rb").read())
    assert "#define EMM_SPEC_NL 8" in text and "#define EMM_SP_in_h 31" in text and "#define EMM_SP_in_w 13" in text
    assert "#define EMM_SP_out_n 10" in text and "static constexpr ed_mm_run_t EMM_SR[EMM_SPEC_NL]" in text
    assert text == net_spec_source(open(_lib.DEFAULT_MODEL, "rb").read())           # deterministic: the cache key hangs on it
    assert text != net_spec_source(_alt_blob("kws_small"))
    with pytest.raises(_lib.EdisonError):
        net_spec_source(b"not a model")
    L = _lib.lib()
    for sym, path in (("ed_jit_src_kernel", "edison_amd/csrc/cnn_net_mfma_kernels.hip"), ("ed_jit_src_edison_hip_h", "include/edison_hip.h"),
                      ("ed_jit_src_edison_internal_h", "edison_amd/csrc/edison_internal.h")):
        n = ctypes.c_size_t.in_dll(L, sym + "_len").value
        data = ctypes.string_at(ctypes.addressof((ctypes.c_ubyte * 1).in_dll(L, sym)), n)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        assert data == open(os.path.join(root, path), "rb").read(), sym            # the library is in step with the tree


@pytest.mark.parametrize("name", ["(shipped)", "same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"])
def test_own_kernel_compiles_for_every_fixture_graph(built_lib, name):
    """The run-time compilation of edison_net_specialize, done here through hipRTC with the same text, headers and options
    (it needs no GPU): every fixture graph's own kernel builds, exports ed_net_mfma_spec and spills nothing to scratch."""
    import ctypes
    from edison_amd import _lib
    from edison_amd.context import net_spec_source
    L = _lib.lib()
    def text(sym):
        n = ctypes.c_size_t.in_dll(L, sym + "_len").value
        return ctypes.string_at(ctypes.addressof((ctypes.c_ubyte * 1).in_dll(L, sym)), n)
    blob = open(_lib.DEFAULT_MODEL, "rb").read() if name == "(shipped)" else _alt_blob(name)
    stdint = (b"#pragma once\ntypedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
              b"typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n"
              b"typedef unsigned long uintptr_t;\n")
    headers = [(b"stdint.h", stdint), (b"stddef.h", b"#pragma once\n"), (b"edison_hip.h", text("ed_jit_src_edison_hip_h")),
               (b"edison_internal.h", text("ed_jit_src_edison_internal_h")), (b"emm_spec.h", net_spec_source(blob).encode())]
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-fno-slp-vectorize", b"-DEMM_JIT=1", b"-DEMM_SPEC=1", b'-DEMM_SPEC_HEADER="emm_spec.h"',
            b"-mllvm", b"-pragma-unroll-threshold=1000000"]
    r, log, code = _hiprtc_compile(text("ed_jit_src_kernel"), headers, opts)
    assert r == 0, log[:2000]
    assert b"ed_net_mfma_spec" in code and len(code) > 4096
    assert "loop not unrolled" not in log, log[:2000]   # the layer loop MUST unroll: the layer records only become constants then


def test_hot_kernels_use_no_scratch(tmp_path):
    """The frame loops of the hot kernels hold everything in registers. A 16-wave build of the Q15 kernel once ran with 19 registers of
    hoisted per-lane addresses in scratch (20 MB of writes per launch that only the HBM counters showed, profiles/r03_wave_priorities.txt):
    the compiler's resource remarks are checked here, on the build flags the library uses (hipcc cross-compiles without a GPU)."""
    from concurrent.futures import ThreadPoolExecutor
    from edison_amd import build as B
    want = {  # file -> (substring of the mangled kernel name, maximum VGPRs for the occupancy the launch code assumes)
        "mfcc_kernels.hip": [("ed_mfcc2_kernel", 168), ("ed_mfcc2_list_kernel", 168), ("ed_mfcc2_window_kernel", 168)],   # 12 waves per CU; the list kernel and variant TF's instances (round 5) likewise
        "mfcc_q15_kernels.hip": ("ed_mfcc_q15_kernelILb0E", 128),  # 16 waves per CU; the stage-dump instances (ILb1E) are diagnostics
        "cnn_mfma_kernels.hip": ("ed_cnn_mfma_kernel", 256),   # 8 waves per CU (LDS-bound)
        # both instances of the general network kernel, 12 waves per CU: round 4 found 33 spilled registers in its input / output stages,
        # whose scratch reloads waited on the next batch's prefetch (-10 %)
        "cnn_net_mfma_kernels.hip": ("ed_net_mfma_kernel", 168),
    }

    def remarks(name):
        cmd = [B._hipcc(), "--offload-arch=" + B.ARCH, "--cuda-device-only", "-c", "-std=c++17", "-fno-slp-vectorize", "-O3", "-I" + B.CSRC,
               "-Rpass-analysis=kernel-resource-usage"] + B.PER_FILE_FLAGS.get(name, []) + ["-x", "hip", os.path.join(B.CSRC, name), "-o", str(tmp_path / (name + ".o"))]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        texts = dict(zip(want, ex.map(remarks, want)))
    for name, entries in want.items():
        blocks = re.split(r"remark: Function Name: ", texts[name])[1:]
        for kernel, max_vgprs in (entries if isinstance(entries, list) else [entries]):
            seen = 0
            for b in blocks:
                if kernel not in b.split()[0]:
                    continue
                seen += 1
                scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
                vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
                assert scratch == 0, (name, b.split()[0], "scratch bytes per lane", scratch)
                assert vgprs <= max_vgprs, (name, b.split()[0], vgprs)
            assert seen >= 1, (name, kernel)


# ---------------------------------------------------------------------------------------------------------------------------
# lab knobs never reach the product library (round-3 review: "nothing tests that the shipped .so was built with every knob
# at its default")
KNOB_FILES = ["mfcc_kernels.hip", "mfcc_one_frame.h", "mfcc_q15_kernels.hip", "mfcc_f32_kernels.hip", "cnn_mfma_kernels.hip",
              "cnn_net_kernels.hip", "cnn_net_mfma_kernels.hip"]
# macros with an `#ifndef X / #define X` default that are MODES set by the library itself, not knobs: the run-time compiler of
# edison_net_specialize defines them (edison_net_jit.hip)
MODE_MACROS = {"EMM_JIT", "EMM_SPEC"}


def _knobs_of(text):
    """(guarded names, {name: default}) of a kernel file: the names in its `#if !defined(ED_LAB) && (defined(A) || ...)` guard and
    every `#ifndef NAME / #define NAME value` default block it holds."""
    g = re.search(r"#if !defined\(ED_LAB\) && \(([^\n]*)\)\n#error", text)
    guarded = set(re.findall(r"defined\((\w+)\)", g.group(1))) if g else set()
    defaults = dict(re.findall(r"#ifndef (E[A-Z0-9]*_\w+)[^\n]*\n#define \1[ \t]+([^\n/]*?)[ \t]*(?:/\*[^\n]*)?\n", text))
    return guarded, defaults


def test_every_compile_time_knob_of_a_kernel_file_is_behind_ed_lab():
    """A macro a kernel file gives a default to (`#ifndef X / #define X v`) is either a mode the library sets itself or a lab knob --
    and a lab knob must be named in the file's guard, which refuses it in a build without ED_LAB."""
    from edison_amd import build as B
    seen = 0
    for f in KNOB_FILES:
        guarded, defaults = _knobs_of(open(os.path.join(B.CSRC, f)).read())
        free = set(defaults) - guarded - MODE_MACROS
        assert not free, (f, "knobs outside the ED_LAB guard", sorted(free))
        assert guarded <= set(defaults) | {"EQ_WAVES_PER_EU"}, (f, sorted(guarded - set(defaults)))
        seen += len(guarded)
    assert seen >= 15
    # the retired switches are gone for good: nothing may bring back a "results are WRONG when non-zero" macro without the guard
    mk = open(os.path.join(B.CSRC, "mfcc_kernels.hip")).read()
    for gone in ("ED2_ABLATE", "ED2_SKIP", "ED2_T1_LDS", "ED2_TW_LDS", "ED2_LATE_DRAW", "ED2_STAGGER ", "ED2_MAX_WGS"):
        assert gone not in mk, gone


def test_product_compile_refuses_a_lab_knob_and_a_lab_build_says_so(tmp_path):
    """Preprocess kernel files the way edison_amd/build.py compiles them: (1) as they are -- every knob has the value its file gives
    it and ED_LAB is undefined; (2) with a knob on the command line -- the guard's #error; (3) the same with -DED_LAB -- accepted,
    and the translation unit then carries the ed_lab_build_* marker a product library must not export."""
    from edison_amd import build as B

    def pp(name, extra, want_ok):
        cmd = [B._hipcc(), "--offload-arch=" + B.ARCH, "--cuda-device-only", "-std=c++17", "-I" + B.CSRC, "-E", "-dD"] + extra + ["-x", "hip", os.path.join(B.CSRC, name)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert (r.returncode == 0) == want_ok, (name, extra, r.stderr[-1500:])
        return r.stdout if want_ok else r.stderr

    for name, knob in (("mfcc_kernels.hip", "ED2_PRIO"), ("mfcc_q15_kernels.hip", "EQ_ABLATE"), ("cnn_mfma_kernels.hip", "EDM_PRIO"),
                       ("cnn_net_mfma_kernels.hip", "EMM_SKIP")):
        text = open(os.path.join(B.CSRC, name)).read()
        guarded, defaults = _knobs_of(text)
        out = pp(name, [], True)
        assert not re.search(r"#define ED_LAB\b", out), name
        for k in guarded & set(defaults):
            m = re.findall(r"#define %s[ \t]+([^\n]*)" % k, out)
            assert m and m[-1].split("/*")[0].strip() == defaults[k].strip(), (name, k, m, defaults[k])
        assert "ed_lab_build_" not in re.sub(r"/\*.*?\*/", "", out, flags=re.S).split("#define")[0] or True
        assert not re.search(r"const int ed_lab_build_\w+ = 1", out), name
        err = pp(name, ["-D%s=0" % knob], False)
        assert "lab knob defined without ED_LAB" in err, err[-500:]
        lab = pp(name, ["-DED_LAB", "-D%s=0" % knob], True)
        assert re.search(r"const int ed_lab_build_\w+ = 1", lab), name


def test_product_library_is_not_a_lab_build_and_build_py_refuses_lab_flags(monkeypatch):
    from edison_amd import build as B, _lib
    r = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0
    assert "ed_lab_build_" not in r.stdout, [l for l in r.stdout.splitlines() if "ed_lab_build_" in l]
    for flags in ("-DED2_PRIO=0", "-O2 -DEQ_ABLATE=4", "-DED_LAB", "-UEMM_PRIO"):
        monkeypatch.setenv("ED_CFLAGS", flags)
        with pytest.raises(RuntimeError, match="lab macros"):
            B.product_flags()
    monkeypatch.setenv("ED_CFLAGS", "-g -fno-omit-frame-pointer")
    assert B.product_flags() == ["-g", "-fno-omit-frame-pointer"]
    # the text the run-time compiler of edison_net_specialize receives is the product text, and it is handed no knob
    jit = open(os.path.join(B.CSRC, "edison_net_jit.hip")).read()
    i = jit.index('getenv("EDISON_JIT_DEFINE")')
    assert "#ifdef ED_LAB" in jit[i - 200:i], "EDISON_JIT_DEFINE must only exist in a lab build of edison_net_jit.hip"
    assert B.JIT_TEXTS[0][1].endswith("cnn_net_mfma_kernels.hip")


def test_cnn_column_tables_cover_every_column_and_meet_no_bank_conflict(built_lib):
    """csrc/cnn_mfma_cols.h (the order in which the lanes of a FULL group take the columns of conv1 / conv2 / conv3) and what model.c
    makes of it: every live column exactly once, the packed offsets are the columns' LDS addresses, idle lanes re-read a live
    column of their tile -- and, under the LDS model that the counters confirmed for the natural order (tools/dev/cnn_lds_model.py),
    the operand reads and epilogue stores of a group lose at most 40 cycles to bank conflicts (natural order: 457)."""
    import importlib.util
    from edison_amd import _lib
    spec = importlib.util.spec_from_file_location("cnn_lds_model", os.path.join(ROOT, "tools", "dev", "cnn_lds_model.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    tabs = M.read_header()
    total_nat = total = 0
    for L in (M.Conv1, M.Conv2, M.Conv3):
        cols = tabs[L.name]
        assert len(cols) == L.tiles and sorted(q for r in cols for q in r if q is not None) == list(range(L.n))
        total_nat += M.layer_cost(L, M.natural(L))
        total += M.layer_cost(L, cols)
    assert total_nat == 457 and total <= 40, (total_nat, total)
    # the tables as the kernel gets them (ed_parse_model -> ed_cnn_mfma_model_t.cols1 / cols2 / cols3)
    blob = open(_lib.DEFAULT_MODEL, "rb").read()
    plain, mfma = (ctypes.c_char * (1 << 17))(), (ctypes.c_char * (1 << 17))()
    err = ctypes.create_string_buffer(256)
    built_lib.ed_parse_model.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    assert built_lib.ed_parse_model(blob, len(blob), plain, mfma, err, 256) == _lib.OK, err.value
    off = (15 + 5 + 18 + 18 + 2) * 1024 + (16 + 32 + 64 + 32 + 16) * 4 + 8 * 4
    u32 = np.frombuffer(mfma, dtype=np.uint32, count=(2 + 5 + 2) * 32, offset=off)
    UTT, REGA = 2992, 1120
    want = {
        "conv1": lambda q: ((q // 13) * UTT + (q % 13) * 16, (q // 13) * UTT + REGA + (q % 13) * 9 * 16),
        "conv2": lambda q: ((q // 35) * UTT + REGA + ((2 * ((q % 35) // 7)) * 9 + (q % 35) % 7) * 16, (q // 35) * UTT + ((q % 35) // 7 * 7 + (q % 35) % 7) * 16),
        "conv3": lambda q: ((q // 15) * UTT + (((q % 15) // 5) * 7 + (q % 15) % 5) * 16, (q // 15) * UTT + REGA + (((q % 15) // 5) * 5 + (q % 15) % 5) * 16),
    }
    at = 0
    for name, tiles in (("conv1", 2), ("conv2", 5), ("conv3", 2)):
        for t in range(tiles):
            live_reads = {want[name](q)[0] for q in tabs[name][t] if q is not None}
            for c in range(32):
                e, q = int(u32[at]), tabs[name][t][c]
                at += 1
                if q is None:
                    assert e >> 31 and (e & 0x7fff) in live_reads, (name, t, c, hex(e))
                else:
                    assert (e & 0x7fff, (e >> 16) & 0x7fff, e >> 31) == want[name](q) + (0,), (name, t, c, q, hex(e))
