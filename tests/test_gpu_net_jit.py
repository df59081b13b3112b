"""GPU tests of edison_net_specialize (csrc/edison_net_jit.hip, net_spec.c): the loaded graph's OWN kernel -- the general
matrix-core kernel's source compiled by hipRTC with the graph's plan as constants. Same arithmetic, so every comparison is
bit for bit: against the reference-generated fixtures (tests/golden/net_golden.npz, NNoM 0.3.0 + CMSIS-NN built around the
generated headers), against the numpy restatement on seeded batches, and against the general kernel on the same context."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"]


def _header(name):
    return os.path.join(GOLDEN, "alt_models", name + ".h")


def _blob(name):
    from edison_amd import nnom_import
    with open(_header(name)) as f:
        shape, layers = nnom_import.parse_weights_h(f.read())
    return nnom_import.build_blob(shape, layers)


@pytest.fixture()
def jit_cache(tmp_path, monkeypatch):
    d = tmp_path / "jit"
    monkeypatch.setenv("EDISON_JIT_CACHE", str(d))
    monkeypatch.setenv("EDISON_NET_SPECIALIZE", "cache")           # loads only LOOK into the cache: the tests decide when to compile
    return d


@pytest.mark.parametrize("name", NAMES)
def test_own_kernel_matches_reference_fixtures_and_general_kernel(built_lib, jit_cache, name):
    from edison_amd.context import Context
    from oracle import net_ref
    g = np.load(os.path.join(GOLDEN, "net_golden.npz"))
    c = Context(0, model_path=None)
    c.load_weights_h(_header(name))
    info = c.net_info()
    n_in = info["in_h"] * info["in_w"] * info["in_c"]
    rng = np.random.default_rng(77)
    x = rng.integers(-128, 128, (3001, n_in)).astype(np.int8)      # more than one pass of the persistent grid, ragged end
    x[:300] = rng.integers(-12, 13, (300, n_in))
    x[300], x[301], x[302] = 0, 127, -128
    assert c.net_specialized() == 0
    general = c.net(x)
    assert c.net_specialize() in (1, 3) and c.net_specialized() in (1, 3)   # compiled here (the cache directory was empty)
    assert len(list(jit_cache.glob("net_gfx950_*.hsaco"))) == 1
    own = c.net(x)
    ref = net_ref.run(_blob(name), x)
    for k in ("logits", "argmax"):
        assert np.array_equal(own[k], general[k]) and np.array_equal(own[k], ref[k]), k
    if info["has_softmax"]:
        assert np.array_equal(own["softmax"], general["softmax"])
    # the reference's own outputs for this graph
    xg, acts = g["in_" + name], g["acts_" + name]
    last = info["layers"][-1]
    final = acts[:, last["acts_offset"]:last["acts_offset"] + info["n_out"]]
    out = c.net(xg)
    if info["has_softmax"]:
        pre = info["layers"][-2]
        assert np.array_equal(out["softmax"], final)
        assert np.array_equal(out["logits"], acts[:, pre["acts_offset"]:pre["acts_offset"] + info["n_out"]])
    else:
        assert np.array_equal(out["logits"], final)
    assert np.array_equal(out["argmax"], np.argmax(final, axis=1))
    assert c.net_specialize() in (1, 3)                             # a second call changes nothing
    c.close()
    # a new context finds the code object in the cache: already at load time, without being asked
    c2 = Context(0, model_path=None)
    c2.load_weights_h(_header(name))
    assert c2.net_specialized() == 2
    assert c2.net_specialize() == 2
    again = c2.net(x)
    assert np.array_equal(again["logits"], ref["logits"]) and np.array_equal(again["argmax"], ref["argmax"])
    c2.close()


def test_reload_drops_the_own_kernel_and_other_entry_points_use_it(built_lib, jit_cache, oracle_mod, oracle_model, monkeypatch):
    """The shipped kws_conv graph kept off its hand-written kernel (EDISON_NET_FORCE_GENERAL) runs on the general kernel,
    then on its own; a model load in between invalidates the own kernel of the previous graph."""
    from edison_amd.context import Context
    monkeypatch.setenv("EDISON_NET_FORCE_GENERAL", "1")
    c = Context(0)
    rng = np.random.default_rng(5)
    feat = rng.integers(-128, 128, (2500, 403)).astype(np.int8)
    o = oracle_mod.cnn(oracle_model, feat)
    a = c.net(feat)
    assert c.net_specialize() in (1, 2, 3)
    b = c.net(feat)
    for k in ("logits", "softmax", "argmax"):
        assert np.array_equal(a[k], o[k]) and np.array_equal(b[k], o[k]), k
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialized() == 0                                 # another graph: the old code object is gone
    assert c.net_specialize() in (1, 2, 3) and c.net_specialized() in (1, 2, 3)
    c.close()


def test_cache_can_be_switched_off_and_damaged_entries_are_replaced(built_lib, tmp_path, monkeypatch):
    from edison_amd.context import Context
    monkeypatch.setenv("EDISON_JIT_CACHE", "off")
    monkeypatch.setenv("EDISON_NET_SPECIALIZE", "cache")
    c = Context(0, model_path=None)
    c.load_weights_h(_header("tiny_conv"))
    assert c.net_specialize() in (1, 3)
    c.close()
    d = tmp_path / "jit2"
    monkeypatch.setenv("EDISON_JIT_CACHE", str(d))
    c = Context(0, model_path=None)
    c.load_weights_h(_header("tiny_conv"))
    assert c.net_specialize() in (1, 3)
    c.close()
    (entry,) = list(d.glob("*.hsaco"))
    entry.write_bytes(b"not a code object")
    c = Context(0, model_path=None)
    c.load_weights_h(_header("tiny_conv"))                          # the load looks into the cache by itself: a damaged entry is
    assert c.net_specialized() == 0 and not entry.exists()          # not the load's problem; it is removed ...
    assert c.net_specialize() in (1, 3)                             # ... and the next call compiles again
    entry.write_bytes(b"not a code object")
    monkeypatch.setenv("EDISON_JIT_COMPILER", "neither")
    c2 = Context(0, model_path=None)
    c2.load_weights_h(_header("tiny_conv"))
    with pytest.raises(Exception):
        c2.net_specialize()                                         # an unknown compiler name is an argument error
    c2.close()
    monkeypatch.delenv("EDISON_JIT_COMPILER")
    x = np.zeros((3, c.net_info()["in_h"] * c.net_info()["in_w"] * c.net_info()["in_c"]), np.int8)
    c.net(x)
    c.close()


@pytest.mark.parametrize("compiler,state", [("hipcc", 1), ("hiprtc", 3)])
def test_both_compilers_give_the_same_answers(built_lib, jit_cache, monkeypatch, compiler, state):
    """EDISON_JIT_COMPILER: the installed hipcc as a child process, or hipRTC inside this process (which, beside PyTorch, is the
    compiler library the wheel bundles: other code, same results)."""
    import shutil
    from edison_amd.context import Context
    from oracle import net_ref
    if compiler == "hipcc" and not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc on this machine")
    monkeypatch.setenv("EDISON_JIT_COMPILER", compiler)
    c = Context(0, model_path=None)
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialize() == state
    (entry,) = list(jit_cache.glob("*.hsaco"))
    assert ("_%s-" % compiler) in entry.name and entry.name.endswith(".hsaco")     # ..._<compiler>-<its identity>.hsaco
    info = c.net_info()
    x = np.random.default_rng(3).integers(-128, 128, (2000, info["in_h"] * info["in_w"] * info["in_c"])).astype(np.int8)
    ref = net_ref.run(_blob("kws_small"), x)
    out = c.net(x)
    assert np.array_equal(out["logits"], ref["logits"]) and np.array_equal(out["argmax"], ref["argmax"])
    c.close()


def test_a_failing_hipcc_falls_through_to_hiprtc(built_lib, jit_cache, monkeypatch, tmp_path):
    """EDISON_HIPCC pointing at a program that exits non-zero: edison_net_specialize still succeeds, through hipRTC (state 3)."""
    from edison_amd.context import Context
    bad = tmp_path / "hipcc"
    bad.write_text("#!/bin/sh\necho broken compiler >&2\nexit 3\n")
    bad.chmod(0o755)
    monkeypatch.setenv("EDISON_HIPCC", str(bad))
    c = Context(0, model_path=None)
    c.load_weights_h(_header("tiny_conv"))
    assert c.net_specialize() == 3
    x = np.zeros((5, c.net_info()["in_h"] * c.net_info()["in_w"] * c.net_info()["in_c"]), np.int8)
    c.net(x)
    c.close()


def test_a_model_load_compiles_only_when_asked(built_lib, tmp_path, monkeypatch, oracle_mod, oracle_model):
    """The library's default (EDISON_NET_SPECIALIZE unset): a model load starts no compiler, writes no file and takes nothing from a
    cache -- not even an entry an explicit call left there. =1: every load ends with edison_net_specialize (compiled at the first
    load on a machine, from the cache afterwards); =cache: a load only looks into the cache; =0: as unset."""
    from edison_amd.context import Context
    from oracle import net_ref
    monkeypatch.setenv("EDISON_JIT_CACHE", str(tmp_path / "jit3"))
    monkeypatch.delenv("EDISON_NET_SPECIALIZE", raising=False)
    c = Context(0, model_path=None)
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialized() == 0 and not list((tmp_path / "jit3").glob("*.hsaco"))
    c.close()
    monkeypatch.setenv("EDISON_NET_SPECIALIZE", "1")
    c = Context(0, model_path=None)
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialized() in (1, 3) and len(list((tmp_path / "jit3").glob("*.hsaco"))) == 1
    info = c.net_info()
    x = np.random.default_rng(8).integers(-128, 128, (1501, info["in_h"] * info["in_w"] * info["in_c"])).astype(np.int8)
    ref = net_ref.run(_blob("kws_small"), x)
    out = c.net(x)
    assert np.array_equal(out["logits"], ref["logits"]) and np.array_equal(out["argmax"], ref["argmax"])
    c.close()
    c = Context(0, model_path=None)
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialized() == 2                                  # the second load on this "machine": a file read
    c.close()
    for v, want in (("cache", 2), ("0", 0), (None, 0)):
        if v is None:
            monkeypatch.delenv("EDISON_NET_SPECIALIZE")
        else:
            monkeypatch.setenv("EDISON_NET_SPECIALIZE", v)
        c = Context(0, model_path=None)
        c.load_weights_h(_header("kws_small"))
        assert c.net_specialized() == want, (v, c.net_specialized())
        c.close()
    # the cache belongs to this user alone: a directory others can write is not used (every call compiles), and neither is an entry
    # others can write
    import os, stat
    monkeypatch.setenv("EDISON_NET_SPECIALIZE", "cache")
    entry = next((tmp_path / "jit3").glob("*.hsaco"))
    os.chmod(entry, 0o666)
    c = Context(0, model_path=None)
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialized() == 0
    c.close()
    os.chmod(entry, 0o600)
    os.chmod(tmp_path / "jit3", 0o777)
    c = Context(0, model_path=None)
    c.load_weights_h(_header("kws_small"))
    assert c.net_specialized() == 0
    c.close()
    os.chmod(tmp_path / "jit3", 0o700)
    assert stat.S_IMODE(os.stat(tmp_path / "jit3").st_mode) == 0o700


def test_kws_and_stream_entry_points_on_another_graphs_own_kernel(built_lib, jit_cache):
    """A 31 x 13 x 1 -> 10 softmax graph that is NOT kws_conv serves edison_kws_* and edison_stream_* through the network kernels:
    windows 13 bytes apart (the stream's sliding window), 403-byte utterances, one-window host pushes. Same answers from the
    general kernel and from the graph's own."""
    from edison_amd.context import Context
    from edison_amd.stream import Stream
    rng = np.random.default_rng(21)
    audio = np.clip(rng.normal(0, 3000, 9 * 31744), -32768, 32767).astype(np.int16)
    hop, chunk, n_push = 512, 5, 11
    st_audio = np.clip(rng.normal(0, 2500, n_push * chunk * hop), -32768, 32767).astype(np.int16)
    res = {}
    for own in (False, True):
        c = Context(0, model_path=None)
        c.load_weights_h(_header("kws_small"))
        if own:
            assert c.net_specialize() in (1, 2, 3)
        assert (c.net_specialized() != 0) == own
        r = c.kws(audio, n_utt=9, utt_stride=31744)
        st = Stream(c, hop=hop, chunk_frames=chunk)
        outs = [st.push(st_audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
        st.close()
        st1 = Stream(c, hop=hop, chunk_frames=1)
        one = [st1.push(st_audio[i * hop:(i + 1) * hop]) for i in range(40)]
        st1.close()
        res[own] = (r["logits"], r["softmax"], r["argmax"], np.concatenate([o["softmax"] for o in outs]), np.concatenate([o["softmax"] for o in one]))
        c.close()
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[True][3][:40], res[True][4])           # chunk 5 and chunk 1 streams see the same windows
