"""CPU tests of the matrix-core PLANNER (edison_amd/csrc/model_net_mm.c -- host logic of the general network path): the plan
of a graph (tables, fragments, seeds, region offsets) is walked in numpy exactly as the GPU kernel walks it
(tests/plan_emulator.py) and must reproduce oracle/net_ref.py bit for bit -- the restatement that
tests/golden/gen_fixtures_net.py checked against the reference's own NNoM build. No GPU, no kernel: what is tested is every
address and every packed byte the planner hands to the kernel, in all of its forms (row-Toeplitz first layers, pixel gaps,
16 x 16 x 64 tiles, fused pooling windows, per-wave batches of 1 / 2 / 4, ragged last batches)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
NAMES = ["same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"]
KNOBS = [{}, {"EDISON_NET_NO_TOEPLITZ": "1"}, {"EDISON_NET_NO_PIXEL_GAP": "1"}, {"EDISON_NET_NO_SMALL_TILES": "1"},
         {"EDISON_NET_BATCH": "1"}, {"EDISON_NET_BATCH": "4", "EDISON_NET_MIN_WAVES": "1"}]


def _blob(name):
    from edison_amd import _lib, nnom_import
    if name == "(shipped)":
        return open(_lib.DEFAULT_MODEL, "rb").read()
    with open(os.path.join(HERE, "golden", "alt_models", name + ".h")) as f:
        shape, layers = nnom_import.parse_weights_h(f.read())
    return nnom_import.build_blob(shape, layers)


def _check(blob, x, what):
    import plan_emulator
    from oracle import net_ref
    plan = plan_emulator.Plan(blob)
    got, ref = plan_emulator.run(plan, x), net_ref.run(blob, x)
    assert np.array_equal(got["logits"], ref["logits"]), what
    assert np.array_equal(got["argmax"], ref["argmax"]), what
    return plan


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: "+".join("%s=%s" % kv for kv in k.items()) or "default")
@pytest.mark.parametrize("name", ["(shipped)"] + NAMES)
def test_plan_walk_equals_the_restatement(built_lib, monkeypatch, name, knobs):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    blob = _blob(name)
    rng = np.random.default_rng(len(name) + len(knobs))
    import plan_emulator
    n_in = plan_emulator.Plan(blob).P.in_n
    x = rng.integers(-128, 128, (5, n_in)).astype(np.int8)            # 5: ragged against batches of 2 and 4
    x[0] = rng.integers(-9, 10, n_in)
    x[1] = 127
    plan = _check(blob, x, (name, knobs))
    if name == "(shipped)" and not knobs:
        # the shipped graph's plan, as DESIGN 4.5b describes it: Toeplitz first layer, two inputs per wave at 12 waves, gaps
        assert plan.ML[0].toep == 1 and plan.ML[0].expand == 0 and plan.M.x_bytes == 0
        assert (plan.M.batch, plan.M.waves) == (2, 12) and plan.M.lds_bytes <= 156 * 1024
        assert plan.ML[4].pp == 48 and plan.ML[5].pp == 80


def test_plan_walk_on_random_graphs(built_lib):
    """Seeded random graphs (the generator of tools/fuzz_net.py, which runs them on the GPU): every graph the planner accepts
    walks to the restatement's answer; the ones it refuses are refused with the documented codes."""
    import fuzz_net
    import plan_emulator
    from edison_amd import _lib, nnom_import
    rng = np.random.default_rng(2024)
    walked = refused = toep = gap = small = 0
    while walked < 60:
        g = fuzz_net.random_graph(rng)
        if g is None:
            continue
        shape, layers = g
        try:
            blob = nnom_import.build_blob(shape, [dict(L) for L in layers])
            plan = plan_emulator.Plan(blob)
        except _lib.EdisonError as e:
            assert e.code in (_lib.E_SIZE, _lib.E_NO_IMPL), str(e)
            refused += 1
            continue
        x = rng.integers(-128, 128, (3, shape[0] * shape[1] * shape[2])).astype(np.int8)
        x[0] = rng.integers(-10, 11, x.shape[1])
        _check(blob, x, (shape, [(L["type"], {k: v for k, v in L.items() if k not in ("w", "b")}) for L in layers]))
        walked += 1
        toep += any(m.toep for m in plan.ML)
        gap += any(m.mm and m.pp not in (0, plan.PL[i].in_c, plan.PL[i].in_n) for i, m in enumerate(plan.ML))
        small += any(r.small for r in plan.R)
    assert toep >= 5 and small >= 5, (toep, gap, small)      # the sample reaches the planner's special forms


def test_high_byte_requantisation_is_offered_only_where_it_is_exact(built_lib):
    """ed_mm_run_t.rs carries ED_RUN_RS_HI (0x100) when sat8(v >> rs) may be taken as the high byte of sat16(v >> (rs - 8)): always for
    rs >= 8; for rs < 8 the inner shift goes left, and the planner allows it only when 128 * sum|w| + |seed| of the layer's worst
    output channel stays inside 32 bits after it. The emulator walks either plan with the plain formula."""
    import plan_emulator
    from edison_amd import nnom_import
    rng = np.random.default_rng(11)

    def conv(oc, k, c, rs, wmax, bl=0):
        return dict(type=1, out_ch=oc, kh=k, kw=k, sh=1, sw=1, w=rng.choice([-wmax, wmax], oc * k * k * c).astype(np.int8),
                    b=np.full(oc, 100, np.int8), out_rshift=rs, bias_lshift=bl, relu=1, same=0)

    seen = set()
    for rs, wmax, bl in ((8, 127, 0), (12, 127, 0), (7, 10, 0), (2, 1, 0), (7, 127, 0), (1, 127, 0), (0, 127, 0), (3, 127, 0), (6, 127, 23), (7, 127, 23), (4, 3, 22),
                         (8, 127, 23)):
        # 3 x 3 x 64 taps of +-wmax and a bias of 100 << bl: the bound is 128 * 576 * wmax + (100 << bl) + the rounding constant
        layers = [conv(64, 3, 16, 9, 50), conv(8, 3, 64, rs, wmax, bl)]
        blob = nnom_import.build_blob((9, 7, 16), [dict(L) for L in layers])
        plan = plan_emulator.Plan(blob)
        bound = 128 * 576 * wmax + (100 << bl) + (1 << rs >> 1)
        exact = rs >= 8 or (bound << (8 - rs)) < 2 ** 31
        seen.add((rs >= 8, exact))
        assert (plan.R[1].rs & 0xff) == rs and bool(plan.R[1].rs & 0x100) == exact, (rs, wmax, bl, hex(plan.R[1].rs))
        assert plan.R[0].rs == 9 | 0x100
        x = rng.integers(-128, 128, (2, 9 * 7 * 16)).astype(np.int8)
        _check(blob, x, (rs, wmax, bl))
    assert seen == {(True, True), (False, True), (False, False)}


def test_planners_under_address_and_ub_sanitizers(built_lib):
    """tools/verify/asan_planner.sh: the library's host C (planners, kws_conv model parser, the table builders of all four MFCC
    variants) rebuilt with gcc -fsanitize=address,undefined and run on the shipped graph, the seven fixture graphs, 80 random
    graphs, truncated copies of every blob (exact-size heap copies: all refused) and every filterbank / frame length the tests
    configure: no report. (Sanitizers exist on the CPU build only.)"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(HERE)
    r = subprocess.run(["bash", os.path.join(root, "tools", "verify", "asan_planner.sh"), "80"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "no sanitizer report" in r.stdout and "ERROR" not in r.stderr
