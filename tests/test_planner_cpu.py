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
