"""world_size-2 gloo tests of the N>1 path: sharding + the single all-gather of int8 logits (CPU tensors)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from edison_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(99)
    full = torch.from_numpy(rng.integers(-128, 128, (n_total, 10)).astype(np.int8))   # every rank knows the answer
    lo, hi = parallel.shard_range(n_total, rank, world)
    local = full[lo:hi].clone()
    got = parallel.all_gather_logits(local, n_total=n_total)
    ok1 = bool(torch.equal(got, full))
    got2 = parallel.all_gather_logits(local)            # sizes discovered with a tiny extra all_gather
    ok2 = bool(torch.equal(got2, full))
    ok3 = True
    if n_total % world == 0:
        g = parallel.LogitsGatherer(n_total // world)
        ok3 = bool(torch.equal(g(local), full))
        with pytest.raises(ValueError):
            g(local[:-1])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok1, ok2, ok3))


@pytest.mark.parametrize("n_total", [64, 101, 1])
def test_all_gather_logits_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def test_single_process_gather_is_identity():
    from edison_amd import parallel
    x = torch.arange(40, dtype=torch.int8).reshape(4, 10)
    assert torch.equal(parallel.all_gather_logits(x), x)
    g = parallel.LogitsGatherer(4)
    assert torch.equal(g(x), x)


def test_c_abi_shard_range_equals_the_python_one():
    """edison_dist_shard_range (what a C host uses) and parallel.shard_range must cut the batch identically, and the
    shards must tile [0, n) in rank order (the layout the all-gather assumes)."""
    sys.path.insert(0, ROOT)
    from edison_amd import parallel
    for n in (0, 1, 7, 10, 403, 262144, 2097152, 2097153):
        for w in (1, 2, 3, 4, 8):
            edge = 0
            for r in range(w):
                lo, hi = parallel.shard_range_c(n, r, w)
                assert (lo, hi) == parallel.shard_range(n, r, w)
                assert lo == edge
                edge = hi
            assert edge == n
    with pytest.raises(ValueError):
        parallel.shard_range_c(10, 2, 2)


def _id_worker(rank, world, port, q):
    """The C-ABI communicator bootstrap over gloo: rank 0's RCCL unique id reaches every rank unchanged (the
    ncclCommInitRank that follows needs GPUs and is exercised by tests/test_gpu_dist.py and bench.py --gpus N)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from edison_amd import parallel
    parallel.init_from_env(backend="gloo")
    payload = torch.zeros(128, dtype=torch.uint8)
    mine = None
    if rank == 0:
        mine = parallel.dist_unique_id()
        payload = torch.frombuffer(bytearray(mine), dtype=torch.uint8).clone()
    dist.broadcast(payload, src=0)
    got = bytes(payload.numpy().tobytes())
    gathered = [None] * world
    dist.all_gather_object(gathered, got)
    q.put((rank, len(got) == 128 and all(g == gathered[0] for g in gathered) and (mine is None or mine == got) and any(got)))
    dist.destroy_process_group()


def test_unique_id_reaches_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _run_bench(argv, env_extra=None, launcher=None, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + argv
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _json_lines(text):
    import json
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            out.append(json.loads(ln))
    return out


def test_bench_starts_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher above it must start two ranks itself (BASELINE configs[3] is launched the
    same way with 8). --dry-run rehearses that control flow on CPU: two child processes rendezvous over gloo, rank 0's
    128-byte RCCL id reaches rank 1, the C-ABI's shard ranges tile the batch, W + K gather steps run between barriers,
    and rank 0 alone prints ONE line -- with n_gpus = 2, dry_run = true and no throughput."""
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "2"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["config"]["parallelism"] == "dp2"
    c = d["checks"]
    assert c["ranks"] == 2 and c["gathered_logits_correct"] is True
    assert c["shard_ranges_rank0"] == [[0, 4096], [0, 4097]]
    if c["rccl_bindable"]:
        assert c["rccl_id_reached_every_rank"] is True


def test_bench_dry_run_with_eight_ranks_and_a_batch_the_world_does_not_divide():
    """BASELINE configs[3]'s world size, rehearsed: `python bench.py --gpus 8 --dry-run` starts eight ranks over gloo; the shard
    ranges of a batch that 8 does not divide (8 * 4096 + 1 utterances) tile it in rank order, the unequal-shard gather
    (8 * 7 + 1 rows: shards of 8, 7, 7, ...) returns every row in place, and ONE line with n_gpus = 8 comes back."""
    r = _run_bench(["--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"], timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["value"] is None and d["config"]["parallelism"] == "dp8"
    c = d["checks"]
    assert c["ranks"] == 8 and c["gathered_logits_correct"] is True
    assert c["shard_ranges_rank0"] == [[0, 4096], [0, 4097]]          # the odd utterance goes to the first rank
    assert c["unequal_shard_gather"] == dict(rows=57, ranks=8, correct=True)
    assert c["cabi_collective"]["status"].startswith("ok")


def test_bench_dry_run_under_torchrun():
    """The driver's own launch form for N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N):
    WORLD_SIZE comes from the launcher, bench.py must not start ranks of its own on top."""
    port = str(_free_port())
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", port]
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], launcher=launcher)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["checks"]["ranks"] == 2


def test_bench_refuses_a_world_it_cannot_build():
    """Fewer devices than --gpus: non-zero exit, a message, and NO JSON line (never `n_gpus: 1` for `--gpus 8`).
    Likewise a WORLD_SIZE that contradicts --gpus."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    r = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and _json_lines(r.stdout) == [] and "GPU(s) visible" in r.stderr
    r = _run_bench(["--gpus", "8", "--dry-run"], env_extra=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and _json_lines(r.stdout) == []


def test_bench_launcher_reports_a_failed_rank():
    """If a rank dies the parent stops the others and exits non-zero (here: rank 1 is told to fail before the rendezvous)."""
    r = _run_bench(["--gpus", "2", "--dry-run"], env_extra=dict(EDISON_BENCH_FAIL_RANK="1"), timeout=120)
    assert r.returncode != 0 and _json_lines(r.stdout) == []


@pytest.mark.parametrize("hang_rank", [None, "1"])
def test_bench_stuck_collective_prints_the_line_and_exits_non_zero(hang_rank):
    """The leg behind the watchdog (the collective behind the C-ABI: a second RCCL communicator, ncclAllGather on the context's
    stream -- never run at N > 1 on hardware) hangs past the watchdog: rank 0 must still print the ONE line, with the
    time-out recorded in it, and the run must end NON-zero -- a process that gave up on a stuck call does not report success
    (round-3 review: `os._exit(0)` recorded a hang as rc 0). Either every rank hangs, or one does and the others wait for it
    inside the collective; under bench.py's own launcher and with the exit code of the watchdog (3) reaching the caller."""
    env = dict(EDISON_BENCH_CABI_HANG_S="30", EDISON_BENCH_WATCHDOG_S="2")
    if hang_rank is not None:
        env["EDISON_BENCH_CABI_HANG_RANK"] = hang_rank
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], env_extra=env, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and "timed out" in lines[0]["checks"]["cabi_collective"]["status"]


def test_bench_guarded_leg_that_returns_is_not_disturbed():
    """The same leg without a hang: status ok on the line, exit code 0, and the watchdog timer is gone (the run ends at once)."""
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], env_extra=dict(EDISON_BENCH_WATCHDOG_S="60"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert _json_lines(r.stdout)[0]["checks"]["cabi_collective"]["status"].startswith("ok")


def test_bench_collectives_do_not_sit_under_rank_local_conditions():
    """At N > 1 `timed_region(..., world, ...)` holds barriers and an all-reduce, so every rank must reach every such call: none may sit
    under an `if` whose test reads something a rank finds out for itself (the queue calibration's outcome, the parity sample that only
    rank 0 keeps, the rank). A test may name such a value only when it also pins world == 1. Round 5 had `if n_queues == 2:` around one."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    rank_local = {"n_queues", "pair_kept", "calibration", "sample", "rank", "same", "same_l", "parity", "local_rank"}
    found = []

    def names(node):
        return {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}

    def pins_world_one(test):
        # `... world == 1 ...` anywhere in the test: at N > 1 the rank-local part cannot decide alone only if it is the `X if world == 1 else Y`
        # form (the rank-local X is then dead at N > 1) or an `and` with world == 1 (the whole branch is dead at N > 1)
        if isinstance(test, ast.IfExp):
            return "world" in names(test.test) and not (names(test.orelse) & rank_local)
        if isinstance(test, ast.BoolOp) and isinstance(test.op, ast.And):
            return any(isinstance(v, ast.Compare) and "world" in names(v) for v in test.values)
        return False

    def walk(node, guards):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "timed_region":
            if len(node.args) >= 4 and "world" in names(node.args[3]):
                found.append(node.lineno)
                for g in guards:
                    assert not (names(g) & rank_local) or pins_world_one(g), \
                        "bench.py:%d: timed_region(..., world) under a rank-local condition (line %d)" % (node.lineno, g.lineno)
        if isinstance(node, ast.If):
            walk(node.test, guards)
            for c in node.body:
                walk(c, guards + [node.test])
            for c in node.orelse:
                walk(c, guards + [node.test])
            return
        if isinstance(node, (ast.FunctionDef, ast.Lambda)) and node is not main:
            return        # nested step functions: called from timed_region, not the other way round
        for c in ast.iter_child_nodes(node):
            walk(c, guards)

    walk(main, [])
    assert len(found) >= 10, found
