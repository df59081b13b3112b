#!/usr/bin/env python3
"""bench.py -- throughput of edison's keyword-spotting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --dry-run          (CPU rehearsal of the N > 1 control flow over gloo: no GPU, no numbers)

With N > 1 and no WORLD_SIZE in the environment the script starts the N ranks ITSELF: child processes created before the
parent has made any GPU call, the parent only waits, relays rank 0's JSON line and exits non-zero if any rank failed or
if fewer than N devices are visible (it never prints an n_gpus = 1 line for --gpus 8).

One rank per GPU; every rank works on its own shard (weak scaling: per-GPU work is fixed as N grows).
Two workloads of BASELINE.json are timed in the same run and reported on ONE JSON line:

  * primary (`value`, `ms_per_step`, `roofline`): BASELINE configs[1] -- batched MFCC only, 65 536 x 1024-sample
    synthetic int16 frames per GPU, variant B, 13 fp32 coefficients out. A step = one pass over the batch.
    No collective (frames are independent).
  * `kws`: BASELINE configs[2]/[3] -- full KWS, 262 144 utterances per GPU (31 frames each -> MFCC B -> int8
    -> int8 CNN -> logits/softmax/argmax) followed, for N > 1, by the single RCCL all-gather of the int8 logits.

Inputs are resident in HBM before the timed region. The MFCC batch (134 MB) would fit the 256 MiB Infinity
Cache, so the bench rotates over several distinct batches to make every step read from HBM.
Kernel time is taken with HIP events on the stream the kernels are launched on (torch's current stream, which
the context is told to use); `roofline.achieved` = algorithmic bytes per launch / average launch duration.
`cpu_baseline` times the oracle's C restatement (kind "port") on a bounded sample on rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("EDISON_NET_SPECIALIZE", "0")  # model loads do not take a cached own kernel by themselves: the general-kernel leg times the general kernel
import numpy as np  # noqa: E402
import torch        # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# dense int8 MFMA peak: v_mfma_i32_32x32x32_i8 = 32 768 MAC in 32 cycles per SIMD (guide: I8 = 2 x BF16 per clock; measured
# 32.0 cycles, profiles/r02_ubench_mfma_valu_coexec.txt) x 1024 SIMDs x 2.4 GHz x 2 op/MAC
MFMA_I8_PEAK_TOPS = 1024 * 1024 * 2.4e9 * 2 / 1e12
CNN_MACS_PER_UTT = 784752         # NNoM compile log / SURVEY.md A.2
# MACs the issued MFMAs of ed_cnn_mfma_kernel can do, per utterance: per group of 4 utterances 146 v_mfma_i32_32x32x32_i8 (conv1-3)
# and 20 v_mfma_i32_16x16x64_i8 (conv4, dense) -- cnn_mfma_kernels.hip; counters: profiles/r03_cnn_counters.txt
CNN_MFMA_MACS_PER_UTT = (146 * 32768 + 20 * 16384) / 4.0
MFCC_BYTES_PER_FRAME = 2048 + 52  # 1024 int16 in + 13 fp32 out
MFCC_KERNEL = "ed_mfcc2_kernel<true, true, 2, 5>"          # the instantiation a plain 65 536-frame batch of variant B runs
MFCC_KERNEL_KWS = "ed_mfcc2_kernel<true, false, 2, 5>"     # ... and the grouped one (31 frames per utterance)
KWS_BYTES_PER_UTT = 63488 + 10 + 10 + 4  # 31*1024 int16 in + logits + softmax + argmax out


def synth_frames(n_frames, seed, device, chunk=16384):
    """clip(N(0, 3000^2)) + the two-tone of mfcc_on_mcu.py:314-315 with a random phase per frame -> int16."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_frames, 1024), dtype=torch.int16, device=device)
    t = torch.arange(1024, device=device, dtype=torch.float32) / 16000.0
    for lo in range(0, n_frames, chunk):
        n = min(chunk, n_frames - lo)
        x = torch.randn((n, 1024), generator=g, device=device) * 3000.0
        ph = torch.rand((n, 2), generator=g, device=device) * (2 * np.pi)
        x += 1000.0 * torch.cos(2 * np.pi * 1000.0 * t[None, :] + ph[:, 0:1])
        x += 500.0 * torch.cos(2 * np.pi * 125.0 * t[None, :] + ph[:, 1:2])
        out[lo:lo + n] = x.clamp_(-32768, 32767).to(torch.int16)
    return out


def synth_utterances(n_utt, seed, device, chunk=4096):
    """Mix of the three regimes of SURVEY.md 8(d): speech-level noise + two-tone (80 %), 1 %-FS noise (15 %),
    silence (5 %); 31*1024 samples per utterance (the part of the 32000 the reference uses)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = 31 * 1024
    out = torch.empty((n_utt, L), dtype=torch.int16, device=device)
    t = torch.arange(L, device=device, dtype=torch.float32) / 16000.0
    for lo in range(0, n_utt, chunk):
        n = min(chunk, n_utt - lo)
        kind = torch.rand((n, 1), generator=g, device=device)
        amp = torch.where(kind < 0.80, 3000.0, torch.where(kind < 0.95, 327.67, 0.0))
        x = torch.randn((n, L), generator=g, device=device) * amp
        ph = torch.rand((n, 2), generator=g, device=device) * (2 * np.pi)
        tone = 1000.0 * torch.cos(2 * np.pi * 1000.0 * t[None, :] + ph[:, 0:1]) + 500.0 * torch.cos(2 * np.pi * 125.0 * t[None, :] + ph[:, 1:2])
        x += tone * (kind < 0.80)
        out[lo:lo + n] = x.clamp_(-32768, 32767).to(torch.int16)
    return out


def hbm_traffic_from_profiles(key):
    """(bytes per launch, source file) from the newest profiles/r*_rocprof_summary.json, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_summary.json")))
    if not files:
        return None
    try:
        hbm = json.load(open(files[-1]))["hbm"]
        if key not in hbm:  # "name<...>:short" also matches by kernel name + size class when the template arguments changed
            name, _, cls = key.partition(":")
            cands = [k for k in hbm if k.split("<")[0] == name.split("<")[0] and k.endswith(":" + cls)]
            key = cands[0] if len(cands) == 1 else key
        h = hbm[key]
        return (round(h["FETCH_SIZE_bytes_corrected"] + h["WRITE_SIZE_bytes_corrected"]), os.path.relpath(files[-1], ROOT))
    except (KeyError, ValueError):
        return None


def cnn_counters_from_profiles():
    """Matrix-core counters of ed_cnn_mfma_kernel from the newest committed counter pass (tools/profile_cnn.sh), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cnn_counters.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        d["source"] = os.path.relpath(files[-1], ROOT)
        return d
    except ValueError:
        return None


def timed_region(step_fn, steps, warmup, world, settle_ms=0.0, settle_chunk=256, cuda=True, fork=None, join=None):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides; returns (wall ms/step, event ms/step).
    settle_ms > 0 (only for steps WITHOUT collectives: the count differs per rank): before the W warm-up steps the
    same step is repeated, untimed, for that long -- the board's power
    management needs ~100 ms of sustained load before its clocks stop moving; with microsecond
    steps a small W alone would time the transient."""
    sync = torch.cuda.synchronize if cuda else (lambda: None)    # cuda=False: the --dry-run rehearsal on CPU tensors
    # fork / join (the two-queue step: edison_queues_fork / edison_queues_join) bracket every run of steps that ends in a sync: the
    # steps of such a run go to the context's two queues, the join makes the bench's stream -- and its closing event -- wait for both
    fork = fork or (lambda: None)
    join = join or (lambda: None)
    if settle_ms > 0:
        t_end = time.perf_counter() + settle_ms * 1e-3
        i = 0
        while time.perf_counter() < t_end:
            fork()
            for _ in range(settle_chunk):
                step_fn(i)
                i += 1
            join()
            sync()  # bounds the queue; one ~20 us gap per chunk
    fork()
    for i in range(warmup):
        step_fn(i)
    join()
    e0 = e1 = None
    if cuda:
        # the two events exist BEFORE the clock starts: torch creates a HIP event at its first record, and the first such creation after a pause of the
        # process (the blocking calibration, a sleep, the start of the process) costs 25-65 us -- harness time that a 20-step region of 45 us steps
        # would carry as 3-6 % (tools/lab/region_overhead.py, profiles/r05_timed_region_overhead.txt); recorded once here, re-recorded in the region
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    if cuda:
        e0.record()        # (a host call of ~5 us on an idle GPU: recorded before the clock starts, so the event interval opens at most that much earlier than the wall interval)
    t0 = time.perf_counter()
    fork()
    for i in range(steps):
        step_fn(warmup + i)
    join()
    if cuda:
        e1.record()
    sync()
    # this rank's K steps are done: its clock stops here, the MAX over ranks below is the job's time. The closing barrier is
    # the fence in front of whatever is timed next; inside the interval it would add one more collective's latency (tens of
    # microseconds) to a region that is 1 ms long at the default K for the 50 us MFCC step -- and only for N > 1.
    wall = (time.perf_counter() - t0) * 1e3 / steps
    if world > 1:
        dist.barrier()
    sync()
    ev = e0.elapsed_time(e1) / steps if cuda else wall
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device="cuda" if cuda else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    return wall, ev


def _c_host_latency(graph):
    """examples/host_stream_latency.c (plain C on the streaming entry points, no Python): its own JSON line, or an error."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "bin", "host_stream_latency")
    if not os.path.exists(exe):
        return dict(error="examples/bin/host_stream_latency not built (python -m edison_amd.build)")
    try:
        r = subprocess.run([exe, "2000", "512", "1" if graph else "0"], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return dict(error="exit %d: %s" % (r.returncode, r.stderr.strip()[-200:]))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 -- a side figure must never cost the line
        return dict(error=repr(e))


def _c_host_pipeline():
    """examples/host_mfcc_pipeline.c: 8 independent 65 536-frame batches from a plain C host, one call per batch / one list launch / the context's two calibrated
    queues, medians of five interleaved passes, wall clock, results compared bit for bit -- its own JSON line, or an error."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "bin", "host_mfcc_pipeline")
    if not os.path.exists(exe):
        return dict(error="examples/bin/host_mfcc_pipeline not built (python -m edison_amd.build)")
    try:
        r = subprocess.run([exe, "8", "65536", "100"], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return dict(error="exit %d: %s" % (r.returncode, (r.stderr or r.stdout).strip()[-200:]))
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["what"] = ("examples/host_mfcc_pipeline.c (no Python): microseconds of wall clock per batch for the serial sequence, one edison_mfcc_batches_dev launch per 8 batches, "
                     "and one edison_mfcc_batch_queue_dev call per batch on the calibrated queues; a child process with its own context and its own calibration")
        return d
    except Exception as e:  # noqa: BLE001 -- a side figure must never cost the line
        return dict(error=repr(e))


def stream_bench(ctx, dev):
    """BASELINE configs[4]: 1 h of synthetic 16 kHz audio (57.6 M samples), 1024-sample frames at hop 512
    (50 % overlap) -> 112 499 frames, an inference on the newest 31 frames after every frame.
    (a) latency: one frame per push, host-timed push -> result on the host, through the Python mirror and from a C host,
        each with the kernels launched directly (the default) and as the captured hipGraph the config names;
    (b) throughput: the whole hour in pushes of 4096 frames, direct launches and graph replay."""
    from edison_amd.stream import Stream
    hop, total = 512, 57600000
    n_frames = (total - 1024) // hop + 1                      # 112 499
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    audio = (torch.randn((total,), generator=g, device=dev) * 3000.0).clamp_(-32768, 32767).to(torch.int16)
    host = audio[:2100 * hop].cpu().numpy()

    def latency(graph):
        # chunk = 1, host pointers (includes the 1 KB upload and 24 B download of a real microphone loop)
        st = Stream(ctx, hop=hop, chunk_frames=1, graph=graph)
        lat = []
        for i in range(2100):
            t0 = time.perf_counter()
            st.push(host[i * hop:(i + 1) * hop])
            lat.append(time.perf_counter() - t0)
        st.close()
        lat = np.array(lat[100:]) * 1e6
        return dict(p50=round(float(np.percentile(lat, 50)), 1), p90=round(float(np.percentile(lat, 90)), 1),
                    p99=round(float(np.percentile(lat, 99)), 1), pushes=int(lat.size))

    def throughput(graph, chunk=4096, min_ms=50.0):
        """The WHOLE hour, every frame of it, in both launch modes: 27 pushes of 4096 frames and the ragged last push of the remaining
        1 907 (edison_stream_push_n_dev; under the captured graph the short push runs the same kernels launched directly). One hour is ~0.7 ms of
        GPU time: the hour is streamed again and again (audio continuing seamlessly: later hours start at sample 0 and have
        112 500 frames) until the timed region is at least min_ms long."""
        st = Stream(ctx, hop=hop, chunk_frames=chunk, graph=graph)
        am = torch.empty((chunk,), dtype=torch.int32, device=dev)

        def hour(first):
            body = audio[1024 - hop:] if first else audio          # the stream starts from 1024-hop samples of silence
            n = body.numel() // hop                                  # 112 499 (first hour) / 112 500
            full, rest = divmod(n, chunk)
            for i in range(full):
                st.push_t(body[i * chunk * hop:(i + 1) * chunk * hop], argmax=am)
            if rest:
                st.push_t(body[full * chunk * hop:(full * chunk + rest) * hop], argmax=am, n_frames=rest)
            return n
        hour(True)                                                   # warm-up: one whole hour
        torch.cuda.synchronize()
        st.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = hour(True)
        hours = 1
        torch.cuda.synchronize()
        while (time.perf_counter() - t0) * 1e3 < min_ms:
            for _ in range(8):                                       # 8 hours per look at the clock
                done += hour(False)
                hours += 1
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st.close()
        return dict(frames_per_s=round(done / dt, 1), inferences_per_s=round(done / dt, 1), frames=done, hours_streamed=hours,
                    frames_first_hour=n_frames, chunk_frames=chunk,
                    ragged_last_push_frames=n_frames % chunk, seconds=round(dt, 4),
                    realtime_factor=round(done * hop / 16000.0 / dt, 1))
    lat_direct = latency(False)
    lat_direct["what"] = ("host push of 512 new samples -> softmax/argmax on the host through the Python mirror (edison_amd/stream.py): MFCC + CNN "
                          "in ONE launch (ed_kws1_kernel) against host-mapped buffers, no copy nodes, completion flag written by the kernel, the host spins on it")
    lat_graph = latency(True)
    lat_graph["what"] = "the same push with launch_mode = EDISON_STREAM_LAUNCH_GRAPH: one hipGraphLaunch of the captured upload + MFCC + CNN + shift + download nodes, then hipStreamSynchronize"
    thr_direct, thr_graph = throughput(False), throughput(True)
    del audio
    return dict(workload="1 h stream, 16 kHz, frame 1024, hop 512, window 31 frames, inference per frame",
                latency_us=lat_direct, latency_us_graph=lat_graph,
                latency_us_c_host=dict(direct=_c_host_latency(False), graph=_c_host_latency(True),
                                       what="examples/host_stream_latency.c: the same one-frame pushes from a plain C program on edison_stream_push (no Python, no ctypes)"),
                throughput=thr_direct, throughput_graph=thr_graph,
                default_launch_mode="direct (the hipGraph replay of the same nodes is what configs[4] names; it measures slower on this platform, both are reported)")


def _synth_frames_np(n_frames, seed):
    """SURVEY.md 8(d) generator on the host: clip(N(0, 3000^2)) + the two-tone of mfcc_on_mcu.py:314-315 with a random
    phase per frame -> int16 (the same distribution as synth_frames; the GPU tensors never leave the device)."""
    rng = np.random.default_rng(seed)
    t = np.arange(1024, dtype=np.float32) / 16000.0
    x = rng.normal(0, 3000, (n_frames, 1024)).astype(np.float32)
    ph = rng.random((n_frames, 2), dtype=np.float32) * (2 * np.pi)
    x += 1000.0 * np.cos(2 * np.pi * 1000.0 * t[None, :] + ph[:, 0:1])
    x += 500.0 * np.cos(2 * np.pi * 125.0 * t[None, :] + ph[:, 1:2])
    return np.clip(x, -32768, 32767).astype(np.int16).reshape(-1)


def _cgroup_cpu_quota():
    """CPU quota of this container in cores (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except (OSError, ValueError):
        return None


def _median_rate(fn, units, passes=3):
    """one warm-up pass, then >= 3 timed passes, the median."""
    fn()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return units / sorted(ts)[len(ts) // 2]


def _cnn_reference_all_cores(n_proc, per_proc=4000):
    """The reference NNoM build keeps its model in globals (weights.h:136-137, ai_nnom.c:40): not reentrant, so "all cores"
    is n_proc PROCESSES, each with its own copy of oracle/_ref/libnnom_ref.so, started together on a wall-clock mark
    (oracle/ref_worker.py); rate = all inferences / the slowest worker's time."""
    import subprocess
    start = time.time() + 3.0
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_worker.py"), str(per_proc), "%.6f" % start]
    ps = [subprocess.Popen(cmd + [str(i)], stdout=subprocess.PIPE, text=True) for i in range(n_proc)]
    outs = [json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in ps]
    if any(p.returncode != 0 for p in ps):
        raise RuntimeError("a reference worker failed")
    t_end = max(o["t_end"] for o in outs)
    t_begin = min(o["t_begin"] for o in outs)
    return dict(value=n_proc * per_proc / (t_end - t_begin), unit="inferences/s", cores=n_proc, kind="reference",
                sample="%d processes x %d random inputs each (the reference model is a global singleton: one process per core), "
                       "first start to last finish" % (n_proc, per_proc))


def cpu_baseline():
    """The oracle's C restatement (checker code, timed here only as the reported CPU baseline), as BASELINE.md
    section 3 prescribes: built -O3 -march=native on this host, (i) one thread, (ii) all host cores (static partition
    over frames / utterances, OpenMP), one warm-up pass and the median of 3 timed passes, core count printed."""
    from oracle import oracle
    oracle.use_native_build()
    nproc = os.cpu_count()
    n_aff = len(os.sched_getaffinity(0))
    quota = _cgroup_cpu_quota()
    # "all host cores" = the cores this process may actually run on at once: its affinity mask, cut by the container's CPU
    # quota when there is one (a GPU box hands a 1-GPU job a 16-core share of a 256-thread host: 256 OpenMP threads on
    # it measure the scheduler, not the code)
    n_all = max(1, min(n_aff, int(quota + 0.999))) if quota else n_aff
    print("cpu_baseline: nproc=%d, affinity=%d cores, cgroup CPU quota=%s; legs: 1 thread and %d threads" % (
        nproc, n_aff, "%.1f cores" % quota if quota else "none", n_all), file=sys.stderr, flush=True)
    host = dict(nproc=nproc, affinity_cores=n_aff, cgroup_quota_cores=quota, threads_all_cores_leg=n_all,
                build="gcc -O3 -march=native -fopenmp (oracle/Makefile native)", passes=3)
    res = {}
    x = _synth_frames_np(65536, 20)                                   # seed 20: SURVEY.md 8(d) config 2
    n1 = 4096                                                         # 1-thread sample: ~0.1-0.2 s per pass

    def leg(fn_1, units_1, fn_all, units_all, unit, what, arith="float64: the reference's CPU path (numpy / scipy, mfcc_utils.py) computes in float64 and the port restates THAT; an fp32 "
                                                                "port would time a program the reference does not run (SURVEY 8(d) suggested one: not built, for this reason)"):
        r1 = _median_rate(fn_1, units_1)
        ra = _median_rate(fn_all, units_all)
        return dict(value=ra, unit=unit, cores=n_all, kind="port", arithmetic=arith, single_thread=dict(value=r1, cores=1, sample_units=units_1),
                    sample="%s; all-cores leg %d units, 1-thread leg %d units; median of 3 passes after 1 warm-up" % (what, units_all, units_1),
                    host=host)
    # MFCC variant B (the features the net was trained on), float64 like the reference's numpy path it restates
    res["mfcc"] = leg(lambda: oracle.mfcc(x[:n1 * 1024], oracle.VARIANT_B, n_threads=1), n1,
                      lambda: oracle.mfcc(x, oracle.VARIANT_B, n_threads=n_all), 65536, "frames/s",
                      "oracle/mfcc_ref.c variant B float64, SURVEY 8(d) generator (noise + two-tone), seed 20")
    res["mfcc_variant_a"] = leg(lambda: oracle.mfcc(x[:n1 * 1024], oracle.VARIANT_A, n_threads=1), n1,
                                lambda: oracle.mfcc(x, oracle.VARIANT_A, n_threads=n_all), 65536, "frames/s",
                                "oracle/mfcc_ref.c variant A float64, same frames")
    # MFCC variant C = the firmware's own C path (Q15 CMSIS-DSP arithmetic restated)
    res["q15"] = leg(lambda: oracle.mfcc_q15(x[:n1 * 1024], n_threads=1), n1,
                     lambda: oracle.mfcc_q15(x, n_threads=n_all), 65536, "frames/s",
                     "oracle/mfcc_q15_ref.c (firmware audioCalcMFCCs arithmetic), same frames", arith="int16 / int32 (Q15 / Q31), the firmware's own")
    # full KWS: MFCC B + int8 CNN restatement; 4096 utterances (126 976 frames) for the all-cores leg keeps the whole
    # baseline inside ~30 s of CPU work on a 16-core host
    nu_all, nu_1 = 4096, 128
    a = _synth_frames_np(nu_all * 31, 21)                             # utterances of 31 frames, seed 21 (config 3)
    model = oracle.Model()

    def kws(n, th):
        m = oracle.mfcc(a[:n * 31744], oracle.VARIANT_B, n_threads=th)[:, :13]
        return oracle.cnn(model, oracle.net_input(m).reshape(n, 403), n_threads=th)
    res["kws"] = leg(lambda: kws(nu_1, 1), nu_1, lambda: kws(nu_all, n_all), nu_all, "inferences/s",
                     "oracle MFCC B + int8 CNN restatement (NNoM arithmetic), utterances of 31 frames, seed 21", arith="float64 MFCC (as the reference's host path) + int8 / int32 CNN (NNoM's)")
    if oracle.have_ref():
        rng = np.random.default_rng(5)
        f = rng.integers(-128, 128, (2000, 403)).astype(np.int8)
        r = _median_rate(lambda: oracle.nnom_ref_batch(f), 2000)
        res["cnn_reference"] = dict(value=r, unit="inferences/s", cores=1, kind="reference",
                                    sample="2000 random inputs, reference NNoM 0.3.0 + CMSIS-NN + weights.h (oracle/_ref, gcc -O2), CNN only, 1 thread, median of 3")
        try:
            res["cnn_reference"]["all_cores"] = _cnn_reference_all_cores(n_all)
        except Exception as e:  # noqa: BLE001
            res["cnn_reference"]["all_cores"] = dict(error=repr(e))
    return res


MFCC_TOL = {"A": (1e-3, 1e-4), "B": (1e-2, 1e-5)}   # |d| <= abs + rel * |ref| against the float64 reference arithmetic


def parity_check(sample):
    """Part of the cpu_baseline leg (the oracle is loaded there, never inside a timed region): what the LAST TIMED step of each
    workload wrote, against the oracle on the very samples that step read -- the first 4 096 frames of the MFCC batch (variants B,
    A, C) and the first 128 utterances of the KWS batch. Reference arithmetic: mfcc_utils.py:287-322 (B), :160-197 (A),
    audioprocessing.c:116-215 (C), kws_nnom.py:354-361 (net input), NNoM + CMSIS-NN (CNN; nnom_utils.c:275-284 for the argmax).
    SURVEY.md 7: feature flips (fp32 landing on the other side of an x.5 of the float64 path) and argmax flips are counted
    separately; CNN exactness is asserted on the features the GPU produced. ok = False fails the run (exit code 5, after the line)."""
    from oracle import oracle
    res = dict(ok=True)

    def mfcc_leg(tag, variant, frames, got):
        ref = oracle.mfcc(frames.reshape(-1), variant, n_threads=8)[:, :got.shape[1]]
        d = np.abs(got.astype(np.float64) - ref)
        a_, r_ = MFCC_TOL[tag]
        bound = a_ + r_ * np.abs(ref)
        worst = int(np.argmax(d / bound))
        inside = bool((d <= bound).all())
        res["mfcc_%s" % tag.lower()] = dict(frames_checked=int(frames.shape[0]), max_abs_err=float(d.max()), at_ref=float(ref.reshape(-1)[worst]),
                                            tolerance="|d| <= %g + %g |ref|" % (a_, r_), worst_err_over_tolerance=float((d / bound).max()),
                                            within_tolerance=inside)
        res["ok"] = res["ok"] and inside
    if "mfcc_b" in sample:
        mfcc_leg("B", oracle.VARIANT_B, sample["frames"], sample["mfcc_b"])
        res["mfcc_max_abs_err"] = res["mfcc_b"]["max_abs_err"]
        res["mfcc_tolerance"] = res["mfcc_b"]["tolerance"]
    if "mfcc_a" in sample:
        mfcc_leg("A", oracle.VARIANT_A, sample["frames_a"], sample["mfcc_a"])
    if "mfcc_q15" in sample:
        ref = oracle.mfcc_q15(sample["frames_q15"].reshape(-1), n_threads=8)[:, :sample["mfcc_q15"].shape[1]]
        same = bool(np.array_equal(ref, sample["mfcc_q15"]))
        res["q15"] = dict(frames_checked=int(ref.shape[0]), bit_exact=same, values_differing=int((ref != sample["mfcc_q15"]).sum()))
        res["ok"] = res["ok"] and same
    if "utt_audio" in sample:
        au = sample["utt_audio"]
        n = au.shape[0]
        m = oracle.mfcc(au.reshape(-1), oracle.VARIANT_B, n_threads=8)[:, :13]
        feat_ref = oracle.net_input(m).reshape(n, 403)
        feat_gpu = sample["utt_feat"].reshape(n, 403)
        step = np.abs(feat_gpu.astype(np.int32) - feat_ref.astype(np.int32))
        model = oracle.Model()
        on_gpu_feat = oracle.cnn(model, feat_gpu)
        exact = bool(np.array_equal(on_gpu_feat["logits"], sample["utt_logits"]) and np.array_equal(on_gpu_feat["softmax"], sample["utt_softmax"])
                     and np.array_equal(on_gpu_feat["argmax"], sample["utt_argmax"]))
        on_ref_feat = oracle.cnn(model, feat_ref)
        kws = dict(utterances_checked=int(n), feature_values=int(feat_ref.size), feature_flips=int((step > 0).sum()), feature_flip_max_step=int(step.max()),
                   cnn_bit_exact_on_gpu_features=exact, argmax_flips=int((on_ref_feat["argmax"] != sample["utt_argmax"]).sum()),
                   what="features: float64 reference arithmetic vs the GPU's fp32, rounded to int8 (a flip = fp32 on the other side of an x.5); CNN: the "
                        "oracle's int8 restatement on the GPU's own features, logits + softmax + argmax bit for bit; argmax flips: end to end, CPU features -> CPU CNN vs the GPU")
        if oracle.have_ref():
            r = oracle.nnom_ref_batch(feat_gpu)
            kws["cnn_bit_exact_vs_reference_nnom"] = bool(np.array_equal(r["logits"], sample["utt_logits"]) and np.array_equal(r["softmax"], sample["utt_softmax"])
                                                          and np.array_equal(r["argmax"], sample["utt_argmax"]))
            exact = exact and kws["cnn_bit_exact_vs_reference_nnom"]
        res["kws"] = kws
        # a feature may flip by one step on a rounding boundary (measured 0-2 per 26 000); more than one step, or more than 1 in 1000, is an error
        res["ok"] = res["ok"] and exact and kws["feature_flip_max_step"] <= 1 and kws["feature_flips"] <= max(2, feat_ref.size // 1000)
    res["what"] = "the outputs of the last TIMED step of each workload against the oracle on the samples that step read (bench.py parity_check)"
    return res


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks (fresh child processes of this script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), wait, relay rank 0's stdout (the JSON line),
    send the other ranks' stdout to stderr. The parent makes no GPU call (torch.cuda.device_count() only enumerates).
    Returns the exit code: 0 only if every rank returned 0; if one fails the rest are stopped by PID."""
    import subprocess
    n = args.gpus
    if not args.dry_run:
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but %d GPU(s) visible on this node; not starting (no n_gpus = %d line will be printed for a "
                  "smaller world)" % (n, have, n), file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    out0 = []
    t = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    t.start()
    rc = 0
    alive = set(range(n))
    while alive and rc == 0:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is not None:
                alive.discard(r)
                if c != 0:
                    print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, c), file=sys.stderr, flush=True)
                    rc = c if c > 0 else 1
        time.sleep(0.05)
    for r in alive:             # only after a failure: exact PIDs, never a pattern
        procs[r].terminate()
    for r in alive:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    t.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    return rc


WATCHDOG_EXIT_CODE = 3


def guarded_leg(rank, line, where, leg, timeout_s):
    """Run `leg()` (something that may hang inside a GPU / collective call that no Python exception will ever leave: a second
    RCCL communicator's init, an all-gather on the context's stream) under a watchdog. If it does not return within
    timeout_s, rank 0 prints the JSON line as it stands with `where`["cabi_collective"].status = "timed out ..." and EVERY rank
    leaves with WATCHDOG_EXIT_CODE (non-zero: a process that gave up on a stuck call must not report success; launch_ranks
    and the driver then see the failure). No re-exec, no restart: the process has touched the GPU. Returns leg()'s result."""
    import threading

    def bail():
        if rank == 0:
            where["cabi_collective"] = dict(status="timed out after %g s; the figures above use torch.distributed's collective" % timeout_s)
            print(json.dumps(line), flush=True)
        else:
            time.sleep(1.0)   # rank 0's line first: a parent that sees this rank fail stops the others by PID
        os._exit(WATCHDOG_EXIT_CODE)
    dog = threading.Timer(timeout_s, bail)
    dog.daemon = True
    dog.start()
    try:
        return leg()
    finally:
        dog.cancel()


def dry_run(args):
    """The N > 1 control flow WITHOUT GPUs (gloo, CPU tensors): what can be rehearsed in a container that has no device.
    Every rank: rendezvous -> can RCCL be bound (edison_dist_available)? -> rank 0's 128-byte RCCL id reaches every rank
    -> the C-ABI's shard ranges tile the batch in rank order (divisible and not) -> W + K steps of "make this rank's
    logits, all-gather them" between barriers -> rank 0 checks the gathered rows and prints a line with n_gpus = world,
    dry_run = true and value = null. Nothing here is a measurement."""
    from edison_amd import parallel
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))
    if os.environ.get("EDISON_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit("rank %s: told to fail (EDISON_BENCH_FAIL_RANK; tests/test_distributed_cpu.py)" % os.environ.get("RANK"))
    rank, world, _ = parallel.init_from_env(backend="gloo")
    ok = torch.tensor([1 if parallel.dist_available() else 0], dtype=torch.int32)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    rccl = bool(int(ok.item()))
    id_ok = None
    if rccl and world > 1:
        payload = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            payload = torch.frombuffer(bytearray(parallel.dist_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(payload, src=0)
        got = [None] * world
        dist.all_gather_object(got, bytes(payload.numpy().tobytes()))
        id_ok = len(got[0]) == 128 and any(got[0]) and all(g == got[0] for g in got)
    nu = min(args.utts, 4096)
    shards = []
    for n_total in (world * nu, world * nu + 1):
        edge = 0
        for r in range(world):
            lo, hi = parallel.shard_range_c(n_total, r, world)
            assert lo == edge and (lo, hi) == parallel.shard_range(n_total, r, world)
            edge = hi
        assert edge == n_total
        shards.append(list(parallel.shard_range_c(n_total, rank, world)))

    def logits_of(r, i):
        return torch.from_numpy(np.random.default_rng(1000 * r + i).integers(-128, 128, (nu, 10)).astype(np.int8))
    gather = parallel.LogitsGatherer(nu, 10, device="cpu")
    last = {}

    def step(i):
        last["i"] = i
        last["all"] = gather(logits_of(rank, i))
    wall, _ = timed_region(step, min(args.steps, 20), min(args.warmup, 5), world, cuda=False)
    want = torch.cat([logits_of(r, last["i"]) for r in range(world)])
    good = torch.tensor([1 if torch.equal(last["all"], want) else 0], dtype=torch.int32)
    # ... and the unequal-shard gather (what a batch the world does not divide takes)
    n_odd = world * 7 + 1
    lo, hi = parallel.shard_range(n_odd, rank, world)
    full = torch.from_numpy(np.random.default_rng(7).integers(-128, 128, (n_odd, 10)).astype(np.int8))
    odd_ok = torch.tensor([1 if torch.equal(parallel.all_gather_logits(full[lo:hi].clone(), n_total=n_odd), full) else 0], dtype=torch.int32)
    if world > 1:
        dist.all_reduce(odd_ok, op=dist.ReduceOp.MIN)
    if not int(odd_ok.item()):
        good.zero_()
    if world > 1:
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        dist.barrier()
    line = dict(metric="MFCC frames/sec + KWS inferences/sec (whole node) at 1/2/4/8 MI355X", value=None, unit="frames/s",
                n_gpus=world, steps=min(args.steps, 20), warmup=min(args.warmup, 5), ms_per_step=None, higher_is_better=True,
                scaling="weak", vs_baseline=None, dtype="f32", data="synthetic", dry_run=True,
                config=dict(workload="DRY RUN on CPU over gloo: control flow of the N > 1 bench only, no kernels, no numbers",
                            parallelism="dp%d" % world, collective="all_gather int8 logits (gloo stand-in for RCCL)"),
                checks=dict(rccl_bindable=rccl, rccl_id_reached_every_rank=id_ok, shard_ranges_rank0=shards,
                            gathered_logits_correct=bool(int(good.item())), ranks=world,
                            unequal_shard_gather=dict(rows=n_odd, ranks=world, correct=bool(int(odd_ok.item())))))

    # the leg that sits behind the watchdog in the real run (the collective behind the C-ABI), rehearsed with the gloo stand-in.
    # EDISON_BENCH_CABI_HANG_S makes it sleep (on EDISON_BENCH_CABI_HANG_RANK, default every rank) as a stuck ncclCommInitRank /
    # ncclAllGather would: with EDISON_BENCH_WATCHDOG_S shorter than that the run must print its line AND leave non-zero.
    def cabi_rehearsal():
        hang = float(os.environ.get("EDISON_BENCH_CABI_HANG_S", "0"))
        if hang > 0 and os.environ.get("EDISON_BENCH_CABI_HANG_RANK", str(rank)) == str(rank):
            time.sleep(hang)
        got = gather(logits_of(rank, 0))
        return dict(status="ok (dry run: gloo stand-in)" if torch.equal(got, torch.cat([logits_of(r, 0) for r in range(world)])) else "MISMATCH")
    cabi = guarded_leg(rank, line, line["checks"], cabi_rehearsal, float(os.environ.get("EDISON_BENCH_WATCHDOG_S", "120")))
    line["checks"]["cabi_collective"] = cabi
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if int(good.item()) == 1 and id_ok is not False and cabi["status"].startswith("ok") else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults sized for the steady state: a 65 536-frame step is ~60 us, and the board's power management needs
    # ~100 ms of sustained load to settle (first 1 ms: 61 us/step, next 10 ms: 63 us, after 100 ms: 56 us)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--frames", type=int, default=65536, help="MFCC frames per GPU per step (BASELINE configs[1])")
    ap.add_argument("--utts", type=int, default=262144, help="KWS utterances per GPU per step (BASELINE configs[2])")
    ap.add_argument("--settle-ms", type=float, default=200.0,
                    help="untimed repetition of each workload's step before its W warm-up steps, until the clocks have settled")
    ap.add_argument("--rotate", type=int, default=3, help="distinct MFCC input batches cycled through (defeats the 256 MiB L3)")
    ap.add_argument("--queues", type=int, default=0, choices=[0, 1, 2],
                    help="HIP queues of the headline MFCC step. 0 (default): edison_queues_calibrate decides -- two queues (the next batch's launch in flight while "
                         "this one drains) when a pair of the context's streams beats the serial sequence by 1 %, else one; 1: the serial sequence; 2: two queues whatever "
                         "the calibration found")
    ap.add_argument("--skip-kws", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-stream", action="store_true")
    ap.add_argument("--skip-q15", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N > 1 control flow over gloo (rendezvous, RCCL id broadcast, shard ranges, barriers, "
                         "the gather of the logits): prints a line with n_gpus = N, dry_run = true and NO throughput")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher above us: start the ranks ourselves, BEFORE this process makes any GPU call
        raise SystemExit(launch_ranks(args))
    if args.dry_run:
        raise SystemExit(dry_run(args))

    from edison_amd import parallel, _lib
    from edison_amd.context import Context
    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (or leave WORLD_SIZE unset and let bench.py start them)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = Context(local_rank)
    # one explicit stream for everything: torch fills, the HIP kernels behind the C-ABI, the HIP events, RCCL
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.use_torch_stream(stream)
    info = ctx.device_info()

    # ------------------------------------------------------------------ primary: MFCC only (configs[1])
    nf = args.frames
    bufs = [synth_frames(nf, 20 + 1000 * r + rank, dev) for r in range(max(1, args.rotate))]
    out = torch.empty((nf, 13), dtype=torch.float32, device=dev)

    out_b = torch.empty((nf, 13), dtype=torch.float32, device=dev)

    def mfcc_step(i):
        ctx.mfcc_t(bufs[i % len(bufs)], nf, 1024, _lib.MFCC_B, 13, out=out)
    # ---- one 65 536-frame batch per launch either way; with two queues the NEXT batch's launch is in flight while this one drains
    # (edison_queues_fork / edison_mfcc_batch_queue_dev / edison_queues_join: independent batches, own output each, alternating
    # queues). Whether two HIP streams overlap profitably depends on where runtime and driver put their hardware queues -- a third of
    # all stream pairs gains, a third loses 8-10 % (profiles/r05_mfcc_two_queues_notes.txt) -- so the library measures it once
    # (edison_queues_calibrate, ~0.15 s on this batch) and the step uses two queues only if a pair beat the serial sequence.
    outs2 = [out, out_b]
    q_calls = [ctx.mfcc_queue_call(i & 1, bufs[i % len(bufs)], nf, 1024, _lib.MFCC_B, 13, out=outs2[i & 1]) for i in range(2 * len(bufs))]

    def q_step(i):
        q_calls[i % len(q_calls)]()
    # Order of the first measurements, and why. (1) `serial_cold`: W + K serial steps on the board AS IT COMES -- what the headline of rounds
    # 1-4 was (a board coming from idle runs its first ~25 launches slower than settled: profiles/r03_cold_start_notes.txt). (2) the
    # calibration: 0.15-0.2 s of GPU work, which also wakes the board. (3) THE HEADLINE: W + K steps of the step the calibration chose.
    # (4) `serial`: the same W + K on one queue right after it -- the board in the same state, so headline vs `serial` is the queues'
    # doing and `serial` vs `serial_cold` is the board's. (5) `settled`, both ways.
    calibration = serial_cold = None
    if args.queues != 1:
        cw_ms, cev_ms = timed_region(mfcc_step, args.steps, args.warmup, world, 0.0)
        serial_cold = dict(value=round(world * nf / (cw_ms * 1e-3), 1), unit="frames/s", ms_per_step=round(cw_ms, 4), kernel_ms=round(cev_ms, 4),
                           roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (cev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           what="W + K steps, one call per batch on one queue, as the FIRST GPU work of the process: the measurement that was the headline of "
                                "rounds 1-4 (BENCH_r01-r04), before the queue calibration has run")
        t_c = time.perf_counter()
        try:
            calibration = ctx.queues_calibrate(bufs[0], nf)
            calibration = dict(serial_us=round(calibration["serial_us"], 2), best_us=round(calibration["best_us"], 2), pair=calibration["pair"],
                               seconds=round(time.perf_counter() - t_c, 3),
                               what="edison_queues_calibrate: the serial sequence and every pair of the context's 5 candidate streams (3 least, 2 greatest priority), "
                                    "interleaved blocks of 32-64 launches of this batch; pair = the candidates kept, null = no pair was 1 % faster: one queue")
        except Exception as e:  # noqa: BLE001 -- a calibration that cannot run costs the second queue, not the line
            calibration = dict(pair=None, error=repr(e))
        torch.cuda.synchronize()
    # a pair was kept (or --queues 2 insists): the two-queue step exists on this rank
    pair_kept = (args.queues == 2 and "error" not in (calibration or {})) or (args.queues == 0 and calibration["pair"] is not None)
    # ... and it pays only over a region long enough: forking the two queues from the stream and joining them back costs ~25-30 us of cross-queue
    # signalling per timed region, the queues gain 1-2 us per step -- K = 10: -4.8 %, 20: -0.4 %, 40: +1.3 %, 80 and more: +2 ... +4 % against the
    # serial sequence, wall clock, interleaved (tools/lab/k_sweep.py, profiles/r05_two_queues_region_length.txt). The headline uses the two queues
    # from K = 40 on (--queues 2: always); below that it is the serial step, and `two_queues_same_wk` beside it says what two queues would have read.
    MIN_STEPS_TWO_QUEUES = 40
    n_queues = 2 if pair_kept and (args.queues == 2 or args.steps >= MIN_STEPS_TWO_QUEUES) else 1
    q_kw = dict(fork=ctx.queues_fork, join=ctx.queues_join)
    head_step, head_kw = (q_step, q_kw) if n_queues == 2 else (mfcc_step, {})
    # N > 1: every rank calibrates its own streams, so `config.queues` is rank 0's finding; how many ranks run two queues rides beside it
    ranks_two_queues = None
    if world > 1:
        tq = torch.tensor([1 if n_queues == 2 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(tq)
        ranks_two_queues = int(tq.item())
    # THE HEADLINE is what the command asked for: W warm-up + K timed steps of the step, and nothing else inside the region. Since round 5
    # it is NOT the first GPU work of the process any more (with --queues 1 it still is): `serial_cold` above is, and the calibration
    # between them has kept the board busy for ~0.2 s -- with the driver's small counts (W 5, K 20 = 1.5 ms) the headline is then a burst on
    # an awake board (boost clocks: faster than settled), where `serial_cold` is a burst on an idle one (slower than settled, DESIGN.md
    # section 5). `serial` right after it shows what of the difference is the board's; `settled` is the figure that lasts.
    wall_ms, ev_ms = timed_region(head_step, args.steps, args.warmup, world, 0.0, **head_kw)
    # what the LAST timed step left in its output, and the samples it read: checked against the oracle in the cpu_baseline leg (`parity`)
    sample = {}
    n_par = min(4096, nf)
    if rank == 0 and world == 1 and not args.skip_cpu:
        last = args.warmup + args.steps - 1
        sample["frames"] = bufs[last % len(bufs)][:n_par].cpu().numpy()
        sample["mfcc_b"] = (outs2[last & 1] if n_queues == 2 else out)[:n_par].cpu().numpy()
    frames_per_s = world * nf / (wall_ms * 1e-3)
    ach = MFCC_BYTES_PER_FRAME * nf / (ev_ms * 1e-3) / 1e9
    roofline = dict(bound="hbm", kernel=MFCC_KERNEL, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, bytes_per_unit=MFCC_BYTES_PER_FRAME,
                    units_per_launch=nf, kernel_ms=round(ev_ms, 4))
    if n_queues == 2:
        roofline["kernel_ms_what"] = ("HIP-event time of the K steps / K = the pitch at which launches complete; with two launches in flight a launch's own "
                                      "duration (what rocprofv3 --kernel-trace reports per dispatch) is about twice that -- it includes the wait for the CUs the "
                                      "other launch still holds. `serial.kernel_ms` is the one-queue figure that rocprofv3's average agrees with.")
    serial = None
    if args.queues != 1:
        sw_ms, sev1_ms = timed_region(mfcc_step, args.steps, args.warmup, world, 0.0)
        serial = dict(value=round(world * nf / (sw_ms * 1e-3), 1), unit="frames/s", ms_per_step=round(sw_ms, 4), kernel_ms=round(sev1_ms, 4),
                      roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (sev1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      what="the same W + K steps as one call per batch on ONE queue (edison_mfcc_batch_dev), measured right after the headline (same board state)"
                           + ("" if n_queues == 2 else "; the headline was this same serial step: the two differ by what a W + K burst scatters from run to run"))
    # K too small for the queues to pay although a pair was kept: the same W + K on the two queues, right after `serial` -- what the headline would have read
    # (N > 1: every rank runs this leg or none does -- timed_region holds barriers, and whether a pair was kept is each rank's own finding; a rank without
    # a pair runs the queue calls on one stream, which is the serial sequence)
    two_queues_same_wk = None
    if (pair_kept and n_queues == 1) if world == 1 else (args.queues == 0 and args.steps < MIN_STEPS_TWO_QUEUES):
        tw_ms, tev_ms = timed_region(q_step, args.steps, args.warmup, world, 0.0, **q_kw)
        two_queues_same_wk = dict(value=round(world * nf / (tw_ms * 1e-3), 1), unit="frames/s", ms_per_step=round(tw_ms, 4), kernel_ms=round(tev_ms, 4),
                                  roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (tev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  what="the same W + K steps over the context's two queues (fork ... join inside the timed region): not the headline because K < %d -- "
                                       "a fork / join pair costs ~25-30 us per region, the queues gain 1-2 us per step (profiles/r05_two_queues_region_length.txt)" % MIN_STEPS_TWO_QUEUES)
    settled = None
    if args.settle_ms > 0:
        # at least 400 timed steps here whatever K is: a 20-step block of this 45 us step scatters by +-3 % from block to block (profiles/r03_bench_repeatability.txt),
        # and `settled` is the figure meant to be compared from run to run. Over 400 steps the two queues pay whenever a pair was kept: `settled` uses them then,
        # whatever the headline's K made it use.
        n_settled = max(args.steps, 400)
        set_step, set_kw = (q_step, q_kw) if pair_kept else (mfcc_step, {})
        s_ms, sev_ms = timed_region(set_step, n_settled, args.warmup, world, args.settle_ms, **set_kw)
        settled = dict(value=round(world * nf / (s_ms * 1e-3), 1), unit="frames/s", ms_per_step=round(s_ms, 4), kernel_ms=round(sev_ms, 4), steps=n_settled,
                       roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (sev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), settle_ms=args.settle_ms,
                       queues=2 if pair_kept else 1,
                       what="the same workload after settle_ms of untimed repetition of the step in front of the W warm-up steps (clocks and "
                            "power settled), on two queues when the calibration kept a pair; a side figure, the headline is the W + K run above")
        # (N > 1: whether a pair was kept is each rank's own finding, and timed_region holds barriers -- every rank runs this leg or none does)
        if pair_kept if world == 1 else (args.queues != 1):
            s1_ms, s1ev_ms = timed_region(mfcc_step, n_settled, args.warmup, world, args.settle_ms)
            settled["serial"] = dict(value=round(world * nf / (s1_ms * 1e-3), 1), ms_per_step=round(s1_ms, 4), kernel_ms=round(s1ev_ms, 4),
                                     roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (s1ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    # ---- the same batches, several per launch (edison_mfcc_rows_dev: batches that sit at a constant stride are the rows of ONE launch).
    # The 65 536-frame launch above pays per launch what a longer one amortises -- 2.4 us until the median wave computes, then
    # waves leaving over 4-5 us at the end: 15 % of the launch window idle, of which the power manager gives ~6 % back as clock
    # (profiles/r04_mfcc_launch_structure_notes.txt) -- so a caller that streams config-2-sized batches gets the loop's own rate
    # by handing over a few at a time. A side figure: the headline stays one batch per launch.
    rows_launch = None
    try:
        n_b = 8
        big = torch.empty((n_b, nf, 1024), dtype=torch.int16, device=dev)
        for b in range(n_b):
            big[b] = bufs[b % len(bufs)]
        out_b = torch.empty((n_b * nf, 13), dtype=torch.float32, device=dev)

        def rows_step(i):
            ctx.mfcc_rows_t(big, n_b, nf * 1024, nf, 1024, _lib.MFCC_B, 13, out=out_b)
        r_ms, rev_ms = timed_region(rows_step, max(20, args.steps // n_b), max(3, min(args.warmup, 2000) // n_b), world, args.settle_ms)
        rows_launch = dict(batches_per_launch=n_b, value=round(world * n_b * nf / (r_ms * 1e-3), 1), unit="frames/s", ms_per_launch=round(r_ms, 4),
                           us_per_65536_frames=round(rev_ms * 1e3 / n_b * 65536 / nf, 2),
                           roofline_frac=round(MFCC_BYTES_PER_FRAME * n_b * nf / (rev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           what="%d batches of %d frames, %d samples apart, as the rows of one edison_mfcc_rows_dev launch (settled); the kernel is the grouped instantiation %s" % (n_b, nf, nf * 1024, MFCC_KERNEL_KWS))
        del big, out_b
    except Exception as e:  # a side figure must never cost the headline numbers
        rows_launch = dict(error=repr(e))

    # ---- ... and as a LIST of independent batches (edison_mfcc_batches_dev, round 5): separate allocations, own output each, no common
    # stride -- what a caller has whose batches come from different producers. Same loop, one launch per list.
    batch_list = None
    try:
        n_b = 8
        pads, ins, outs_l = [], [], []
        for b in range(n_b):
            pads.append(torch.empty((4099 * (b + 1),), dtype=torch.int8, device=dev))   # keeps the allocations from lining up
            ins.append(bufs[b % len(bufs)].clone())
            outs_l.append(torch.empty((nf, 13), dtype=torch.float32, device=dev))

        def list_step(i):
            ctx.mfcc_batches_t(ins, nf, 1024, _lib.MFCC_B, 13, outs=outs_l)
        l_ms, lev_ms = timed_region(list_step, max(20, args.steps // n_b), max(3, min(args.warmup, 2000) // n_b), world, args.settle_ms)
        ctx.mfcc_t(ins[n_b - 1], nf, 1024, _lib.MFCC_B, 13, out=out)
        same_l = bool(torch.equal(out, outs_l[n_b - 1]))
        batch_list = dict(batches_per_launch=n_b, value=round(world * n_b * nf / (l_ms * 1e-3), 1), unit="frames/s", ms_per_launch=round(l_ms, 4),
                          us_per_65536_frames=round(lev_ms * 1e3 / n_b * 65536 / nf, 2),
                          roofline_frac=round(MFCC_BYTES_PER_FRAME * n_b * nf / (lev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          outputs_bit_identical_to_one_call_per_batch=same_l,
                          what="%d independent batches of %d frames (separate allocations, own outputs) as ONE edison_mfcc_batches_dev launch (settled); kernel ed_mfcc2_list_kernel<true, 2, 5>" % (n_b, nf))
        del pads, ins, outs_l
    except Exception as e:  # a side figure must never cost the headline numbers
        batch_list = dict(error=repr(e))

    # ---- two queues against one, interleaved (8 blocks of 300 batches each, medians): the evidence behind `config.queues`, whatever the
    # calibration chose. The library's own two queues (the calibrated pair, or one stream of each priority without a calibration).
    two_queues = None
    try:
        if args.queues == 1:
            raise RuntimeError("--queues 1: every launch of this run is on one queue (the profiled passes whose per-launch durations are quoted)")
        ctx.queues_fork()
        for i in range(4):
            q_step(i)
        ctx.queues_join()
        ref_o = torch.empty((nf, 13), dtype=torch.float32, device=dev)
        ctx.mfcc_t(bufs[2 % len(bufs)], nf, 1024, _lib.MFCC_B, 13, out=ref_o)
        same = bool(torch.equal(outs2[0], ref_o))
        ctx.mfcc_t(bufs[3 % len(bufs)], nf, 1024, _lib.MFCC_B, 13, out=ref_o)
        same = same and bool(torch.equal(outs2[1], ref_o))
        torch.cuda.synchronize()
        n2 = 300     # batches per timed block; a 40-batch block of this step scatters by +-5 %

        def block(step, two):
            for i in range(30):
                if not two:
                    step(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if two:
                ctx.queues_fork()
            for i in range(n2):
                step(i)
            if two:
                ctx.queues_join()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n2
        t_end = time.perf_counter() + 0.2           # settle the clocks on the serial step
        while time.perf_counter() < t_end:
            for i in range(256):
                mfcc_step(i)
            torch.cuda.synchronize()
        ts2, ts1 = [], []
        for r in range(8):
            for leg in ((1, 2) if r % 2 == 0 else (2, 1)):
                (ts2 if leg == 2 else ts1).append(block(q_step, True) if leg == 2 else block(mfcc_step, False))
        t2, t1 = sorted(ts2)[len(ts2) // 2], sorted(ts1)[len(ts1) // 2]
        two_queues = dict(queues=2, ms_per_batch=round(t2, 4), serial_ms_per_batch=round(t1, 4), vs_serial=round(t1 / t2, 4),
                          value=round(world * nf / (t2 * 1e-3), 1), unit="frames/s", outputs_bit_identical_to_serial=same,
                          roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          what="one %d-frame batch per launch, own output each, launches alternating over the context's two queues (edison_queues_fork / "
                               "edison_mfcc_batch_queue_dev / edison_queues_join); medians of 8 interleaved blocks of %d batches each, settled, the serial control beside it" % (nf, n2))
        del ref_o
    except Exception as e:  # a side figure must never cost the headline numbers
        two_queues = dict(error=repr(e))

    # ------------------------------------------------------------------ variant A (log-mel, mfcc_utils.mfcc), same batch
    def mfcc_a_step(i):
        ctx.mfcc_t(bufs[i % len(bufs)], nf, 1024, _lib.MFCC_A, 13, out=out)
    a_ms, aev_ms = timed_region(mfcc_a_step, args.steps, args.warmup, world, args.settle_ms)
    if sample:
        sample["frames_a"] = bufs[(args.warmup + args.steps - 1) % len(bufs)][:n_par].cpu().numpy()
        sample["mfcc_a"] = out[:n_par].cpu().numpy()
    variant_a = dict(metric="MFCC frames/sec, variant A (ln + DCT, mfcc_utils.mfcc)", unit="frames/s",
                     value=round(world * nf / (a_ms * 1e-3), 1), ms_per_step=round(a_ms, 4), kernel_ms=round(aev_ms, 4),
                     roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (aev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))

    # ------------------------------------------------------------------ variant TF (windowed frames: the TensorFlow curve of `main.py mfcc host`), same batch
    def mfcc_tf_step(i):
        ctx.mfcc_t(bufs[i % len(bufs)], nf, 1024, _lib.MFCC_TF, 13, out=out)
    t_ms, tev_ms = timed_region(mfcc_tf_step, args.steps, args.warmup, world, args.settle_ms)
    variant_tf = dict(metric="MFCC frames/sec, variant TF (Hann window + ln + DCT, mfcc_utils.mfcc_tf)", unit="frames/s",
                      value=round(world * nf / (t_ms * 1e-3), 1), ms_per_step=round(t_ms, 4), kernel_ms=round(tev_ms, 4),
                      roofline_frac=round(MFCC_BYTES_PER_FRAME * nf / (tev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      parity="UNPINNED: TensorFlow is not in this image and the reference holds no output of it; checked against a float64 restatement of tf.signal's definitions only",
                      what="the two-frame loop with the window multiplied in at the unpack (ed_mfcc2_window_kernel; round 5)")

    # ------------------------------------------------------------------ variant C: the firmware's Q15 MFCC, same batch
    q15 = None
    if not args.skip_q15:
        out16 = torch.empty((nf, 13), dtype=torch.int16, device=dev)

        def q15_step(i):
            ctx.mfcc_q15_t(bufs[i % len(bufs)], nf, 1024, 13, out=out16)
        q_ms, qev_ms = timed_region(q15_step, args.steps, args.warmup, world, args.settle_ms)
        if sample:
            sample["frames_q15"] = bufs[(args.warmup + args.steps - 1) % len(bufs)][:n_par].cpu().numpy()
            sample["mfcc_q15"] = out16[:n_par].cpu().numpy()
        qbytes = 2048 + 13 * 2
        qach = qbytes * nf / (qev_ms * 1e-3) / 1e9
        q15 = dict(metric="MFCC frames/sec, variant C (firmware Q15 arithmetic, bit-exact)", unit="frames/s",
                   value=round(world * nf / (q_ms * 1e-3), 1), ms_per_step=round(q_ms, 4), dtype="q15/q31",
                   config=dict(workload="mfcc_batch_%dx1024_int16_per_gpu_variantC_13coef" % nf, global_batch=world * nf),
                   roofline=dict(bound="hbm", kernel="ed_mfcc_q15_kernel<false>", achieved=round(qach, 1), peak=HBM_PEAK_GBS,
                                 unit="GB/s", frac=round(qach / HBM_PEAK_GBS, 4), traffic=None, bytes_per_unit=qbytes,
                                 units_per_launch=nf, kernel_ms=round(qev_ms, 4)),
                   )
        q15["roofline"]["kernel"] = "ed_mfcc_q15_kernel<false, true, 6, 18>"
        qtr = hbm_traffic_from_profiles("ed_mfcc_q15_kernel<false, true, 6, 18>:short") if nf == 65536 else None
        if qtr is not None:
            q15["roofline"]["traffic"] = qtr[0]
            q15["roofline"]["traffic_source"] = qtr[1]
    # ------------------------------------------------------------------ variant D: the firmware's float32 extractor, 512-sample frames at hop 256
    variant_d = None
    try:
        from edison_amd.mfcc.mfcc_f32 import MfccF32
        md = MfccF32(ctx=ctx)
        flat = bufs[0].reshape(-1)
        nfd = (flat.numel() - 512) // 256 + 1
        outd = torch.empty((nfd, md.n_out), dtype=torch.int8, device=dev)

        def d_step(i):
            md.compute_t(bufs[i % len(bufs)].reshape(-1), nfd, 256, outd)
        d_ms, dev_ms = timed_region(d_step, max(20, args.steps // 4), min(args.warmup, 50), world)
        variant_d = dict(metric="MFCC frames/sec, variant D (firmware float32 ML-KWS extractor, 512-sample frames, hop 256)", unit="frames/s",
                         value=round(world * nfd / (d_ms * 1e-3), 1), ms_per_step=round(d_ms, 4), kernel_ms=round(dev_ms, 4),
                         frames_per_step=nfd, parity="pinned on the reference's object code (mfcc_compute + CMSIS float transform; tests/golden/mfccf32_golden.npz)")
        md.close()
    except Exception as e:  # an extra workload must never cost the headline numbers
        variant_d = dict(error=repr(e))
    del bufs, q_calls, outs2
    # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    # this command, FETCH_SIZE doubled as the gfx950 guide prescribes); only valid for the default batch size
    tr = hbm_traffic_from_profiles(MFCC_KERNEL + ":short") if nf == 65536 else None
    if tr is not None:
        roofline["traffic"] = tr[0]
        roofline["traffic_source"] = tr[1]

    # ------------------------------------------------------------------ kws: MFCC + int8 CNN (+ all-gather) (configs[2]/[3])
    kws = None
    if not args.skip_kws:
        nu = args.utts
        audio = synth_utterances(nu, 21 + rank, dev)
        logits = torch.empty((nu, 10), dtype=torch.int8, device=dev)
        soft = torch.empty((nu, 10), dtype=torch.int8, device=dev)
        am = torch.empty((nu,), dtype=torch.int32, device=dev)
        feat = torch.empty((nu, 403), dtype=torch.int8, device=dev)
        # N > 1: the step ends in ONE all-gather of the int8 logits over RCCL. The reported step uses torch.distributed's
        # all_gather_into_tensor (backend "nccl" = RCCL); the same step with the collective behind the C-ABI
        # (edison_kws_batch_sharded_dev -> ncclAllGather on the context's stream) is set up, verified against it and timed
        # at the very END of the run under a watchdog (`kws.cabi_collective`), because that path has never executed at
        # N > 1 on hardware and a hang there must not cost the line.
        gather, collective = None, "none"
        if world > 1:
            gather = parallel.LogitsGatherer(nu, 10, device=dev)
            collective = "all_gather int8 logits, %d B per rank (RCCL via torch.distributed all_gather_into_tensor)" % (nu * 10)

        def kws_step(i):
            ctx.kws_t(audio, nu, 31 * 1024, feat=feat, logits=logits, softmax=soft, argmax=am)
            if gather is not None:
                gather(logits)
        # 7 ms steps: up to 50 warm-up steps = 0.35 s. No time-based settle phase here: for N > 1 the step ends in a
        # collective, and ranks must execute the same number of steps.
        kw_ms, kev_ms = timed_region(kws_step, args.steps, min(args.warmup, 50), world)
        if sample:
            n_pu = min(128, nu)
            sample["utt_audio"] = audio[:n_pu].cpu().numpy()
            sample["utt_feat"], sample["utt_logits"] = feat[:n_pu].cpu().numpy(), logits[:n_pu].cpu().numpy()
            sample["utt_softmax"], sample["utt_argmax"] = soft[:n_pu].cpu().numpy(), am[:n_pu].cpu().numpy()
        inf_per_s = world * nu / (kw_ms * 1e-3)
        kach = KWS_BYTES_PER_UTT * nu / (kev_ms * 1e-3) / 1e9
        hist = torch.bincount(am.to(torch.int64), minlength=10).tolist()
        kws = dict(metric="KWS inferences/sec (whole node)", value=round(inf_per_s, 1), unit="inferences/s",
                   mfcc_frames_per_s=round(inf_per_s * 31, 1), ms_per_step=round(kw_ms, 4),
                   config=dict(workload="kws_full_%d_utt_per_gpu_x31_frames_mfccB_int8cnn" % nu, global_batch=world * nu,
                               collective=collective),
                   roofline=dict(bound="hbm", kernel=MFCC_KERNEL_KWS + " + ed_cnn_mfma_kernel", achieved=round(kach, 1),
                                 peak=HBM_PEAK_GBS, unit="GB/s", frac=round(kach / HBM_PEAK_GBS, 4), traffic=None,
                                 bytes_per_unit=KWS_BYTES_PER_UTT, units_per_launch=nu, kernel_ms=round(kev_ms, 4)),
                   class_histogram=hist,
                   # which gather `value` used: the product's own (edison_kws_batch_sharded_dev: ncclAllGather behind the C-ABI) is the
                   # guarded leg at the end of the run; when that leg returns ok its figure is `value_cabi` right here
                   collective_path="torch.distributed" if world > 1 else "none (one rank: nothing to gather)")
        # HBM bytes of the step's dominant kernel (the MFCC over 8.1 M frames: 95 % of the step) from the committed PMC pass;
        # the CNN adds 403 B read + 24 B written per utterance (its counters: kws.cnn.roofline.counters)
        ktr = hbm_traffic_from_profiles(MFCC_KERNEL_KWS + ":long") if nu == 262144 else None
        if ktr is not None:
            kws["roofline"]["traffic"] = ktr[0]
            kws["roofline"]["traffic_source"] = ktr[1]
            kws["roofline"]["traffic_what"] = "FETCH_SIZE x 2 + WRITE_SIZE of the MFCC kernel of the step; + 427 B per utterance for the CNN"
        # ---- the CNN alone on the features of the last step: the matrix-core figure north_star asks for
        def cnn_step(i):
            ctx.cnn_t(feat, nu, logits=logits, softmax=soft, argmax=am)
        c_ms, cev_ms = timed_region(cnn_step, max(100, min(args.steps, 200)), 20, 1 if world == 1 else world)  # 0.3 ms steps: 100 of them = 30 ms
        useful = nu * CNN_MACS_PER_UTT * 2 / (cev_ms * 1e-3) / 1e12
        issued = nu * CNN_MFMA_MACS_PER_UTT * 2 / (cev_ms * 1e-3) / 1e12
        kws["cnn"] = dict(metric="int8 CNN alone (ed_cnn_mfma_kernel)", value=round(world * nu / (c_ms * 1e-3), 1), unit="inferences/s",
                          ms_per_step=round(c_ms, 4),
                          roofline=dict(bound="mfma_i8", kernel="ed_cnn_mfma_kernel", achieved=round(useful, 1), peak=round(MFMA_I8_PEAK_TOPS, 1),
                                        unit="TOP/s", frac=round(useful / MFMA_I8_PEAK_TOPS, 4), kernel_ms=round(cev_ms, 4),
                                        what="achieved = real network MACs (784 752 per utterance) x 2; issued = the MACs of the MFMA instructions issued (32 768 per 32x32x32, 16 384 per 16x16x64) x 2",
                                        issued=round(issued, 1), issued_frac=round(issued / MFMA_I8_PEAK_TOPS, 4),
                                        counters=cnn_counters_from_profiles()))
        # ---- the same graph through the GENERAL matrix-core kernel (what any other retrained graph runs on)
        os.environ["EDISON_NET_FORCE_GENERAL"] = "1"
        try:
            def net_step(i):
                ctx.net_t(feat, nu, logits=logits, argmax=am)
            g_ms, _ = timed_region(net_step, 10, 2, 1 if world == 1 else world)
            kws["general_net_kernel"] = dict(value=round(world * nu / (g_ms * 1e-3), 1), unit="inputs/s", ms_per_step=round(g_ms, 4),
                                             what="kws_conv graph forced onto ed_net_mfma_kernel (edison_net_batch_dev, EDISON_NET_FORCE_GENERAL=1)")
            # ... and through the graph's OWN kernel: the same source compiled at run time with this graph's plan as constants
            # (edison_net_specialize); checked bit for bit against the hand-written kernel's logits of the same features
            # (N = 1 only: a per-GPU kernel figure, and a compiler that fails on ONE rank must not leave the others in a barrier)
            try:
                if world > 1:
                    raise RuntimeError("measured at N = 1 only")
                ref_logits = logits.clone()
                ctx.cnn_t(feat, nu, logits=ref_logits, softmax=soft, argmax=am)
                t_c = time.perf_counter()
                how = ctx.net_specialize()
                t_c = time.perf_counter() - t_c
                o_ms, _ = timed_region(net_step, 20, 3, 1 if world == 1 else world)
                torch.cuda.synchronize()
                kws["general_net_kernel"]["own_kernel"] = dict(
                    value=round(world * nu / (o_ms * 1e-3), 1), unit="inputs/s", ms_per_step=round(o_ms, 4),
                    how={1: "compiled by a hipcc child process", 2: "code object from the on-disk cache", 3: "compiled by hipRTC in this process"}[how],
                    specialize_s=round(t_c, 2), logits_equal_hand_written_kernel=bool(torch.equal(logits, ref_logits)),
                    what="ed_net_mfma_spec: ed_net_mfma_kernel's source with this graph's plan as constants (edison_net_specialize)")
            except Exception as e:  # no compiler on this machine: the graph stays on the general kernel
                kws["general_net_kernel"]["own_kernel"] = dict(value=None, error=str(e)[:200])
        finally:
            del os.environ["EDISON_NET_FORCE_GENERAL"]
        if not args.skip_q15:
            # the same utterances with the firmware's own features (variant C): what the board would answer, at GPU speed
            def kws_q15_step(i):
                ctx.kws_t(audio, nu, 31 * 1024, feat=feat, logits=logits, softmax=soft, argmax=am, q15=True)
                if gather is not None:
                    gather(logits)
            qsteps = max(5, args.steps // 5)
            kq_ms, _ = timed_region(kws_q15_step, qsteps, min(args.warmup, 3), world)
            kws["q15_features"] = dict(value=round(world * nu / (kq_ms * 1e-3), 1), unit="inferences/s", steps=qsteps,
                                       ms_per_step=round(kq_ms, 4),
                                       class_histogram=torch.bincount(am.to(torch.int64), minlength=10).tolist())
        if world == 1:
            del audio

    # ------------------------------------------------------------------ streaming (configs[4]): rank 0, N = 1 only
    streaming = c_pipeline = None
    if rank == 0 and world == 1 and not args.skip_stream:
        streaming = stream_bench(ctx, dev)
        c_pipeline = _c_host_pipeline()

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    cpu = parity = None
    if rank == 0 and world == 1 and not args.skip_cpu:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU numbers
            cpu = dict(error=repr(e))
        try:
            parity = parity_check(sample)
        except Exception as e:  # noqa: BLE001 -- a checker that cannot run is reported as such and fails the run
            parity = dict(ok=False, error=repr(e))

    line = None
    exit_code = 0
    if rank == 0:
        line = dict(metric="MFCC frames/sec + KWS inferences/sec (whole node) at 1/2/4/8 MI355X",
                    value=round(frames_per_s, 1), unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(wall_ms, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype="f32", data="synthetic",
                    config=dict(workload="mfcc_batch_%dx1024_int16_per_gpu_variantB_13coef" % nf, global_batch=world * nf,
                                frame_len=1024, parallelism="dp%d" % world, rotate_buffers=args.rotate, queues=n_queues,
                                **({} if ranks_two_queues is None else {"ranks_with_two_queues": ranks_two_queues})),
                    roofline=roofline, device=info["name"])
        if parity is not None:
            line["parity"] = parity
            if not parity.get("ok"):
                print("bench.py: the timed batch does NOT agree with the oracle: %s" % json.dumps(parity), file=sys.stderr, flush=True)
                exit_code = 5
        if serial is not None:
            line["serial"] = serial
        if two_queues_same_wk is not None:
            line["two_queues_same_wk"] = two_queues_same_wk
            line["config"]["queues_why"] = ("a pair of queues was kept by the calibration, but K = %d < %d: the fork / join of a timed region costs more than the queues gain over so few "
                                            "steps; `settled` (>= 400 steps) runs on the two queues" % (args.steps, MIN_STEPS_TWO_QUEUES))
        if serial_cold is not None:
            line["serial_cold"] = serial_cold
        if calibration is not None:
            line["queue_calibration"] = calibration
        if settled is not None:
            line["settled"] = settled
        if rows_launch is not None:
            line["rows_launch"] = rows_launch
        if batch_list is not None:
            line["batch_list"] = batch_list
        if two_queues is not None:
            line["two_queues"] = two_queues
        line["mfcc_variant_a"] = variant_a
        line["mfcc_variant_tf"] = variant_tf
        if variant_d is not None:
            line["mfcc_variant_d"] = variant_d
        if q15 is not None:
            line["mfcc_q15"] = q15
            if cpu is not None and "q15" in cpu:
                q15["cpu_baseline"] = cpu["q15"]
        if kws is not None:
            line["kws"] = kws
        if streaming is not None:
            line["streaming"] = streaming
        if c_pipeline is not None:
            line["batches_from_a_c_host"] = c_pipeline
        if cpu is not None:
            if "mfcc" in cpu:
                line["cpu_baseline"] = cpu["mfcc"]
                line["cpu_baseline_kws"] = cpu.get("kws")
                variant_a["cpu_baseline"] = cpu.get("mfcc_variant_a")
                if "cnn_reference" in cpu:
                    line["cpu_baseline_cnn_reference"] = cpu["cnn_reference"]
            else:
                line["cpu_baseline"] = cpu
    if world > 1 and kws is not None:
        # ---- the same KWS step with the collective behind the C-ABI. Never run at N > 1 before the first multi-GPU node:
        # everything that could hang (ncclCommInitRank on a second communicator, ncclAllGather on the context's stream)
        # sits behind a watchdog that prints the line as it stands and ends every rank with a NON-ZERO code (guarded_leg).
        def cabi_leg():
            cabi = dict(status="not available")
            try:
                ok = torch.tensor([1 if parallel.dist_available() else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    parallel.init_context_comm(ctx, rank, world, dev)       # raises on every rank if any rank failed
                    logits_all = torch.empty((world * nu, 10), dtype=torch.int8, device=dev)
                    ctx.kws_sharded_t(audio, nu, 31 * 1024, logits_all, feat=feat, logits=logits, softmax=soft, argmax=am)
                    ref_all = gather(logits).clone()
                    torch.cuda.synchronize()
                    same = torch.tensor([1 if torch.equal(ref_all, logits_all) else 0], dtype=torch.int32, device=dev)
                    dist.all_reduce(same, op=dist.ReduceOp.MIN)
                    if int(same.item()) == 1:
                        def kws_cabi_step(i):
                            ctx.kws_sharded_t(audio, nu, 31 * 1024, logits_all, feat=feat, logits=logits, softmax=soft, argmax=am)
                        c_ms, _ = timed_region(kws_cabi_step, args.steps, min(args.warmup, 50), world)
                        cabi = dict(status="ok: rows identical to torch.distributed's all-gather on every rank", value=round(world * nu / (c_ms * 1e-3), 1),
                                    unit="inferences/s", ms_per_step=round(c_ms, 4),
                                    what="edison_kws_batch_sharded_dev: MFCC + CNN + ncclAllGather on the context's stream, one call per rank per step")
                    else:
                        cabi = dict(status="MISMATCH against torch.distributed's all-gather on the first step: not timed")
            except Exception as e:  # noqa: BLE001
                cabi = dict(status="failed: %r" % (e,))
            return cabi
        cabi = guarded_leg(rank, line, line["kws"] if rank == 0 else {}, cabi_leg, float(os.environ.get("EDISON_BENCH_WATCHDOG_S", "120")))
        if rank == 0:
            line["kws"]["cabi_collective"] = cabi
            if cabi.get("status", "").startswith("ok") and "value" in cabi:
                line["kws"]["value_cabi"] = cabi["value"]            # the product's path: shard scoring + ncclAllGather in one C call
                line["kws"]["ms_per_step_cabi"] = cabi["ms_per_step"]
        # the line is printed either way (the figures above do not depend on this leg), but a collective behind the C-ABI that
        # failed or disagreed on ANY rank is a failed run: every rank learns it and leaves non-zero after the line
        bad = torch.tensor([0 if cabi["status"].startswith(("ok", "not available")) else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        exit_code = 4 if int(bad.item()) else 0
    if rank == 0:
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
