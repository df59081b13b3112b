"""GPU tests at BASELINE.json's full sizes, through size-independent properties.

The oracle cannot run 65 536 frames / 262 144 utterances in seconds, so the full-size batches are built from a
small oracle-checked base set placed at rotating positions: element n of the big batch is base[(n + n // B) % B].
Then (a) every output row must be bit-identical to the small-batch output of the base row it came from
(position independence over the whole grid / persistent loop), and (b) the small batch is checked against the
oracle. Device tensors come from torch (memory + stream plumbing only); the computation is the C-ABI call.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rot_index(torch, n, base, device):
    i = torch.arange(n, device=device, dtype=torch.int64)
    return (i + i // base) % base


def test_mfcc_full_size_65536_frames(ctx, oracle_mod):
    import torch
    from edison_amd import _lib
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    B, N = 64, 65536
    rng = np.random.default_rng(20)
    base = np.clip(rng.normal(0, 3000, (B, 1024)), -32768, 32767).astype(np.int16)
    base[0] = 0
    base[1] = 32767
    base[2] = -32768
    tb = torch.from_numpy(base).to(dev)
    idx = _rot_index(torch, N, B, dev)
    audio = tb[idx].contiguous()                                   # [65536, 1024] int16 = 134 217 728 B
    assert audio.numel() * 2 == 134217728
    for variant, ov, atol, rtol in ((_lib.MFCC_A, 0, 1e-3, 1e-4), (_lib.MFCC_B, 1, 1e-2, 1e-5)):
        small = torch.empty((B, 13), dtype=torch.float32, device=dev)
        ctx.mfcc_t(tb, B, 1024, variant, 13, out=small)
        big = torch.empty((N, 13), dtype=torch.float32, device=dev)
        feat = torch.empty((N, 13), dtype=torch.int8, device=dev)
        ctx.mfcc_t(audio, N, 1024, variant, 13, out=big, feat=feat)
        torch.cuda.synchronize()
        assert torch.equal(big, small[idx]), "output depends on the frame's position in the batch"
        q = torch.clamp(big, -128, 127).round().to(torch.int8)
        assert torch.equal(feat, q)
        ref = oracle_mod.mfcc(base.reshape(-1), ov)[:, :13]
        got = small.cpu().numpy()
        assert np.all(np.abs(got - ref) <= atol + rtol * np.abs(ref))
        # a checksum of checksums, for the log
        print("variant", variant, "sum", float(big.double().sum()), "expected", float(small.double()[idx].sum()))


def test_kws_full_size_262144_utterances(ctx, oracle_mod, oracle_model):
    import torch
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    B, N, L = 64, 262144, 31 * 1024
    rng = np.random.default_rng(21)
    base = np.concatenate([np.clip(rng.normal(0, 3000, (B - 8, L)), -32768, 32767),
                           np.clip(rng.normal(0, 0.01 * 32767, (6, L)), -32768, 32767),
                           np.zeros((2, L))]).astype(np.int16)
    tb = torch.from_numpy(base).to(dev)
    idx = _rot_index(torch, N, B, dev)
    audio = torch.empty((N, L), dtype=torch.int16, device=dev)     # 16.6 GB
    for lo in range(0, N, 16384):                                  # chunked gather keeps the temporary small
        audio[lo:lo + 16384] = tb[idx[lo:lo + 16384]]
    feat_s = torch.empty((B, 403), dtype=torch.int8, device=dev)
    log_s = torch.empty((B, 10), dtype=torch.int8, device=dev)
    sm_s = torch.empty((B, 10), dtype=torch.int8, device=dev)
    am_s = torch.empty((B,), dtype=torch.int32, device=dev)
    ctx.kws_t(tb, B, L, feat=feat_s, logits=log_s, softmax=sm_s, argmax=am_s)
    feat = torch.empty((N, 403), dtype=torch.int8, device=dev)
    logits = torch.empty((N, 10), dtype=torch.int8, device=dev)
    sm = torch.empty((N, 10), dtype=torch.int8, device=dev)
    am = torch.empty((N,), dtype=torch.int32, device=dev)
    ctx.kws_t(audio, N, L, feat=feat, logits=logits, softmax=sm, argmax=am)
    torch.cuda.synchronize()
    assert torch.equal(feat, feat_s[idx]) and torch.equal(logits, log_s[idx])
    assert torch.equal(sm, sm_s[idx]) and torch.equal(am, am_s[idx])
    # without the optional feature buffer the library uses its own scratch: same answers
    am2 = torch.empty((N,), dtype=torch.int32, device=dev)
    ctx.kws_t(audio, N, L, argmax=am2)
    torch.cuda.synchronize()
    assert torch.equal(am2, am)
    del audio
    # the base batch against the oracle: CNN bit-exact on the GPU's features, features within one rounding step
    f = feat_s.cpu().numpy()
    o = oracle_mod.cnn(oracle_model, f, n_threads=4)
    assert np.array_equal(o["logits"], log_s.cpu().numpy()) and np.array_equal(o["softmax"], sm_s.cpu().numpy())
    assert np.array_equal(o["argmax"], am_s.cpu().numpy())
    ref_feat = np.stack([oracle_mod.net_input(oracle_mod.mfcc(base[u], 1)[:, :13]).reshape(-1) for u in range(B)])
    d = np.abs(ref_feat.astype(int) - f.astype(int))
    assert d.max() <= 1 and (d != 0).sum() <= 3
    counts = np.bincount(am.cpu().numpy(), minlength=10)
    print("class histogram over 262144 utterances:", counts.tolist())


def test_mfcc_full_size_scaling_property(ctx):
    """Variant B without the log is homogeneous: every stage (FFT, |.|, mel dot, DCT) is linear or a magnitude, so
    doubling the samples doubles every coefficient -- and in binary floating point a factor 2 is exact at every
    step, so the two launches must agree BIT FOR BIT. Independent of the oracle, at the full 65 536-frame size."""
    import torch
    from edison_amd import _lib
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    N = 65536
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    x = (torch.randn((N, 1024), generator=g, device=dev) * 2500.0).clamp_(-16000, 16000).to(torch.int16)
    a = torch.empty((N, 32), dtype=torch.float32, device=dev)
    b = torch.empty((N, 32), dtype=torch.float32, device=dev)
    ctx.mfcc_t(x, N, 1024, _lib.MFCC_B, 32, out=a)
    ctx.mfcc_t(x * 2, N, 1024, _lib.MFCC_B, 32, out=b)
    torch.cuda.synchronize()
    assert torch.equal(b, a * 2)
    assert torch.isfinite(a).all() and float(a.abs().max()) > 10.0
    # silence in, zeros out (no log, no offset)
    z = torch.zeros((4096, 1024), dtype=torch.int16, device=dev)
    ctx.mfcc_t(z, 4096, 1024, _lib.MFCC_B, 32, out=a[:4096])
    torch.cuda.synchronize()
    assert float(a[:4096].abs().max()) == 0.0


def test_independent_batches_in_flight_together_equal_serial_calls(ctx):
    """Two 65 536-frame batches in flight at once: edison_mfcc_batch_dev on two HIP streams of ONE context (edison_set_stream per
    call; streams of different priority are on different hardware queues by construction), each batch with its own input and
    output, forked from and joined into one stream by events. The MFCC entry points share only read-only state (tables), so the
    outputs must be bit-identical to serial calls on the same batches -- whatever the launches' overlap does to the speed
    (it does not pay on this hardware; profiles/r05_mfcc_two_queues_notes.txt)."""
    import torch
    from edison_amd import _lib
    dev = torch.device("cuda", ctx.device)
    main = torch.cuda.Stream(device=dev)
    qs = [torch.cuda.Stream(device=dev, priority=0), torch.cuda.Stream(device=dev, priority=-1)]
    N = 65536
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    with torch.cuda.stream(main):
        bufs = [(torch.randn((N, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(4)]
        ref = []
        ctx.use_torch_stream(main)
        for b in bufs:
            o = torch.empty((N, 13), dtype=torch.float32, device=dev)
            ctx.mfcc_t(b, N, 1024, _lib.MFCC_B, 13, out=o)
            ref.append(o)
        outs = [torch.zeros((N, 13), dtype=torch.float32, device=dev) for _ in bufs]
        feats = [torch.zeros((N, 13), dtype=torch.int8, device=dev) for _ in bufs]
        ev = torch.cuda.Event()
        ev.record(main)
    for q in qs:
        q.wait_event(ev)
    try:
        for rep in range(3):                               # 12 launches, two in flight at any time
            for i, b in enumerate(bufs):
                ctx.use_torch_stream(qs[i % 2])
                ctx.mfcc_t(b, N, 1024, _lib.MFCC_B, 13, out=outs[i], feat=feats[i])
        for q in qs:
            e = torch.cuda.Event()
            e.record(q)
            main.wait_event(e)
    finally:
        ctx.use_torch_stream()
    torch.cuda.synchronize()
    for i in range(len(bufs)):
        assert torch.equal(outs[i], ref[i]), i
        assert torch.equal(feats[i].to(torch.float32), ref[i].clamp(-128, 127).round()), i


@pytest.mark.parametrize("n_batches,n_each,variant_log", [(5, 1001, (1, False)), (3, 64, (0, False)), (19, 33, (1, True)), (1, 7, (1, False)), (2, 1, (0, False))])
def test_batch_list_launch_equals_one_call_per_batch(ctx, n_batches, n_each, variant_log):
    """edison_mfcc_batches_dev: independent batches at unrelated addresses (separate allocations, one of them 2-byte aligned only
    in the last case set), own fp32 and int8 outputs each, ONE launch per 16 batches -- bit-identical to edison_mfcc_batch_dev per
    batch. Odd frame counts make frame pairs straddle two batches (frame A the last row of batch g, frame B row 0 of batch g + 1);
    19 batches take two launches; guard rows behind every output stay untouched."""
    import torch
    from edison_amd import _lib
    variant, use_log = variant_log
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    g = torch.Generator(device=dev)
    g.manual_seed(100 * n_batches + n_each)
    audios, pads = [], []
    for b in range(n_batches):
        pads.append(torch.empty((1 + 37 * b,), dtype=torch.int8, device=dev))          # unrelated addresses between the batches
        raw = (torch.randn((n_each * 1024 + 2,), generator=g, device=dev) * (3000 if b % 3 else 30)).clamp_(-32768, 32767).to(torch.int16)
        audios.append(raw[1:1 + n_each * 1024] if (b == 1 and n_batches > 1) else raw[:n_each * 1024])   # batch 1: 2-byte aligned only
    GUARD = 4321.0
    outs = [torch.full((n_each + 2, 13), GUARD, dtype=torch.float32, device=dev) for _ in range(n_batches)]
    feats = [torch.full((n_each + 2, 13), 99, dtype=torch.int8, device=dev) for _ in range(n_batches)]
    ctx.mfcc_batches_t(audios, n_each, 1024, variant, 13, outs=[o[:n_each] for o in outs], feats=[f[:n_each] for f in feats], use_log=use_log)
    torch.cuda.synchronize()
    for b in range(n_batches):
        ro = torch.empty((n_each, 13), dtype=torch.float32, device=dev)
        rf = torch.empty((n_each, 13), dtype=torch.int8, device=dev)
        ctx.mfcc_t(audios[b], n_each, 1024, variant, 13, out=ro, feat=rf, use_log=use_log)
        torch.cuda.synchronize()
        assert torch.equal(outs[b][:n_each], ro), b
        assert torch.equal(feats[b][:n_each], rf), b
        assert (outs[b][n_each:] == GUARD).all() and (feats[b][n_each:] == 99).all(), b
    # only one kind of output; and the argument checks
    o2 = [torch.zeros((n_each, 13), dtype=torch.float32, device=dev) for _ in range(n_batches)]
    ctx.mfcc_batches_t(audios, n_each, 1024, variant, 13, outs=o2, use_log=use_log)
    torch.cuda.synchronize()
    assert all(torch.equal(o2[b], outs[b][:n_each]) for b in range(n_batches))
    with pytest.raises(_lib.EdisonError) as ei:
        ctx.mfcc_batches_t(audios, n_each, 1024, _lib.MFCC_C, 13, outs=o2)
    assert ei.value.code == _lib.E_NO_IMPL


def test_the_contexts_two_queues(ctx):
    """edison_queues_calibrate / fork / edison_mfcc_batch_queue_dev / join: whatever pair of streams the calibration keeps (or none: both
    indices then mean one stream), the outputs equal serial calls bit for bit; what was kept is never slower than the serial sequence by
    the calibration's own measure; the queue call outside fork ... join and a calibration between them are refused."""
    import torch
    from edison_amd import _lib
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    N = 65536
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    bufs = [(torch.randn((N, 1024), generator=g, device=dev) * 3000).clamp_(-32768, 32767).to(torch.int16) for _ in range(3)]
    ref = []
    for b in bufs:
        o = torch.empty((N, 13), dtype=torch.float32, device=dev)
        ctx.mfcc_t(b, N, 1024, _lib.MFCC_B, 13, out=o)
        ref.append(o)
    cal = ctx.queues_calibrate(bufs[0], N)
    assert 20.0 < cal["serial_us"] < 200.0 and cal["best_us"] <= cal["serial_us"]
    assert cal["pair"] is None or (0 <= cal["pair"][0] < cal["pair"][1] <= 4 and cal["best_us"] < 0.99 * cal["serial_us"])
    outs = [torch.zeros((N, 13), dtype=torch.float32, device=dev) for _ in bufs]
    calls = [ctx.mfcc_queue_call(i & 1, bufs[i], N, 1024, _lib.MFCC_B, 13, out=outs[i]) for i in range(3)]
    with pytest.raises(_lib.EdisonError) as ei:
        calls[0]()                                             # not forked
    assert ei.value.code == _lib.E_ARGUMENT
    ctx.queues_fork()
    with pytest.raises(_lib.EdisonError):
        ctx.queues_calibrate(bufs[0], N)                       # not between fork and join
    for rep in range(4):
        for c in calls:
            c()
    ctx.queues_join()
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(outs[i], ref[i]), i
    ctx.queues_join()                                          # a second join is a no-op


def test_queue_calibration_of_a_long_launch_is_short(ctx):
    """A batch whose launch takes milliseconds (2 M frames: ~1.4 ms) keeps one queue after one short probe: the calibration must not spend
    seconds on launches that have nothing to gain from a second queue."""
    import time
    import torch
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    N = 1 << 21
    x = torch.zeros((N, 1024), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cal = ctx.queues_calibrate(x, N)
    dt = time.perf_counter() - t0
    assert cal["pair"] is None and cal["serial_us"] > 1000.0 and dt < 1.0, (cal, dt)
    del x
    torch.cuda.empty_cache()
