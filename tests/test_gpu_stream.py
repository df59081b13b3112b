"""GPU tests of the continuous-microphone mode (edison_stream_*): a stream fed in chunks must give, for every new
frame, exactly what the batch path gives on the corresponding 31-frame window (the firmware's sliding window,
app.c:706-719), and the CNN outputs must be bit-exact against the oracle on the streamed features."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _windows_from_features(rows):
    """rows: [n_frames, 13] int8 (oldest first). Window i ends with frame i; missing history = zeros."""
    n = rows.shape[0]
    padded = np.concatenate([np.zeros((30, 13), np.int8), rows])
    return np.stack([padded[i:i + 31].reshape(-1) for i in range(n)])


@pytest.mark.parametrize("hop,chunk", [(1024, 1), (1024, 7), (512, 5), (512, 64)])
def test_stream_matches_batch_windows(ctx, oracle_mod, oracle_model, hop, chunk):
    from edison_amd import _lib
    from edison_amd.stream import Stream
    rng = np.random.default_rng(hop + chunk)
    n_push = 9 if chunk < 32 else 3
    n_frames = n_push * chunk
    audio = np.clip(rng.normal(0, 2500, n_frames * hop), -32768, 32767).astype(np.int16)
    st = Stream(ctx, hop=hop, chunk_frames=chunk)
    outs = [st.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
    assert st.frames_seen == n_frames
    soft = np.concatenate([o["softmax"] for o in outs])
    logits = np.concatenate([o["logits"] for o in outs])
    am = np.concatenate([o["argmax"] for o in outs])
    # the stream implicitly starts with 1024-hop samples of silence in front of the first hop
    full = np.concatenate([np.zeros(1024 - hop, np.int16), audio])
    _, feat = ctx.mfcc(full, n_frames=n_frames, frame_step=hop, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
    win = _windows_from_features(feat)
    ref = ctx.cnn(win)                                  # batch path of the same library on the same windows
    assert np.array_equal(soft, ref["softmax"]) and np.array_equal(logits, ref["logits"]) and np.array_equal(am, ref["argmax"])
    o = oracle_mod.cnn(oracle_model, win, n_threads=4)  # and the oracle on those windows
    assert np.array_equal(soft, o["softmax"]) and np.array_equal(am, o["argmax"])
    # reset returns to the silent start
    st.reset()
    again = st.push(audio[:chunk * hop])
    assert np.array_equal(again["softmax"], outs[0]["softmax"])
    st.close()


@pytest.mark.parametrize("hop,chunk", [(1024, 1), (512, 6)])
def test_stream_q15_and_output_filter(ctx, oracle_mod, oracle_model, hop, chunk):
    """Continuous mode as the firmware runs it: variant C features (audioCalcMFCCs), sliding window, inference, then the
    output post-processing of app.c:332-356 (moving average in double arithmetic, first maximum, threshold) -- the
    filtered floats must equal the oracle's bit for bit, with the filter state carried across pushes."""
    from edison_amd.stream import Stream
    rng = np.random.default_rng(7 * hop + chunk)
    n_push = 12
    n_frames = n_push * chunk
    audio = np.clip(rng.normal(0, 2500, n_frames * hop), -32768, 32767).astype(np.int16)
    st = Stream(ctx, hop=hop, chunk_frames=chunk, q15=True, output_filter=True)
    outs = [st.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
    soft = np.concatenate([o["softmax"] for o in outs])
    full = np.concatenate([np.zeros(1024 - hop, np.int16), audio])
    feat = oracle_mod.net_input_q15(oracle_mod.mfcc_q15(full, n_frames=n_frames, frame_step=hop, n_threads=4))
    o = oracle_mod.cnn(oracle_model, _windows_from_features(feat), n_threads=4)
    assert np.array_equal(soft, o["softmax"])
    filt, likely, spotted, _ = oracle_mod.output_filter(soft)
    assert np.array_equal(np.concatenate([x["filtered"] for x in outs]).view(np.uint32), filt.view(np.uint32))
    assert np.array_equal(np.concatenate([x["likely"] for x in outs]), likely)
    assert np.array_equal(np.concatenate([x["spotted"] for x in outs]), spotted)
    # reset clears netOutFilt as the firmware does when the mode starts (app.c:299-300)
    st.reset()
    again = st.push(audio[:chunk * hop])
    assert np.array_equal(again["filtered"], outs[0]["filtered"])
    st.close()
    # a different alpha / threshold
    st = Stream(ctx, hop=hop, chunk_frames=chunk, output_filter=True, alpha=0.5, threshold=40.0)
    outs = [st.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(4)]
    filt, likely, spotted, _ = oracle_mod.output_filter(np.concatenate([x["softmax"] for x in outs]), alpha=0.5, threshold=40.0)
    assert np.array_equal(np.concatenate([x["filtered"] for x in outs]), filt)
    assert np.array_equal(np.concatenate([x["spotted"] for x in outs]), spotted)
    st.close()


def test_one_launch_push_filters_like_the_filter_kernel(ctx, oracle_mod):
    """A one-frame float push is ONE launch that also applies the output filter (ed_kws1_kernel); a five-frame stream uses the
    separate filter kernel. Over 200 frames the two and the oracle's filter (app.c:341-356) agree bit for bit, state carried."""
    from edison_amd.stream import Stream
    rng = np.random.default_rng(41)
    audio = np.clip(rng.normal(0, 3000, 200 * 512), -32768, 32767).astype(np.int16)
    s1 = Stream(ctx, hop=512, chunk_frames=1, output_filter=True)
    s5 = Stream(ctx, hop=512, chunk_frames=5, output_filter=True)
    o1 = [s1.push(audio[i * 512:(i + 1) * 512]) for i in range(200)]
    o5 = [s5.push(audio[i * 2560:(i + 1) * 2560]) for i in range(40)]
    soft = np.concatenate([o["softmax"] for o in o1])
    assert np.array_equal(soft, np.concatenate([o["softmax"] for o in o5]))
    filt, likely, spotted, _ = oracle_mod.output_filter(soft)
    for outs in (o1, o5):
        assert np.array_equal(np.concatenate([o["filtered"] for o in outs]).view(np.uint32), filt.view(np.uint32))
        assert np.array_equal(np.concatenate([o["likely"] for o in outs]), likely)
        assert np.array_equal(np.concatenate([o["spotted"] for o in outs]), spotted)
    s1.close()
    s5.close()


@pytest.mark.parametrize("alpha,threshold", [(0.0, 0.0), (1.0, 0.0), (0.25, 126.5)])
def test_one_launch_push_filter_edge_parameters(ctx, oracle_mod, alpha, threshold):
    """alpha 0 (the filtered value IS the newest softmax), alpha 1 (the state never leaves zero) and a threshold nothing reaches, on the
    one-launch path (chunk 1) and on the filter kernel (chunk 3): equal to the oracle's filter bit for bit."""
    from edison_amd.stream import Stream
    rng = np.random.default_rng(int(alpha * 100) + 5)
    audio = np.clip(rng.normal(0, 6000, 60 * 512), -32768, 32767).astype(np.int16)
    for chunk in (1, 3):
        st = Stream(ctx, hop=512, chunk_frames=chunk, output_filter=True, alpha=alpha, threshold=threshold)
        outs = [st.push(audio[i * chunk * 512:(i + 1) * chunk * 512]) for i in range(60 // chunk)]
        soft = np.concatenate([o["softmax"] for o in outs])
        filt, likely, spotted, _ = oracle_mod.output_filter(soft, alpha=alpha, threshold=threshold)
        assert np.array_equal(np.concatenate([o["filtered"] for o in outs]).view(np.uint32), filt.view(np.uint32))
        assert np.array_equal(np.concatenate([o["likely"] for o in outs]), likely)
        assert np.array_equal(np.concatenate([o["spotted"] for o in outs]), spotted)
        st.close()


def test_kws_live_replay(ctx, kws_golden, oracle_mod, oracle_model, tmp_path, capsys):
    """`kws live mcu <wav>`: the firmware's continuous loop on a file -- features (variant C), sliding window, network,
    output filter, FSM. The reference wav says "edison": the filtered wake-word output crosses the threshold, the FSM
    goes IDLE -> HOT, and every printed stage equals the oracle chain."""
    import scipy.io.wavfile as wavfile
    from edison_amd import main as cli
    from edison_amd.kws import kws_live
    a = kws_golden["kws_zero_audio"][:31 * 1024]
    wav = str(tmp_path / "e.wav")
    wavfile.write(wav, 16000, a)
    r = kws_live.run(wav, q15=True, ctx=ctx)
    out = capsys.readouterr().out
    assert len(out.splitlines()) == 31 and "[FSM IDLE -> HOT]" in out and "spotted edison" in out
    feat = oracle_mod.net_input_q15(oracle_mod.mfcc_q15(a))
    o = oracle_mod.cnn(oracle_model, _windows_from_features(feat), n_threads=4)
    assert np.array_equal(r["result"]["softmax"], o["softmax"])
    filt, likely, spotted, _ = oracle_mod.output_filter(o["softmax"])
    assert np.array_equal(r["result"]["filtered"], filt) and np.array_equal(r["result"]["spotted"], spotted)
    assert r["state"] == "HOT" and r["commands"] == []
    assert cli.main(["main.py", "kws", "live", "host", wav]) == 0
    assert "likely:" in capsys.readouterr().out


def test_stream_utterance_equals_kws(ctx, kws_golden):
    """Feeding the 31 frames of the reference wav (hop 1024) ends on the same decision as the batch KWS call."""
    from edison_amd.stream import Stream
    a = kws_golden["kws_zero_audio"][:31 * 1024]
    st = Stream(ctx, hop=1024, chunk_frames=31)
    o = st.push(a)
    assert np.array_equal(o["logits"][-1], kws_golden["kws_zero_logits"])
    assert np.array_equal(o["softmax"][-1], kws_golden["kws_zero_softmax"]) and o["keywords"][-1] == "edison"
    st.close()


def test_stream_argument_checks(ctx):
    from edison_amd import _lib
    from edison_amd.stream import Stream
    with pytest.raises(_lib.EdisonError):
        Stream(ctx, hop=333, chunk_frames=1)      # odd hop: frames would not be 4-byte aligned
    with pytest.raises(_lib.EdisonError):
        Stream(ctx, hop=2048, chunk_frames=1)
    st = Stream(ctx, hop=512, chunk_frames=2)
    with pytest.raises(ValueError):
        st.push(np.zeros(1000, np.int16))
    st.close()


def test_stream_refuses_pushes_after_reconfigure(oracle_mod, oracle_model):
    """A captured stream graph bakes in the MFCC kernel instance chosen for the table shape and the table addresses.
    edison_mfcc_configure after the stream was created must therefore stop that stream (a new one works and agrees
    with the batch path on the new filterbank) instead of silently running the old kernel on re-laid tables."""
    from edison_amd import _lib
    from edison_amd._lib import EdisonError
    from edison_amd.context import Context
    from edison_amd.stream import Stream
    c = Context(0)                                         # private context: the shared fixture keeps its filterbank
    try:
        rng = np.random.default_rng(77)
        audio = np.clip(rng.normal(0, 2500, 8 * 1024), -32768, 32767).astype(np.int16)
        st = Stream(c, hop=1024, chunk_frames=4)
        st.push(audio[:4096])
        c.configure_mfcc(16000, 20.0, 4000.0, 128)         # a filterbank that needs the wide table shape
        with pytest.raises(EdisonError) as e:
            st.push(audio[4096:])
        assert "edison_mfcc_configure" in str(e.value)
        st.close()
        st2 = Stream(c, hop=1024, chunk_frames=4)
        outs = [st2.push(audio[:4096]), st2.push(audio[4096:])]
        soft = np.concatenate([o["softmax"] for o in outs])
        _, feat = c.mfcc(audio, n_frames=8, frame_step=1024, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
        ref = c.cnn(_windows_from_features(feat))
        assert np.array_equal(soft, ref["softmax"])
        st2.close()
    finally:
        c.close()


@pytest.mark.parametrize("hop,chunk,filt", [(512, 1, True), (1024, 3, False)])
def test_stream_host_and_device_pushes_share_one_state(ctx, hop, chunk, filt):
    """Host-pointer pushes of a few frames run against host-mapped buffers (the history lives in pinned host memory),
    device-pointer pushes against device buffers: a caller that alternates them must get what a stream fed by host
    pushes alone answers -- the history moves with the caller (stream_state_to)."""
    import torch
    from edison_amd.stream import Stream
    rng = np.random.default_rng(99 + chunk)
    n_push = 10
    audio = np.clip(rng.normal(0, 2500, n_push * chunk * hop), -32768, 32767).astype(np.int16)
    a = Stream(ctx, hop=hop, chunk_frames=chunk, output_filter=filt)
    b = Stream(ctx, hop=hop, chunk_frames=chunk, output_filter=filt)
    dev = torch.device("cuda", 0)
    for i in range(n_push):
        x = audio[i * chunk * hop:(i + 1) * chunk * hop]
        want = a.push(x)
        if i % 3 == 1:                                 # every third push through the device entry point
            soft = torch.zeros((chunk, 10), dtype=torch.int8, device=dev)
            am = torch.zeros((chunk,), dtype=torch.int32, device=dev)
            fl = torch.zeros((chunk, 10), dtype=torch.float32, device=dev) if filt else None
            b.push_t(torch.from_numpy(x.copy()).to(dev), softmax=soft, argmax=am, filtered=fl)
            torch.cuda.synchronize()
            assert np.array_equal(soft.cpu().numpy(), want["softmax"]) and np.array_equal(am.cpu().numpy(), want["argmax"]), i
            if filt:
                assert np.array_equal(fl.cpu().numpy().view(np.uint32), want["filtered"].view(np.uint32)), i
        else:
            got = b.push(x)
            assert np.array_equal(got["softmax"], want["softmax"]) and np.array_equal(got["argmax"], want["argmax"]), i
            if filt:
                assert np.array_equal(got["filtered"].view(np.uint32), want["filtered"].view(np.uint32)), i
    a.close()
    b.close()


@pytest.mark.parametrize("hop,chunk,filt", [(512, 1, False), (1024, 3, True), (512, 40, False)])
def test_stream_launch_modes_agree(ctx, hop, chunk, filt):
    """edison_stream_opts.launch_mode: EDISON_STREAM_LAUNCH_GRAPH replays the hipGraph captured at creation (what BASELINE
    configs[4] names), EDISON_STREAM_LAUNCH_DIRECT (the default) launches the same kernels one by one -- for one-window host
    pushes with the CNN kernel signalling completion itself. Same samples through both, host pushes and device pushes:
    every output of every push identical, state carried across pushes included."""
    import torch
    from edison_amd.stream import Stream
    rng = np.random.default_rng(3 * hop + chunk)
    n_push = 8
    audio = np.clip(rng.normal(0, 2500, n_push * chunk * hop), -32768, 32767).astype(np.int16)
    res = {}
    for graph in (False, True):
        st = Stream(ctx, hop=hop, chunk_frames=chunk, output_filter=filt, graph=graph)
        outs = [st.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
        res[graph] = {k: np.concatenate([np.asarray(o[k]).reshape(chunk, -1) for o in outs]) for k in outs[0] if k != "keywords"}
        st.close()
    for k in res[False]:
        assert np.array_equal(res[False][k], res[True][k]), k
    # device pushes
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    a = torch.from_numpy(audio).to(dev)
    got = {}
    for graph in (False, True):
        st = Stream(ctx, hop=hop, chunk_frames=chunk, graph=graph)
        so = torch.zeros((n_push * chunk, 10), dtype=torch.int8, device=dev)
        am = torch.zeros((n_push * chunk,), dtype=torch.int32, device=dev)
        for i in range(n_push):
            st.push_t(a[i * chunk * hop:(i + 1) * chunk * hop], softmax=so[i * chunk:(i + 1) * chunk], argmax=am[i * chunk:(i + 1) * chunk])
        torch.cuda.synchronize()
        got[graph] = (so.cpu().numpy(), am.cpu().numpy())
        st.close()
    assert np.array_equal(got[False][0], got[True][0]) and np.array_equal(got[False][1], got[True][1])
    assert np.array_equal(got[False][0], res[False]["softmax"])


@pytest.mark.parametrize("hop,chunk", [(512, 3), (1024, 1), (300, 7)])
def test_device_pushes_slide_through_the_history_buffer_and_wrap(ctx, hop, chunk):
    """Direct device pushes do not move the history back to the front after every push: they slide through buffers eight
    pushes long and shift once when the next push would not fit (edison_stream.hip: slots). 27 device pushes = three wraps,
    with a host push thrown in at the 6th and the 14th (which first brings the history to the front): every output equals
    what a stream fed through host pushes alone answers."""
    import torch
    from edison_amd.stream import Stream
    rng = np.random.default_rng(1000 + hop + chunk)
    n_push = 27
    audio = np.clip(rng.normal(0, 2500, n_push * chunk * hop), -32768, 32767).astype(np.int16)
    ref = Stream(ctx, hop=hop, chunk_frames=chunk)
    want = [ref.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
    ref.close()
    dev = torch.device("cuda", ctx.device)
    ctx.use_torch_stream()
    a = torch.from_numpy(audio).to(dev)
    st = Stream(ctx, hop=hop, chunk_frames=chunk)
    so = torch.zeros((n_push * chunk, 10), dtype=torch.int8, device=dev)
    am = torch.zeros((n_push * chunk,), dtype=torch.int32, device=dev)
    for i in range(n_push):
        if i in (5, 13):
            got = st.push(audio[i * chunk * hop:(i + 1) * chunk * hop])
            assert np.array_equal(got["softmax"], want[i]["softmax"]) and np.array_equal(got["argmax"], want[i]["argmax"]), i
        else:
            st.push_t(a[i * chunk * hop:(i + 1) * chunk * hop], softmax=so[i * chunk:(i + 1) * chunk], argmax=am[i * chunk:(i + 1) * chunk])
    torch.cuda.synchronize()
    st.close()
    so, am = so.cpu().numpy(), am.cpu().numpy()
    for i in range(n_push):
        if i in (5, 13):
            continue
        assert np.array_equal(so[i * chunk:(i + 1) * chunk], want[i]["softmax"]), i
        assert np.array_equal(am[i * chunk:(i + 1) * chunk], np.asarray(want[i]["argmax"]).reshape(-1)), i


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("hop,chunk,tail", [(512, 64, 17), (1024, 9, 1), (512, 4096, 1907)])
def test_ragged_last_push(ctx, hop, chunk, tail, graph):
    """A recording the chunk does not divide: whole chunks through edison_stream_push_dev (sliding history / the captured graph), the
    remaining `tail` frames through edison_stream_push_n_dev, more whole chunks after it -- every window equal to the batch path on
    the same samples (BASELINE configs[4]'s hour is 27 x 4096 + 1907 frames), in BOTH launch modes: under the captured graph (the
    mode configs[4] names) a short push runs the same kernels launched directly (round 5; it used to be refused)."""
    import torch
    from edison_amd import _lib
    from edison_amd.stream import Stream
    rng = np.random.default_rng(hop + chunk + tail)
    plan = [chunk, chunk, tail, chunk, tail, chunk]
    n_frames = sum(plan)
    audio = np.clip(rng.normal(0, 2500, n_frames * hop), -32768, 32767).astype(np.int16)
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(audio).to(dev)
    st = Stream(ctx, hop=hop, chunk_frames=chunk, graph=graph)
    soft, am = [], []
    at = 0
    for n in plan:
        so = torch.zeros((n, 10), dtype=torch.int8, device=dev)
        ar = torch.zeros((n,), dtype=torch.int32, device=dev)
        st.push_t(a[at * hop:(at + n) * hop], softmax=so, argmax=ar, n_frames=None if n == chunk else n)
        soft.append(so); am.append(ar); at += n
    torch.cuda.synchronize()
    assert st.frames_seen == n_frames
    soft = torch.cat(soft).cpu().numpy(); am = torch.cat(am).cpu().numpy()
    full = np.concatenate([np.zeros(1024 - hop, np.int16), audio])
    _, feat = ctx.mfcc(full, n_frames=n_frames, frame_step=hop, variant=_lib.MFCC_B, n_coef=13, want_feat=True)
    ref = ctx.cnn(_windows_from_features(feat))
    assert np.array_equal(soft, ref["softmax"]) and np.array_equal(am, ref["argmax"])
    with pytest.raises(ValueError):
        st.push_t(a[:chunk * hop], n_frames=max(1, chunk - 1))          # sample count must match n_frames
    st.close()


@pytest.mark.parametrize("hop,chunk,tail", [(512, 64, 17), (1024, 9, 1)])
def test_ragged_last_push_with_filter_and_state_machine(ctx, oracle_mod, hop, chunk, tail):
    """The filtered outputs and the machine's states of a ragged push are ITS n entries, nothing more: the caller's device buffers are
    sized [n][..] inside a larger allocation whose remainder must stay untouched (before the fix the stream copied `chunk` entries:
    stale rows of an earlier push, written past an [n]-sized buffer), and the chain over all pushes equals the host chain."""
    import torch
    from edison_amd.stream import Stream
    rng = np.random.default_rng(3 * hop + chunk + tail)
    plan = [chunk, tail, chunk, tail]
    audio = np.clip(rng.normal(0, 2500, sum(plan) * hop), -32768, 32767).astype(np.int16)
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(audio).to(dev)
    thr, dt_us = 0.5, hop * 1_000_000 // 16000
    st = Stream(ctx, hop=hop, chunk_frames=chunk, fsm=True, threshold=thr)
    GUARD = 7777.0
    soft, filt, likely, spotted, states = [], [], [], [], []
    at = 0
    for n in plan:
        so = torch.zeros((n, 10), dtype=torch.int8, device=dev)
        fl = torch.full((chunk + 8, 10), GUARD, dtype=torch.float32, device=dev)       # [n] used, the rest is the guard
        li = torch.full((chunk + 8,), -77, dtype=torch.int32, device=dev)
        sp = torch.full((chunk + 8,), -77, dtype=torch.int32, device=dev)
        sd = torch.full((chunk + 8,), -77, dtype=torch.int32, device=dev)
        st.push_t(a[at * hop:(at + n) * hop], softmax=so, filtered=fl, likely=li, spotted=sp, n_frames=None if n == chunk else n)
        ctx._check(st._L.edison_stream_fsm_dev(st._h, sd.data_ptr()))
        torch.cuda.synchronize()
        assert (fl[n:] == GUARD).all() and (li[n:] == -77).all() and (sp[n:] == -77).all() and (sd[n:] == -77).all(), n
        soft.append(so.cpu().numpy()); filt.append(fl[:n].cpu().numpy()); likely.append(li[:n].cpu().numpy())
        spotted.append(sp[:n].cpu().numpy()); states.append(sd[:n].cpu().numpy())
        at += n
        # the host-pointer form after a ragged device push answers with n entries too
        hf, hl, hs = np.full((chunk, 10), GUARD, np.float32), np.full(chunk, -77, np.int32), np.full(chunk, -77, np.int32)
        hst = np.full(chunk, -77, np.int32)
        ctx._check(st._L.edison_stream_filtered(st._h, hf.ctypes.data, hl.ctypes.data, hs.ctypes.data))
        ctx._check(st._L.edison_stream_fsm(st._h, None, hst.ctypes.data))
        assert np.array_equal(hf[:n].view(np.uint32), filt[-1].view(np.uint32)) and (hf[n:] == GUARD).all() and (hl[n:] == -77).all() and (hst[n:] == -77).all()
        assert np.array_equal(hst[:n], states[-1])
    rf, rl, rs, _, rstates, _ = _host_chain(oracle_mod, np.concatenate(soft), 0.9, thr, dt_us)
    assert np.array_equal(np.concatenate(filt).view(np.uint32), rf.view(np.uint32))
    assert np.array_equal(np.concatenate(likely), rl) and np.array_equal(np.concatenate(spotted), rs)
    assert np.array_equal(np.concatenate(states), rstates)
    st.close()


def _host_chain(oracle_mod, soft, alpha, threshold, dt_us, state=None, fsm=None):
    """The chain on the host, independently of the GPU stage AND of the product's state machine: the oracle's output filter
    (oracle/postproc_ref.c), then oracle/fsm_ref.py -- edisonFSM restated from the reference's app.c:727-928 with the firmware's own
    tables and pointer walks; it shares no code with csrc/edison_fsm_core.h, which host and device compile. `fsm`: an
    oracle machine (fsm_ref.EdisonFsmRef) carried in from an earlier call."""
    from oracle import fsm_ref
    filt, likely, spotted, st = oracle_mod.output_filter(soft, state=state, alpha=alpha, threshold=threshold)
    states, fsm = fsm_ref.walk(filt[np.arange(len(likely)), likely], likely, dt_us, true_threshold=threshold, machine=fsm)
    return filt, likely, spotted, st, np.array(states, dtype=np.int32), fsm


def _fsm_tuple(f):
    """The product's machine (edison_fsm, or the `raw` tuple of Stream.fsm_snapshot) or the oracle's, reduced to what both have whatever
    the history: state, counter, wake word, the pending location while it can still be consumed (LOC, SET), the pending value in
    SET, the last executed command, the number of commands."""
    if hasattr(f, "ediState"):
        last = f.last_command_idx()
        return (f.ediState, f.hotTimeout, f.wakeWordIdx, f.pending_location_idx() if f.ediState in (3, 4) else None,
                f.pending_value_idx() if f.ediState == 4 else None, last[0], last[1], len(f.executed))
    t = tuple(f) if isinstance(f, (tuple, list)) else (f.state, f.hot_timeout_ms, f.wake_idx, f.loc_idx, f.val_idx, f.last_loc, f.last_val, f.commands)
    return (t[0], t[1], t[2], t[3] if t[0] in (3, 4) else None, t[4] if t[0] == 4 else None, t[5], t[6], t[7])


def _scenario(rng, n_segments, dt_us):
    """int8 softmax rows in time order: random segments in which one class is confident (or nobody is), and between them scripted
    episodes -- a whole command (wake word, location, value), a wake word followed by silence (the 5 s time-out in HOT), a wake
    word and a location followed by silence (the time-out in LOC) -- long enough for the moving average to cross any threshold
    used here, so that the machine is walked through every state and both time-outs."""
    per_5s = 5_000_000 // dt_us
    rows = []

    def hold(cls, length, conf=127):
        for _ in range(length):
            r = rng.integers(0, 8, 10)
            r[cls] = conf
            rows.append(r)
    for k in range(n_segments):
        if k % 20 == 5:
            hold(0, 30); hold(int(rng.choice([1, 2, 3, 4, 5])), 30); hold(int(rng.choice([6, 7])), 30); hold(9, 10)   # a command
        elif k % 20 == 11:
            hold(0, 30); hold(9, per_5s + 40, conf=int(rng.choice([127, 40])))                                         # time-out in HOT
        elif k % 20 == 17:
            hold(0, 30); hold(int(rng.choice([1, 2, 3, 4, 5])), 30); hold(8, per_5s + 40)                               # time-out in LOC
        else:
            cls = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7])) if rng.random() < 0.45 else int(rng.choice([8, 9]))
            hold(cls, int(rng.choice([1, 2, 5, 20, per_5s // 2, per_5s + 5])), conf=int(rng.choice([127, 100, 60, 20])))
    return np.array(rows, dtype=np.int8)


@pytest.mark.parametrize("seed,threshold,dt_us", [(1, 0.5, 64000), (2, 30.0, 64000), (3, 50.0, 32000), (4, 80.0, 1_000_000), (5, 30.0, 999)])
def test_postproc_chain_with_the_state_machine_on_the_gpu(ctx, oracle_mod, seed, threshold, dt_us):
    """moving average -> first maximum -> threshold -> edisonFSM as ONE GPU stage (edison_postproc = the stream's filter
    kernel with the machine behind it), against the host chain over scripted scenario streams: filtered floats bit for bit, the
    state after EVERY inference, the machine itself at the end (time-out counter, pending location / value, executed commands) --
    incl. both 5 s time-outs, the value that arrives at the very step that times out (dropped), dt below one millisecond (the
    firmware's `hotTimeout += dt/1000` truncates to 0: the machine never times out), and the state carried across calls. The
    reference holds no vectors for this chain: parity unpinned; the checker is oracle/postproc_ref.c + oracle/fsm_ref.py, neither of
    which shares code with the product (the product's own host machine, edison_fsm_step, is held against the same oracle on the CPU:
    tests/test_host_cpu.py::test_fsm_equals_the_independent_restatement_of_edisonFSM)."""
    from edison_amd.context import postproc
    rng = np.random.default_rng(seed)
    soft = _scenario(rng, 80, dt_us if dt_us >= 1000 else 64000)
    alpha = 0.9
    filt, likely, spotted, st, states, fsm = _host_chain(oracle_mod, soft, alpha, threshold, dt_us)
    got = postproc(ctx, soft, alpha=alpha, threshold=threshold, dt_us=dt_us)
    assert np.array_equal(got["filtered"].view(np.uint32), filt.view(np.uint32))
    assert np.array_equal(got["likely"], likely) and np.array_equal(got["spotted"], spotted)
    assert np.array_equal(got["fsm_states"], states), np.flatnonzero(got["fsm_states"] != states)[:5]
    assert _fsm_tuple(got["fsm"]) == _fsm_tuple(fsm)
    if 1000 <= dt_us <= 64000:   # the scenario really gets everywhere: commands executed, both time-outs taken
        timeouts = int(((states[1:] == 1) & (states[:-1] >= 2) & (states[:-1] <= 3)).sum())
        assert len(fsm.executed) >= 3 and set(states.tolist()) >= {1, 2, 3, 4} and timeouts >= 4, (len(fsm.executed), set(states.tolist()), timeouts)
    # in three pieces, the filter state and the machine carried along: the same answers
    cuts = [0, len(soft) // 3, len(soft) // 3 + 1, len(soft)]
    state, m, parts = None, None, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = postproc(ctx, soft[a:b], alpha=alpha, threshold=threshold, dt_us=dt_us, state=state, fsm=m)
        state, m = g["state"], g["fsm"]
        parts.append(g["fsm_states"])
    assert np.array_equal(np.concatenate(parts), states) and _fsm_tuple(m) == _fsm_tuple(fsm)


def test_postproc_argument_checks(ctx):
    """edison_postproc refuses what edison_fsm_step refuses: a machine in a state that does not exist (it used to run and answer n
    states of -1 with EDISON_OK), a threshold that is not a number, n beyond 2^30."""
    import ctypes
    from edison_amd import _lib
    from edison_amd.context import postproc
    L = _lib.lib()
    soft = np.zeros((4, 10), np.int8)
    bad = _lib.Fsm(); L.edison_fsm_init(ctypes.byref(bad)); bad.state = 9
    with pytest.raises(_lib.EdisonError) as ei:
        postproc(ctx, soft, fsm=bad)
    assert ei.value.code == _lib.E_ARGUMENT
    with pytest.raises(_lib.EdisonError) as ei:
        postproc(ctx, soft, threshold=float("nan"))
    assert ei.value.code == _lib.E_ARGUMENT
    st = np.zeros(10, np.float32)
    assert L.edison_postproc(ctx._h, soft.ctypes.data, 1 << 30, 0.9, 0.5, 64000, st.ctypes.data, None, None, None, None, None) == _lib.E_ARGUMENT
    ok = postproc(ctx, soft)
    assert ok["fsm_states"].tolist() == [1, 1, 1, 1]


@pytest.mark.parametrize("hop,chunk", [(512, 1), (512, 6), (1024, 64)])
def test_stream_with_the_state_machine_as_its_last_stage(ctx, oracle_mod, hop, chunk):
    """A stream created with fsm = 1: host pushes (chunk 1: the one-launch kernel steps the machine itself; larger chunks: the filter
    kernel's last lane walks the push) and device pushes give, inference by inference, the states of the host chain run on the
    stream's own softmax outputs; the machine survives pushes and edison_stream_reset puts it back into RESET."""
    import torch
    from edison_amd.stream import Stream
    rng = np.random.default_rng(17 * hop + chunk)
    n_push = 40 if chunk == 1 else 6
    audio = np.clip(rng.normal(0, 2500, n_push * chunk * hop), -32768, 32767).astype(np.int16)
    dt_us = hop * 1_000_000 // 16000
    thr = 0.5
    st = Stream(ctx, hop=hop, chunk_frames=chunk, fsm=True, threshold=thr)
    outs = [st.push(audio[i * chunk * hop:(i + 1) * chunk * hop]) for i in range(n_push)]
    soft = np.concatenate([o["softmax"] for o in outs])
    got = np.concatenate([o["fsm_states"] for o in outs])
    filt, likely, spotted, _, states, fsm = _host_chain(oracle_mod, soft, 0.9, thr, dt_us)
    assert np.array_equal(np.concatenate([o["filtered"] for o in outs]).view(np.uint32), filt.view(np.uint32))
    assert np.array_equal(got, states)
    assert _fsm_tuple(outs[-1]["fsm"]["raw"]) == _fsm_tuple(fsm)
    assert got[0] == 1                                   # RESET -> IDLE at the first inference (app.c:766-791)
    # device pushes on the same stream object continue the same machine
    dev = torch.device("cuda", 0)
    more = np.clip(rng.normal(0, 2500, 2 * chunk * hop), -32768, 32767).astype(np.int16)
    a = torch.from_numpy(more).to(dev)
    so = torch.zeros((chunk, 10), dtype=torch.int8, device=dev)
    sd = torch.zeros((chunk,), dtype=torch.int32, device=dev)
    dstates, dsoft = [], []
    for i in range(2):
        st.push_t(a[i * chunk * hop:(i + 1) * chunk * hop], softmax=so)
        ctx._check(st._L.edison_stream_fsm_dev(st._h, sd.data_ptr()))
        torch.cuda.synchronize()
        dstates.append(sd.cpu().numpy().copy()); dsoft.append(so.cpu().numpy().copy())
    _, _, _, _, states2, _ = _host_chain(oracle_mod, np.concatenate([soft] + dsoft), 0.9, thr, dt_us)
    assert np.array_equal(np.concatenate(dstates), states2[len(soft):])
    st.reset()
    again = st.push(audio[:chunk * hop])
    assert np.array_equal(again["fsm_states"], outs[0]["fsm_states"])
    st.close()
    with pytest.raises(Exception):
        Stream(ctx, hop=hop, chunk_frames=chunk, fsm=True, output_filter=False, alpha=2.0)
