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
