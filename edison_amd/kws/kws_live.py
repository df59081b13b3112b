"""``kws live <host|mcu> <wav>`` -- the firmware's continuous mode (firmware/src/app.c:288-371) replayed on a wav file.

The reference's ``audio/edison/kws/kws_live.py`` listens to a microphone; the signal path behind the microphone is
what this module runs, frame by frame, on the GPU: 1024-sample frames -> MFCC -> 31-row sliding window -> int8 network
-> moving average over the outputs -> maximum / threshold -> wake-word state machine. ``host`` uses the host float
model of the features (variant B), ``mcu`` the firmware's own Q15 arithmetic (variant C). The printed lines follow the
firmware's UART log (app.c:330-353).
"""
import sys

import numpy as np

from .. import config as cfg
from ..context import KEYWORDS, default_context
from ..stream import Fsm, Stream
from .kws_host import read_wav


def netOutFilt(net_outs, alpha):
    """The host-side moving average of the reference's live view (kws_live.py:139-152): row 0 is all zeros, row i+1 =
    alpha * row i + (1 - alpha) * net_outs[i], in float64 like the reference's Python floats. (The firmware's own
    filter -- float32 state, double arithmetic, app.c:332-356 -- is the stream's `output_filter` option.)"""
    x = np.asarray(net_outs, dtype=np.float64)
    flt = np.zeros((x.shape[0] + 1, x.shape[1]), dtype=np.float64)
    for i in range(x.shape[0]):
        flt[i + 1] = alpha * flt[i] + (1.0 - alpha) * x[i]
    return flt


def run(path, q15=False, ctx=None, out=None, alpha=0.9, threshold=0.5):
    out = out or sys.stdout
    ctx = ctx or default_context()
    data = read_wav(path)
    hop = cfg.frame_length
    n = -(-data.shape[0] // hop)
    data = np.pad(data, (0, n * hop - data.shape[0]))
    # the firmware's loop body behind the network (app.c:341-371) is part of the push: moving average, maximum, threshold and
    # edisonFSM run as the last GPU stages (Stream(fsm=True)); what comes back is the state after every inference
    st = Stream(ctx, hop=hop, chunk_frames=n, q15=q15, output_filter=True, alpha=alpha, threshold=threshold, fsm=True)
    res = st.push(data)
    st.close()
    events = []
    before, loc, val = "RESET", -1, -1
    for i in range(n):
        likely, spotted = int(res["likely"][i]), int(res["spotted"][i])
        line = "pred: [ " + " ".join("%2.2f" % float(v) for v in res["softmax"][i]) + " ] likely: %s" % KEYWORDS[likely]
        if spotted >= 0:
            line += " spotted %s" % KEYWORDS[spotted]
        after = Fsm.STATES[int(res["fsm_states"][i])]
        if after != before:
            line += "   [FSM %s -> %s]" % (before, after)
        if before == "HOT" and after == "LOC":
            loc = likely                                          # the location that was spotted (app.c:812-816)
        if before == "LOC" and after == "SET":
            val = likely                                          # ... and the value (app.c:836-840)
        if before == "SET":                                       # the step that executes the command (app.c:850-872)
            line += "   [%s %s]" % (KEYWORDS[loc], KEYWORDS[val])
            events.append((KEYWORDS[loc], KEYWORDS[val]))
        before = after
        print(line, file=out)
    assert len(events) == res["fsm"]["commands"]
    return dict(result=res, commands=events, state=before)


def main(argv):
    if len(argv) < 3 or argv[1] not in ("host", "mcu"):
        print("usage: kws live <host|mcu> <wav>   (the microphone front end of the reference is not part of this port)")
        return 0 if len(argv) >= 2 and argv[1] in ("host", "mcu") else 1
    run(argv[2], q15=(argv[1] == "mcu"))
    return 0
