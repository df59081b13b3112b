#!/usr/bin/env python3
"""The output filter done by the one-launch kernel (chunk 1) against the filter kernel (chunk 5) and the oracle, bit for bit; then the latency."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from edison_amd.context import Context
from edison_amd.stream import Stream
from oracle import oracle
c = Context(0)
rng = np.random.default_rng(4)
a = np.clip(rng.normal(0, 3000, 300 * 512), -32768, 32767).astype(np.int16)
s1, s5 = Stream(c, hop=512, chunk_frames=1, output_filter=True), Stream(c, hop=512, chunk_frames=5, output_filter=True)
o1 = [s1.push(a[i * 512:(i + 1) * 512]) for i in range(300)]
o5 = [s5.push(a[i * 2560:(i + 1) * 2560]) for i in range(60)]
f1 = np.concatenate([o["filtered"] for o in o1]); f5 = np.concatenate([o["filtered"] for o in o5])
so = np.concatenate([o["softmax"] for o in o1])
ref = oracle.output_filter(so)
print("filtered: chunk 1 == chunk 5:", np.array_equal(f1.view(np.uint32), f5.view(np.uint32)), "| == oracle:", np.array_equal(f1.view(np.uint32), ref[0].view(np.uint32)),
      "| likely:", np.array_equal(np.concatenate([o["likely"] for o in o1]), ref[1]), "| spotted:", np.array_equal(np.concatenate([o["spotted"] for o in o1]), ref[2]))
x = np.zeros(512, np.int16); t = []
for i in range(2200):
    t0 = time.perf_counter(); s1.push(x); t.append(time.perf_counter() - t0)
print("filtered one-frame push through the Python mirror: p50 %.1f us" % (np.median(t[200:]) * 1e6))
