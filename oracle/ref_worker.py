"""oracle/ref_worker.py -- TEST / BASELINE INFRASTRUCTURE. One worker of bench.py's all-cores leg of the reference-CNN CPU
baseline: loads oracle/_ref/libnnom_ref.so (the reference NNoM + CMSIS-NN + weights.h, compiled by oracle/Makefile), waits
for the wall-clock mark every worker was given, runs `n` inferences on seeded random int8 inputs and prints when it began
and ended. The reference model lives in globals (firmware/src/ai/nnom/kws_nnom/weights.h:136-137), hence one process per core.

    python oracle/ref_worker.py <n> <start epoch seconds> <seed>
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from oracle import oracle  # noqa: E402


def main():
    n, start, seed = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    f = np.random.default_rng(100 + seed).integers(-128, 128, (n, 403)).astype(np.int8)
    oracle.nnom_ref_batch(f[:64])                 # warm-up: page the library and the weights in
    while time.time() < start:
        pass
    t0 = time.time()
    out = oracle.nnom_ref_batch(f)
    t1 = time.time()
    print(json.dumps(dict(t_begin=t0, t_end=t1, n=n, checksum=int(np.asarray(out["softmax"], dtype=np.int64).sum()))))


if __name__ == "__main__":
    main()
