#!/bin/bash
# The host-side C of the library (planners, model parser, table builders) under ASan + UBSan: fixture graphs, N random ones,
# truncated blobs, every filterbank / frame length the tests configure.
# usage: tools/verify/asan_planner.sh [n_random=300]
set -e
cd "$(dirname "$0")/../.."
OUT=${TMPDIR:-/tmp}/edison_asan_planner
rm -rf "$OUT"; mkdir -p "$OUT/blobs"
python3 - "$OUT/blobs" "${1:-300}" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
from edison_amd import _lib, nnom_import
import fuzz_net
out, n = sys.argv[1], int(sys.argv[2])
open(os.path.join(out, "0_shipped.ednn"), "wb").write(open(_lib.DEFAULT_MODEL, "rb").read())
for name in ("same_stride", "odd_no_softmax", "square", "kws_small", "tiny_conv", "low_latency_small", "even_same"):
    shape, layers = nnom_import.parse_weights_h(open("tests/golden/alt_models/%s.h" % name).read())
    open(os.path.join(out, name + ".ednn"), "wb").write(nnom_import.build_blob(shape, layers))
rng, k = np.random.default_rng(99), 0
while k < n:
    g = fuzz_net.random_graph(rng)
    if g is None: continue
    try:
        blob = nnom_import.build_blob(g[0], [dict(L) for L in g[1]])
    except Exception:
        continue
    open(os.path.join(out, "rnd%04d.ednn" % k), "wb").write(blob); k += 1
PY
C=edison_amd/csrc
gcc -std=gnu11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wall -Wextra -Wno-unused-parameter -Wno-unknown-pragmas \
    tools/verify/asan_planner.c $C/model_net.c $C/model_net_mm.c $C/net_spec.c $C/model.c $C/tables.c $C/tables_q15.c $C/tables_f32.c -lm -o "$OUT/asan_planner"
"$OUT/asan_planner" "$OUT"/blobs/*.ednn
