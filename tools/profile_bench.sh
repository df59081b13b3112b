#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own
# passes, as the gfx950 guide prescribes). Output under gpurun_out/$1; condense with tools/summarize_profiles.py.
# Since round 5 the headline step may run on two queues (config.queues = 2: a launch's own duration then includes the wait for the CUs the
# other launch still holds -- about twice the pitch); the passes whose per-launch durations and counters are quoted run with --queues 1.
set -e
ONLY=${2:-all}   # "serial": only the two --queues 1 stats passes
OUT=gpurun_out/${1:-prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --skip-cpu --skip-stream"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS --queues 1 > "$OUT/stats.json" 2> "$OUT/stats.err"
[ "$ONLY" = serial ] || rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_q" -- python3 bench.py $ARGS > "$OUT/stats_q.json" 2> "$OUT/stats_q.err"
[ "$ONLY" = serial ] || rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS --queues 1 > /dev/null 2> "$OUT/fetch.err"
[ "$ONLY" = serial ] || rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS --queues 1 > /dev/null 2> "$OUT/write.err"
# the default (steady-state) step counts, MFCC workloads only: one queue (the per-launch durations the roofline quotes), then as the bench runs by default
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_mfcc" -- python3 bench.py --skip-cpu --skip-kws --skip-stream --queues 1 > "$OUT/stats_mfcc.json" 2> "$OUT/stats_mfcc.err"
[ "$ONLY" = serial ] || rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_mfcc_q" -- python3 bench.py --skip-cpu --skip-kws --skip-stream > "$OUT/stats_mfcc_q.json" 2> "$OUT/stats_mfcc_q.err"
find "$OUT" -name "*.csv" | head -30
