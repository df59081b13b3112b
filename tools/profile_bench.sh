#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own
# passes, as the gfx950 guide prescribes). Output under gpurun_out/$1; condense with tools/summarize_profiles.py.
set -e
OUT=gpurun_out/${1:-prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --skip-cpu --skip-stream"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/write.err"
# the default (steady-state) step counts, MFCC workloads only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_mfcc" -- python3 bench.py --skip-cpu --skip-kws --skip-stream > "$OUT/stats_mfcc.json" 2> "$OUT/stats_mfcc.err"
find "$OUT" -name "*.csv" | head -20
