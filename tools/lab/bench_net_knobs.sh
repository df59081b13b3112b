# the planner's A/B knobs on the kws_conv graph (general kernel and the graph's own), one process per setting
export EDISON_NET_FORCE_GENERAL=1 EDISON_JIT_CACHE=off
for k in "" "EDISON_NET_BATCH=4 EDISON_NET_MIN_WAVES=4" "EDISON_NET_BATCH=1" "EDISON_NET_NO_TOEPLITZ=1" "EDISON_NET_NO_PIXEL_GAP=1" "" "EDISON_NET_BATCH=4 EDISON_NET_MIN_WAVES=4"; do
  echo "== ${k:-default}"
  env $k python tools/bench_net.py --specialize --reps 30 2>&1 | grep -v "amdgpu.ids\|layers_dev\|own kernel =="
done
