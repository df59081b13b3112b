# pixel gap A/B on the kws_conv graph (general kernel and the graph's own), alternating processes
export EDISON_NET_FORCE_GENERAL=1 EDISON_JIT_CACHE=off
for rep in 1 2; do for k in "" "EDISON_NET_NO_PIXEL_GAP=1"; do
  echo "== ${k:-default (pixel gap)}"
  env $k python tools/bench_net.py --specialize --reps 30 2>&1 | grep -v "amdgpu.ids\|layers_dev\|own kernel =="
done; done
