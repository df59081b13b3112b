set -e
export EDISON_NET_FORCE_GENERAL=1 EDISON_JIT_CACHE=off
echo "== shipped kws_conv"; python tools/bench_net.py --specialize --reps 20
for b in 2 4; do echo "== shipped kws_conv, EDISON_NET_BATCH=$b"; EDISON_NET_BATCH=$b python tools/bench_net.py --specialize --reps 20; done
for m in same_stride odd_no_softmax square kws_small tiny_conv low_latency_small even_same; do echo "== $m"; python tools/bench_net.py --specialize --reps 20 --header tests/golden/alt_models/$m.h; done
