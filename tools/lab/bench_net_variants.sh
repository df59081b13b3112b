for v in net512 net1024; do EDISON_LIB=edison_amd/csrc/abl/libedison_hip_$v.so EDISON_NET_FORCE_GENERAL=1 timeout -k 10 200 python tools/bench_net.py 2>&1 | grep batch_dev | sed "s/^/$v: /"; done
