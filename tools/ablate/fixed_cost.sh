for n in 4096 8192 16384 32768 65536 131072 262144 1048576; do python3 tools/bench_mfcc.py --frames $n --utts 64 --reps 50 2>&1 | grep "mfcc B"; done
