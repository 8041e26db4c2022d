for n in 8192 16384 32768 49152 65536 98304; do echo "slices $n"; timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu --slices $n 2>&1 | grep -o '"kernels_ms": {[^}]*}'; done
for w in 4 8 16; do echo "waves/CU $w"; KMP_MATCH_WAVES_PER_CU=$w timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | grep -o '"kernels_ms": {[^}]*}'; done
