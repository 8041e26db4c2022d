for f in 2 6 10 14; do echo "flags $f"; KMP_MATCH_FLAGS=$f timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | grep -o '"kernels_ms": {[^}]*}'; done
