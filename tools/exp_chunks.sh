timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for t in 4 2; do for ch in 1 2 3; do echo "team $t chunks $ch"; KMP_ZSTD_CHUNKS=$ch timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu --team $t 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_ms": {[^}]*}'; done; done
