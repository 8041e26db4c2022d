timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for t in 2 4; do echo "team $t"; KMP_MATCH_WAVES_PER_CU=12 timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu --team $t 2>&1 | grep -o '"kernels_ms": {[^}]*}'; done
echo "team 2, 131072 slices"; KMP_MATCH_WAVES_PER_CU=12 timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu --team 2 --slices 131072 2>&1 | grep -o '"kernels_ms": {[^}]*}'
