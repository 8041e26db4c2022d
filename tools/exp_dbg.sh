timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "multiblock" 2>&1 | grep -v "^$" | tail -40
