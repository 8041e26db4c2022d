#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "$@" timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_ms'))"; }
run KMP_ENTROPY_PAD_LDS=0
run KMP_ENTROPY_PAD_LDS=300
run KMP_ENTROPY_PAD_LDS=0
run KMP_ENTROPY_PAD_LDS=300
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or ladder or special or multiblock or full_batch" 2>&1 | tail -2
