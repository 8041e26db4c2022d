#!/bin/bash
# the old parser with its position-dependent loads merged: small batches, the full batch, big frames; then the GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; timeout -k 10 300 python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_ms'))"; }
run --slices 64
run --slices 1024
run --slices 4096
run --slices 16384
run
run
run --slice-kib 1024 --slices 8192
run --slice-kib 256 --slices 32768
run --slice-kib 1024 --slices 16384
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
