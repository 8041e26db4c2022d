#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "## python bench.py --config 3 --steps 3 --warmup 1   (N = 1 anchor of BASELINE configs[3]: 131 072 text / binary slices per GPU)"
timeout -k 10 500 python bench.py --config 3 --steps 3 --warmup 1 2>gpurun_out/r03_cfg3.err | tail -n 1
grep "HBM plan" gpurun_out/r03_cfg3.err
echo "## python bench.py --slices 64 --steps 5 --warmup 2 --no-cpu --no-pcie --no-stream   (a small batch: 8 lanes per slice chosen by the library)"
timeout -k 10 300 python bench.py --slices 64 --steps 5 --warmup 2 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
