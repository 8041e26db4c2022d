#!/bin/bash
# k_zstd_l3_fused against the two-kernel pipeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_fuse1.txt; : > $O
run() { echo "## $ENVV python bench.py $*" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py "$@" --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['config']['ratio'])" >> $O; }
ENVV="KMP_FUSE=0" run --steps 3 --warmup 1 &&
ENVV="KMP_FUSE=1" run --steps 3 --warmup 1 &&
ENVV="KMP_FUSE=1" run --steps 3 --warmup 1 --slice-class T &&
ENVV="KMP_FUSE=1" run --steps 3 --warmup 1 --slices 4096 &&
ENVV="KMP_FUSE=0" run --steps 3 --warmup 1 --slices 4096 &&
(cd $R && KMP_FUSE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_batch or golden" 2>&1 | tail -3 >> $O)
echo done >> $O
