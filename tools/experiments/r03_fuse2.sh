#!/bin/bash
# k_zstd_l3_fused on configs[3] and single classes; the parity test with the fused section
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_fuse2.txt; : > $O
run() { echo "## $ENVV python bench.py $*" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py "$@" --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['config']['ratio'])" >> $O; }
ENVV="KMP_FUSE=0" run --config 3 --steps 2 --warmup 1 &&
ENVV="KMP_FUSE=1" run --config 3 --steps 2 --warmup 1 &&
for c in B X I; do ENVV="KMP_FUSE=1" run --slice-class $c --steps 2 --warmup 1; done
(cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_batch" 2>&1 | tail -3 >> $O)
echo done >> $O
