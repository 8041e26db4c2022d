#!/bin/bash
# match kernel at fewer resident waves per CU (team slots = waves x 16; the kernel loops over the slices with its work counter)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_occ.txt; : > $O
run() { echo "## $ENVV python bench.py $*" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py "$@" --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['config']['ratio'])" >> $O; }
for w in 16 14 12 10 8; do ENVV="KMP_MATCH_WAVES_PER_CU_L3=$w" run --steps 3 --warmup 1; done
echo done >> $O
