#!/bin/bash
# tiny batches: team width against latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_small.txt; : > $O
run() { echo "## $ENVV python bench.py $*" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py "$@" --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['config']['ratio'])" >> $O; }
for n in 1 16 256; do for g in 8 16 32 64; do ENVV="KMP_TEAM_LANES=$g" run --slices $n --steps 5 --warmup 2; done; done
for c in T B X; do for g in 8 32; do ENVV="KMP_TEAM_LANES=$g" run --slices 1 --steps 5 --warmup 2 --slice-class $c; done; done
echo done >> $O
