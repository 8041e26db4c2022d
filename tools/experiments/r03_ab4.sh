#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ab4.txt
: > $O
cd $R
run() { echo "## $*" >> $O; env "$@" timeout -k 10 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])" >> $O; }
run KMP_MATCH_FLAGS=6
run KMP_MATCH_FLAGS=134
run KMP_MATCH_FLAGS=6
run KMP_MATCH_FLAGS=134
run KMP_MATCH_FLAGS=134 KMP_TEAM_LANES=8
run KMP_MATCH_FLAGS=6 KMP_TEAM_LANES=8
cat $O
