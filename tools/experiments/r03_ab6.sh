#!/bin/bash
# small batches are latency-bound: does the split-phase parser win there?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "${@:2}" timeout -k 10 300 python $R/bench.py --slices $1 --steps 5 --warmup 2 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])"; }
for n in 64 1024 4096 16384 32768; do
  run $n KMP_MATCH_V2=0
  run $n KMP_MATCH_V2=1
  run $n KMP_MATCH_V2=2
done
run 1024 KMP_MATCH_V2=0 KMP_TEAM_LANES=8
run 1024 KMP_MATCH_V2=2 KMP_TEAM_LANES=8
run 64 KMP_MATCH_V2=0 KMP_TEAM_LANES=8
run 64 KMP_MATCH_V2=2 KMP_TEAM_LANES=8
