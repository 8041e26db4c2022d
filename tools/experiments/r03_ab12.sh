#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "$@" timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_ms'))"; }
run KMP_MATCH_FLAGS=6
run KMP_MATCH_FLAGS=2
run KMP_MATCH_FLAGS=6
run KMP_MATCH_FLAGS=2
