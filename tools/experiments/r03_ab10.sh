#!/bin/bash
# arena layouts: pieces evenly spread (0) against two at the bottom and two at the top (1), by span; table pair rate and parser time
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "$@" timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('random_access_roofline') or {}; print(d['kernels_ms']['k_zstd_match'], r.get('pairs_per_s_on_these_tables'), r.get('measured_over_model'))"; }
for span in 0 60 80 100 140; do
  run KMP_TABLE_SPAN_GIB=$span KMP_TABLE_LAYOUT=0
  run KMP_TABLE_SPAN_GIB=$span KMP_TABLE_LAYOUT=1
done
