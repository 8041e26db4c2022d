#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "$@" timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('random_access_roofline') or {}; print(d['value'], d['kernels_ms'], r.get('pairs_per_s_on_these_tables'), r.get('reads_per_s_on_these_tables'), d['config'].get('table_span_gib'))"; }
run KMP_MATCH_FLAGS=6
run KMP_TABLE_SPAN_GIB=80
run KMP_TABLE_SPAN_GIB=100
run KMP_TABLE_SPAN_GIB=0
run KMP_TABLE_SPAN_GIB=60
run KMP_TABLE_SPAN_GIB=140
run KMP_TABLE_ARENA=0
