#!/bin/bash
# arena placement: the arena's own span against KMP_TABLE_SPAN_GIB=80 / 120, old and new parser; then the default bench line with the CPU leg
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ab3.txt
: > $O
cd $R
run() { echo "## $*" >> $O; env "$@" timeout -k 10 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-pcie 2>>$O | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d.get('random_access_roofline'))" >> $O; }
run KMP_MATCH_V2=0
run KMP_MATCH_V2=0 KMP_TABLE_SPAN_GIB=80
run KMP_MATCH_V2=0 KMP_TABLE_SPAN_GIB=120
run KMP_MATCH_V2=2 KMP_TABLE_SPAN_GIB=80
run KMP_MATCH_V2=0 KMP_TABLE_ARENA=0
run KMP_MATCH_V2=0
echo "## default line" >> $O
timeout -k 10 400 python $R/bench.py >> $O 2>&1
echo done >> $O
