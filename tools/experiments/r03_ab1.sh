#!/bin/bash
# round 3, first GPU contact of the split-phase parser: parity tests of the level-3 path, then old vs new in one call (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ab1.txt
: > $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $R/gpurun_out/r03_gpu_parity.log 2>&1; echo "pytest rc $?" >> $O; tail -n 3 $R/gpurun_out/r03_gpu_parity.log >> $O
run() { echo "## $*" >> $O; env "$@" KMP_ZSTD_AUTOTUNE=0 timeout -k 10 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-pcie 2>/dev/null | tail -n 1 >> $O; echo >> $O; }
run KMP_MATCH_V2=0
run KMP_MATCH_V2=1
run KMP_MATCH_V2=2
run KMP_MATCH_V2=1 KMP_TEAM_LANES=2
run KMP_MATCH_V2=1 KMP_TEAM_LANES=8
run KMP_MATCH_V2=1 KMP_TABLE_SPREAD=0
echo done >> $O
