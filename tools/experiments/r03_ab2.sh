#!/bin/bash
# the whole GPU suite with the split-phase parser (512-byte window), then old vs new twice in alternation (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ab2.txt
: > $O
cd $R
KMP_MATCH_V2=2 timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $R/gpurun_out/r03_gpu_all_v2.log 2>&1; echo "pytest (KMP_MATCH_V2=2) rc $?" >> $O; tail -n 3 $R/gpurun_out/r03_gpu_all_v2.log >> $O
run() { echo "## $*" >> $O; env "$@" KMP_ZSTD_AUTOTUNE=0 timeout -k 10 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-pcie 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])" >> $O; }
run KMP_MATCH_V2=0
run KMP_MATCH_V2=2
run KMP_MATCH_V2=0
run KMP_MATCH_V2=2
run KMP_MATCH_V2=2 KMP_MATCH_WAVES_PER_CU_L3=8
run KMP_MATCH_V2=2 KMP_MATCH_WAVES_PER_CU_L3=12
run KMP_MATCH_V2=0 KMP_MATCH_WAVES_PER_CU_L3=8
bash tools/r03_probe.sh >> $O 2>&1
echo done >> $O
