#!/bin/bash
# host bulk batch: worker threads x engine size
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_bulk.txt; : > $O
run() { echo "## $ENVV" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-pcie 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['streaming_abi']['host_batch_bulk'])" >> $O; }
ENVV="KMP_HOST_BULK_WORKERS=3 KMP_HOST_BULK_SLICES=16384" run
ENVV="KMP_HOST_BULK_WORKERS=4 KMP_HOST_BULK_SLICES=16384" run
ENVV="KMP_HOST_BULK_WORKERS=4 KMP_HOST_BULK_SLICES=8192" run
ENVV="KMP_HOST_BULK_WORKERS=6 KMP_HOST_BULK_SLICES=8192" run
ENVV="KMP_HOST_BULK_WORKERS=8 KMP_HOST_BULK_SLICES=4096" run
ENVV="KMP_HOST_BULK_WORKERS=2 KMP_HOST_BULK_SLICES=32768" run
echo done >> $O
