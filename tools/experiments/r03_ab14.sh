#!/bin/bash
# the arena's layout chosen by measurement: five fresh processes each, chosen layout against layout 1 fixed
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { env "$@" KMP_PLACE_VERBOSE=1 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --no-stream 2>gpurun_out/ab14.err | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('random_access_roofline') or {}; print('$*', d['kernels_ms']['k_zstd_match'], r.get('pairs_per_s_on_these_tables'))"; grep "arena layout" gpurun_out/ab14.err | tr '\n' ';'; echo; }
for i in 1 2 3 4 5; do run KMP_TABLE_LAYOUT=0; run KMP_TABLE_LAYOUT=1; done
run KMP_TABLE_SPAN_GIB=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_batch or golden" 2>&1 | tail -2
