#!/bin/bash
# arena retry: eight fresh processes with the retry, verbose; free memory before / after creation
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { env "$@" KMP_PLACE_VERBOSE=1 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --no-stream 2>gpurun_out/ab16.err | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('random_access_roofline') or {}; print('$*', d['kernels_ms']['k_zstd_match'], r.get('pairs_per_s_on_these_tables'))"; grep "arena retry" gpurun_out/ab16.err | tr '\n' ';'; grep -c "arena 0x" gpurun_out/ab16.err; }
for i in 1 2 3 4 5 6 7 8; do run KMP_TABLE_LAYOUT=0; done
