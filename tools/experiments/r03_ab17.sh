#!/bin/bash
# arena retry threshold: fresh processes, 23.5 against 25.7 G pairs/s in the order A B B A A B B A (memory alternates between processes on some boxes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { env "$@" KMP_PLACE_VERBOSE=1 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --no-stream 2>gpurun_out/ab17.err | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('random_access_roofline') or {}; print('$*', d['kernels_ms']['k_zstd_match'], r.get('pairs_per_s_on_these_tables'))"; grep "arena retry" gpurun_out/ab17.err | tr '\n' ';'; echo; }
for t in 235 257 257 235 235 257 257 235 235 257; do run KMP_TABLE_RETRY_BELOW=$t; done
