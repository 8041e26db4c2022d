#!/bin/bash
# per-kernel times of the compress step (rocprofv3 --kernel-trace --stats), printed; extra bench.py arguments pass through
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cmp -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --no-stream "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_cmp.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python3 -c "
import sqlite3,glob
db=glob.glob('gpurun_out/prof_cmp/**/run_results.db',recursive=True)[0]
c=sqlite3.connect(db)
for r in c.execute('select name,total_calls,total_duration,average from top_kernels').fetchall()[:5]: print(r[0][:50],r[1],round(r[2]/1000,1),round(r[3]/1000,2))
"
find gpurun_out/prof_cmp -name "*.db" -delete
tail -1 gpurun_out/prof_cmp.json | cut -c1-200
