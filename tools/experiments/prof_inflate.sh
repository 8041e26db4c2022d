#!/bin/bash
# per-kernel times of the inflate path (rocprofv3 --kernel-trace --stats over bench.py --mode deflate --slices 16384)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_inf -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode deflate --deflate-level 4 ${SLICES:+--slices $SLICES} --steps 1 --warmup 0 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_inf.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python3 -c "
import sqlite3,glob
db=glob.glob('gpurun_out/prof_inf/**/run_results.db',recursive=True)[0]
c=sqlite3.connect(db)
for r in c.execute('select name,total_calls,total_duration,average from top_kernels').fetchall()[:9]: print(r[0][:50],r[1],round(r[2]/1000,1),round(r[3]/1000,2))
"
find gpurun_out/prof_inf -name "*.db" -delete
