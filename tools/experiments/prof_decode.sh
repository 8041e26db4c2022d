#!/bin/bash
# per-kernel times of the decode path (rocprofv3 --kernel-trace --stats), printed; no database travels back
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dec -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode decompress --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_dec.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python3 -c "
import sqlite3,glob
db=glob.glob('gpurun_out/prof_dec/**/run_results.db',recursive=True)[0]
c=sqlite3.connect(db)
for r in c.execute('select name,total_calls,total_duration,average from top_kernels').fetchall()[:6]: print(r[0][:50],r[1],round(r[2]/1000,1),round(r[3]/1000,2))
t=[x[0] for x in c.execute(\"select name from sqlite_master where name like 'kernels%' or name like '%kernel_dispatch%'\").fetchall()]
print(t)
try:
    rows=c.execute('select name,start,end from kernels order by start').fetchall()
    rows=[r for r in rows if 'decode' in r[0]][-10:]
    t0=rows[0][1]
    for r in rows: print(r[0][:30],round((r[1]-t0)/1e6,2),round((r[2]-t0)/1e6,2))
except Exception as e: print('timeline:',e)
"
find gpurun_out/prof_dec -name "*.db" -delete
tail -1 gpurun_out/prof_dec.json | cut -c1-260
