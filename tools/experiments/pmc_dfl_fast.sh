#!/bin/bash
# HBM traffic / request counts and the wait profile of k_deflate_fast (raw DEFLATE level 1): -> stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_f; rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=$O/$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $grp -d $d -o run -- python3 $R/bench.py --mode deflate --deflate-level ${LEVEL:-1} --steps 1 --warmup 0 --no-cpu > $d.out 2>$d.err || echo "pass $grp failed"
done
cd $R && python3 - <<'PY'
import sqlite3,glob
for db in sorted(glob.glob('gpurun_out/pmc_f/**/*.db',recursive=True)):
    c=sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    t=[x for x in tabs if x.startswith('counters_collection')][0]
    rows=c.execute(f"select kernel_name,counter_name,sum(value),count(*) from {t} group by kernel_name,counter_name").fetchall()
    for r in rows:
        if 'deflate' in r[0]: print(r[0][:32], r[1], int(r[2]), r[3])
PY
find gpurun_out/pmc_f -name "*.db" -delete
