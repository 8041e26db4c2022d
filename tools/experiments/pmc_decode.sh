#!/bin/bash
# instruction mix of the decode kernels: one rocprofv3 --pmc pass per counter group, sums per kernel printed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_dec; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS"; do
  d=$O/$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $grp -d $d -o run -- python3 $R/bench.py --mode decompress --steps 1 --warmup 0 --no-cpu > $d.out 2>$d.err || echo "pass $grp failed"
done
cd $R && python3 - <<'PY'
import sqlite3,glob,collections
for db in sorted(glob.glob('gpurun_out/pmc_dec/**/*.db',recursive=True)):
    c=sqlite3.connect(db)
    try:
        rows=c.execute("select kernel_name,counter_name,sum(value),count(*) from counters_collection group by kernel_name,counter_name").fetchall()
    except Exception as e:
        tabs=[r[0] for r in c.execute("select name from sqlite_master").fetchall()]
        print('schema?',e,tabs[:40]); continue
    for r in rows:
        if 'decode' in r[0]: print(r[0][:28], r[1], int(r[2]), r[3])
PY
find gpurun_out/pmc_dec -name "*.db" -delete
