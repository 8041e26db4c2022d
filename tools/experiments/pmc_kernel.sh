#!/bin/bash
# instruction mix / wait counters of the kernels of one bench mode: tools/pmc_kernel.sh <kernel-name-substring> <bench.py args...>
cd /tmp && export TMPDIR=/tmp
K=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_k; rm -rf $O; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  d=$O/$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $grp -d $d -o run -- python3 $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu > $d.out 2>$d.err || echo "pass $grp failed"
done
cd $R && K=$K python3 - <<'PY'
import sqlite3,glob,os
for db in sorted(glob.glob('gpurun_out/pmc_k/**/*.db',recursive=True)):
    c=sqlite3.connect(db)
    rows=c.execute("select kernel_name,counter_name,sum(value),count(*) from counters_collection group by kernel_name,counter_name").fetchall()
    for r in rows:
        if os.environ['K'] in r[0]: print(r[0][:32], r[1], int(r[2]), r[3])
PY
find gpurun_out/pmc_k -name "*.db" -delete
