#!/bin/bash
# PMC passes of the level-3 parser, old (KMP_MATCH_V2=0) and split-phase (1, 2): instruction mix, waits, EA requests, L2 hits.
# One launch per process (autotune off, no PCIe pass): sums are per launch of 65 536 slices.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_m; rm -rf $O; mkdir -p $O
export KMP_ZSTD_AUTOTUNE=0
VARIANTS=${VARIANTS:-"0 1"}
for v in $VARIANTS; do
 export KMP_MATCH_V2=$v
 i=0
 for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=$O/v${v}_g$i
  timeout -k 10 300 rocprofv3 --pmc $grp -d $d -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-pcie "$@" > $d.out 2>$d.err || echo "pass v$v $grp failed"
 done
done
cd $R && python3 - <<'PY'
import sqlite3,glob,os
for db in sorted(glob.glob('gpurun_out/pmc_m/**/*.db',recursive=True)):
    c=sqlite3.connect(db)
    try: rows=c.execute("select kernel_name,counter_name,sum(value),count(*) from counters_collection group by kernel_name,counter_name").fetchall()
    except Exception as e: print(db, e); continue
    tag=db.split('/')[2]
    for r in rows:
        if 'k_zstd_match' in r[0] or 'k_zstd_entropy' in r[0]: print(tag, r[0][:30], r[1], f"{r[2]:.6g}", r[3])
PY
find gpurun_out/pmc_m -name "*.db" -delete
