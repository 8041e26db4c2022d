#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_probe; rm -rf $O; mkdir -p $O
python3 $R/tools/r03_probe_pmc.py
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  timeout -k 10 300 rocprofv3 --pmc $grp -d $O/g1 -o run -- python3 $R/tools/r03_probe_pmc.py > $O/g1.out 2>$O/g1.err || echo "probe pmc failed"
done
cd $R && python3 - <<'PY'
import sqlite3,glob
for db in sorted(glob.glob('gpurun_out/pmc_probe/**/*.db',recursive=True)):
    c=sqlite3.connect(db)
    rows=c.execute("select kernel_name,counter_name,sum(value),count(*),min(value),max(value) from counters_collection group by kernel_name,counter_name").fetchall()
    for r in rows:
        if 'probe' in r[0]: print(r[0][:24], r[1], f"sum {r[2]:.6g} over {r[3]} launches, min {r[4]:.6g} max {r[5]:.6g}")
PY
find gpurun_out/pmc_probe -name "*.db" -delete
