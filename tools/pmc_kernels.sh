#!/bin/bash
# rocprofv3 --pmc passes over one bench.py command, summed per kernel: bash tools/pmc_kernels.sh <outname> "<counters pass 1>" "<counters pass 2>" ... -- <bench.py args>
# (each pass its own run, as the MI355X guide prescribes; results: gpurun_out/<outname>.txt)
NAME=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
PASSES=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do PASSES+=("$1"); shift; done
shift
O=$R/gpurun_out/pmc_$NAME; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/$NAME.txt
i=0
for ctr in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $ctr -d $O/p$i -o run -- python3 $R/bench.py "$@" > $O/p$i.out 2>&1 || echo "pass $i failed" >> $R/gpurun_out/$NAME.txt
  python3 - $O/p$i >> $R/gpurun_out/$NAME.txt <<'PY'
import sqlite3, sys, glob
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    rows = sqlite3.connect(db).execute("select kernel_name,counter_name,sum(value),count(*) from counters_collection group by kernel_name,counter_name").fetchall()
    for k, c, v, n in sorted(rows):
        if k.startswith("void k_") or k.startswith("k_"):
            print(f"{k.split('(')[0][:44]:44s} {c:24s} {v:.6g}   ({n} launches)")
PY
  find $O/p$i -name "*.db" -delete
done
cat $R/gpurun_out/$NAME.txt
