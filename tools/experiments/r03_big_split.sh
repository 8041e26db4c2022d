#!/bin/bash
# where does k_zstd_big's time go?  The same frames built round by round (KMP_BIG_ROUNDS=1: a parse launch and a frame launch per round of blocks)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/big_split; rm -rf $O; mkdir -p $O
KMP_BIG_ROUNDS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/rounds -o run -- python3 $R/bench.py --slice-kib 1024 --slices 8192 --steps 2 --warmup 1 --no-cpu > $O/rounds.out 2>$O/rounds.err
tail -n 1 $O/rounds.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('by rounds:', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
cd $R && python3 - <<'PY'
import sqlite3,glob
for db in glob.glob('gpurun_out/big_split/**/*.db', recursive=True):
    rows = sqlite3.connect(db).execute("select name,total_calls,total_duration,average from top_kernels").fetchall()
    for r in rows[:8]: print(r[0][:60], r[1], round(r[2]/1e6,1), 'ms total', round(r[3]/1e3,1), 'us avg')
PY
find gpurun_out/big_split -name "*.db" -delete
for spw in 1 2 3 4 8; do echo "## KMP_BIG_SLICES_PER_WAVE=$spw"; KMP_BIG_SLICES_PER_WAVE=$spw timeout -k 10 300 python3 $R/bench.py --slice-kib 1024 --slices 8192 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for g in 4 16; do echo "## KMP_BIG_TEAM_LANES=$g"; KMP_BIG_TEAM_LANES=$g timeout -k 10 300 python3 $R/bench.py --slice-kib 1024 --slices 8192 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
