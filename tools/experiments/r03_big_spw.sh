#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "${@:4}" timeout -k 10 300 python3 $R/bench.py --slice-kib $1 --slices $2 --steps 2 --warmup 1 --no-cpu $3 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for spw in 4 6 8 12 16; do run 1024 16384 "" KMP_BIG_SLICES_PER_WAVE=$spw; done
for spw in 8 11 16; do run 256 32768 "" KMP_BIG_SLICES_PER_WAVE=$spw; done
for spw in 1 2 3 4; do run 1024 2048 "" KMP_BIG_SLICES_PER_WAVE=$spw; done
run 1024 8192 "" KMP_BIG_SLICES_PER_WAVE=4 KMP_BIG_TEAM_LANES=16
run 1024 8192 "" KMP_BIG_SLICES_PER_WAVE=5
run 1024 8192 "" KMP_BIG_SLICES_PER_WAVE=6
