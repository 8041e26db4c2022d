#!/bin/bash
# level 4: throughput against the number of teams its table set holds (1 MiB each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_l4.txt; : > $O
run() { echo "## $ENVV python bench.py $*" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py "$@" --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['ratio'])" >> $O; }
for t in 8192 16384 32768 65536; do ENVV="KMP_L4_TEAMS=$t" run --level 4 --steps 3 --warmup 1; done
ENVV="KMP_L4_TEAMS=32768" run --level 4 --steps 3 --warmup 1 --slice-kib 128 --slices 32768
ENVV="KMP_L4_TEAMS=16384" run --level 4 --steps 3 --warmup 1 --slice-kib 128 --slices 32768
echo done >> $O
