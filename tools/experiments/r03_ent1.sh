#!/bin/bash
# entropy kernel: the configuration's mix against single classes, with the timing-only ablation flags
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_ent1.txt; : > $O
run() { echo "## $ENVV python bench.py $*" >> $O; env $ENVV timeout -k 10 300 python $R/bench.py "$@" --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['config']['ratio'])" >> $O; }
ENVV="A=1" run --steps 3 --warmup 1 &&
ENVV="A=1" run --config 3 --steps 2 --warmup 1 &&
for c in T B X S D I Z R; do ENVV="A=1" run --slice-class $c --steps 2 --warmup 1; done
for f in 1 2 16 3; do ENVV="KMP_ENTROPY_FLAGS=$f" run --slice-class T --steps 2 --warmup 1; ENVV="KMP_ENTROPY_FLAGS=$f" run --slice-class B --steps 2 --warmup 1; done
echo done >> $O
