#!/bin/bash
# small batches: wider teams (more positions speculated per step)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "## $*"; env "${@:2}" timeout -k 10 300 python $R/bench.py --slices $1 --steps 5 --warmup 2 --no-cpu --no-pcie --no-stream 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])"; }
for n in 64 256 1024 4096; do
  for g in 4 8 16 32 64; do run $n KMP_TEAM_LANES=$g; done
done
