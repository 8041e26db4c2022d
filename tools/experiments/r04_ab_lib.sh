# A/B on one box: bench.py <args> with the product library and with another build of it (KMP_LIB_PATH): bash tools/experiments/r04_ab_lib.sh <other .so> <reps> -- <bench args>
R=$GRAFT_REPO_ROOT; LIB=$1; REPS=$2; shift 3
O=$R/gpurun_out/r04_ab_lib.txt; : > $O
b() { echo "## ${KMP_LIB_PATH:-product} $*" >> $O; "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('kernels_ms_first_workspace_chunk') or d.get('kernels_ms'))" >> $O 2>&1; }
for rep in $(seq 1 $REPS); do
b python bench.py "$@"
KMP_LIB_PATH=$R/$LIB b python bench.py "$@"
done
cat $O
