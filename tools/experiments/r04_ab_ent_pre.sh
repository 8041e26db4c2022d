R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ab_ent_pre.txt; : > $O
b() { echo "## $*" >> $O; "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('kernels_ms'), d['roofline'].get('avg_launch_ms'))" >> $O 2>&1; }
for rep in 1 2; do
b python bench.py --steps 4 --warmup 1 --no-cpu --no-pcie --no-stream --no-extra
KMP_LIB_PATH=$R/kompressor_amd/libkompressor_hip_F8.so b python bench.py --steps 4 --warmup 1 --no-cpu --no-pcie --no-stream --no-extra
b python bench.py --mode decompress --steps 5 --warmup 2 --no-cpu
KMP_LIB_PATH=$R/kompressor_amd/libkompressor_hip_F8.so b python bench.py --mode decompress --steps 5 --warmup 2 --no-cpu
KMP_LIB_PATH=$R/kompressor_amd/libkompressor_hip_F4.so b python bench.py --mode decompress --steps 5 --warmup 2 --no-cpu
done
cat $O
