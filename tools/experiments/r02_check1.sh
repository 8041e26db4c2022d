#!/bin/bash
# round 2, first GPU call: the GPU suite, the default bench line, the N = 2 control flow rehearsed on one GPU
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest1.txt
tail -5 gpurun_out/r02_pytest1.txt
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/r02_bench1.json 2> gpurun_out/r02_bench1.err; echo "bench rc=$?"
cat gpurun_out/r02_bench1.json
KMP_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --slices 4096 --steps 2 --warmup 1 --no-cpu > gpurun_out/r02_rehearsal_n2.json 2> gpurun_out/r02_rehearsal_n2.err; echo "rehearsal rc=$?"
cat gpurun_out/r02_rehearsal_n2.json; tail -5 gpurun_out/r02_rehearsal_n2.err
