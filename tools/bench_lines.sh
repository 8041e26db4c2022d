#!/bin/bash
# Unprofiled bench lines of every mode, one gpurun call (same box): -> gpurun_out/bench_lines.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/bench_lines.txt
: > $O
run() { echo "## python bench.py $*" >> $O; timeout -k 10 400 python $R/bench.py "$@" 2>/dev/null | tail -n 1 >> $O; echo >> $O; }
run --steps 5 --warmup 2
run --config 3 --steps 3 --warmup 1 --no-cpu --no-stream
run --mode decompress --steps 5 --warmup 2 --no-cpu
run --mode deflate --steps 2 --warmup 1
run --mode inflate --steps 3 --warmup 1 --no-cpu
run --level 1 --steps 3 --warmup 1 --no-cpu
run --level 2 --steps 3 --warmup 1 --no-cpu
run --level 4 --steps 3 --warmup 1 --no-cpu
run --level 5 --steps 2 --warmup 1
run --level 7 --steps 2 --warmup 1
run --level 10 --steps 2 --warmup 1
run --level -1 --steps 3 --warmup 1 --no-cpu
run --level -5 --steps 3 --warmup 1 --no-cpu
run --dict-kib 16 --steps 3 --warmup 1 --no-cpu
run --dict-kib 64 --slice-kib 8 --slices 262144 --steps 3 --warmup 1 --no-cpu
run --dict-kib 64 --dict-trained --slice-kib 8 --slices 262144 --steps 3 --warmup 1 --no-cpu
run --dict-kib 16 --dict-trained --slice-kib 2 --slices 524288 --steps 3 --warmup 1 --no-cpu
run --slice-kib 2 --slices 524288 --steps 3 --warmup 1 --no-cpu --no-stream --no-pcie --no-extra
run --slice-kib 128 --slices 32768 --steps 3 --warmup 1 --no-cpu
run --slice-kib 256 --slices 32768 --steps 2 --warmup 1 --no-cpu
run --slice-kib 1024 --slices 8192 --steps 2 --warmup 1
run --slice-kib 1024 --slices 16384 --steps 2 --warmup 1 --no-cpu
run --slice-kib 1024 --slices 32768 --steps 2 --warmup 1 --no-cpu --no-stream
run --slice-kib 256 --slices 32768 --level 1 --steps 2 --warmup 1 --no-cpu
run --slice-kib 256 --slices 32768 --level 2 --steps 2 --warmup 1 --no-cpu
run --slice-kib 1024 --slices 8192 --level 2 --steps 2 --warmup 1 --no-cpu
run --slice-kib 1024 --slices 8192 --level 4 --steps 2 --warmup 1 --no-cpu --no-stream
run --slice-kib 512 --slices 16384 --level -1 --steps 2 --warmup 1 --no-cpu --no-stream
run --slice-kib 1024 --slices 8192 --level 1 --steps 2 --warmup 1 --no-cpu --no-stream
run --mode deflate --deflate-level 4 --steps 2 --warmup 1 --no-cpu
run --mode deflate --deflate-level 1 --steps 2 --warmup 1 --no-cpu
run --mode deflate --deflate-level 9 --steps 1 --warmup 1 --no-cpu
run --mode deflate --slice-kib 256 --slices 8192 --steps 2 --warmup 1 --no-cpu
run --mode deflate --slice-kib 1024 --slices 2048 --steps 2 --warmup 1 --no-cpu
run --mode deflate --deflate-window-bits 12 --deflate-mem-level 5 --steps 2 --warmup 1 --slices 16384
run --mode deflate --deflate-level 1 --deflate-window-bits 9 --deflate-mem-level 9 --steps 2 --warmup 1 --slices 16384 --no-cpu
run --mode decompress --slice-kib 256 --slices 16384 --steps 3 --warmup 1 --no-cpu
echo done
