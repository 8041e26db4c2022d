#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest4.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest4.txt
tail -4 gpurun_out/r02_pytest4.txt
for pre in 0 1 2 3; do echo "KMP_DECODE_PRE=$pre"; KMP_DECODE_PRE=$pre timeout -k 10 300 python bench.py --mode decompress --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 | cut -c1-230; done
./tools/prof_decode.sh
