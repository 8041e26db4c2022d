#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest3.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest3.txt
tail -6 gpurun_out/r02_pytest3.txt
timeout -k 10 300 python bench.py --mode decompress --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/dec_pre.json; cat gpurun_out/dec_pre.json
KMP_DECODE_PRE=0 timeout -k 10 300 python bench.py --mode decompress --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/dec_nopre.json; cat gpurun_out/dec_nopre.json
