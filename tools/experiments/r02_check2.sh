#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest2.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest2.txt
tail -15 gpurun_out/r02_pytest2.txt
