#!/bin/bash
# decode parity + per-kernel times under the pre-decode settings given as arguments (KMP_DECODE_PRE values)
set -e
for v in "$@"; do
  export KMP_DECODE_PRE=$v
  echo "== KMP_DECODE_PRE=$v"
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predecode or decoder or mutated or roundtrip or decompress" 2>&1 | tail -2
  bash tools/prof_decode.sh 2>&1 | grep -v "k_zstd_match\|k_zstd_entropy\|k_compact\|fillBuffer\|rocpd" | cut -c1-200
done
