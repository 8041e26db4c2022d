#!/bin/bash
# timing-only ablations of k_zstd_entropy alone on the GPU (one chunk: no overlap with the match kernel); KMP_ENTROPY_FLAGS values as arguments
for v in "$@"; do
  export KMP_ZSTD_CHUNKS=1 KMP_ENTROPY_FLAGS=$v
  echo "== KMP_ENTROPY_FLAGS=$v"
  bash tools/prof_compress.sh 2>&1 | grep "k_zstd_entropy"
done
