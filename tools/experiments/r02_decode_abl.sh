#!/bin/bash
# timing-only ablations of k_zstd_decode (results wrong by design): "PRE:FLAGS" pairs as arguments
for v in "$@"; do
  export KMP_DECODE_PRE=${v%%:*} KMP_DECODE_FLAGS=${v##*:}
  echo "== KMP_DECODE_PRE=$KMP_DECODE_PRE KMP_DECODE_FLAGS=$KMP_DECODE_FLAGS"
  bash tools/prof_decode.sh 2>&1 | grep "^k_zstd" | grep -v "k_zstd_match\|k_zstd_entropy" | awk '$NF ~ /^[0-9.]+$/ && NF==4' | cut -c1-200
done
