#!/bin/bash
# k_zstd_decode under different launch bounds (waves per SIMD): rebuilt on the box, per-kernel times printed
export KMP_DECODE_PRE=${KMP_DECODE_PRE:-3}
for b in "$@"; do
  sed -i "s/__launch_bounds__(64, [0-9]*) void k_zstd_decode/__launch_bounds__(64, $b) void k_zstd_decode/" kompressor_amd/csrc/kmp_api.hip
  python -c "from kompressor_amd import build; build.build_hip(force=True)" 2>&1 | grep -E "error" 
  echo "== launch bound $b"
  bash tools/prof_decode.sh 2>&1 | grep -v "k_zstd_match\|k_zstd_entropy\|k_compact\|fillBuffer\|rocpd\|reduce_kernel" | cut -c1-150
done
