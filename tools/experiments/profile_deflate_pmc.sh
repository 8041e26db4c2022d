#!/bin/bash
# PMC passes for the DEFLATE kernels (each its own run, no traces): -> gpurun_out/prof_deflate/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_deflate
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  d=$O/pmc_$(echo $ctr | tr ' ' '_' | cut -c1-40)
  echo "== pmc $ctr" >> $O/log.txt
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/bench.py --mode deflate --steps 1 --warmup 0 --slices 16384 --no-cpu > $d.out 2>> $O/log.txt || echo "pmc pass $ctr failed" >> $O/log.txt
done
echo done
