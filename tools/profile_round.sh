#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats for each bench mode + separate PMC passes for the compress step.
# Usage (through gpurun): bash tools/profile_round.sh [name]   -> gpurun_out/prof/, summary gpurun_out/prof_summary/<name>.txt
NAME=${1:-r04_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*" >> $O/log.txt; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/$name -o run -- python3 $R/bench.py "$@" > $O/$name.out 2>> $O/log.txt; tail -1 $O/$name.out > $O/$name.json; }
run compress --steps 5 --warmup 1 --no-stream --no-extra &&
run config3 --config 3 --steps 3 --warmup 1 --no-cpu --no-stream &&
run decompress --mode decompress --steps 3 --warmup 1 --no-cpu &&
run deflate --mode deflate --steps 1 --warmup 0 &&
run big1m --slice-kib 1024 --slices 8192 --steps 2 --warmup 1 &&
run big256k --slice-kib 256 --slices 32768 --steps 2 --warmup 1 --no-cpu &&
run level1 --level 1 --steps 3 --warmup 1 --no-cpu &&
run level7 --level 7 --slices 16384 --steps 2 --warmup 1 --no-cpu &&
run dict_trained --dict-kib 64 --dict-trained --slice-kib 8 --slices 262144 --steps 3 --warmup 1 --no-cpu &&
run deflate1 --mode deflate --deflate-level 1 --steps 1 --warmup 0 --no-cpu &&
run deflate256k --mode deflate --slice-kib 256 --slices 8192 --steps 2 --warmup 1 --no-cpu &&
run deflate_w12m5 --mode deflate --deflate-window-bits 12 --deflate-mem-level 5 --slices 16384 --steps 2 --warmup 1 --no-cpu &&
run inflate --mode inflate --steps 3 --warmup 1 --no-cpu &&
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$O/pmc_$(echo $ctr | tr ' ' '_' | cut -c1-40)
  echo "== pmc $ctr" >> $O/log.txt
  # (every launch of such a pass is one full batch: no PCIe pass, no streaming figures, no trial of launch settings)
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-pcie --no-stream --no-extra > $d.out 2>> $O/log.txt || echo "pmc pass $ctr failed" >> $O/log.txt
done
# HBM bytes of the decoder and of the DEFLATE kernels (same counters, their own runs)
for mode in decompress deflate inflate; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc2_${mode}_$ctr
    echo "== pmc $mode $ctr" >> $O/log.txt
    if [ $mode = deflate ]; then extra="--slices 16384"; else extra=""; fi
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/bench.py --mode $mode --steps 1 --warmup 0 --no-cpu $extra > $d.out 2>> $O/log.txt || echo "pmc pass $mode $ctr failed" >> $O/log.txt
  done
done
# instruction mix of the decode pipeline (VERDICT r1 item 3 asks for SQ_INSTS_VALU per slice)
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  d=$O/pmc2_decompress_$(echo $ctr | tr ' ' '_' | cut -c1-30)
  echo "== pmc decompress $ctr" >> $O/log.txt
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/bench.py --mode decompress --steps 1 --warmup 0 --no-cpu > $d.out 2>> $O/log.txt || echo "pmc pass decompress $ctr failed" >> $O/log.txt
done
# ... and of the other parsers: levels 1 / 2 ("fast"), the dictionary parser, the block-chain kernel (frames of several blocks)
pmc3() { tag=$1; shift; for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc3_${tag}_$ctr; echo "== pmc $tag $ctr" >> $O/log.txt
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-stream --no-extra "$@" > $d.out 2>> $O/log.txt || echo "pmc pass $tag $ctr failed" >> $O/log.txt
  done; }
pmc3 level1 --level 1
pmc3 dict16 --dict-kib 16
pmc3 big256k --slice-kib 256 --slices 32768
pmc3 big1m --slice-kib 1024 --slices 8192
ls -R $O | head -120 >> $O/log.txt
# summarise here (the result databases exceed what gpurun carries back), keep only text
cd $R && python3 tools/summarize_profiles.py $NAME $R/gpurun_out/prof_summary > /dev/null 2>> $O/log.txt
find $O -name 'run_results.db' -delete
echo done
