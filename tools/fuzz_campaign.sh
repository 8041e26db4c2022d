#!/bin/bash
# A fuzz campaign in one gpurun call: bash tools/fuzz_campaign.sh <first seed> <small-slice seeds> <multi-block seeds> <decoder seeds>  -> gpurun_out/fuzz_campaign_<seed>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; S=${1:-41}; A=${2:-4}; B=${3:-2}; C=${4:-1}
O=$R/gpurun_out/fuzz_campaign_$S.txt; : > $O
run() { echo "## python $*" >> $O; timeout -k 10 420 python "$@" >> $O 2>&1 || { echo "FUZZ FAILED: $*" >> $O; return 1; }; }
for i in $(seq 0 $((A-1))); do run $R/tools/fuzz_gpu.py $((S+i)) 32000 || exit 1; done
for i in $(seq 0 $((B-1))); do run $R/tools/fuzz_gpu_big.py $((S+i)) || exit 1; done
for i in $(seq 0 $((C-1))); do run $R/tools/fuzz_gpu_dec.py $((S+i)) || exit 1; done
grep -c "FUZZ OK" $O; grep -n "different: [1-9]\|FAILED\|Error" $O | head
