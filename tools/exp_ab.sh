for rep in 1 2; do
for cfg in "4 1" "2 1" "4 2" "2 1" "4 1"; do set -- $cfg; echo "rep $rep team $1 chunks $2"; KMP_ZSTD_CHUNKS=$2 timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu --team $1 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_ms": {[^}]*}' | tr '\n' ' '; echo; done; done
