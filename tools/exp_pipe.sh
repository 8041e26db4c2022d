for rep in 1 2; do
echo "pipelined steps (two caller streams)"; timeout -k 10 150 python bench.py --steps 6 --warmup 1 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernels_ms": {[^}]*}' | tr '\n' ' '; echo
done
timeout -k 10 150 python bench.py --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-1500
echo; echo "dist path, 1 rank"; KMP_BENCH_FORCE_DIST=1 MASTER_PORT=29612 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
timeout -k 10 150 python bench.py --mode decompress --steps 3 --warmup 1 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"roundtrip_ok": [a-z]*' | tr '\n' ' '; echo
