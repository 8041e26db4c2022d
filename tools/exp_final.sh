timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "dist path, 1 rank"; KMP_BENCH_FORCE_DIST=1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
