#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host_batch or streaming or check_raises" > gpurun_out/r03_stream_tests.log 2>&1; echo "pytest rc $?"; tail -n 5 gpurun_out/r03_stream_tests.log
timeout -k 10 300 python bench.py --slices 4096 --steps 2 --warmup 1 --no-cpu --no-pcie 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['streaming_abi'], indent=1))"
KMP_COALESCE=0 timeout -k 10 300 python bench.py --slices 4096 --steps 2 --warmup 1 --no-cpu --no-pcie 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('KMP_COALESCE=0', json.dumps(d['streaming_abi']['one_context']), json.dumps(d['streaming_abi']['contexts_64']))"
