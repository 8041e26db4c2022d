#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python tools/r03_hostbatch.py 65536
KMP_HOST_BULK_WORKERS=3 python tools/r03_hostbatch.py 65536
KMP_HOST_BULK_WORKERS=1 python tools/r03_hostbatch.py 65536
KMP_HOST_BULK_WORKERS=0 python tools/r03_hostbatch.py 16384
