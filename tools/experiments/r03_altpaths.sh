#!/bin/bash
# the GPU suite with the non-default level-3 paths switched on for every test that compresses at level 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r03_altpaths.txt; : > $O
echo "## KMP_FUSE=1" >> $O; KMP_FUSE=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $O
echo "## KMP_MATCH_V2=1" >> $O; KMP_MATCH_V2=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $O
echo done >> $O
