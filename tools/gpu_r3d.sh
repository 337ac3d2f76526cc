#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3d
timeout 900 python tools/prof_search.py 1 > gpurun_out/r3d/prof.txt 2>&1
NM_SEARCH_TIMING=1 timeout 600 python tools/bg_probe.py > gpurun_out/r3d/bg.txt 2>&1
tail -5 gpurun_out/r3d/bg.txt
