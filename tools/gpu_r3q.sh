#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3q
bash tools/ab_bench.sh "libnmscan.so libnmscan_skip.so" 3 2>&1 | tee gpurun_out/r3q/ab_skip_empty.txt
