#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python tools/cli_probe.py 20000000 5 2>&1 | tail -6 | tee gpurun_out/r2i/cli_probe.txt
bash profiles/run_profile.sh r2_cfg5all --workload cfg5_all --steps 2 --warmup 1 > /dev/null 2>&1
ls gpurun_out/prof_r2_cfg5all | head -3
