#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3m/tests.log 2>&1
tail -25 gpurun_out/r3m/tests.log
df -h /dev/shm | tail -1
timeout 900 python tools/cli_probe.py 20000000 5 device 2>&1 | tail -14
timeout 900 python tools/cli_probe.py 20000000 5 host 2>&1 | tail -3
