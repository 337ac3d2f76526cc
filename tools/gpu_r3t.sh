#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/bed_probe.py 20000000 5 2>&1 | grep "device parse"
timeout 900 python tools/cli_probe.py 20000000 5 device 2>&1 | tail -8
