#!/bin/bash
# round 4: bgzip inflated on the device: tests, then the file-to-bin-motifs legs of bench.py
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -15
bash tools/gpu_cli_gz.sh 2>&1 | tail -12
