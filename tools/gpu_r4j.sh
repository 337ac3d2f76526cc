#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_bed_device.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_r4c.sh 2>&1 | grep -E "buffers allocated|rep|plain" | head -14
bash tools/gpu_cli_gz.sh 2>&1 | grep -E "^device|^gz_device|gz_over"
bash tools/gpu_cli_gz.sh 2>&1 | grep -E "^device|^gz_device|gz_over"
