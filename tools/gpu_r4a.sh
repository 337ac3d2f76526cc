#!/bin/bash
# round 4: inflate prototype v2 + the ingest kernels after the key-cache change (tests, then the per-kernel split at 1e9 rows)
cd ${GRAFT_REPO_ROOT:-.}
bash tools/gpu_inflate_proto.sh 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_bed_device.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_ingest_prof.sh r4_ingest 2>&1 | tail -14
