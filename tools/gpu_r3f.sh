#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_ingest_prof.sh r3f
