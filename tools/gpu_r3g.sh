#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -5
bash tools/gpu_ingest_prof.sh r3g
NM_INGEST_ATOMIC=1 bash tools/gpu_ingest_prof.sh r3g_atomic
