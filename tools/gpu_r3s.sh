#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k extra_wide 2>&1 | tail -15
