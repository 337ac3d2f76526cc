#!/bin/bash
# nm_hit_positions / nm_contig_methylation out to 95 positions from the modified base: the touched parity tests.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r4l
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_read_methylation.py tests/test_gpu_per_contig.py -q -m gpu -x > gpurun_out/r4l/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4l/tests.log
tail -5 gpurun_out/r4l/tests.log
