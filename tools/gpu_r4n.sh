#!/bin/bash
# ingest_decide with the contig memo: the ingest / e2e / CLI parity tests, then the per-kernel split at 1e9 rows
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r4n
timeout 1200 python3 -m pytest tests/test_gpu_ingest.py tests/test_gpu_synth.py tests/test_gpu_baseline_configs.py -q -m gpu -x > gpurun_out/r4n/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4n/tests.log
tail -3 gpurun_out/r4n/tests.log
timeout 500 bash tools/gpu_ingest_prof.sh r4n_ingest 2>&1 | tail -12
