#!/bin/bash
# round 5, first call: the bgzip / tabix parser tests after the NM_EINDEX + ISIZE changes, the product's bin-motifs.tsv of ALL 500
# bins of the 1 Gbp full loop (for tools/cfg5_all_bins_parity.py on CPUs), a default bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r5a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
timeout 900 python tools/gpu_cfg5_rows.py > gpurun_out/r5a/cfg5_rows.log 2>&1
echo "rows rc=$?"; tail -3 gpurun_out/r5a/cfg5_rows.log
timeout 600 python bench.py > gpurun_out/r5a/bench.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r5a/bench.log | cut -c1-1500
