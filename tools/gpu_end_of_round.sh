#!/bin/bash
# end-of-round evidence: the lanes tests, the 1 Gbp full loop's bin-motifs.tsv (sha256 against the text the oracle was diffed with,
# profiles/r5/cfg5_all_bins_parity.txt), the same through two lanes
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/end_of_round
timeout 1500 python -m pytest tests/test_gpu_e2e_lanes.py -x -q > gpurun_out/end_of_round/lanes_tests.log 2>&1
echo "lanes tests rc=$?"; tail -3 gpurun_out/end_of_round/lanes_tests.log
timeout 900 python tools/gpu_cfg5_rows.py end_of_round/cfg5_rows > gpurun_out/end_of_round/cfg5_rows.log 2>&1
echo "cfg5 rows rc=$?"; sha256sum gpurun_out/end_of_round/cfg5_rows/bin-motifs.tsv
