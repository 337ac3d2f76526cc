#!/bin/bash
# the two device parsers + the command line after a change to their setup: their tests, then the default bench's from-files leg with timing lines
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/parsers_check
timeout 2400 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_fasta_device.py tests/test_gpu_cli.py -x -q > gpurun_out/parsers_check/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/parsers_check/tests.log
bash tools/gpu_files_legs.sh "t1:NM_BED_TIMING=1;NM_FASTA_TIMING=1,t2:NM_BED_TIMING=1;NM_FASTA_TIMING=1" parsers_check 2>&1 | grep -E "fasta\]|device inflate|wall|rc=|parity|True" | cut -c1-700
