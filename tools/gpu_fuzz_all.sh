#!/bin/bash
# every GPU-side differential fuzzer over a fresh range of seeds (first seed = $1, default 1000), one summary line each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
s=${1:-1000}
mkdir -p gpurun_out/fuzz_all
timeout 900 python3 tools/e2e_fuzz.py $s 150 12 > gpurun_out/fuzz_all/e2e.log 2>&1; tail -1 gpurun_out/fuzz_all/e2e.log
timeout 900 python3 tools/cli_fuzz.py $s 100 > gpurun_out/fuzz_all/cli.log 2>&1; tail -1 gpurun_out/fuzz_all/cli.log
timeout 600 python3 tools/ingest_fuzz.py $s 800 > gpurun_out/fuzz_all/ingest.log 2>&1; tail -1 gpurun_out/fuzz_all/ingest.log
timeout 600 python3 tools/meth_fuzz.py $s 300 > gpurun_out/fuzz_all/meth.log 2>&1; tail -1 gpurun_out/fuzz_all/meth.log
grep -h MISMATCH gpurun_out/fuzz_all/*.log | cut -c1-1500 | head -5
