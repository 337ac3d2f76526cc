#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests/test_gpu_synth.py tests/test_gpu_cli.py tests/test_gpu_windows.py tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/r2e/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2e/tests.log
tail -8 gpurun_out/r2e/tests.log
timeout 300 python bench.py --workload e2e > gpurun_out/r2e/e2e.json 2> gpurun_out/r2e/e2e.log
NANOMOTIF_PY_SEARCH=1 timeout 300 python bench.py --workload e2e > gpurun_out/r2e/e2e_py.json 2> gpurun_out/r2e/e2e_py.log
timeout 300 python -m cProfile -o gpurun_out/r2e/e2e.prof bench.py --workload e2e > /dev/null 2>&1
python - <<'PY'
import json
for f in ("e2e", "e2e_py"):
    d = json.load(open(f"gpurun_out/r2e/{f}.json"))
    print(f, d["value"], {k: d["timings_rank0"][k] for k in ("search_s", "upload_filter_s", "gpu_busy_s", "rounds", "candidates")}, d["per_rank"][0]["motif_rows"])
PY
