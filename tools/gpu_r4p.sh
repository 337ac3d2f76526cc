#!/bin/bash
# round 4: window batches of a search on the context's second stream (default) against everything on the ctx stream
# (NM_WIN_STREAM=0): the window / search / CLI tests in both modes, then the end-to-end run three times each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r4p
for m in 1 0; do
  NM_WIN_STREAM=$m timeout 1500 python3 -m pytest tests/test_gpu_windows.py tests/test_gpu_synth.py tests/test_gpu_baseline_configs.py tests/test_gpu_comm.py tests/test_gpu_lanes.py -q -m gpu -x > gpurun_out/r4p/tests_$m.log 2>&1
  echo "NM_WIN_STREAM=$m tests rc=$? $(tail -1 gpurun_out/r4p/tests_$m.log)"
done
for rep in 1 2 3; do
  for m in 1 0; do
    NM_WIN_STREAM=$m NM_SEARCH_TIMING=1 timeout 600 python3 bench.py --workload e2e > gpurun_out/r4p/e2e_${m}_$rep.json 2> gpurun_out/r4p/e2e_${m}_$rep.log
    python3 - <<PY
import json
d = json.load(open("gpurun_out/r4p/e2e_${m}_$rep.json"))
e = d["e2e"]; t = e["timings_rank0"]
print("NM_WIN_STREAM=$m rep $rep wall %.4f search_s %.4f native %.4f rounds %d gpu_busy %.4f rows %d" % (e["wall_s"], e["search_s"], t["native_search_s"], e["rounds"], e["gpu_busy_s"], e["motif_rows"]))
PY
    grep nm_search gpurun_out/r4p/e2e_${m}_$rep.log | tail -1
  done
done
