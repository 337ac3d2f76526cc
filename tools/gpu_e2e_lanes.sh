#!/bin/bash
# e2e with 1..6 lanes on one GPU (tools/e2e_lanes_probe.py)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/e2e_lanes
timeout 1500 python tools/e2e_lanes_probe.py > gpurun_out/e2e_lanes/probe.jsonl 2> gpurun_out/e2e_lanes/probe.err
echo "rc=$?"; cat gpurun_out/e2e_lanes/probe.jsonl | cut -c1-700; tail -5 gpurun_out/e2e_lanes/probe.err | cut -c1-300
timeout 900 python bench.py --extras e2e > gpurun_out/e2e_lanes/bench.json 2> gpurun_out/e2e_lanes/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/e2e_lanes/bench.json").read().strip().split("\n")[-1])
e = d.get("e2e", {})
print({k: e.get(k) for k in ("wall_s", "lanes", "gpu_busy_s", "gpu_busy_over_wall", "gpu_busy_sum_of_phases_s", "lanes_rows_equal_single_lane", "lane_pipeline_s", "single_lane", "motif_rows", "planted_recovered")})
print(d.get("extra_errors"), d["value"])
PY
