#!/bin/bash
# the default bench line (as the driver runs it), timed
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/bench_default
start=$(date +%s.%N)
timeout 1200 python bench.py > gpurun_out/bench_default/bench.json 2> gpurun_out/bench_default/bench.log
echo "bench rc=$? in $(echo "$(date +%s.%N) - $start" | bc) s"
python3 - gpurun_out/bench_default/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "vs_baseline")}, "kernel_ms", d["roofline"].get("kernel_ms"), "frac", d["roofline"].get("frac"))
e = d.get("e2e", {})
print("e2e wall", e.get("wall_s"), "gpu_busy_over_wall", e.get("gpu_busy_over_wall"), {k: round(v, 4) for k, v in (e.get("timings_rank0") or {}).items() if isinstance(v, float)})
f = d.get("e2e_files", {})
print("e2e_files wall", f.get("wall_s"), f.get("wall_s_second_process"), "parity", f.get("parity"), "written in", f.get("files_written_in_s"))
print("   phases", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (f.get("phases") or {}).items() if not isinstance(v, dict)})
print("errors", d.get("extra_errors"))
PY
